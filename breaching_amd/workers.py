"""Worker processes for trial-parallel restarts from a single-process entry point ("Process model").

``simulate_breach.py`` is one Python process on one device.  To spread ``restarts.num_trials`` over the GPUs of a node
without touching that script, the attacker keeps a pool of persistent worker processes -- one per additional GPU, started
lazily by the first ``reconstruct`` that has more than one trial and more than one device to use.  The calling process is
rank 0 and runs its own share of the trials; rank r > 0 runs trials {t : t mod W == r} on its device with its own model
replica.  The ranks form a ``torch.distributed`` group ("nccl" = RCCL over xGMI when every rank has its own GPU, "gloo"
when ranks share a device, which is how the path is exercised on a 1-GPU box) whose only traffic is the trial selection
of ``trials.TrialShard.select``.  The rendezvous (RCCL unique id included) goes through a TCP store on 127.0.0.1.

reference: the sequential trial loop of breaching/attacks/optimization_based_attack.py:70-78 -- the reference has no
multi-device code, this is an MI355X-native addition behind the unchanged ``reconstruct`` signature.

Protocol (one duplex pipe per worker):  parent -> worker ``("job", dict)`` | ``("stop",)``;  worker -> parent
``("ready",)`` once its process group is up, ``("trials_done",)`` when its trials are finished -- it then waits for
``("go",)`` from the parent before entering the selection collective -- ``("ok",)`` when the job is finished,
``("error", traceback)`` on failure.  The parent sends ``go`` only after every worker reported ``trials_done``: all ranks
enter the collective within milliseconds of each other (no rank sits in RCCL while another still optimises, so the
collective watchdog never fires), and a crashed worker raises in the parent instead of hanging it.
"""

import logging
import os
import socket
import traceback

import torch

log = logging.getLogger(__name__)


_ACTIVE_POOL = None


def active_pool():
    """The worker pool that owns this process's default process group, or None.  While a pool is idle (no job submitted),
    `torch.distributed.is_initialized()` is true although nobody else will join a collective -- callers that did not submit
    work to the pool must not shard over that group."""
    pool = _ACTIVE_POOL
    return pool if pool is not None and not getattr(pool, "closed", True) else None


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def to_cpu(obj):
    """Deep copy of nested lists / tuples / dicts with every tensor moved to host memory (what travels over the pipe)."""
    if torch.is_tensor(obj):
        return obj.detach().to("cpu")
    if isinstance(obj, dict):
        return type(obj)((k, to_cpu(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_cpu(v) for v in obj)
    return obj


def requested_devices(cfg, device):
    """Device indices the restarts may use: ``cfg.impl.trial_devices`` / ``BREACH_HIP_TRIAL_DEVICES`` ("all", "0,1,2", or a
    list; an index may repeat to put several ranks on one GPU), default: every visible GPU, the caller's own first."""
    spec = os.environ.get("BREACH_HIP_TRIAL_DEVICES")
    if spec is None:
        try:
            spec = cfg.impl["trial_devices"]
        except (KeyError, AttributeError, TypeError):
            spec = None
    own = device.index if device.index is not None else torch.cuda.current_device()
    if spec is None or spec == "all":
        return [own] + [i for i in range(torch.cuda.device_count()) if i != own]
    if isinstance(spec, str):
        spec = [int(tok) for tok in spec.replace(" ", "").split(",") if tok != ""]
    devices = [int(d) for d in spec]
    if len(devices) == 0 or devices[0] != own:
        raise ValueError(f"trial_devices={devices} must start with the attacker's own device index {own} (rank 0 is the caller).")
    return devices


def _worker_main(rank, world, port, backend, device_index, conn, runner_factory, factory_args):
    """Entry point of rank `rank` > 0."""
    import torch.distributed as dist

    try:
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        if device_index is not None:
            torch.cuda.set_device(device_index)
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
        runner = runner_factory(rank, world, device_index, conn, *factory_args)
        conn.send(("ready",))
        while True:
            message = conn.recv()
            if message[0] == "stop":
                break
            try:
                runner(message[1])
                conn.send(("ok",))
            except Exception:  # report and stay alive for the next job
                conn.send(("error", traceback.format_exc()))
        dist.destroy_process_group()
    except (EOFError, KeyboardInterrupt):
        pass
    except Exception:
        try:
            conn.send(("error", traceback.format_exc()))
        except Exception:
            pass


class TrialWorkerPool:
    """W - 1 persistent worker processes plus the calling process as rank 0 of one process group."""

    def __init__(self, devices, runner_factory, factory_args=(), backend=None, start_timeout=600.0):
        import datetime

        import torch.distributed as dist
        import torch.multiprocessing as mp

        if dist.is_initialized():
            raise RuntimeError("TrialWorkerPool needs to own the default process group of the calling process.")
        self.devices = list(devices)
        self.world = len(self.devices)
        if self.world < 2:
            raise ValueError("A worker pool needs at least two ranks.")
        if backend is None:
            on_gpu = all(d is not None for d in self.devices)
            backend = "nccl" if on_gpu and len(set(self.devices)) == self.world else "gloo"
        self.backend = backend
        self.port = free_port()
        ctx = mp.get_context("spawn")
        self.workers = []
        for rank in range(1, self.world):
            parent_end, child_end = ctx.Pipe(duplex=True)
            proc = ctx.Process(target=_worker_main, daemon=True,
                               args=(rank, self.world, self.port, backend, self.devices[rank], child_end, runner_factory, factory_args))
            proc.start()
            child_end.close()
            self.workers.append((proc, parent_end))
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(self.port)
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", self.devices[0])
        try:
            dist.init_process_group(backend, rank=0, world_size=self.world,
                                    timeout=datetime.timedelta(seconds=start_timeout), **kwargs)
            for message in self._collect(start_timeout):
                if message[0] != "ready":
                    raise RuntimeError(f"worker failed to start:\n{message[-1]}")
        except Exception:
            self.close(force=True)
            raise
        self.closed = False
        global _ACTIVE_POOL
        _ACTIVE_POOL = self
        log.info(f"Trial worker pool up: {self.world} ranks on devices {self.devices} ({backend}).")

    # -- messaging -------------------------------------------------------------------------------------------------
    def submit(self, jobs):
        """`jobs[r - 1]` goes to rank r."""
        for (proc, conn), job in zip(self.workers, jobs):
            conn.send(("job", job))

    def broadcast(self, message):
        for proc, conn in self.workers:
            conn.send(message)

    def _collect(self, timeout=None, poll=0.2):
        """One message from every worker, raising if a worker died or reported an error."""
        import time

        out = [None] * len(self.workers)
        deadline = None if timeout is None else time.time() + timeout
        while any(m is None for m in out):
            for idx, (proc, conn) in enumerate(self.workers):
                if out[idx] is not None:
                    continue
                if conn.poll(poll):
                    message = conn.recv()
                    if message[0] == "error":
                        raise RuntimeError(f"trial worker (rank {idx + 1}) failed:\n{message[1]}")
                    out[idx] = message
                elif not proc.is_alive():
                    raise RuntimeError(f"trial worker (rank {idx + 1}) died with exit code {proc.exitcode}")
            if deadline is not None and time.time() > deadline:
                raise RuntimeError("timed out waiting for the trial workers")
        return out

    def expect(self, tag, timeout=None):
        for message in self._collect(timeout):
            if message[0] != tag:
                raise RuntimeError(f"trial worker protocol error: expected {tag!r}, got {message[0]!r}")

    def close(self, force=False):
        import torch.distributed as dist

        global _ACTIVE_POOL
        if getattr(self, "closed", False):
            return
        self.closed = True
        if _ACTIVE_POOL is self:
            _ACTIVE_POOL = None
        for proc, conn in self.workers:
            try:
                conn.send(("stop",))
            except Exception:
                pass
        if dist.is_initialized():
            # Also when forced: a default process group left behind would make the next attacker of this process believe it
            # is one rank of a (dead) group.  Tearing the group down is a local operation.
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        for proc, conn in self.workers:
            proc.join(timeout=1.0 if force else 20.0)
            if proc.is_alive():
                proc.kill()
            conn.close()

    def __del__(self):
        try:
            self.close(force=True)
        except Exception:
            pass


# ---- the runner the attacker installs in its workers -----------------------------------------------------------------
def attacker_runner_factory(rank, world, device_index, conn, attack_class_name, model, loss_fn, cfg):
    """Build this rank's attacker (own model replica on its own device) once; every job is one ``reconstruct`` call."""
    from . import attacker as attacker_module

    device = torch.device("cuda", device_index)
    att = getattr(attacker_module, attack_class_name)(model, loss_fn, cfg, dict(device=device, dtype=torch.float))
    att._is_trial_worker = True

    def rendezvous():
        conn.send(("trials_done",))
        message = conn.recv()
        if message[0] != "go":
            raise RuntimeError(f"trial worker protocol error: expected 'go', got {message[0]!r}")

    def run(job):
        att._preset = dict(inits=job["inits"], labels=job["labels"])
        att._before_select = rendezvous
        try:
            att.reconstruct(job["server_payload"], job["shared_data"], job["server_secrets"], job["initial_data"], job["dryrun"])
        finally:
            att._preset, att._before_select = None, None

    return run
