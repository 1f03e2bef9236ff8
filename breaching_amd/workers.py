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

Job inputs do NOT travel over the pipes (round 6): the model parameters, the observed gradients, buffers, labels and starting points
of a call are broadcast from rank 0 over the communicator the pool already owns (`TrialWorkerPool.ship`: one flat buffer per
dtype, device to device over xGMI with "nccl"); the pipe carries the skeleton of the containers -- shapes, dtypes, metadata.
reference: what has to arrive is what the server / user hand to `reconstruct`, breaching/cases/servers.py:138-147 and
breaching/cases/users.py:176-183.

Protocol (one duplex pipe per worker):  parent -> worker ``("ship", key, skeleton, specs)`` | ``("ship_go",)`` | ``("job", dict)`` |
``("go",)`` | ``("abort",)`` | ``("stop",)``;
worker -> parent ``("ship_ready",)`` once the receive buffers of a shipment are allocated (the parent answers ``ship_go`` when
every worker said so and all ranks enter the broadcast together; a failure on any rank turns into ``abort`` instead),
``("booted",)`` as the first statement of the child (its arguments unpickled, i.e. the victim model class is
importable there), ``("ready",)`` once its process group is up, ``("trials_done",)`` when its trials are finished -- it then
waits for ``("go",)`` from the parent before entering the selection collective -- ``("ok",)`` when the job is finished,
``("error", traceback)`` on failure, ``("aborted",)`` in answer to ``abort``.  The parent sends ``go`` only after every
worker reported ``trials_done``: all ranks enter the collective within milliseconds of each other (no rank sits in RCCL while
another still optimises, so a short collective timeout is safe), and a crashed worker raises in the parent instead of
hanging it.  When a job fails anywhere between ``submit`` and the last ``ok`` -- a worker error, or rank 0's own trials
raising -- the parent calls ``abort()``: every worker leaves the job (the ones waiting for ``go`` raise `JobAborted`
instead of entering a collective nobody else will join), acknowledges, and the pool is clean for the next call; workers that
do not acknowledge within ``drain_timeout`` are killed and the pool is closed (the next ``reconstruct`` starts a new one).
"""

import logging
import os
import socket
import traceback
import weakref

import torch

log = logging.getLogger(__name__)

# Collectives are entered only after every rank reported `trials_done` (see the protocol above), so a rank never waits in
# one for longer than the others need to get there: seconds.
DEFAULT_COLLECTIVE_TIMEOUT = 180.0
# Start-up is a different matter: spawn, `import torch` on a cold box (1-2 minutes for the first process of a fresh
# container), RCCL communicator creation over up to 8 GPUs and building the attacker in every worker.  Its own, longer limit
# (BREACH_HIP_POOL_START_TIMEOUT); exceeding it raises in `TrialWorkerPool.__init__` and the attacker falls back to one GPU
# (or raises with impl.trial_pool=required).
DEFAULT_START_TIMEOUT = 600.0

_ACTIVE_POOL = None  # weak reference: a pool dies with the attacker that owns it


class JobAborted(Exception):
    """Raised inside a worker's job when the parent cancels it at the rendezvous."""


def active_pool():
    """The worker pool that owns this process's default process group, or None.  While a pool is idle (no job submitted),
    `torch.distributed.is_initialized()` is true although nobody else will join a collective -- callers that did not submit
    work to the pool must not shard over that group."""
    pool = _ACTIVE_POOL() if _ACTIVE_POOL is not None else None
    return pool if pool is not None and not getattr(pool, "closed", True) else None


def rendezvous(conn):
    """Worker side of the barrier in front of the selection collective: report `trials_done`, wait for `go`."""
    conn.send(("trials_done",))
    message = conn.recv()
    if message[0] == "abort":
        raise JobAborted()
    if message[0] != "go":
        raise RuntimeError(f"trial worker protocol error: expected 'go', got {message[0]!r}")


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# ---- shipping job inputs over the process group instead of the pipes ---------------------------------------------------
SHIP_ALIGN_BYTES = 512  # every tensor of a shipment starts on a 512-byte boundary of its flat buffer (the caching allocator's grain)


class _Slot:
    """Placeholder of tensor number `index` in the skeleton of a shipment."""

    __slots__ = ("index",)

    def __init__(self, index):
        self.index = index

    def __reduce__(self):
        return (_Slot, (self.index,))


def split_tensors(obj, tensors=None, seen=None):
    """(skeleton, tensors): nested lists / tuples / dicts rebuilt with every tensor replaced by a `_Slot`; a tensor object that
    occurs twice (public buffers named by the payload and by the user update) is shipped once."""
    if tensors is None:
        tensors, seen = [], {}
    if torch.is_tensor(obj):
        index = seen.get(id(obj))
        if index is None:
            index = seen[id(obj)] = len(tensors)
            tensors.append(obj)
        return _Slot(index), tensors
    if isinstance(obj, dict):
        return type(obj)((k, split_tensors(v, tensors, seen)[0]) for k, v in obj.items()), tensors
    if isinstance(obj, (list, tuple)):
        return type(obj)(split_tensors(v, tensors, seen)[0] for v in obj), tensors
    return obj, tensors


def join_tensors(skeleton, tensors):
    """Inverse of `split_tensors`."""
    if isinstance(skeleton, _Slot):
        return tensors[skeleton.index]
    if isinstance(skeleton, dict):
        return type(skeleton)((k, join_tensors(v, tensors)) for k, v in skeleton.items())
    if isinstance(skeleton, (list, tuple)):
        return type(skeleton)(join_tensors(v, tensors) for v in skeleton)
    return skeleton


def shipment_layout(specs):
    """Flat-buffer layout of a shipment: {dtype name: (total elements, [(tensor index, element offset, numel), ...])}, one
    buffer per dtype in order of first appearance, every tensor on a SHIP_ALIGN_BYTES boundary."""
    layout = {}
    for index, (shape, dtype_name) in enumerate(specs):
        grain = max(SHIP_ALIGN_BYTES // torch.empty(0, dtype=getattr(torch, dtype_name)).element_size(), 1)
        total, members = layout.get(dtype_name, (0, []))
        numel = int(torch.Size(shape).numel())
        members.append((index, total, numel))
        layout[dtype_name] = ((total + numel + grain - 1) // grain * grain, members)
    return layout


def _collective_device(backend, device):
    return torch.device("cpu") if backend == "gloo" or device is None else torch.device(device)


def receive_shipment(conn, skeleton, specs, backend, device):
    """Worker side of `TrialWorkerPool.ship`: allocate the flat receive buffers, report `ship_ready`, wait for `ship_go` (or
    `abort`), join the broadcasts, rebuild the containers around views of the buffers (moved to `device` when the collective
    ran on the host: gloo)."""
    import torch.distributed as dist

    cdev = _collective_device(backend, device)
    layout = shipment_layout(specs)
    flats = {name: torch.empty(total, dtype=getattr(torch, name), device=cdev) for name, (total, _) in layout.items()}
    conn.send(("ship_ready",))
    message = conn.recv()
    if message[0] == "abort":
        raise JobAborted()
    if message[0] != "ship_go":
        raise RuntimeError(f"trial worker protocol error: expected 'ship_go', got {message[0]!r}")
    tensors = [None] * len(specs)
    for name, (total, members) in layout.items():
        if total > 0:
            dist.broadcast(flats[name], src=0)
        for index, offset, numel in members:
            view = flats[name][offset : offset + numel].view(specs[index][0])
            tensors[index] = view if device is None or cdev == torch.device(device) else view.to(device)
    return join_tensors(skeleton, tensors)


def default_miopen_user_db():
    """Where MIOpen keeps the user find-db when nobody says otherwise."""
    return os.path.join(os.path.expanduser("~"), ".config", "miopen")


def seed_miopen_user_db(source, target):
    """Copy the files of one MIOpen user database directory into another (top level only: the per-rank directories of this
    module live INSIDE the default one and must not be copied into each other).  Returns the number of files copied."""
    import shutil

    copied = 0
    try:
        for name in os.listdir(source):
            src = os.path.join(source, name)
            if os.path.isfile(src):
                shutil.copy2(src, os.path.join(target, name))
                copied += 1
    except OSError:
        pass
    return copied


def isolate_miopen_user_db(rank, seed_from=None):
    """Give this rank its own MIOpen user database directory unless the user chose one (MIOPEN_USER_DB_PATH set) or switched
    the isolation off (BREACH_HIP_MIOPEN_ISOLATE=0).  On a fresh box every process runs MIOpen's solver search for each
    convolution configuration on first use and records the result in the user find-db; eight first processes sharing one sqlite
    file serialise on its lock and time their candidate solvers against each other (profiles/r3_miopen_selection.txt: the
    search doubles the dispatches of the first iterations).  Must run before the process's first convolution.

    The directory is ~/.config/miopen/breach_hip_rank<r>.  It starts as a COPY of `seed_from` (default: the user's existing
    find-db, ~/.config/miopen) when that holds anything, so an already tuned database is not thrown away and ranks that start from
    the same database pick the same solvers; results of later searches stay in the rank's own directory."""
    if "MIOPEN_USER_DB_PATH" in os.environ:
        return os.environ["MIOPEN_USER_DB_PATH"]
    if os.environ.get("BREACH_HIP_MIOPEN_ISOLATE", "1").strip().lower() in ("0", "false", "off", "no"):
        return None
    path = os.path.join(default_miopen_user_db(), f"breach_hip_rank{int(rank)}")
    try:
        os.makedirs(path, exist_ok=True)
    except OSError:
        return None
    source = seed_from if seed_from is not None else default_miopen_user_db()
    if os.path.isdir(source) and os.path.abspath(source) != os.path.abspath(path) and not os.listdir(path):
        seed_miopen_user_db(source, path)
    os.environ["MIOPEN_USER_DB_PATH"] = path
    return path


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


def _worker_main(rank, world, port, backend, device_index, conn, runner_factory, factory_args, timeout):
    """Entry point of rank `rank` > 0."""
    import datetime

    import torch.distributed as dist

    try:
        conn.send(("booted",))  # the spawn bootstrap unpickled our arguments: the parent may start its own rendezvous
        isolate_miopen_user_db(rank)
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        if device_index is not None:
            torch.cuda.set_device(device_index)
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", device_index)
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout), **kwargs)
        runner = runner_factory(rank, world, device_index, conn, *factory_args)
        conn.send(("ready",))
        device = None if device_index is None else torch.device("cuda", device_index)
        shipped = {}  # what `TrialWorkerPool.ship` delivered for the job that follows
        while True:
            message = conn.recv()
            if message[0] == "stop":
                break
            if message[0] == "abort":  # the job this refers to is already over here (finished or failed): acknowledge
                shipped = {}
                conn.send(("aborted",))
                continue
            if message[0] == "ship":
                try:
                    shipped[message[1]] = receive_shipment(conn, message[2], message[3], backend, device)
                except JobAborted:
                    shipped = {}
                    conn.send(("aborted",))
                except Exception:
                    shipped = {}
                    conn.send(("error", traceback.format_exc()))
                continue
            if message[0] != "job":  # e.g. a `go` that crossed an error report
                continue
            try:
                job, shipped = dict(message[1], **shipped), {}
                runner(job)
                del job
                conn.send(("ok",))
            except JobAborted:
                conn.send(("aborted",))
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

    def __init__(self, devices, runner_factory, factory_args=(), backend=None, start_timeout=None, collective_timeout=None,
                 drain_timeout=15.0):
        import datetime

        import time

        import torch.distributed as dist
        import torch.multiprocessing as mp

        if dist.is_initialized():
            raise RuntimeError("TrialWorkerPool needs to own the default process group of the calling process.")
        t_start = time.perf_counter()
        self.timing = {}  # pool_start_s here; job_ship_s / trials_wait_s / select_s of the latest job (set by the attacker)
        self.devices = list(devices)
        self.world = len(self.devices)
        if self.world < 2:
            raise ValueError("A worker pool needs at least two ranks.")
        if backend is None:
            on_gpu = all(d is not None for d in self.devices)
            backend = "nccl" if on_gpu and len(set(self.devices)) == self.world else "gloo"
        self.backend = backend
        if collective_timeout is None:
            collective_timeout = float(os.environ.get("BREACH_HIP_COLLECTIVE_TIMEOUT", DEFAULT_COLLECTIVE_TIMEOUT))
        self.collective_timeout = float(collective_timeout)
        if start_timeout is None:
            start_timeout = float(os.environ.get("BREACH_HIP_POOL_START_TIMEOUT", DEFAULT_START_TIMEOUT))
        self.start_timeout = start_timeout = float(start_timeout)
        self.drain_timeout = float(drain_timeout)
        self.busy = False  # a job is in flight (between submit and the last `ok`)
        self.port = free_port()
        ctx = mp.get_context("spawn")
        self.workers = []
        for rank in range(1, self.world):
            parent_end, child_end = ctx.Pipe(duplex=True)
            proc = ctx.Process(target=_worker_main, daemon=True,
                               args=(rank, self.world, self.port, backend, self.devices[rank], child_end, runner_factory,
                                     factory_args, self.collective_timeout))
            proc.start()
            child_end.close()
            self.workers.append((proc, parent_end))
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(self.port)
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        # A process started by torch.distributed.run (a rank that left its group, or a tool run under torchrun) carries
        # TORCHELASTIC_USE_AGENT_STORE=True: rank 0 would then CONNECT to the elastic agent's store instead of hosting this pool's own
        # (round 6: the pool leg of `bench.py --gpus N` timed out in the rendezvous).  This group is the pool's, not the agent's.
        os.environ.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", self.devices[0])
        try:
            # A child that cannot even start (its arguments do not unpickle there: an unimportable victim model class) dies
            # in the spawn bootstrap; `_collect` polls `is_alive`, so that is noticed here in a fraction of a second and
            # not after the rendezvous timeout below.
            self.expect("booted", start_timeout)
            # the group's own timeout is the COLLECTIVE limit on every rank (it bounds the store rendezvous and RCCL's eager
            # communicator creation too: by now every worker has imported torch and is at the same call)
            dist.init_process_group(backend, rank=0, world_size=self.world,
                                    timeout=datetime.timedelta(seconds=self.collective_timeout), **kwargs)
            for message in self._collect(start_timeout):
                if message[0] != "ready":
                    raise RuntimeError(f"worker failed to start:\n{message[-1]}")
        except Exception:
            self.close(force=True)
            raise
        self.closed = False
        global _ACTIVE_POOL
        _ACTIVE_POOL = weakref.ref(self)
        self.timing["pool_start_s"] = round(time.perf_counter() - t_start, 3)
        log.info(f"Trial worker pool up in {self.timing['pool_start_s']:.1f} s: {self.world} ranks on devices {self.devices} "
                 f"({backend}, collective timeout {self.collective_timeout:.0f} s).")

    def describe(self):
        """What a caller gets to see in `stats["execution"]["pool"]`: the group, and where the wall time outside the trials
        went -- `pool_start_s` (spawn + imports + communicator + worker attackers; paid by the first call only),
        `job_ship_s` (inputs and starting points: skeletons down the pipes, `job_pipe_bytes` each, tensor contents in
        broadcasts over the group, `job_ship_bytes`), `trials_wait_s` (rank 0 waiting for the slowest
        worker after its own trials), `select_s` (all-reduce + broadcast + history gather)."""
        return dict(backend=self.backend, world=self.world, devices=list(self.devices), **self.timing)

    # -- messaging -------------------------------------------------------------------------------------------------
    def submit(self, jobs):
        """`jobs[r - 1]` goes to rank r."""
        import time

        t0 = time.perf_counter()
        self.busy = True
        for (proc, conn), job in zip(self.workers, jobs):
            conn.send(("job", job))
        self.timing["job_ship_s"] = round(self.timing.get("job_ship_s", 0.0) + time.perf_counter() - t0, 4)

    def ship(self, key, tree, device=None):
        """Deliver `tree` (nested lists / tuples / dicts of tensors and small picklable leaves) to every worker as
        ``job[key]`` of the job submitted next.  The pipes carry the skeleton (shapes, dtypes, metadata); the tensor contents
        travel in ONE broadcast per dtype from rank 0 over the pool's process group -- device to device (RCCL over xGMI) with
        "nccl", host buffers with "gloo" (ranks sharing a GPU: the functional path of a 1-GPU box).  Every rank enters the
        broadcast only after ALL workers allocated their receive buffers (`ship_ready` -> `ship_go`); a worker that died or
        failed raises here, before anyone waits in a collective.  Adds to `timing["job_ship_s"]`; `job_ship_bytes` /
        `job_pipe_bytes` count what went over the group and over each pipe."""
        import pickle
        import time

        import torch.distributed as dist

        t0 = time.perf_counter()
        self.busy = True
        skeleton, tensors = split_tensors(tree)
        specs = [(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in tensors]
        cdev = _collective_device(self.backend, device)
        message = ("ship", key, skeleton, specs)
        self.timing["job_pipe_bytes"] = self.timing.get("job_pipe_bytes", 0) + len(pickle.dumps(message))
        for proc, conn in self.workers:
            conn.send(message)
        self.expect("ship_ready")
        self.broadcast(("ship_go",))
        for name, (total, members) in shipment_layout(specs).items():
            if total == 0:
                continue
            flat = torch.empty(total, dtype=getattr(torch, name), device=cdev)
            for index, offset, numel in members:
                flat[offset : offset + numel].copy_(tensors[index].detach().reshape(-1))
            dist.broadcast(flat, src=0)
            self.timing["job_ship_bytes"] = self.timing.get("job_ship_bytes", 0) + flat.numel() * flat.element_size()
        if cdev.type == "cuda":
            torch.cuda.synchronize(cdev)
        self.timing["job_ship_s"] = round(self.timing.get("job_ship_s", 0.0) + time.perf_counter() - t0, 4)

    def begin_job(self):
        """Reset the per-job timing record (everything but `pool_start_s`)."""
        self.timing = {k: v for k, v in self.timing.items() if k == "pool_start_s"}

    def finish(self):
        """Every worker reported `ok`: the job is over."""
        self.expect("ok")
        self.busy = False

    def abort(self):
        """Cancel the job in flight after a failure on any rank: workers waiting for `go` leave the job instead of entering
        the collective, every worker acknowledges, and whatever else they had sent for that job (`trials_done`, `error`,
        `ok`) is discarded -- so the next `submit` starts from a clean pipe.  Workers that do not acknowledge within
        `drain_timeout` (still optimising, or stuck in a collective rank 0 never joined) are killed with the pool."""
        import time

        if getattr(self, "closed", True):
            return
        deadline = time.time() + self.drain_timeout
        try:
            for proc, conn in self.workers:
                conn.send(("abort",))
            for idx, (proc, conn) in enumerate(self.workers):
                while True:
                    remaining = deadline - time.time()
                    if remaining <= 0 or not proc.is_alive():
                        raise TimeoutError(f"trial worker (rank {idx + 1}) did not acknowledge the abort")
                    if conn.poll(min(remaining, 0.2)) and conn.recv()[0] == "aborted":  # EOFError: handled below
                        break
            self.busy = False
        except Exception as exc:
            log.warning(f"Trial worker pool could not be drained after a failed job ({exc!r}); closing it.")
            self.close(force=True)

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
                    try:
                        message = conn.recv()
                    except (EOFError, ConnectionError):  # the child closed its end: it is gone
                        proc.join(timeout=5.0)
                        raise RuntimeError(f"trial worker (rank {idx + 1}) died with exit code {proc.exitcode}") from None
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
        self.busy = False
        if _ACTIVE_POOL is not None and _ACTIVE_POOL() in (self, None):
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

    def run(job):
        # `inputs` and `starts` arrived through `TrialWorkerPool.ship`: tensors already on this rank's device (the reference
        # only casts gradients and buffers, base_attack.py:214-220, because its caller's tensors live on the attack device;
        # e.g. the FedAvg labels in metadata.local_hyperparams are used as they are), containers private to this job.
        inputs, starts = job["inputs"], job["starts"]
        mine = {t: state for t, state in starts["inits"].items() if t % world == rank}
        att._preset = dict(inits=mine, labels=starts["labels"])
        att._before_select = lambda: rendezvous(conn)
        try:
            att.reconstruct(inputs["server_payload"], inputs["shared_data"], inputs["server_secrets"], inputs["initial_data"],
                            job["dryrun"])
        finally:
            att._preset, att._before_select = None, None

    return run
