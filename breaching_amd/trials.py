"""Trial-parallel restarts: shard ``restarts.num_trials`` over the GPUs of a node, pick the winner with one all-reduce.

reference: the sequential loop ``for trial in range(self.cfg.restarts.num_trials)`` and the argmin selection in
breaching/attacks/optimization_based_attack.py:70-78, :206-218.  The reference has no multi-device code at all
("Parallelism strategies ... none"), so this is an MI355X-native addition:

  * one process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo" in CPU tests),
  * rank r runs trials {t : t mod W == r}; trials share nothing but read-only inputs, so there is no data-path
    collective,
  * selection is ONE ``all_reduce(MIN)`` on a packed int64 key ``(float_bits(score) << 32) | trial`` -- scores are
    >= 0 or +inf, for which IEEE-754 bit patterns order like the floats; NaN is mapped to +inf exactly like
    ``_score_trial`` does (:204) -- followed by ONE broadcast of the winning candidate from its owner (all parts of a
    joint data+label solution travel in one flat buffer; every rank knows the shapes from its own trials).  The per-trial
    loss histories are merged afterwards with a host-side object gather (off the data path, optional).
"""

import struct

import torch

_INF_BITS = 0x7F800000


def score_key(score, trial):
    """Pack a non-negative (or +inf / NaN) fp32 score and a trial index into one ordered int64 key."""
    value = float(score)
    if value != value or value == float("inf"):
        bits = _INF_BITS
    elif value < 0:
        # Negative scores cannot come out of the supported scorings (cosine in [0,2], euclidean >= 0, TV >= 0); order
        # them below every non-negative score while keeping their relative order.
        bits = 0
        return -(((struct.unpack("<I", struct.pack("<f", -value))[0]) << 32) | (0xFFFFFFFF - trial))
    else:
        bits = struct.unpack("<I", struct.pack("<f", value))[0]
    return (bits << 32) | int(trial)


def unpack_key(key):
    """Inverse of :func:`score_key` for non-negative keys: returns (score, trial)."""
    key = int(key)
    if key < 0:
        mag = -key
        trial = 0xFFFFFFFF - (mag & 0xFFFFFFFF)
        return -struct.unpack("<f", struct.pack("<I", mag >> 32))[0], trial
    return struct.unpack("<f", struct.pack("<I", key >> 32))[0], key & 0xFFFFFFFF


class TrialShard:
    """Which trials this process runs, and how the best one is agreed on."""

    def __init__(self, num_trials, rank=0, world=1, group=None, always_collective=False):
        self.num_trials, self.rank, self.world, self.group = int(num_trials), int(rank), int(world), group
        # world 1 normally never touches torch.distributed; `always_collective` sends even a one-rank selection through the
        # all-reduce and the broadcast (how the RCCL code path is exercised on a 1-GPU box, see `dry_collective`)
        self.always_collective = bool(always_collective)

    @classmethod
    def current(cls, num_trials, group=None):
        """Shard over the default process group when one is initialised and larger than one rank."""
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            return cls(num_trials, dist.get_rank(group), dist.get_world_size(group), group)
        return cls(num_trials)

    def local_trials(self):
        return range(self.rank, self.num_trials, self.world)

    def owner(self, trial):
        return trial % self.world

    def select(self, local_solutions, local_scores, stats, device, gather_stats=True):
        """Return (optimal_value: float, optimal_solution) identically on every rank.

        ``local_solutions`` / ``local_scores`` map trial index -> tensor (or tuple of tensors) / score.  ``stats`` gets the
        per-trial loss lists of the other ranks merged in (host-side object gather, small) unless ``gather_stats`` is off.
        """
        if len(local_scores) > 0:
            local_key = min(score_key(_to_float(s), t) for t, s in local_scores.items())
        else:
            local_key = score_key(float("inf"), 0xFFFFFFFF)
        collective = self.world > 1 or self.always_collective
        if collective and torch.device(device).type == "cuda":
            # object collectives (the stats gather, the shapes fallback) place their buffers on the CURRENT device: make that
            # the rank's own GPU for the whole selection, whatever the caller's current device is
            with torch.cuda.device(device):
                return self._select(local_key, local_solutions, stats, device, gather_stats, collective)
        return self._select(local_key, local_solutions, stats, device, gather_stats, collective)

    def _select(self, local_key, local_solutions, stats, device, gather_stats, collective):
        if not collective:
            best_key = local_key
        else:
            import torch.distributed as dist

            # second word: does this rank hold a finished trial whose shapes can serve as the broadcast layout?  (A rank whose
            # trials were all interrupted, or one of more ranks than trials, does not.)
            key = torch.tensor([local_key, 1 if len(local_solutions) > 0 else 0], dtype=torch.int64,
                               device=_collective_device(device, self.group))
            dist.all_reduce(key, op=dist.ReduceOp.MIN, group=self.group)  # the single selection collective
            best_key, every_rank_has_a_template = (int(v) for v in key.tolist())
        value, trial = unpack_key(best_key)
        if trial == 0xFFFFFFFF or trial >= self.num_trials:
            # no rank produced a solution (interrupted before the first trial finished)
            raise RuntimeError("No trial finished; nothing to select.")

        if not collective:
            solution = local_solutions[trial]
        else:
            solution = self._broadcast_solution(local_solutions, trial, device, bool(every_rank_has_a_template))
            if gather_stats:
                self._merge_stats(stats)
        return value, solution

    # -- distributed helpers ---------------------------------------------------------------------------------------
    def _broadcast_solution(self, local_solutions, trial, device, every_rank_has_a_template=True):
        """One broadcast: the parts of the winning solution, flattened into one buffer of their common dtype (fp32 for every
        supported attack; parts of another dtype -- an fp64 set-up, integer label parts -- travel in one more buffer per
        dtype, unrounded).  The layout comes from any local solution (all trials of an attack have the same shapes).  Only
        when some rank holds no finished trial (more ranks than trials, or a rank interrupted before its first trial
        ended: the flag all-reduced together with the key says so on every rank) are the shapes sent first, by one host-side
        object broadcast that every rank then takes part in."""
        import torch.distributed as dist

        src = self.owner(trial)
        cdev = _collective_device(device, self.group)
        if not every_rank_has_a_template:
            meta = [None]
            if self.rank == src:
                sol = local_solutions[trial]
                parts = [sol] if torch.is_tensor(sol) else list(sol)
                meta = [[(tuple(p.shape), str(p.dtype).replace("torch.", "")) for p in parts] + [torch.is_tensor(sol)]]
            dist.broadcast_object_list(meta, src=_global_rank(src, self.group), group=self.group)
            *shapes, is_single = meta[0]
        else:
            template = next(iter(local_solutions.values()))
            parts = [template] if torch.is_tensor(template) else list(template)
            shapes = [(tuple(p.shape), str(p.dtype).replace("torch.", "")) for p in parts]
            is_single = torch.is_tensor(template)
        sizes = [int(torch.Size(shape).numel()) for shape, _ in shapes]
        if self.rank == src:
            sol = local_solutions[trial]
            parts = [sol] if torch.is_tensor(sol) else list(sol)
        out = [None] * len(shapes)
        for dtype_name in dict.fromkeys(dtype for _, dtype in shapes):  # one flat buffer per dtype, in order of appearance
            members = [i for i, (_, dtype) in enumerate(shapes) if dtype == dtype_name]
            dtype = getattr(torch, dtype_name)
            flat = torch.empty(sum(sizes[i] for i in members), dtype=dtype, device=cdev)
            if self.rank == src:
                torch.cat([parts[i].detach().reshape(-1).to(device=cdev, dtype=dtype) for i in members], out=flat)
            dist.broadcast(flat, src=_global_rank(src, self.group), group=self.group)  # the single winner broadcast
            offset = 0
            for i in members:
                out[i] = flat[offset : offset + sizes[i]].view(shapes[i][0]).to(device=device)
                offset += sizes[i]
        return out[0] if is_single else tuple(out)

    def _merge_stats(self, stats):
        import torch.distributed as dist

        mine = {k: v for k, v in stats.items() if k.startswith("Trial_")}
        gathered = [None] * self.world
        dist.all_gather_object(gathered, (mine, stats.get("execution_trials")), group=self.group)
        for other, execution in gathered:
            for k, v in other.items():
                if k not in stats or len(stats[k]) == 0:
                    stats[k] = v
            if execution is not None:  # how each rank issued its trials (hipGraph replay / eager launches)
                stats.setdefault("execution_trials", {}).update(execution)


def _to_float(score):
    if torch.is_tensor(score):
        return float(score.detach().reshape(-1)[0].item())
    return float(score)


def _collective_device(device, group):
    """Tensors for collectives live on the GPU for RCCL and on the host for gloo."""
    import torch.distributed as dist

    backend = dist.get_backend(group)
    return torch.device("cpu") if backend == "gloo" else torch.device(device)


def _global_rank(group_rank, group):
    import torch.distributed as dist

    if group is None:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def dry_collective(device, backend="nccl", timeout_s=120.0):
    """Run the selection's two collectives -- int64 `all_reduce(MIN)` on the packed key, flat broadcast of the winner -- on
    DEVICE tensors through a ONE-rank process group of `backend` ("nccl" = RCCL accepts a one-rank communicator).  This is
    what a 1-GPU box can execute of the multi-GPU path: communicator creation, both collectives and the teardown go through
    RCCL exactly as they will at 2 / 4 / 8 ranks.  Must run in a process without a default process group.  Returns a record
    of what ran."""
    import datetime
    import os
    import socket
    import time

    import torch.distributed as dist

    device = torch.device(device)
    if dist.is_initialized():
        raise RuntimeError("dry_collective needs a process without a default process group")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
    kwargs = dict(device_id=device) if backend == "nccl" else {}
    t0 = time.perf_counter()
    dist.init_process_group(backend, rank=0, world_size=1, timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
    try:
        shard = TrialShard(3, rank=0, world=1, always_collective=True)
        solutions = {t: (torch.full((1, 3, 8, 8), float(t), device=device), torch.full((1, 5), -float(t), device=device))
                     for t in range(3)}
        scores = {0: torch.tensor(0.75, device=device), 1: torch.tensor(0.25, device=device), 2: float("nan")}
        stats = {f"Trial_{t}_Val": [float(t)] for t in range(3)}
        t1 = time.perf_counter()
        value, solution = shard.select(solutions, scores, stats, device)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        ok = (value == 0.25 and solution[0].device.type == device.type and float(solution[0].flatten()[0]) == 1.0
              and float(solution[1].flatten()[0]) == -1.0 and sorted(stats) == [f"Trial_{t}_Val" for t in range(3)])
        # the collective of `TrialWorkerPool.ship` (job inputs by broadcast): one flat buffer per dtype on the device, packed and
        # unpacked with the pool's own layout code -- mixed dtypes, a ragged and an empty tensor
        from . import workers

        tree = dict(gradients=[torch.arange(7, dtype=torch.float32, device=device), torch.zeros(0, device=device),
                               torch.arange(6, dtype=torch.float32, device=device).view(2, 3)], labels=torch.tensor([3, 1], device=device))
        skeleton, tensors = workers.split_tensors(tree)
        specs = [(tuple(t.shape), str(t.dtype).replace("torch.", "")) for t in tensors]
        received = [None] * len(specs)
        for name, (total, members) in workers.shipment_layout(specs).items():
            flat = torch.zeros(total, dtype=getattr(torch, name), device=device)
            for index, offset, numel in members:
                flat[offset : offset + numel].copy_(tensors[index].reshape(-1))
            dist.broadcast(flat, src=0)
            for index, offset, numel in members:
                received[index] = flat[offset : offset + numel].view(specs[index][0])
        back = workers.join_tensors(skeleton, received)
        ship_ok = all(torch.equal(a, b) and a.device == b.device for a, b in zip(back["gradients"] + [back["labels"]], tree["gradients"] + [tree["labels"]]))
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        return dict(backend=dist.get_backend(), world=dist.get_world_size(), device=str(device), ok=bool(ok and ship_ok), value=value,
                    ship_ok=bool(ship_ok), init_s=round(t1 - t0, 3), select_s=round(t2 - t1, 3))
    finally:
        dist.destroy_process_group()


def parse_dry_collective(stdout):
    """The record printed by ``python -m breaching_amd.trials --dry-collective`` out of a captured stdout (None if absent)."""
    import json

    for line in reversed(stdout.splitlines()):
        if line.startswith("DRY_COLLECTIVE "):
            return json.loads(line[len("DRY_COLLECTIVE "):])
    return None


if __name__ == "__main__":  # python -m breaching_amd.trials --dry-collective [nccl|gloo] [device]
    import json
    import sys

    if len(sys.argv) >= 2 and sys.argv[1] == "--dry-collective":
        backend = sys.argv[2] if len(sys.argv) > 2 else "nccl"
        device = sys.argv[3] if len(sys.argv) > 3 else ("cuda:0" if backend == "nccl" else "cpu")
        # one marked line: RCCL / gloo write their own chatter to stdout around it
        print("DRY_COLLECTIVE " + json.dumps(dry_collective(device, backend)), flush=True)
