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

    def __init__(self, num_trials, rank=0, world=1, group=None):
        self.num_trials, self.rank, self.world, self.group = int(num_trials), int(rank), int(world), group

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
        if self.world == 1:
            best_key = local_key
        else:
            import torch.distributed as dist

            key = torch.tensor([local_key], dtype=torch.int64, device=_collective_device(device, self.group))
            dist.all_reduce(key, op=dist.ReduceOp.MIN, group=self.group)  # the single selection collective
            best_key = int(key.item())
        value, trial = unpack_key(best_key)
        if trial == 0xFFFFFFFF or trial >= self.num_trials:
            # no rank produced a solution (interrupted before the first trial finished)
            raise RuntimeError("No trial finished; nothing to select.")

        if self.world == 1:
            solution = local_solutions[trial]
        else:
            solution = self._broadcast_solution(local_solutions, trial, device)
            if gather_stats:
                self._merge_stats(stats)
        return value, solution

    # -- distributed helpers ---------------------------------------------------------------------------------------
    def _broadcast_solution(self, local_solutions, trial, device):
        """One broadcast: the parts of the winning solution, flattened into one fp32 buffer.  The layout comes from any
        local solution (all trials of an attack have the same shapes).  Only a rank without a single finished trial -- more
        ranks than trials -- needs the shapes sent first; that rare case costs one extra host-side object broadcast, which
        every rank then takes part in (the all-reduced flag below tells them)."""
        import torch.distributed as dist

        src = self.owner(trial)
        cdev = _collective_device(device, self.group)
        template = next(iter(local_solutions.values())) if len(local_solutions) > 0 else None
        if self.world > self.num_trials:  # some rank may hold no template: agree on the shapes the slow way
            meta = [None]
            if self.rank == src:
                sol = local_solutions[trial]
                parts = [sol] if torch.is_tensor(sol) else list(sol)
                meta = [[(tuple(p.shape), str(p.dtype).replace("torch.", "")) for p in parts] + [torch.is_tensor(sol)]]
            dist.broadcast_object_list(meta, src=_global_rank(src, self.group), group=self.group)
            *shapes, is_single = meta[0]
        else:
            parts = [template] if torch.is_tensor(template) else list(template)
            shapes = [(tuple(p.shape), str(p.dtype).replace("torch.", "")) for p in parts]
            is_single = torch.is_tensor(template)
        sizes = [int(torch.Size(shape).numel()) for shape, _ in shapes]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=cdev)
        if self.rank == src:
            sol = local_solutions[trial]
            parts = [sol] if torch.is_tensor(sol) else list(sol)
            torch.cat([p.detach().reshape(-1).to(device=cdev, dtype=torch.float32) for p in parts], out=flat)
        dist.broadcast(flat, src=_global_rank(src, self.group), group=self.group)  # the single winner broadcast
        out, offset = [], 0
        for (shape, dtype), n in zip(shapes, sizes):
            out.append(flat[offset : offset + n].view(shape).to(device=device, dtype=getattr(torch, dtype)))
            offset += n
        return out[0] if is_single else tuple(out)

    def _merge_stats(self, stats):
        import torch.distributed as dist

        mine = {k: v for k, v in stats.items() if k.startswith("Trial_")}
        gathered = [None] * self.world
        dist.all_gather_object(gathered, mine, group=self.group)
        for other in gathered:
            for k, v in other.items():
                if k not in stats or len(stats[k]) == 0:
                    stats[k] = v


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
