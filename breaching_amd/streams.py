"""Side streams for trials in flight, chosen so that no two of them share a hardware compute pipe.

Measured on MI355X (profiles/r4_inflight_pipes_probe.jsonl, scripts/inflight_pipes_probe.py): HIP streams map to hardware
queues, and the queues are dealt round-robin onto FOUR compute pipes of the command processor.  Two streams that replay
hipGraphs at the same time on ONE pipe do not overlap -- they run far slower than one after the other (two ResNet-18 trials
on the 1st and 5th stream created: 80 iterations/s; on the 1st and 2nd: 356; one trial alone: 228) -- and as soon as any two
busy streams collide the whole round costs ~24 ms whatever the number of trials (4 trials on streams 1-4: 526 it/s, on
1,2,3,5: 173; 5 / 6 / 8 trials: 216 / 248 / 277).  That is the "collapse beyond four trials in flight" of round 3, and it
is why `MAX_TRIALS_IN_FLIGHT` is 4: a fifth busy stream necessarily shares a pipe.

Which pipe a stream lands on depends on every stream the process created before it, so taking "the next four streams of
torch's pool" is only right by luck.  `side_streams(device, n)` therefore picks its streams by MEASUREMENT, once per process
and device: candidates are probed pairwise with two tiny captured graphs (64 dependent one-element kernels each, replayed
concurrently; a clean pair takes ~1.2x the time of one graph alone, a colliding pair ~3.1x), and the first `n` candidates that collide with none of
the others are kept and reused by every later group of trials.  Cost: a few milliseconds, outside any timed loop.

reference: none (the reference runs its restarts one after the other, optimization_based_attack.py:70-78).
"""

import logging
import os

import torch

log = logging.getLogger(__name__)

PIPES = 4                 # concurrently busy streams that can each have a compute pipe to themselves
_CANDIDATES = 12          # streams looked at before giving up on finding PIPES clean ones
_CHAIN, _REPLAYS = 64, 6  # probe graph: dependent one-element kernels per replay, replays per measurement
_COLLISION_FACTOR = 1.9   # pair time / solo time above which two streams are taken to share a pipe (measured: 1.2 clean, 3.1 colliding)

_CHOSEN = {}              # device index -> (list of streams, report dict)


def _probe_graphs(device, stream):
    graphs = []
    for _ in range(2):
        cell = torch.zeros(1, device=device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for _ in range(_CHAIN):
                cell.add_(1.0)
        graphs.append((graph, cell))
    return graphs


def _timed(device, work):
    """Milliseconds from now until every stream of `work` = [(stream, graph), ...] has finished _REPLAYS replays."""
    torch.cuda.synchronize(device)
    start = torch.cuda.Event(enable_timing=True)
    ends = []
    start.record(torch.cuda.current_stream(device))
    for stream, _ in work:
        stream.wait_event(start)
    for _ in range(_REPLAYS):
        for stream, graph in work:
            with torch.cuda.stream(stream):
                graph.replay()
    for stream, _ in work:
        end = torch.cuda.Event(enable_timing=True)
        end.record(stream)
        ends.append(end)
    torch.cuda.synchronize(device)
    return max(start.elapsed_time(end) for end in ends)


def calibrate(device, n=PIPES, candidates=None):
    """Find `n` streams on pairwise different pipes.  Returns (streams, report).  `candidates`: an iterator of streams to look
    at instead of fresh ones from torch's pool (tests feed a deliberately scrambled order)."""
    device = torch.device(device)
    n = max(1, min(int(n), PIPES))
    report = dict(method="pairwise concurrent replay of two 64-node probe graphs", candidates=0, collisions=[], solo_ms=None)
    supply = iter(candidates) if candidates is not None else None

    def fresh():
        if supply is not None:
            return next(supply, None) or torch.cuda.Stream(device)
        return torch.cuda.Stream(device)

    with torch.cuda.device(device):
        first = fresh()
        chosen = [first]
        (graph_a, _), (graph_b, _) = _probe_graphs(device, first)
        _timed(device, [(first, graph_a)])  # warm-up
        solo = min(_timed(device, [(first, graph_a)]) for _ in range(3))
        report["solo_ms"] = round(solo, 4)
        looked_at = 1
        while len(chosen) < n and looked_at < _CANDIDATES:
            candidate = fresh()
            looked_at += 1
            clash = None
            for idx, kept in enumerate(chosen):
                pair = min(_timed(device, [(kept, graph_a), (candidate, graph_b)]) for _ in range(2))
                if pair > _COLLISION_FACTOR * solo:
                    clash = (idx, round(pair, 4))
                    break
            if clash is None:
                chosen.append(candidate)
            else:
                report["collisions"].append(dict(candidate=looked_at, with_chosen=clash[0], pair_ms=clash[1]))
        report["candidates"] = looked_at
        while len(chosen) < n:  # not enough clean streams found (a profiler serialising everything, a busy GPU): take what comes
            chosen.append(fresh())
            report["incomplete"] = True
    if report["collisions"] or report.get("incomplete"):
        log.info(f"Trial streams on {device}: {len(chosen)} chosen out of {report['candidates']} candidates; {report}")
    return chosen, report


def side_streams(device, n):
    """`n` (<= 4) streams for trials in flight on `device`, on pairwise different hardware pipes; cached per process and device.
    BREACH_HIP_STREAM_CALIBRATION=0 skips the measurement and takes the next streams of torch's pool (round 3's behaviour)."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    n = int(n)
    if n > PIPES or os.environ.get("BREACH_HIP_STREAM_CALIBRATION", "1") == "0":
        return [torch.cuda.Stream(device) for _ in range(n)]
    entry = _CHOSEN.get(index)
    if entry is None or len(entry[0]) < n:
        # The probe captures and replays two small hipGraphs.  Where that cannot work (capture unsupported, a profiler that refuses
        # it) or its timing cannot be trusted (another tenant on the GPU, ranks sharing one device and calibrating at the same time:
        # `incomplete`), the attack must still run: fall back to plain pool streams -- round 3's behaviour, at worst slower --
        # and say so (calibration_report() -> stats["execution"]).
        try:
            streams, report = calibrate(torch.device("cuda", index), PIPES)
        except Exception as exc:  # noqa: BLE001 -- any failure of the measurement is a reason to fall back, never to abort the attack
            streams, report = None, dict(method="fallback: next streams of torch's pool", failed=repr(exc)[:300])
            log.warning(f"Trial-stream calibration on cuda:{index} failed ({exc!r}); using plain pool streams.")
        if streams is not None and report.get("incomplete"):
            log.warning(f"Trial-stream calibration on cuda:{index} found fewer than {PIPES} collision-free streams ({report}); "
                        "using plain pool streams.")
            report = dict(report, method="fallback: next streams of torch's pool (calibration incomplete)")
            streams = None
        if streams is None:
            with torch.cuda.device(index):
                streams = [torch.cuda.Stream(torch.device("cuda", index)) for _ in range(PIPES)]
        entry = _CHOSEN[index] = (streams, report)
    return list(entry[0][:n])


def calibration_report(device):
    """What the measurement found for `device` (None before the first group of trials)."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    entry = _CHOSEN.get(index)
    return None if entry is None else dict(entry[1])
