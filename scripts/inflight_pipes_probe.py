"""Why do more than four trials in flight collapse (round 3: 531 it/s with 4, 172-215 with 6 or 8)?

rocprofv3 cannot answer it: with its kernel tracing attached, graph replays on different streams no longer overlap at all
(profiles/r4_4trials_gap_census.txt: one queue busy 94.5 % of the time, 172 it/s instead of 531).  Hypothesis from the process
experiments of round 3 (4 processes x 1 trial = 464 it/s, but 2 x 2 = 168 and 1 x 6 = 215): a HIP stream is a hardware
queue, hardware queues are dealt round-robin onto the 4 compute pipes of an XCD's command processor, and two permanently
busy queues on one pipe take turns instead of overlapping.  Test: create `--total` streams in order, put a trial on the
streams listed in `--use` (1-based creation order) only, and measure trial-iterations/s.  If the hypothesis holds, {1,2,3,4}
and {2,3,4,5} are fast, {1,5} (same pipe) is no faster than one trial while {1,2} overlaps, and any fifth busy stream costs
throughput.

    python scripts/inflight_pipes_probe.py --total 8 --use 1,2,3,4 [--rounds 60]     ->  one JSON line
    python scripts/inflight_pipes_probe.py --sweep                                   ->  one line per arrangement (fresh process each)
"""
import argparse
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SWEEP = ["1", "1,2", "1,5", "1,3", "1,2,3,4", "2,3,4,5", "1,2,3,5", "1,2,3,4,5", "1,2,3,4,5,6", "1,2,3,4,5,6,7,8", "1,3,5,7", "1,2,5,6"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total", type=int, default=8)
    ap.add_argument("--use", default="1,2,3,4")
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--touch-all", action="store_true", help="launch one tiny kernel on every created stream first (forces its queue to exist)")
    args = ap.parse_args()
    if args.sweep:
        for use in SWEEP:
            proc = subprocess.run([sys.executable, os.path.abspath(__file__), "--total", str(args.total), "--use", use, "--rounds",
                                   str(args.rounds), "--touch-all"], capture_output=True, text=True, timeout=400)
            line = proc.stdout.strip().splitlines()[-1] if proc.stdout.strip() else json.dumps(dict(use=use, error=proc.stderr[-300:]))
            print(line, flush=True)
        return
    import torch

    import breaching_amd
    from breaching_amd.attacker import FusedTrial
    from breaching_amd.cases import build_case, initial_candidate

    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    case = build_case("resnet18", "ImageNet", 1, device=device, gradient_device=device)
    cfg = breaching_amd.get_attack_config("invertinggradients", ["impl.hip_graph=required"])
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=device, dtype=torch.float))
    rec_models, labels, _ = attacker.prepare_attack(case.server_payload, case.shared_data)
    attacker.objective.initialize(attacker.loss_fn, cfg.impl, None)
    for reg in attacker.regularizers:
        reg.initialize(rec_models, case.shared_data, labels)
    attacker.objective.prepare(rec_models, case.shared_data)
    main_stream = torch.cuda.current_stream(device)
    streams = [torch.cuda.Stream(device) for _ in range(args.total)]
    if args.touch_all:
        for s in streams:
            with torch.cuda.stream(s):
                torch.zeros(1, device=device).add_(1)
        torch.cuda.synchronize()
    use = [int(tok) for tok in args.use.split(",")]
    runs = []
    for j, idx in enumerate(use):
        stream = streams[idx - 1]
        stream.wait_stream(main_stream)
        x = initial_candidate(case.data_cfg, 1, trial=j).to(device).requires_grad_(True)
        with torch.cuda.stream(stream):
            runs.append((stream, FusedTrial(attacker, [x], labels, rec_models, case.shared_data)))

    def round_():
        for stream, run in runs:
            with torch.cuda.stream(stream):
                run.step()

    for _ in range(12):
        round_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.rounds):
        round_()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(total_streams=args.total, use=use, touch_all=args.touch_all, trials=len(use), rounds=args.rounds,
                          trial_iterations_per_s=round(len(use) * args.rounds / dt, 1), ms_per_round=round(dt / args.rounds * 1e3, 3),
                          mode=runs[0][1].execution_mode(), stream_handles=[hex(s.cuda_stream) for s in streams])), flush=True)


if __name__ == "__main__":
    main()
