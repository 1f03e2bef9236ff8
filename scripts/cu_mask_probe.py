"""Do trials in flight run faster when each one owns a slice of the chip?

Four ResNet-18 trials in flight reach 520-545 it/s = 7.4-7.7 ms per round of four, i.e. each trial's iteration takes 1.8x as long
as alone (4.2 ms): their kernels compete for the same 256 CUs and eight L2s.  `hipExtStreamCreateWithCUMask` pins a stream's
kernels to a set of CUs.  This probe gives each trial a quarter of the chip in two ways -- two whole XCDs (CU-mask bit n is
taken to address XCD n % 8: "xcd"), or 64 consecutive mask bits ("block") -- and compares with unmasked streams, for 1 / 2 / 4
trials; it also times ONE trial confined to 64 and 128 CUs (how much of the chip does a batch-1 iteration need?).

    python scripts/cu_mask_probe.py [--rounds 60]   ->  JSON lines (fresh process per arrangement)
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ARRANGEMENTS = [("none", 1), ("none", 2), ("none", 4), ("xcd", 1), ("block", 1), ("xcd", 2), ("xcd", 4), ("block", 4), ("xcd-half", 1), ("xcd-half", 2)]


def masks(kind, trials):
    """One 256-bit CU mask (8 uint32 words) per trial."""
    out = []
    for t in range(trials):
        bits = [0] * 256
        if kind == "xcd":         # two whole XCDs per trial: bits n with n % 8 in {2t, 2t+1}
            for n in range(256):
                if n % 8 in (2 * t % 8, (2 * t + 1) % 8):
                    bits[n] = 1
        elif kind == "xcd-half":  # four whole XCDs per trial
            for n in range(256):
                if (n % 8) // 4 == t % 2:
                    bits[n] = 1
        elif kind == "block":     # 64 consecutive mask bits per trial
            for n in range(64 * (t % 4), 64 * (t % 4) + 64):
                bits[n] = 1
        words = [sum(bits[32 * w + b] << b for b in range(32)) for w in range(8)]
        out.append(words)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=60)
    ap.add_argument("--kind", default=None)
    ap.add_argument("--trials", type=int, default=4)
    args = ap.parse_args()
    if args.kind is None:
        for kind, trials in ARRANGEMENTS:
            proc = subprocess.run([sys.executable, os.path.abspath(__file__), "--kind", kind, "--trials", str(trials), "--rounds", str(args.rounds)],
                                  capture_output=True, text=True, timeout=400)
            line = proc.stdout.strip().splitlines()[-1] if proc.stdout.strip() else json.dumps(dict(kind=kind, trials=trials, error=proc.stderr[-400:]))
            print(line, flush=True)
        return
    import torch

    import breaching_amd
    from breaching_amd.attacker import FusedTrial
    from breaching_amd.cases import build_case, initial_candidate
    from breaching_amd.streams import side_streams

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
    if args.kind == "none":
        streams = side_streams(device, args.trials) if args.trials <= 4 else [torch.cuda.Stream(device) for _ in range(args.trials)]
    else:
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
        hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
        streams = []
        for words in masks(args.kind, args.trials):
            handle = ctypes.c_void_p()
            arr = (ctypes.c_uint32 * 8)(*words)
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), 8, arr)
            assert rc == 0, rc
            streams.append(torch.cuda.ExternalStream(handle.value, device=device))
    runs = []
    for j, stream in enumerate(streams):
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
    print(json.dumps(dict(kind=args.kind, trials=args.trials, rounds=args.rounds, trial_iterations_per_s=round(args.trials * args.rounds / dt, 1),
                          ms_per_round=round(dt / args.rounds * 1e3, 3), mode=runs[0][1].execution_mode())), flush=True)


if __name__ == "__main__":
    main()
