"""Kernels B (candidate step) and C (TV + norm prior) timed alone at B = 1 and B = 8 of 3 x 224 x 224, per launch with events.

    python scripts/step_prior_probe.py [--launches 50]                       # the in-tree library
    BREACH_HIP_LIB=build/libbreach_hip_prev.so python scripts/step_prior_probe.py   # a previous build of the same ABI, for A/B on one box

ALGORITHMIC bytes: kernel B 7 N 4 (+1 N 4 when the best copy is taken), kernel C 2 N 4.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from breaching_amd import _lib, schedules  # noqa: E402
from breaching_amd.priors import launch_tv_norm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=50)
args = ap.parse_args()
dev = torch.device("cuda:0")
lib = _lib.load()
st = _lib.current_stream_handle(dev)


def timed(body, launches):
    for _ in range(5):
        body()
    torch.cuda.synchronize()
    pairs = []
    for _ in range(launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        body()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
    return dict(median_us=round(t[len(t) // 2], 2), best_us=round(t[0], 2))


def burst(body, reps=200):
    for _ in range(10):
        body()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        body()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) * 1e3 / reps, 2)


library = os.environ.get("BREACH_HIP_LIB", "in-tree")
for B in (1, 8):
    x = torch.randn(B, 3, 224, 224, device=dev)
    g = torch.empty_like(x)
    parts = torch.empty(_lib.BH_PRIOR_MAX_GRID * 2, dtype=torch.float64, device=dev)
    for opp in (False, True):
        body = lambda: launch_tv_norm(x, 0.2, 1, 1, 1e-8, opp, 1e-6, 2.0, grad_out=g, partials=parts)  # noqa: E731
        row = dict(library=library, kernel="C tv+norm p=q=1", B=B, double_opponents=opp, bytes=2 * x.numel() * 4, **timed(body, args.launches),
                   burst_us=burst(body))
        row["GBps_burst"] = round(row["bytes"] / row["burst_us"] / 1e3, 1)
        print(json.dumps(row), flush=True)
    n = x.numel()
    xs, gs, gr, m, v, best, noise = (torch.randn(n, device=dev) for _ in range(7))
    v.abs_()
    state = torch.zeros(_lib.BH_STATE_WORDS, dtype=torch.int32, device=dev)
    hist = torch.zeros(64, dtype=torch.float32, device=dev)
    sched = torch.from_numpy(schedules.adam_schedule_table([0.1] * 64, 0.9, 0.999)).to(dev)
    for mode, sign, langevin, loss_value in (("hard sign, best copy taken", 1, 0.0, 0.5), ("plain Adam + Langevin noise, no best copy", 0, 0.01, 2.0)):
        P = _lib.StepParams()
        P.n, P.plane, P.channels, P.boxed, P.sign_mode, P.max_iterations = n, 224 * 224, 3, 1, sign, 64
        for c in range(3):
            P.lo[c], P.hi[c] = -2.0, 2.0
        P.beta1, P.beta2, P.eps, P.langevin, P.grad_clip = 0.9, 0.999, 1e-8, langevin, -1.0
        lib.bh_state_reset(_lib.ptr(state), st)
        for value in (1.0, loss_value):  # second commit: improved (0.5 < 1) or not (2 > 1)
            loss = torch.full((1,), value, device=dev)
            lib.bh_loss_commit(_lib.ptr(state), _lib.ptr(hist), 64, _lib.ptr(loss), None, 0, None, None, st)
        copies = 8 if loss_value < 1.0 else 7
        operands = copies + (1 if langevin > 0 else 0)
        body = lambda: lib.bh_candidate_step(_lib.ptr(state), _lib.ptr(sched), P, _lib.ptr(xs), _lib.ptr(gs), _lib.ptr(gr),  # noqa: E731
                                             _lib.ptr(noise) if langevin > 0 else None, _lib.ptr(m), _lib.ptr(v), _lib.ptr(best), st)
        row = dict(library=library, kernel="B candidate step", mode=mode, B=B, bytes=operands * n * 4, **timed(body, args.launches), burst_us=burst(body))
        row["GBps_burst"] = round(row["bytes"] / row["burst_us"] / 1e3, 1)
        print(json.dumps(row), flush=True)
