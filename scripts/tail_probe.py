"""The attack-side tail of an iteration -- kernel C (TV value + gradient) -> loss commit -> kernel B (step) -- as a replayed hipGraph chain,
with and without the commit launch ("fuse the B = 8 tail or show why it is the floor").

B's best copy depends on the commit's `improved` flag and the commit on the grid-wide sum of C's partials, so a fused form needs a
grid-wide dependency inside one launch.  This probe prices the PRIZE: N repetitions of [C, commit, B] against N of [C, B] (the commit
simply left out: what a free commit would cost) at B = 1 and B = 8 (3 x 224 x 224 images), captured once, replayed R times, event-timed,
no profiler.  The difference is the most any fusion of the commit could save per iteration.

    python scripts/tail_probe.py [--nodes 200] [--replays 20]  ->  JSON lines
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--replays", type=int, default=20)
    args = ap.parse_args()
    from breaching_amd import _lib, schedules
    from breaching_amd.priors import launch_tv_norm

    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for B in (1, 8):
        x = torch.randn(B, 3, 224, 224, device=dev)
        g, m, v, best = torch.randn_like(x), torch.zeros_like(x), torch.zeros_like(x), x.clone()
        prior_grad = torch.empty_like(x)
        partials = torch.empty(_lib.BH_PRIOR_MAX_GRID * _lib.BH_PRIOR_PARTIAL_STRIDE, dtype=torch.float64, device=dev)
        state = torch.zeros(_lib.BH_STATE_WORDS, dtype=torch.int32, device=dev)
        iters = args.nodes * (args.replays + 8) + 8
        history = torch.zeros(iters, dtype=torch.float32, device=dev)
        table = schedules.adam_schedule_table(schedules.lr_sequence(0.1, "step-lr", 0, iters), 0.9, 0.999, 0.0)
        sched = torch.from_numpy(table).to(dev)
        loss = torch.ones(1, device=dev)
        P = _lib.StepParams()
        P.n, P.plane, P.channels, P.boxed, P.sign_mode, P.max_iterations = x.numel(), 224 * 224, 3, 1, 1, iters
        for c in range(3):
            P.lo[c], P.hi[c] = -2.0, 2.0
        P.beta1, P.beta2, P.eps, P.decoupled_wd, P.langevin, P.grad_clip = 0.9, 0.999, 1e-8, 0, 0.0, -1.0

        def chain(n, with_commit):
            stream = _lib.current_stream_handle(dev)
            for _ in range(n):
                _, _, grid = launch_tv_norm(x, 0.2, 1, 1, 1e-8, False, grad_out=prior_grad, partials=partials)
                if with_commit:
                    _lib.check(lib.bh_loss_commit(_lib.ptr(state), _lib.ptr(history), iters, _lib.ptr(loss), _lib.ptr(partials),
                                                  grid * _lib.BH_PRIOR_PARTIAL_STRIDE, None, None, stream), "commit")
                _lib.check(lib.bh_candidate_step(_lib.ptr(state), _lib.ptr(sched), P, _lib.ptr(x), _lib.ptr(g), _lib.ptr(prior_grad), None,
                                                 _lib.ptr(m), _lib.ptr(v), _lib.ptr(best), stream), "step")

        out = dict(batch=B, nodes=args.nodes, replays=args.replays)
        for tag, with_commit in (("C_commit_B", True), ("C_B", False)):
            stream = torch.cuda.Stream(dev)
            stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(stream):
                _lib.check(lib.bh_state_reset(_lib.ptr(state), _lib.current_stream_handle(dev)), "reset")
                chain(4, True)  # the state record needs a committed iteration before a step may read it
                torch.cuda.synchronize(dev)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=stream):
                    chain(args.nodes, with_commit)
                for _ in range(3):
                    graph.replay()
                torch.cuda.synchronize(dev)
                start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record()
                for _ in range(args.replays):
                    graph.replay()
                stop.record()
                torch.cuda.synchronize(dev)
            out[tag + "_us"] = round(start.elapsed_time(stop) * 1e3 / (args.replays * args.nodes), 3)
        out["commit_launch_costs_us"] = round(out["C_commit_B_us"] - out["C_B_us"], 3)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
