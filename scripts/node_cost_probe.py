"""What does ONE node of a replayed hipGraph cost on this box, without a profiler attached?

The attack iteration of BASELINE configs[1] replays ~620 dependent kernel nodes in 4.4 ms = 7.1 us per node, against the
1.45-1.9 us dependent-kernel boundary of MI355X_MICROARCH.md.  This probe captures chains of N dependent nodes of one kind on
one stream and times R replays with events (no rocprofv3: its per-dispatch bookkeeping inflates every duration by 2-4 us):

  * ATen in-place elementwise on 1 / 64 Ki / 784 Ki fp32 elements (a ResNet-18 stem activation is 64 x 112 x 112 = 802 816),
  * ATen out-of-place add (graph-pool allocation pattern of autograd's gradient accumulation),
  * ATen zero_ (FillFunctor), relu / threshold_backward, copy_,
  * libbreach_hip's own smallest kernel (bh_state_reset) through ctypes,
  * a 3x3 convolution 64->64 at 56 x 56 (MIOpen's pick) and the eval-BN kernel E forward of the same activation.

    python scripts/node_cost_probe.py [--nodes 600] [--replays 30]  ->  JSON lines
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402


def time_chain(name, body, nodes, replays, device, note=""):
    stream = torch.cuda.Stream(device)
    stream.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(stream):
        for _ in range(3):
            body(4)
        torch.cuda.synchronize(device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            body(nodes)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize(device)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(replays):
            graph.replay()
        stop.record()
        torch.cuda.synchronize(device)
        graph_us = start.elapsed_time(stop) * 1e3 / (replays * nodes)
        start.record()
        for _ in range(max(replays // 6, 2)):
            body(nodes)
        stop.record()
        torch.cuda.synchronize(device)
        eager_us = start.elapsed_time(stop) * 1e3 / (max(replays // 6, 2) * nodes)
    print(json.dumps(dict(chain=name, nodes=nodes, replays=replays, graph_us_per_node=round(graph_us, 3),
                          eager_us_per_node=round(eager_us, 3), note=note)), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=600)
    ap.add_argument("--replays", type=int, default=30)
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    from breaching_amd import _lib

    lib = _lib.load()
    N, R = args.nodes, args.replays
    for numel in (1, 65536, 802816):
        x = torch.zeros(numel, device=device)

        def inplace(n, x=x):
            for _ in range(n):
                x.add_(1.0)

        time_chain(f"aten add_ in place, {numel} elements", inplace, N, R, device)
    for numel in (65536, 802816):
        a = torch.zeros(numel, device=device)
        b = torch.ones(numel, device=device)

        def outofplace(n, a=a, b=b):
            y = a
            for _ in range(n):
                y = torch.add(y, b)
            return y

        time_chain(f"aten add out of place, {numel} elements", outofplace, N, R, device, "fresh output per node (graph pool)")

        def zero(n, a=a):
            for _ in range(n):
                a.zero_()

        time_chain(f"aten zero_, {numel} elements", zero, N, R, device)

        def relu(n, a=a):
            y = a
            for _ in range(n):
                y = torch.relu(y)
            return y

        time_chain(f"aten relu, {numel} elements", relu, N, R, device)

        def copy(n, a=a, b=b):
            for _ in range(n):
                a.copy_(b)

        time_chain(f"aten copy_, {numel} elements", copy, N, R, device)
    state = torch.zeros(_lib.BH_STATE_WORDS, dtype=torch.int32, device=device)

    def ours(n):
        s = _lib.current_stream_handle(device)
        for _ in range(n):
            lib.bh_state_reset(_lib.ptr(state), s)

    time_chain("libbreach_hip bh_state_reset (one wave)", ours, N, R, device)
    x = torch.randn(1, 64, 56, 56, device=device)
    w = torch.randn(64, 64, 3, 3, device=device) * 0.05

    def conv(n, x=x, w=w):
        y = x
        for _ in range(n):
            y = torch.nn.functional.conv2d(y, w, padding=1)
        return y

    time_chain("conv2d 64->64 3x3 at 1x64x56x56", conv, min(N, 200), R, device, "MIOpen's pick; may be several kernels per node")
    from breaching_amd.attacker import _EvalBNFunction

    wt, bs = torch.ones(64, device=device), torch.zeros(64, device=device)
    inv, mi = torch.ones(64, device=device), torch.zeros(64, device=device)

    def bn(n, x=x):
        y = x
        with torch.no_grad():
            for _ in range(n):
                y = _EvalBNFunction.apply(y, wt, bs, inv, mi, None)[0]
        return y

    time_chain("kernel E forward at 1x64x56x56", bn, N, R, device)


if __name__ == "__main__":
    main()
