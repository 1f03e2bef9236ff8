"""Multi-tensor kernels (`mt_kernel<OP>`, FedAvg local step / Pearlmutter patch): 16-byte loads vs the round-2/3 build.

The instruction census (profiles/r4_kernel_isa_census.txt, no GPU needed) showed the round-2/3 kernel loading every operand
with four branch-guarded 4-byte loads -- the compiler had scalarised `a4 ? a4[i] : zero`.  This probe times both builds on the
gradient lists of ResNet-18 / ResNet-50 / BERT-base and checks that the outputs are bit-identical to each other and to torch's
two separately rounded ops:

    python scripts/mt_kernel_probe.py [--prev build/libbreach_mt_prev.so] [--launches 30]  ->  JSON lines

`--prev`: a shared library holding the previous build of mt_kernels.hip (built by hand from `git show <rev>:...`); without it
only the current library is timed.  ALGORITHMIC bytes: axpy 3 N 4 (read a, b; write out), axpy-minus 4 N 4, scale 2 N 4.
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

PEAK = 8.0e12


def shapes_of(name):
    from bench import bert_base_gradient_shapes
    from breaching_amd.cases import build_model

    if name == "bert_base":
        return bert_base_gradient_shapes()
    return [tuple(p.shape) for p in build_model(name, 1000).parameters()]


def bind_prev(path):
    from breaching_amd import _lib

    lib = ctypes.CDLL(path)
    for name in ("bh_mt_axpy", "bh_mt_scale"):
        restype, argtypes = _lib._PROTOTYPES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def timed(device, launches, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(device)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(launches):
        fn()
    stop.record()
    torch.cuda.synchronize(device)
    return start.elapsed_time(stop) * 1e3 / launches  # us per launch, back to back


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prev", default=None)
    ap.add_argument("--launches", type=int, default=30)
    ap.add_argument("--lists", default="resnet18,resnet50,bert_base")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    from breaching_amd import _lib
    from breaching_amd.gm import ListLayout

    libs = [("current", _lib.load())]
    if args.prev and os.path.exists(args.prev):
        libs.append(("previous", bind_prev(args.prev)))
    stream = _lib.current_stream_handle(device)
    for name in args.lists.split(","):
        try:
            shapes = shapes_of(name)
        except Exception as exc:  # a model class this checkout cannot build on the meta device
            print(json.dumps(dict(list=name, error=repr(exc)[:200])), flush=True)
            continue
        layout = ListLayout(shapes, device)
        gen = torch.Generator(device=device).manual_seed(5)
        a = [torch.randn(s, device=device, generator=gen) for s in shapes]
        b = [torch.randn(s, device=device, generator=gen) for s in shapes]
        c = [torch.randn(s, device=device, generator=gen) for s in shapes]
        alpha = -0.0123
        n = sum(layout.numels)
        want_axpy = [x + alpha * y for x, y in zip(a, b)]
        want_minus = [(x + alpha * y) - z for x, y, z in zip(a, b, c)]
        want_scale = [alpha * x for x in a]
        outs = {}
        for tag, lib in libs:
            row = dict(list=name, tensors=len(shapes), elements=n, build=tag)
            for op, bytes_per_elem, call, want in (
                ("axpy", 12, lambda out, lib=lib: lib.bh_mt_axpy(layout.n_tensors, layout.pointers(a), layout.pointers(b), None, alpha,
                                                                _lib.ptr(layout.chunks_dev), layout.n_chunks, layout.mt_bounds, _lib.ptr(out), stream), want_axpy),
                ("axpy_minus", 16, lambda out, lib=lib: lib.bh_mt_axpy(layout.n_tensors, layout.pointers(a), layout.pointers(b), layout.pointers(c), alpha,
                                                                      _lib.ptr(layout.chunks_dev), layout.n_chunks, layout.mt_bounds, _lib.ptr(out), stream), want_minus),
                ("scale", 8, lambda out, lib=lib: lib.bh_mt_scale(layout.n_tensors, layout.pointers(a), alpha, _lib.ptr(layout.chunks_dev), layout.n_chunks,
                                                                 layout.mt_bounds, _lib.ptr(out), stream), want_scale),
            ):
                out = layout.empty_flat().zero_()
                rc = call(out)
                torch.cuda.synchronize(device)
                assert rc == 0, (op, rc)
                got = layout.split(out)
                row[f"{op}_bit_identical_to_torch"] = all(torch.equal(g, w) for g, w in zip(got, want))
                outs[(tag, op)] = out
                us = timed(device, args.launches, lambda: call(out))
                row[f"{op}_us"] = round(us, 2)
                row[f"{op}_GBps"] = round(bytes_per_elem * n / us / 1e3, 1)
                row[f"{op}_frac_of_8TBps"] = round(bytes_per_elem * n / (us * 1e-6) / PEAK, 3)
            print(json.dumps(row), flush=True)
        if len(libs) == 2:
            same = {op: bool(torch.equal(outs[("current", op)], outs[("previous", op)])) for op in ("axpy", "axpy_minus", "scale")}
            print(json.dumps(dict(list=name, current_equals_previous=same)), flush=True)


if __name__ == "__main__":
    main()
