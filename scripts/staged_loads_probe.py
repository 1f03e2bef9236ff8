"""Kernels E (eval-mode BatchNorm) and F (LayerNorm) with staged loads vs the previous build: same bits, fewer round trips?

The instruction census (profiles/r4_kernel_isa_census.txt) showed both kernel families with ONE load per operand in flight per
lane (load -> s_waitcnt vmcnt(0) -> use); at the attack's batch-1 sizes they run at most one wavefront per SIMD, so a launch
was 3-12 serial memory round trips.  The staged build issues four loads per operand before the first use, in the same
accumulation order.  This probe

  1. calls both builds through the C ABI on a grid of shapes x optional-operand combinations and compares every output bit
     for bit (kernel E: forward with statistics / residual / ReLU, backward with tap / mask / residual gradient / folded
     gradient, backward of the backward with every optional operand; kernel F: three orders, with and without affine),
  2. times chains of dependent launches replayed as a hipGraph (what the attack loop does) for both builds,
  3. with --pytest, runs the kernel parity tests of the suite in this same process (one torch import on a fresh box).

    python scripts/staged_loads_probe.py --prev build/libbreach_hip_prev.so [--pytest]   ->  JSON lines
"""
import argparse
import ctypes
import itertools
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

DEV = torch.device("cuda", 0)

E_SHAPES = [(1, 64, 112, 112), (1, 64, 56, 56), (1, 128, 28, 28), (1, 256, 14, 14), (1, 512, 7, 7), (2, 12, 7, 7), (3, 5, 9, 11), (8, 16, 64, 64),
            (4, 8, 60, 60), (1, 3, 1030, 4)]
F_SHAPES = [(32, 768), (16, 64), (7, 130), (15, 37), (300, 96), (5, 1030)]


def bind(path):
    from breaching_amd import _lib

    lib = ctypes.CDLL(path)
    for name in ("bh_bn_eval_fwd", "bh_bn_eval_slabs", "bh_bn_eval_bwd", "bh_bn_eval_bwd_bwd", "bh_ln_fwd", "bh_ln_bwd", "bh_ln_bwd_bwd"):
        restype, argtypes = _lib._PROTOTYPES[name]
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = restype, argtypes
    return lib


def same(a, b):
    """Bit-for-bit, NaNs included."""
    if a is None and b is None:
        return True
    return bool(torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int64),
                            b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int64)))


def kernel_e(lib, ptr, stream, shape, gen, opts):
    B, C, H, W = shape
    hw, dev = H * W, DEV
    r = lambda *s: torch.randn(*s, device=dev, generator=gen)  # noqa: E731
    x, gy, res, add, ggx, ggr = r(*shape), r(*shape), r(*shape), r(*shape), r(*shape), r(*shape)
    w, b, inv, mi, ggw, ggb = r(C), r(C), r(C).abs() + 0.5, r(C), r(C), r(C)
    coef, gout = r(C, 2), r(1)
    S = lib.bh_bn_eval_slabs(B, C, hw)
    out = {}
    # forward
    y = torch.full(shape, float("nan"), device=dev)
    stats = torch.zeros(2 * C * S, dtype=torch.float64, device=dev) if opts["stats"] else None
    rc = lib.bh_bn_eval_fwd(ptr(x), ptr(w), ptr(b), ptr(inv), ptr(mi), ptr(y), ptr(stats), ptr(res if opts["residual"] else None), int(opts["relu"]), B, C, hw, stream)
    assert rc == 0, ("fwd", rc)
    out["y"], out["stats"] = y, stats
    # backward
    gx = torch.full(shape, float("nan"), device=dev)
    gres = torch.full(shape, float("nan"), device=dev) if opts["residual"] else None
    gw, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.zeros(2 * C * S, dtype=torch.float64, device=dev) if S > 1 else None
    rc = lib.bh_bn_eval_bwd(ptr(gy), ptr(x), ptr(w), ptr(inv), ptr(mi), ptr(gx), ptr(gw), ptr(gb), ptr(ws), ptr(coef if opts["tap"] else None),
                            ptr(gout if opts["tap"] else None), ptr(y if opts["relu"] else None), ptr(gres), ptr(add if opts["add"] else None), B, C, hw, stream)
    assert rc == 0, ("bwd", rc)
    out.update(gx=gx, gres=gres, gw=gw, gb=gb)
    # backward of the backward
    d_gy = torch.full(shape, float("nan"), device=dev)
    d_x = torch.full(shape, float("nan"), device=dev) if opts["d_x"] else None
    d_w = torch.zeros(C, device=dev)
    ws2 = torch.zeros(C * S, dtype=torch.float64, device=dev) if S > 1 else None
    rc = lib.bh_bn_eval_bwd_bwd(ptr(ggx if opts["ggx"] else None), ptr(ggw if opts["d_x"] else None), ptr(ggb if opts["d_x"] else None), ptr(gy), ptr(x), ptr(w),
                                ptr(inv), ptr(mi), ptr(d_gy), ptr(d_x), ptr(d_w if opts["ggx"] else None), ptr(ws2), ptr(y if opts["relu"] else None),
                                ptr(ggr if opts["residual"] else None), B, C, hw, stream)
    assert rc == 0, ("bwd_bwd", rc)
    out.update(d_gy=d_gy, d_x=d_x, d_w=d_w)
    return out


def kernel_f(lib, ptr, stream, shape, gen, opts):
    R, D = shape
    dev = DEV
    r = lambda *s: torch.randn(*s, device=dev, generator=gen)  # noqa: E731
    x, gy, u = r(R, D), r(R, D), r(R, D)
    gamma, beta = (r(D), r(D)) if opts["affine"] else (None, None)
    s, t = r(D), r(D)
    y, gx, d_gy, d_x = (torch.full((R, D), float("nan"), device=dev) for _ in range(4))
    mean, rstd = torch.zeros(R, device=dev), torch.zeros(R, device=dev)
    gg, gbt, d_gamma = torch.zeros(D, device=dev), torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    scal = torch.zeros(2 * R, device=dev)
    assert lib.bh_ln_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), R, D, 1e-5, stream) == 0
    assert lib.bh_ln_bwd(ptr(gy), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(gx), ptr(gg), ptr(gbt), R, D, stream) == 0
    assert lib.bh_ln_bwd_bwd(ptr(u if opts["u"] else None), ptr(s if opts["st"] else None), ptr(t if opts["st"] else None), ptr(gy), ptr(x), ptr(gamma),
                             ptr(mean), ptr(rstd), ptr(d_gy), ptr(d_x), ptr(d_gamma if opts["u"] else None), ptr(scal), R, D, stream) == 0
    return dict(y=y, mean=mean, rstd=rstd, gx=gx, ggamma=gg, gbeta=gbt, d_gy=d_gy, d_x=d_x, d_gamma=d_gamma)


def sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)


class _ArgCountLib:
    """--dry: stands in for a library on a machine without a GPU; checks only that every call passes as many arguments as the
    C prototype has (the Python plumbing of this probe can then be exercised before a GPU-minute is spent on it)."""

    def __getattr__(self, name):
        from breaching_amd import _lib

        n = len(_lib._PROTOTYPES[name][1])

        def call(*args):
            assert len(args) == n, (name, len(args), n)
            return 1 if name == "bh_bn_eval_slabs" else 0

        return call


def chain_us(device, nodes, replays, body):
    if device.type != "cuda":  # --dry
        body(2)
        return 0.0
    stream = torch.cuda.Stream(device)
    stream.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(stream):
        body(2)
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
    return start.elapsed_time(stop) * 1e3 / (replays * nodes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prev", default="build/libbreach_hip_prev.so")
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--replays", type=int, default=20)
    ap.add_argument("--pytest", action="store_true")
    ap.add_argument("--dry", action="store_true", help="no GPU: exercise the plumbing against argument-counting stand-ins")
    args = ap.parse_args()
    from breaching_amd import _lib

    global DEV
    if args.dry:
        device = DEV = torch.device("cpu")
        ptr, stream = (lambda t: None if t is None else t.data_ptr()), None
        libs = dict(current=_ArgCountLib(), previous=_ArgCountLib())
        _lib.current_stream_handle = lambda d: None
    else:
        device = DEV
        torch.cuda.set_device(device)
        ptr = _lib.ptr
        libs = dict(current=_lib.load(), previous=bind(args.prev))
        stream = _lib.current_stream_handle(device)
    for part in (bit_identity, chains):
        try:
            part(args, device, libs, ptr, stream)
        except Exception as exc:  # keep going: the next part is worth having on its own
            print(json.dumps(dict(part=part.__name__, error=repr(exc)[:400])), flush=True)
    if args.pytest:
        import pytest

        os.environ.setdefault("BREACH_HIP_GRAPH_STRICT", "1")
        rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", "tests/test_gpu_kernels.py", "-k",
                          "eval_batchnorm or bn_eval_bwd_tap or layernorm_function"])
        print(json.dumps(dict(pytest_exit_code=int(rc))), flush=True)


def bit_identity(args, device, libs, ptr, stream):
    mismatches, cases = [], 0
    for shape in E_SHAPES:
        for stats, residual, relu, tap, add, ggx, d_x in itertools.product((False, True), repeat=7):
            if (add and not tap and residual) or (stats and relu and not residual):  # thin the grid: 128 -> 80 combinations per shape
                continue
            opts = dict(stats=stats, residual=residual, relu=relu, tap=tap, add=add, ggx=ggx, d_x=d_x)
            outs = {}
            for tag, lib in libs.items():
                gen = torch.Generator(device=device).manual_seed(11)
                outs[tag] = kernel_e(lib, ptr, stream, shape, gen, opts)
            sync(device)
            cases += 1
            for key in outs["current"]:
                if not same(outs["current"][key], outs["previous"][key]):
                    a, b = outs["current"][key], outs["previous"][key]
                    mismatches.append(dict(kernel="E", shape=shape, opts={k: int(v) for k, v in opts.items()}, output=key,
                                           max_abs=float((a.double() - b.double()).abs().nan_to_num(0).max())))
    print(json.dumps(dict(check="kernel E current == previous, bit for bit", cases=cases, mismatching_outputs=len(mismatches), first=mismatches[:4])), flush=True)
    f_mismatches, f_cases = [], 0
    for shape in F_SHAPES:
        for affine, u, st in itertools.product((False, True), repeat=3):
            opts = dict(affine=affine, u=u, st=st)
            outs = {}
            for tag, lib in libs.items():
                gen = torch.Generator(device=device).manual_seed(12)
                outs[tag] = kernel_f(lib, ptr, stream, shape, gen, opts)
            sync(device)
            f_cases += 1
            for key in outs["current"]:
                if not same(outs["current"][key], outs["previous"][key]):
                    a, b = outs["current"][key], outs["previous"][key]
                    f_mismatches.append(dict(kernel="F", shape=shape, opts={k: int(v) for k, v in opts.items()}, output=key,
                                             max_abs=float((a.double() - b.double()).abs().nan_to_num(0).max())))
    print(json.dumps(dict(check="kernel F current == previous, bit for bit", cases=f_cases, mismatching_outputs=len(f_mismatches), first=f_mismatches[:4])), flush=True)


def chains(args, device, libs, ptr, stream):
    """Replayed chains of dependent launches, both builds."""
    from breaching_amd import _lib

    for shape in [(1, 64, 112, 112), (1, 64, 56, 56), (1, 128, 28, 28), (1, 256, 14, 14), (8, 64, 56, 56)]:
        B, C, H, W = shape
        hw = H * W
        gen = torch.Generator(device=device).manual_seed(3)
        x = torch.randn(*shape, device=device, generator=gen)
        bufs = [torch.empty_like(x) for _ in range(2)]
        w, b, inv, mi = (torch.ones(C, device=device) for _ in range(4))
        gw, gb = torch.zeros(C, device=device), torch.zeros(C, device=device)
        S = libs["current"].bh_bn_eval_slabs(B, C, hw)
        ws = torch.zeros(2 * C * S, dtype=torch.float64, device=device)
        row = dict(kernel="E", shape=shape, nodes=args.nodes)
        for tag, lib in libs.items():
            def fwd(n, lib=lib):
                s = _lib.current_stream_handle(device)
                src = x
                for i in range(n):
                    dst = bufs[i % 2]
                    lib.bh_bn_eval_fwd(ptr(src), ptr(w), ptr(b), ptr(inv), ptr(mi), ptr(dst), None, ptr(x), 1, B, C, hw, s)
                    src = dst

            def bwd(n, lib=lib):
                s = _lib.current_stream_handle(device)
                src = x
                for i in range(n):
                    dst = bufs[i % 2]
                    lib.bh_bn_eval_bwd(ptr(src), ptr(x), ptr(w), ptr(inv), ptr(mi), ptr(dst), ptr(gw), ptr(gb), ptr(ws), None, None, ptr(x), None, None, B, C, hw, s)
                    src = dst

            def bwd_bwd(n, lib=lib):
                s = _lib.current_stream_handle(device)
                src = x
                for i in range(n):
                    dst = bufs[i % 2]
                    lib.bh_bn_eval_bwd_bwd(ptr(src), None, None, ptr(x), ptr(x), ptr(w), ptr(inv), ptr(mi), ptr(dst), None, ptr(gw), ptr(ws), ptr(x), None, B, C, hw, s)
                    src = dst

            row[f"{tag}_fwd_relu_res_us"] = round(chain_us(device, args.nodes, args.replays, fwd), 3)
            row[f"{tag}_bwd_mask_us"] = round(chain_us(device, args.nodes, args.replays, bwd), 3)
            row[f"{tag}_bwd_bwd_mask_us"] = round(chain_us(device, args.nodes, args.replays, bwd_bwd), 3)
        print(json.dumps(row), flush=True)
    for shape in [(32, 768), (256, 768)]:
        R, D = shape
        gen = torch.Generator(device=device).manual_seed(4)
        x = torch.randn(R, D, device=device, generator=gen)
        bufs = [torch.empty_like(x) for _ in range(2)]
        gamma, beta = torch.ones(D, device=device), torch.zeros(D, device=device)
        mean, rstd = torch.zeros(R, device=device), torch.ones(R, device=device)
        gg, gbt, scal = torch.zeros(D, device=device), torch.zeros(D, device=device), torch.zeros(2 * R, device=device)
        row = dict(kernel="F", shape=shape, nodes=args.nodes)
        for tag, lib in libs.items():
            def fwd(n, lib=lib):
                s = _lib.current_stream_handle(device)
                src = x
                for i in range(n):
                    dst = bufs[i % 2]
                    lib.bh_ln_fwd(ptr(src), ptr(gamma), ptr(beta), ptr(dst), ptr(mean), ptr(rstd), R, D, 1e-5, s)
                    src = dst

            def bwd(n, lib=lib):  # two launches per node: row kernel + column kernel
                s = _lib.current_stream_handle(device)
                src = x
                for i in range(n):
                    dst = bufs[i % 2]
                    lib.bh_ln_bwd(ptr(src), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dst), ptr(gg), ptr(gbt), R, D, s)
                    src = dst

            def bwd_bwd(n, lib=lib):  # two launches per node
                s = _lib.current_stream_handle(device)
                src = x
                for i in range(n):
                    dst = bufs[i % 2]
                    lib.bh_ln_bwd_bwd(ptr(src), ptr(gamma), ptr(beta), ptr(x), ptr(x), ptr(gamma), ptr(mean), ptr(rstd), ptr(dst), None, ptr(gg), ptr(scal), R, D, s)
                    src = dst

            row[f"{tag}_fwd_us"] = round(chain_us(device, args.nodes, args.replays, fwd), 3)
            row[f"{tag}_bwd_2launches_us"] = round(chain_us(device, args.nodes, args.replays, bwd), 3)
            row[f"{tag}_bwd_bwd_2launches_us"] = round(chain_us(device, args.nodes, args.replays, bwd_bwd), 3)
        print(json.dumps(row), flush=True)

if __name__ == "__main__":
    main()
