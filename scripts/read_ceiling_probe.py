"""What can a launch shaped like kernel A's forward reach at each BASELINE list size -- is 0.65-0.70 of the HBM peak on the ResNet-18 list
the kernel, or the chip?

Kernel A's forward streams two lists once.  On the ResNet-18 list (2 x 46.8 MB) both sit in the 256 MiB Infinity Cache and `rec` has
just been written by autograd: the launch reads from the cache fabric, not from HBM, yet the bench line prices it against 8 TB/s.
Round 3's sweeps showed the launch insensitive to grid size and software pipelining (profiles/r3_kernel_bench_pipeline_sweep.json),
which points at the memory system.  This probe measures the ceiling directly with `diag_read` (scripts/diag/read_ceiling.hip: the
same persistent grid and eight staged 16-byte loads per lane as kernel A, two multiply-adds per element instead of the objective):

  * warm: the same two buffers read again and again,
  * behind a writer: buffer `a` rewritten by `diag_fill` before every read (what autograd does to `rec`),
  * plain vs non-temporal loads, grids of 256 / 512 / 1024 / 2048 workgroups,

next to kernel A's own forward (cosine) on a list of the same size under the same two conditions.

    python scripts/read_ceiling_probe.py [--launches 30]   ->  JSON lines (us per launch, GB/s of 2 * N * 4 bytes, fraction of 8 TB/s)
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK = 8.0e12
CHUNK = 4096


def load_diag():
    src = os.path.join(ROOT, "scripts", "diag", "read_ceiling.hip")
    lib = os.path.join(ROOT, "scripts", "diag", "libread_ceiling.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
        subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", src, "-o", lib], check=True)
    diag = ctypes.CDLL(lib)
    diag.diag_read.restype = ctypes.c_int
    diag.diag_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    diag.diag_fill.restype = ctypes.c_int
    diag.diag_fill.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
    return diag


def per_launch_us(device, launches, before, body):
    """Average event time of `body()` alone; `before()` (untimed) runs ahead of every timed launch."""
    pairs = []
    for _ in range(3):
        before()
        body()
    torch.cuda.synchronize(device)
    for _ in range(launches):
        before()
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        body()
        stop.record()
        pairs.append((start, stop))
    torch.cuda.synchronize(device)
    times = sorted(s.elapsed_time(e) * 1e3 for s, e in pairs)
    return times[len(times) // 2], times[0]  # median, best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=30)
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()
    diag = load_diag()
    if args.build_only:
        return
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    from bench import bert_base_gradient_shapes
    from breaching_amd import _lib
    from breaching_amd.cases import build_model
    from breaching_amd.gm import GradientMatchPlan

    stream = _lib.current_stream_handle(device)
    sizes = dict(resnet18=[tuple(p.shape) for p in build_model("resnet18", 1000).parameters()],
                 resnet50=[tuple(p.shape) for p in build_model("resnet50", 1000).parameters()],
                 bert_base=bert_base_gradient_shapes())
    sink = torch.zeros(4096, device=device)
    for name, shapes in sizes.items():
        n = sum(int(torch.Size(s).numel()) for s in shapes)
        n_chunks = n // CHUNK
        a = torch.randn(n_chunks * CHUNK, device=device)
        b = torch.randn(n_chunks * CHUNK, device=device)
        nbytes = 2 * n_chunks * CHUNK * 4
        for nt in (0, 1):
            for grid in (256, 512, 1024, 2048):
                row = dict(list=name, kernel="diag_read", bytes=nbytes, grid=grid, non_temporal=nt)
                read = lambda: diag.diag_read(a.data_ptr(), b.data_ptr(), n_chunks, grid, nt, sink.data_ptr(), stream)  # noqa: E731
                fill = lambda: diag.diag_fill(a.data_ptr(), a.numel(), 0.5, stream)  # noqa: E731
                for tag, before in (("warm", lambda: None), ("behind_writer", fill)):
                    med, best = per_launch_us(device, args.launches, before, read)
                    row[f"{tag}_us"], row[f"{tag}_best_us"] = round(med, 2), round(best, 2)
                    row[f"{tag}_GBps"], row[f"{tag}_frac_of_8TBps"] = round(nbytes / med / 1e3, 1), round(nbytes / (med * 1e-6) / PEAK, 3)
                print(json.dumps(row), flush=True)
        # kernel A's own forward on a list of this size, same two conditions -- the dispatch's own start / stop events (the C side's
        # hipExtLaunchKernelGGL pair, what bench.py reports) -- on (1) the model's real tensor list (ragged: ResNet-18 has 40 tensors
        # of 64-512 elements among its 62) and (2) ONE tensor of the same number of elements (full chunks only, one allocation):
        # what the chunk-table indirection and the short chunks cost
        gen = torch.Generator(device=device).manual_seed(1)
        for label, shp in (("model list", shapes), ("one tensor of the same size", [(n,)])):
            data = [torch.randn(s, device=device, generator=gen) for s in shp]
            rec = [torch.randn(s, device=device, generator=gen) for s in shp]
            plan = GradientMatchPlan(data)
            fwd_bytes = 2 * n * 4
            row = dict(list=name, kernel=f"kernel A forward (cosine), {label}", tensors=len(shp), chunks=plan.n_chunks, rows=plan.n_rows, bytes=fwd_bytes)

            def rewrite():
                torch._foreach_mul_(rec, 1.0)  # every tensor of `rec` rewritten in place, as autograd leaves it

            for tag, before in (("warm", lambda: None), ("behind_writer", rewrite)):
                for _ in range(3):
                    before()
                    plan.forward(0, rec, 1.0, 0.0, 1e-7, None)
                plan.enable_timing()
                for _ in range(args.launches):
                    before()
                    plan.forward(0, rec, 1.0, 0.0, 1e-7, None)  # BH_GM_COSINE
                torch.cuda.synchronize(device)
                t = sorted(plan.drain_timers()["fwd"])
                med = t[len(t) // 2]
                row[f"{tag}_us"], row[f"{tag}_best_us"] = round(med, 2), round(t[0], 2)
                row[f"{tag}_GBps"], row[f"{tag}_frac_of_8TBps"] = round(fwd_bytes / med / 1e3, 1), round(fwd_bytes / (med * 1e-6) / PEAK, 3)
            print(json.dumps(row), flush=True)
            del data, rec, plan
        # the bare read with the same event mechanism (diag_read_timed), 512 workgroups, plain loads
        import ctypes as _ct

        diag.diag_read_timed.restype = _ct.c_int
        diag.diag_read_timed.argtypes = [_ct.c_void_p, _ct.c_void_p, _ct.c_int64, _ct.c_int32, _ct.c_int32, _ct.c_void_p, _ct.c_void_p, _ct.c_int32,
                                         _ct.c_int32, _ct.POINTER(_ct.c_float)]
        row = dict(list=name, kernel="diag_read, dispatch start/stop events", bytes=nbytes, grid=512)
        for nt in (0, 1):
            for tag, fill in (("warm", 0), ("behind_writer", 1)):
                us = (_ct.c_float * args.launches)()
                rc = diag.diag_read_timed(a.data_ptr(), b.data_ptr(), n_chunks, 512, nt, sink.data_ptr(), stream, fill, args.launches, us)
                t = sorted(us)
                row[f"{'nt_' if nt else ''}{tag}_us"] = round(t[len(t) // 2], 2) if rc == 0 else None
        print(json.dumps(row), flush=True)
        # the backward / multi-tensor shape: read two buffers, write one (3 N 4 bytes), one workgroup per chunk -- next to kernel A's
        # own backward on the model's list (plan.backward after plan.forward; events of the C side)
        diag.diag_rw_timed.restype = _ct.c_int
        diag.diag_rw_timed.argtypes = [_ct.c_void_p, _ct.c_void_p, _ct.c_void_p, _ct.c_int64, _ct.c_int32, _ct.c_void_p, _ct.c_int32, _ct.c_int32,
                                       _ct.POINTER(_ct.c_float)]
        o = torch.empty_like(a)
        row = dict(list=name, kernel="diag_rw (read a, b; write o), dispatch start/stop events", bytes=3 * n_chunks * CHUNK * 4, workgroups=n_chunks)
        for policy, tag in ((0, "plain"), (1, "nt_loads"), (2, "nt_loads_and_stores")):
            for cond, fill in (("warm", 0), ("behind_writer", 1)):
                us = (_ct.c_float * args.launches)()
                rc = diag.diag_rw_timed(a.data_ptr(), b.data_ptr(), o.data_ptr(), n_chunks, policy, stream, fill, args.launches, us)
                t = sorted(us)
                row[f"{tag}_{cond}_us"] = round(t[len(t) // 2], 2) if rc == 0 else None
        print(json.dumps(row), flush=True)
        gen = torch.Generator(device=device).manual_seed(2)
        data = [torch.randn(s, device=device, generator=gen) for s in shapes]
        rec = [torch.randn(s, device=device, generator=gen) for s in shapes]
        plan = GradientMatchPlan(data)
        row = dict(list=name, kernel="kernel A backward (cosine), model list, after its forward", bytes=3 * n * 4, chunks=plan.n_chunks)
        for _ in range(3):
            plan.backward(0, rec, plan.forward(0, rec, 1.0, 0.0, 1e-7, None), None, None)
        plan.enable_timing()
        for _ in range(args.launches):
            plan.backward(0, rec, plan.forward(0, rec, 1.0, 0.0, 1e-7, None), None, None)
        torch.cuda.synchronize(device)
        t = sorted(plan.drain_timers()["bwd"])
        row["us"], row["best_us"], row["frac_of_8TBps"] = round(t[len(t) // 2], 2), round(t[0], 2), round(3 * n * 4 / (t[len(t) // 2] * 1e-6) / PEAK, 3)
        print(json.dumps(row), flush=True)
        del a, b, o, data, rec, plan
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
