"""Kernel E (eval-mode BatchNorm + add + ReLU, csrc/affine_kernels.hip) at BATCH size, layer by layer.

DESIGN section 6 tabulated these kernels at B = 1 only (4-5 us, launch latency).  At BASELINE configs[2] -- ResNet-50, B = 8 -- the
53 BatchNorm layers move 3-77 MB per launch, and the in-loop trace mixes layers of the same grid size.  This probe launches every
DISTINCT (channels, H x W, epilogue) of that model through the C ABI with the operand sets the attack loop uses in each autograd
order, back to back, `--reps` times each, and prints a manifest (launch order, bytes); run under `rocprofv3 --kernel-trace` and joined
with the trace (`--join`), that gives per-layer dispatch durations:

    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -- python scripts/affine_layer_probe.py --manifest m.json
    python scripts/affine_layer_probe.py --join /tmp/prof/.../*_kernel_trace.csv --manifest m.json --out profiles/r6_affine_layers

Operand sets (breaching_amd/victim_layers.py):
  fwd      x (+ residual) -> y, with the DeepInversion statistics sink (per-(channel, slab) sums)
  bwd1     first-order pass under create_graph: gy, x (+ ReLU mask y) -> gx (+ g_residual), gw, gb
  bwd2     outer pass to the candidate: the same launch + the DeepInversion tap (A_c + B_c x) + gx_add (the d_x of this layer's bwd_bwd)
  bwd_bwd  derivative of bwd1: ggx, ggw, ggb (+ ggr), gy, x (+ mask) -> d_gy, d_x, d_w
Bytes = every tensor-sized operand once (algorithmic); fraction of the 8 TB/s HBM peak (MI355X_MICROARCH.md).  The probe also times
`diag_rw` (scripts/diag/read_ceiling.hip: read two, write one, one workgroup per 4096-float chunk, staged 16-byte accesses, outside the
library) on buffers of the three largest activations: the bare-launch ceiling of that shape, warm and behind a writer.
"""
import argparse
import csv
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PEAK = 8.0e12


def layer_inventory(model_name="resnet50", size=224):
    """[(channels, H, W, epilogue, count)] over the BatchNorm layers of the model; epilogue from the block structure (cases._Residual:
    bn1 / bn2 -> relu; the last BatchNorm of a block -> + identity -> relu; projection shortcut -> none; stem -> relu)."""
    import torch

    from breaching_amd.cases import build_model

    model = build_model(model_name, 1000).eval()
    shapes = {}
    hooks = []

    def make(name):
        def hook(module, inputs):
            x = inputs[0]
            if name.startswith("stem"):
                epi = "relu"
            elif "shortcut" in name:
                epi = "none"
            else:
                last = "bn3" if hasattr(owner_of[name], "bn3") else "bn2"
                epi = "residual+relu" if name.endswith(last) else "relu"
            key = (x.shape[1], x.shape[2], x.shape[3], epi)
            shapes[key] = shapes.get(key, 0) + 1
        return hook

    owner_of = {}
    for mod_name, module in model.named_modules():
        for child_name, child in module.named_children():
            if isinstance(child, torch.nn.BatchNorm2d):
                owner_of[f"{mod_name}.{child_name}" if mod_name else child_name] = module
    for name, module in model.named_modules():
        if isinstance(module, torch.nn.BatchNorm2d):
            hooks.append(module.register_forward_pre_hook(make(name)))
    with torch.no_grad():
        model(torch.zeros(1, 3, size, size))
    for h in hooks:
        h.remove()
    return [(c, h, w, epi, n) for (c, h, w, epi), n in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2])]


def run(args):
    import torch

    from breaching_amd import _lib

    lib = _lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stream = _lib.current_stream_handle(dev)
    B = args.batch
    manifest = dict(batch=B, reps=args.reps, model=args.model, launches=[], ceiling=[])
    gen = torch.Generator(device="cpu").manual_seed(0)
    inventory = layer_inventory(args.model)
    for C, H, W, epi, count in inventory:
        hw = H * W
        n = B * C * hw

        def t():
            return torch.randn(B, C, H, W, generator=gen).to(dev)

        x, gy, ggx, resid, ggr, add_in = t(), t(), t(), t(), t(), t()
        weight, bias = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        inv_std, mean_inv = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        ggw, ggb = torch.randn(C, device=dev), torch.randn(C, device=dev)
        coef, gout = torch.randn(C, 2, device=dev), torch.ones(1, device=dev)
        y, gx, gres, d_gy, d_x = (torch.empty_like(x) for _ in range(5))
        gw, gb, d_w = (torch.empty(C, device=dev) for _ in range(3))
        S = lib.bh_bn_eval_slabs(B, C, hw)
        stats = torch.empty(2 * C * S, dtype=torch.float64, device=dev)
        ws = torch.empty(2 * C * S, dtype=torch.float64, device=dev)
        relu = int("relu" in epi)
        has_res = "residual" in epi
        p = _lib.ptr
        null = p(None)
        mask = p(y) if relu else null
        cases = {
            "fwd": (lambda: lib.bh_bn_eval_fwd(p(x), p(weight), p(bias), p(inv_std), p(mean_inv), p(y), p(stats), p(resid) if has_res else null,
                                               relu, B, C, hw, stream), 2 + has_res),
            "bwd1": (lambda: lib.bh_bn_eval_bwd(p(gy), p(x), p(weight), p(inv_std), p(mean_inv), p(gx), p(gw), p(gb), p(ws), null, null, mask,
                                                p(gres) if has_res else null, null, B, C, hw, stream), 3 + relu + has_res),
            "bwd2": (lambda: lib.bh_bn_eval_bwd(p(gy), p(x), p(weight), p(inv_std), p(mean_inv), p(gx), p(gw), p(gb), p(ws), p(coef), p(gout), mask,
                                                p(gres) if has_res else null, p(add_in), B, C, hw, stream), 4 + relu + has_res),
            "bwd_bwd": (lambda: lib.bh_bn_eval_bwd_bwd(p(ggx), p(ggw), p(ggb), p(gy), p(x), p(weight), p(inv_std), p(mean_inv), p(d_gy), p(d_x), p(d_w),
                                                       p(ws), mask, p(ggr) if has_res else null, B, C, hw, stream), 5 + relu + has_res),
        }
        kernel_of = dict(fwd="bn_eval_fwd_kernel", bwd1="bn_eval_bwd_kernel", bwd2="bn_eval_bwd_kernel", bwd_bwd="bn_eval_bwd_bwd_kernel")
        for name, (launch, operands) in cases.items():
            for _ in range(args.reps):
                _lib.check(launch(), name)
            manifest["launches"].append(dict(kernel=kernel_of[name], order=name, C=C, H=H, W=W, epilogue=epi, layers=count, slabs=S,
                                             reps=args.reps, bytes=operands * n * 4, elements=n))
        torch.cuda.synchronize(dev)
        del x, gy, ggx, resid, ggr, add_in, y, gx, gres, d_gy, d_x
    # the bare-launch ceiling for the three largest activations
    lib_path = os.path.join(ROOT, "scripts", "diag", "libread_ceiling.so")
    if os.path.exists(lib_path):
        diag = ctypes.CDLL(lib_path)
        diag.diag_rw_timed.restype = ctypes.c_int
        diag.diag_rw_timed.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.POINTER(ctypes.c_float)]
        seen = set()
        for C, H, W, epi, count in inventory:
            n = B * C * H * W
            if n in seen or len(seen) >= 3:
                continue
            seen.add(n)
            chunks = n // 4096
            a, b, o = (torch.randn(chunks * 4096, device=dev) for _ in range(3))
            rec = dict(elements=chunks * 4096, bytes=3 * chunks * 4096 * 4, shape=f"{B}x{C}x{H}x{W}")
            for tag, fill in (("warm", 0), ("behind_writer", 1)):
                us = (ctypes.c_float * 25)()
                rc = diag.diag_rw_timed(a.data_ptr(), b.data_ptr(), o.data_ptr(), chunks, 0, stream, fill, 25, us)
                vals = sorted(us[5:])
                rec[tag + "_us"] = round(vals[len(vals) // 2], 2) if rc == 0 else None
                if rc == 0:
                    rec[tag + "_frac"] = round(rec["bytes"] / vals[len(vals) // 2] / 1e-6 / PEAK, 4)
            manifest["ceiling"].append(rec)
    with open(args.manifest, "w") as f:
        json.dump(manifest, f)
    print(json.dumps(dict(cases=len(manifest["launches"]), ceiling=manifest["ceiling"])))


def join(args):
    with open(args.manifest) as f:
        manifest = json.load(f)
    rows = []
    with open(args.join) as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            if "bn_eval_" in name and "combine" not in name:
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"]), name))
    rows.sort()
    at = 0
    out = []
    for case in manifest["launches"]:
        mine = rows[at : at + case["reps"]]
        at += case["reps"]
        assert len(mine) == case["reps"] and all(case["kernel"] in name for _, _, name in mine), (case, mine[:2])
        d = sorted(ns for _, ns, _ in mine[2:])  # the first two launches of a case warm the instruction cache / TLB
        avg = sum(d) / len(d) / 1e3
        out.append(dict(case, avg_us=round(avg, 2), min_us=round(d[0] / 1e3, 2), max_us=round(d[-1] / 1e3, 2),
                        GBs=round(case["bytes"] / avg / 1e3, 1), frac=round(case["bytes"] / (avg * 1e-6) / PEAK, 4)))
    assert at == len(rows), (at, len(rows))
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out + ".json", "w") as f:
        json.dump(dict(batch=manifest["batch"], model=manifest["model"], reps=manifest["reps"], rows=out, ceiling=manifest["ceiling"]), f, indent=0)
    with open(args.out + ".csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["order", "C", "H", "W", "epilogue", "layers", "slabs", "MB", "avg_us", "min_us", "max_us", "GB/s", "frac_of_8TBs"])
        for r in out:
            w.writerow([r["order"], r["C"], r["H"], r["W"], r["epilogue"], r["layers"], r["slabs"], round(r["bytes"] / 1e6, 2), r["avg_us"], r["min_us"],
                        r["max_us"], r["GBs"], r["frac"]])
    for order in ("fwd", "bwd1", "bwd2", "bwd_bwd"):
        sel = [r for r in out if r["order"] == order]
        total_us = sum(r["avg_us"] * r["layers"] for r in sel)
        total_b = sum(r["bytes"] * r["layers"] for r in sel)
        big = [r for r in sel if r["bytes"] >= 10e6]
        print(f"{order:8s} all {sum(r['layers'] for r in sel)} layers: {total_b / 1e6:8.1f} MB in {total_us:7.1f} us = {total_b / total_us / 1e6 / 8:.3f} of peak;  "
              f">= 10 MB launches: " + ", ".join(f"{r['C']}x{r['H']}x{r['W']}/{r['epilogue']} {r['bytes'] / 1e6:.0f} MB {r['avg_us']:.1f} us {r['frac']:.2f}" for r in big))
    print("ceiling (diag_rw, read two write one):", manifest["ceiling"])


def join_pmc(args):
    """HBM traffic per launch from two `rocprofv3 --pmc` passes of this probe (FETCH_SIZE, WRITE_SIZE: separate passes, counters only;
    values in KiB; FETCH_SIZE x 2 = the gfx950 half-count of wide coalesced reads, MI355X_MICROARCH.md) against the algorithmic bytes."""
    with open(args.manifest) as f:
        manifest = json.load(f)

    def per_dispatch(path, counter):
        rows = []
        with open(path) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") == counter and "bn_eval_" in row["Kernel_Name"] and "combine" not in row["Kernel_Name"]:
                    rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
        return [v for _, v in sorted(rows)]

    fetch, write = per_dispatch(args.join_pmc[0], "FETCH_SIZE"), per_dispatch(args.join_pmc[1], "WRITE_SIZE")
    at, out = 0, []
    for case in manifest["launches"]:
        f_kib, w_kib = fetch[at : at + case["reps"]], write[at : at + case["reps"]]
        at += case["reps"]
        assert len(f_kib) == len(w_kib) == case["reps"], (case, len(f_kib), len(w_kib))
        traffic = (2.0 * sum(f_kib) / len(f_kib) + sum(w_kib) / len(w_kib)) * 1024.0
        out.append(dict(order=case["order"], C=case["C"], H=case["H"], W=case["W"], epilogue=case["epilogue"], layers=case["layers"],
                        algorithmic_bytes=case["bytes"], traffic_bytes=int(traffic), ratio=round(traffic / case["bytes"], 4)))
    assert at == len(fetch) == len(write), (at, len(fetch), len(write))
    with open(args.out + "_pmc.json", "w") as f:
        json.dump(dict(batch=manifest["batch"], model=manifest["model"], rows=out), f, indent=0)
    big = [r for r in out if r["algorithmic_bytes"] >= 19e6]
    print("PMC traffic / algorithmic bytes, launches >= 19 MB: " + ", ".join(f"{r['order']} {r['C']}x{r['H']}x{r['W']}/{r['epilogue']} {r['ratio']:.3f}" for r in big))
    print(f"all {len(out)} cases: min {min(r['ratio'] for r in out):.3f}, max {max(r['ratio'] for r in out):.3f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--manifest", default="gpurun_out/affine_layer_manifest.json")
    ap.add_argument("--join", default=None, help="kernel_trace.csv of a rocprofv3 run of this probe")
    ap.add_argument("--join-pmc", nargs=2, default=None, metavar=("FETCH_CSV", "WRITE_CSV"), help="counter_collection.csv of the two --pmc passes")
    ap.add_argument("--out", default="gpurun_out/affine_layers")
    a = ap.parse_args()
    if a.join_pmc:
        join_pmc(a)
    elif a.join:
        join(a)
    else:
        run(a)
