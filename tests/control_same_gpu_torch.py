"""Same-GPU, same-arithmetic control for the end-of-run statistics of BASELINE configs[1].

In every free-running comparison of rounds 3-4 the HIP runs' mean final loss / opt_value / PSNR sat slightly below the CPU
reference's (1-2 standard errors each, always the same sign).  Two candidate causes: (a) the HIP kernels of this repo, (b) the
victim model's convolution arithmetic on the GPU (MIOpen) against the CPU reference's (oneDNN).  This script separates them: it
runs oracle/restate.py -- the statement-by-statement restatement of the reference attack, pinned to the unmodified reference on
CPU by tests/test_oracle_pinning.py -- ON THE GPU with PyTorch-ROCm ops throughout (no kernel of libbreach_hip.so involved),
from the same eight starting points (nominal x0 + seven <= 16 ulp away) as the reference's eight CPU runs stored in
tests/golden/attack_resnet18_long.npz, 1 000 iterations each, and the HIP path from the same starts.  If the torch-on-GPU
distribution sits with HIP's and both away from the CPU reference's, the offset is (b).

    python tests/control_same_gpu_torch.py [--starts 8] [--iterations 1000] [--out gpurun_out/control.json]

One worker process per start for the torch control (eager PyTorch is launch-bound on the host: the eight processes share the GPU
almost without slowing each other); the HIP runs go through the product's own restarts (eight trials in flight).
Test infrastructure: lives under tests/ because it imports oracle/.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MARKS = (373, 380, 624, 630, 874, 880, 999)  # around the three step-lr milestones of a 1000-iteration run, and the end


def start_point(data_cfg, idx, seed):
    from breaching_amd.cases import initial_candidate, ulp_perturb

    x0 = initial_candidate(data_cfg, 1)
    return x0 if idx == 0 else ulp_perturb(x0, 16, torch.Generator().manual_seed(seed + idx))


def worker(idx, iterations, seed, out_path):
    import breaching_amd
    from breaching_amd.cases import build_case, psnr
    from oracle import restate

    dev = torch.device("cuda:0")
    case = build_case("resnet18", "ImageNet", 1, device=dev)
    cfg = breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={iterations}", "optim.callback=1000"])
    x0 = start_point(case.data_cfg, idx, seed)
    torch.manual_seed(7)
    timing = []
    t0 = time.perf_counter()
    rec, stats = restate.run_attack(case.model, case.loss_fn, cfg, case.server_payload, case.shared_data, initial_data=x0, device=dev,
                                    timing=timing)
    torch.cuda.synchronize()
    hist = np.asarray(stats["Trial_0_Val"], dtype=np.float64)
    out = dict(idx=idx, history=hist.tolist(), opt_value=float(stats["opt_value"]), wall_s=time.perf_counter() - t0, loop_s=timing[0],
               psnr=float(psnr(rec["data"], case.true_user_data["data"].to(dev), case.data_cfg)))
    with open(out_path, "w") as f:
        json.dump(out, f)


def hip_runs(n, iterations, seed):
    import breaching_amd
    from breaching_amd.cases import build_case, psnr

    dev = torch.device("cuda:0")
    case = build_case("resnet18", "ImageNet", 1, device=dev)
    cfg = breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={iterations}", "optim.callback=1000", f"restarts.num_trials={n}"])
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=dev, dtype=torch.float))
    attacker._preset = dict(inits={t: (start_point(case.data_cfg, t, seed),) for t in range(n)}, labels=None)
    scored, inner = [], attacker._score_trial

    def spy(candidate, labels, rec_model, shared_data):
        score = inner(candidate, labels, rec_model, shared_data)
        scored.append((psnr(candidate, case.true_user_data["data"], case.data_cfg), float(score)))
        return score

    attacker._score_trial = spy
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {})
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return [dict(idx=t, history=list(map(float, stats[f"Trial_{t}_Val"])), opt_value=scored[t][1], psnr=scored[t][0]) for t in range(n)], wall


def table(name, runs):
    cols = {f"loss@{m}": np.asarray([r["history"][m] for r in runs if len(r["history"]) > m]) for m in MARKS}
    cols["opt_value"] = np.asarray([r["opt_value"] for r in runs])
    cols["psnr"] = np.asarray([r["psnr"] for r in runs])
    return name, cols


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--starts", type=int, default=8)
    ap.add_argument("--iterations", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "control_same_gpu_torch.json"))
    ap.add_argument("--worker", nargs=2, default=None)
    ap.add_argument("--hip-only", action="store_true",
                    help="only the HIP runs (cheap: 64 starts x 1000 iterations take about two minutes), for a larger sample of the HIP "
                         "distribution against reference runs generated on CPU (oracle/make_golden.py --long-worker IDX OUT)")
    ap.add_argument("--torch-only", action="store_true", help="only the torch-on-GPU runs (more starts for a firmer control sample)")
    ap.add_argument("--first-start", type=int, default=0, help="index of the first starting point (0 = the nominal x0)")
    ap.add_argument("--processes", type=int, default=3, help="torch-on-GPU worker processes at a time (eight at once ran at 8 it/s each)")
    args = ap.parse_args()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "attack_resnet18_long.npz"))
    seed = int(gold["twin_seed"])
    if args.worker is not None:
        worker(int(args.worker[0]), args.iterations, seed, args.worker[1])
        return
    assert args.iterations == int(gold["iterations"]), "the stored CPU reference runs are 1000 iterations long"
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    if args.hip_only:
        hip, hip_wall = hip_runs(args.starts, args.iterations, seed)
        name, cols = table("hip", hip)
        report = dict(starts=args.starts, iterations=args.iterations, hip_wall_s=round(hip_wall, 1),
                      hip={k: dict(mean=float(v.mean()), sd=float(v.std(ddof=1)), n=int(len(v)), values=[round(float(x), 6) for x in v]) for k, v in cols.items()})
        for k, v in cols.items():
            print(f"{k:10s} hip {v.mean():.6f} +- {v.std(ddof=1):.6f} (n = {len(v)}, standard error {v.std(ddof=1) / np.sqrt(len(v)):.6f})", flush=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)
        return
    t0 = time.perf_counter()
    torch_runs, pending, running = [], list(range(args.first_start, args.first_start + args.starts)), []
    while pending or running:
        while pending and len(running) < args.processes:
            idx = pending.pop(0)
            path = f"{args.out}.worker{idx}.json"
            running.append((subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(idx), path, "--iterations", str(args.iterations)]), path))
        proc, path = running.pop(0)
        proc.wait()
        if proc.returncode == 0:
            with open(path) as f:
                torch_runs.append(json.load(f))
        os.path.exists(path) and os.remove(path)
    torch_wall = time.perf_counter() - t0
    if args.torch_only:
        name, cols = table("torch_on_gpu", torch_runs)
        report = dict(starts=sorted(r["idx"] for r in torch_runs), iterations=args.iterations, torch_on_gpu_wall_s=round(torch_wall, 1),
                      torch_on_gpu={k: dict(mean=float(v.mean()), sd=float(v.std(ddof=1)), n=int(len(v)), values=[round(float(x), 6) for x in v]) for k, v in cols.items()})
        for k, v in cols.items():
            print(f"{k:10s} torch on GPU {v.mean():.6f} +- {v.std(ddof=1):.6f} (n = {len(v)})", flush=True)
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)
        return
    hip, hip_wall = hip_runs(args.starts, args.iterations, seed)
    ref_hist = np.concatenate([gold["history"][None, :], gold["twin_history"]], axis=0)
    ref = [dict(history=h.tolist(), opt_value=float(o), psnr=float(p)) for h, o, p in
           zip(ref_hist, np.concatenate([[gold["opt_value"]], gold["twin_opt_value"]]), np.concatenate([[gold["psnr"]], gold["twin_psnr"]]))]
    tables = dict([table("reference_cpu", ref), table("torch_on_gpu", torch_runs), table("hip", hip)])
    report = dict(starts=args.starts, iterations=args.iterations, torch_on_gpu_wall_s=round(torch_wall, 1), hip_wall_s=round(hip_wall, 1),
                  torch_on_gpu_loop_s_each=[round(r["loop_s"], 1) for r in torch_runs],
                  torch_on_gpu_iterations_per_s_one_process=round(args.iterations / np.mean([r["loop_s"] for r in torch_runs]), 2))
    print(f"{'quantity':10s} " + "".join(f"{n:>34s}" for n in tables) + "   (mean +- sd; differences in pooled standard errors)")
    pooled = {}
    for key in tables["reference_cpu"]:
        row, stats = f"{key:10s} ", {}
        for name, cols in tables.items():
            v = cols[key]
            stats[name] = (v.mean(), v.std(ddof=1), len(v))
            row += f"{v.mean():>20.6f} +- {v.std(ddof=1):<10.6f}"

        def z(a, b):
            (ma, sa, na), (mb, sb, nb) = stats[a], stats[b]
            return (ma - mb) / np.sqrt(sa ** 2 / na + sb ** 2 / nb)

        pooled[key] = dict(hip_minus_reference_se=round(float(z("hip", "reference_cpu")), 2),
                           torch_gpu_minus_reference_se=round(float(z("torch_on_gpu", "reference_cpu")), 2),
                           hip_minus_torch_gpu_se=round(float(z("hip", "torch_on_gpu")), 2),
                           **{f"{n}_mean": float(s[0]) for n, s in stats.items()}, **{f"{n}_sd": float(s[1]) for n, s in stats.items()})
        print(row + f"  hip-ref {pooled[key]['hip_minus_reference_se']:+.2f}  torch-ref {pooled[key]['torch_gpu_minus_reference_se']:+.2f}  "
              f"hip-torch {pooled[key]['hip_minus_torch_gpu_se']:+.2f}", flush=True)
    report["quantities"] = pooled
    report["runs"] = dict(torch_on_gpu=[{k: (v if k != "history" else [v[m] for m in MARKS]) for k, v in r.items()} for r in torch_runs],
                          hip=[{k: (v if k != "history" else [v[m] for m in MARKS]) for k, v in r.items()} for r in hip])
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: report[k] for k in ("torch_on_gpu_wall_s", "hip_wall_s", "torch_on_gpu_iterations_per_s_one_process")}))


if __name__ == "__main__":
    main()
