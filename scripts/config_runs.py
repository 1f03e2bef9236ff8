"""Run the BASELINE.json configurations through the public API on one MI355X and report wall time / iterations per second.

    python scripts/config_runs.py [--full]        # --full: the real 24 000-iteration ResNet-18 run of configs[1]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import breaching_amd
from breaching_amd.cases import build_case, build_text_case, initial_candidate, psnr

parser = argparse.ArgumentParser()
parser.add_argument("--full", action="store_true")
parser.add_argument("--only", default=None)
parser.add_argument("--its", type=int, default=200, help="iterations of the short configurations (configs[2], configs[4])")
parser.add_argument("--per-process", action="store_true", help="one process per configuration instead of one process for all")
parser.add_argument("--starts", type=int, default=0,
                    help="--only 24k: this many full-length (24 000-iteration) ResNet-18 runs of configs[1] from the reference's starting points "
                         "(nominal x0, then starts <= 16 ulp away, seeded like oracle/make_golden.py), four in flight, compared with the "
                         "reference's own full-length runs in tests/golden/attack_resnet18_24k.npz")
args = parser.parse_args()
if args.only is None and args.per_process:
    import subprocess

    for k in "12345":
        subprocess.run([sys.executable, os.path.abspath(__file__), "--only", k] + (["--full"] if args.full else []))
    raise SystemExit(0)
# Default: ALL configurations in this one process, attack after attack, the way benchmark_breaches.py:60-70 drives an attacker.
# (Round 2 had to split them: the 4-in-flight restarts of configs[3] ran 3.6x slower after any earlier attack of the process --
# fixed in round 3, see attacker._run_trial_group and profiles/r3_stall_bisect.jsonl.)
dev = torch.device("cuda:0")
setup = dict(device=dev, dtype=torch.float)
# kernel D tuning for same-box A/B profiles: since ABI 5 these are launch arguments carried by the plan (cfg.impl.bn_grid_cap /
# bn_load_depth / bn_finalize_block), not process-wide library state
BN_TUNING = [f"impl.{key}={os.environ[env]}" for env, key in (("BN_DEPTH", "bn_load_depth"), ("BN_GRID_CAP", "bn_grid_cap"),
                                                               ("BN_FIN_BLOCK", "bn_finalize_block"), ("GM_CACHE", "gm_cache_policy"),
                                                               ("GM_CACHE_BWD", "gm_cache_policy_bwd")) if env in os.environ]
out = {}


def run(name, case, cfg, x0=None):
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {}, initial_data=x0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    its = sum(len(v) for k, v in stats.items() if k.startswith("Trial_"))
    hist = stats["Trial_0_Val"]
    entry = dict(iterations=its, wall_s=round(dt, 2), iterations_per_s=round(its / dt, 1), first_loss=hist[0], last_loss=hist[-1],
                 opt_value=stats["opt_value"], execution=sorted(set(stats["execution"]["trials"].values())))
    if case.data_cfg.modality == "vision":
        entry["psnr_db"] = round(psnr(rec["data"], case.true_user_data["data"], case.data_cfg), 3)
    else:
        entry["token_accuracy"] = float((rec["data"].cpu() == case.true_user_data["data"]).float().mean())
    out[name] = entry
    print(name, json.dumps(entry), flush=True)


if args.only == "24k":
    import numpy as np
    from breaching_amd.cases import ulp_perturb

    gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "attack_resnet18_24k.npz"))
    its, n = int(gold["iterations"]), max(args.starts, 1)
    case = build_case("resnet18", "ImageNet", 1, device=dev)  # gradient on the CPU: the target the reference attacked
    starts = {}
    for idx in range(n):
        x0 = initial_candidate(case.data_cfg, 1)
        starts[idx] = x0 if idx == 0 else ulp_perturb(x0, 16, torch.Generator().manual_seed(int(gold["twin_seed"]) + idx))
    cfg = breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={its}", "optim.callback=4000", f"restarts.num_trials={n}"])
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
    attacker._preset = dict(inits={t: (x,) for t, x in starts.items()}, labels=None)  # every trial from its prescribed start
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
    dt = time.perf_counter() - t0
    marks = [999, 8997, 9100, 14999, 15100, 21014, 21100, its - 1]
    ref_hist = np.concatenate([gold["history"][None, :], gold["twin_history"]], axis=0).astype(np.float64)
    report = dict(starts=n, iterations_each=its, wall_s=round(dt, 1), trial_iterations_per_s=round(n * its / dt, 1),
                  execution=sorted(set(stats["execution"]["trials"].values())),
                  hip={f"loss@{m}": [round(stats[f"Trial_{t}_Val"][m], 6) for t in range(n)] for m in marks},
                  reference={f"loss@{m}": [round(float(v), 6) for v in ref_hist[:, m]] for m in marks})
    report["hip"].update(opt_value=[round(s, 6) for _, s in scored], psnr_db=[round(p, 4) for p, _ in scored])
    report["reference"].update(opt_value=[round(float(v), 6) for v in np.concatenate([[gold["opt_value"]], gold["twin_opt_value"]])],
                               psnr_db=[round(float(v), 4) for v in np.concatenate([[gold["psnr"]], gold["twin_psnr"]])])
    for key in ("opt_value", "psnr_db", f"loss@{its - 1}"):
        h, r = np.asarray(report["hip"][key]), np.asarray(report["reference"][key])
        print(f"  {key:12s} hip {h.mean():.6f} +- {h.std(ddof=1) if len(h) > 1 else 0:.6f} [{h.min():.6f}, {h.max():.6f}]   "
              f"reference {r.mean():.6f} +- {r.std(ddof=1):.6f} [{r.min():.6f}, {r.max():.6f}]", flush=True)
    print(json.dumps(report), flush=True)
    raise SystemExit(0)
if args.only in (None, "1"):
    case = build_case("convnet", "CIFAR10", 1, device=dev)
    run("configs[0] ConvNet CIFAR-10 invertinggradients, 100 its", case,
        breaching_amd.get_attack_config("invertinggradients", ["optim.max_iterations=100"]), initial_candidate(case.data_cfg, 1, seed=6))
if args.only in (None, "2"):
    case = build_case("resnet18", "ImageNet", 1, device=dev, gradient_device=dev)
    its = 24000 if args.full else 2000
    run(f"configs[1] ResNet-18 ImageNet invertinggradients, {its} its", case,
        breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={its}"]), initial_candidate(case.data_cfg, 1))
if args.only in (None, "3"):
    case = build_case("resnet50", "ImageNet", 8, device=dev, gradient_device=dev, provide_buffers=True)
    run(f"configs[2] ResNet-50 ImageNet batch 8 see-through-gradients (+DeepInversion), {args.its} its", case,
        breaching_amd.get_attack_config("seethroughgradients", [f"optim.max_iterations={args.its}", "optim.callback=100", *BN_TUNING]), initial_candidate(case.data_cfg, 8))
if args.only in (None, "4"):
    case = build_case("resnet18", "ImageNet", 1, device=dev, gradient_device=dev)
    its = 24000 if args.full else 500
    run(f"configs[3] (one GPU's share) ResNet-18 invertinggradients, 4 restarts in flight x {its} its", case,
        breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={its}", "restarts.num_trials=4"]))
if args.only in (None, "5"):
    case = build_text_case(device=dev, full_size=True, seq_len=32)
    run(f"configs[4] BERT-base seq 32 TAG joint attack, {args.its} its", case,
        breaching_amd.get_attack_config("tag", [f"optim.max_iterations={args.its}", "optim.callback=100"]))
if args.only == "fedavg":
    # SURVEY section 8 f2 at a BASELINE-sized list: FedAvg user, 2 local SGD steps x 2 images, ResNet-18 / ImageNet -- every local step's
    # parameter update and the final p_local - p_server are multi-tensor launches (mt_kernel<*>) over the 46.8 MB list
    from breaching_amd.cases import build_fedavg_case

    model_name = os.environ.get("FEDAVG_MODEL", "resnet18")
    # ResNet-50 at its random init diverges under two plain-SGD steps of 0.05 (non-finite objective at the first iteration)
    case = build_fedavg_case(device=dev, model_name=model_name, data_name="ImageNet", lr=float(os.environ.get("FEDAVG_LR", "0.05")))
    run(f"FedAvg multi-step objective (f2): {model_name} ImageNet, 4 images, 2 local steps x 2, invertinggradients, {args.its} its", case,
        breaching_amd.get_attack_config("invertinggradients", [f"optim.max_iterations={args.its}", "optim.callback=100"]), initial_candidate(case.data_cfg, 4))
print(json.dumps(out, indent=1))
