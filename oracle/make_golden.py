"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz by running the UNMODIFIED reference in the build container.

    python -m oracle.make_golden [--only configs|kernels|schedules|convnet|resnet18|seethrough|tag|variants|fedavg|labels|dlg|multiquery|pearlmutter]
    python -m oracle.make_golden --only resnet18_long|seethrough_b8|tag_bert_base|resnet18_24k|seethrough_noise   (slow: run by name only)

Needs /root/reference (through oracle/ref_shim.py); the outputs are committed so that the GPU box -- which has no
reference checkout -- can pin oracle/restate.py, oracle/kernels_oracle.c and the HIP path against real reference
behaviour.  Everything is seeded; inputs that tests need are stored next to the reference's outputs.
"""

import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle.ref_shim import import_reference  # noqa: E402

LIST_SHAPES = [(3, 5, 7), (1,), (4097,), (64, 3, 3, 3), (8192,), (130,), (2, 4100), (7,)]


def _coerce(node):
    """PyYAML reads `1e-4` as a string (YAML 1.1); Hydra/OmegaConf read it as a float.  Follow Hydra."""
    import re

    if isinstance(node, dict):
        return {k: _coerce(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_coerce(v) for v in node]
    if isinstance(node, str) and re.fullmatch(r"[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+", node):
        return float(node)
    return node


def _cfg(name, overrides=None):
    """Attack config composed from the reference YAML files (Hydra `defaults:` deep merge restated)."""
    import yaml

    from breaching_amd.config import AttrDict, _apply_overrides, deep_merge

    base_dir = os.path.join(os.environ.get("BREACHING_REFERENCE", "/root/reference"), "breaching", "config", "attack")
    with open(os.path.join(base_dir, "_default_optimization_attack.yaml")) as f:
        base = yaml.safe_load(f)
    with open(os.path.join(base_dir, f"{name}.yaml")) as f:
        spec = yaml.safe_load(f)
    spec.pop("defaults", None)
    cfg = AttrDict(_coerce(deep_merge(base, spec)))
    return _apply_overrides(cfg, overrides)


def _data_cfg(name):
    import yaml

    from breaching_amd.config import AttrDict

    path = os.path.join(os.environ.get("BREACHING_REFERENCE", "/root/reference"), "breaching", "config", "case", "data", f"{name}.yaml")
    with open(path) as f:
        spec = yaml.safe_load(f)
    spec.pop("defaults", None)
    return AttrDict(_coerce(spec))


def golden_configs():
    """The merged attack configs themselves (pins breaching_amd/config.py against the YAML files)."""
    import json

    out = {name: _cfg(name) for name in ("invertinggradients", "seethroughgradients", "tag", "deepleakage", "modern", "legacy",
                                       "clsattack", "beyondinfering", "wei", "sanitycheck")}
    data = {name: {k: _data_cfg(name)[k] for k in ("modality", "task", "classes", "shape", "mean", "std")}
            for name in ("CIFAR10", "ImageNet")}
    with open(os.path.join(GOLDEN, "configs.json"), "w") as f:
        json.dump(dict(attacks=out, data=data), f, indent=1, sort_keys=True)


def golden_kernels():
    breaching = import_reference()
    from breaching.attacks.auxiliaries import objectives as O
    from breaching.attacks.auxiliaries import regularizers as R

    rng = np.random.default_rng(1234)
    out = {}
    rec_np = [rng.standard_normal(s).astype(np.float32) * 0.3 for s in LIST_SHAPES]
    data_np = [(r + rng.standard_normal(r.shape).astype(np.float32) * 0.1) for r in rec_np]
    for d in data_np:  # exact zeros / tiny values exercise the mask of MaskedCosineSimilarity and sign(0)
        flat = d.reshape(-1)
        flat[:: 11] = 0.0
        flat[1:: 13] = 5e-7
    rec_np[1][0] = data_np[1][0]  # r - d == 0 somewhere
    for i, (r, d) in enumerate(zip(rec_np, data_np)):
        out[f"rec_{i}"], out[f"data_{i}"] = r, d
    setup = dict(dtype=torch.float32, device=torch.device("cpu"))
    specs = {
        "cosine-similarity": dict(scale=1.0),
        "masked-cosine-similarity": dict(scale=0.7),
        "fast-cosine-similarity": dict(scale=1.3),
        "angular": dict(scale=2.0),
        "euclidean": dict(scale=1e-2),
        "l1": dict(scale=0.5),
        "tag-euclidean": dict(scale=1.5, tag_scale=0.1, scale_scheme="linear"),
        "tag-euclidean/exp": dict(scale=1.0, tag_scale=0.25, scale_scheme="exp"),
    }
    for name, kw in specs.items():
        obj = O.objective_lookup[name.split("/")[0]](**kw)
        rec = [torch.tensor(r, requires_grad=True) for r in rec_np]
        data = [torch.tensor(d) for d in data_np]
        value = obj.gradient_based_loss(rec, data)
        grads = torch.autograd.grad(value.sum(), rec)
        key = name.replace("/", "_")
        out[f"{key}__value"] = value.detach().numpy().reshape(-1)
        for i, g in enumerate(grads):
            out[f"{key}__grad_{i}"] = g.numpy()
    # total variation / norm
    x_np = rng.standard_normal((2, 3, 13, 9)).astype(np.float32)
    x_np[0, 0, 3, 3:6] = 0.25  # flat run -> zero differences -> sign(0)
    out["tv_x"] = x_np
    tv_specs = {"p1q1": dict(scale=0.2, inner_exp=1, outer_exp=1, double_opponents=False),
                "p1q1_opp": dict(scale=0.3, inner_exp=1, outer_exp=1, double_opponents=True),
                "p2q05_opp": dict(scale=0.1, inner_exp=2, outer_exp=0.5, double_opponents=True),
                "p2q05": dict(scale=1e-4, inner_exp=2, outer_exp=0.5, double_opponents=False)}
    for key, kw in tv_specs.items():
        x = torch.tensor(x_np, requires_grad=True)
        value = R.TotalVariation(setup, **kw)(x)
        (g,) = torch.autograd.grad(value, x)
        out[f"tv_{key}__value"], out[f"tv_{key}__grad"] = value.detach().numpy().reshape(-1), g.numpy()
    for key, kw in {"p2": dict(scale=1e-2, pnorm=2), "p3": dict(scale=0.3, pnorm=3.0)}.items():
        x = torch.tensor(x_np, requires_grad=True)
        value = R.NormRegularization(setup, **kw)(x)
        (g,) = torch.autograd.grad(value, x)
        out[f"norm_{key}__value"], out[f"norm_{key}__grad"] = value.detach().numpy().reshape(-1), g.numpy()
    # deep inversion statistic on a single BN layer + the full regulariser on a two-BN model
    for tag, (B, C, H, W) in {"a": (3, 5, 6, 4), "b": (2, 300, 7, 7)}.items():
        feat = rng.standard_normal((B, C, H, W)).astype(np.float32) * 1.7 + 0.3
        bn = torch.nn.BatchNorm2d(C)
        bn.running_mean.copy_(torch.tensor(rng.standard_normal(C).astype(np.float32) * 0.2))
        bn.running_var.copy_(torch.tensor(rng.random(C).astype(np.float32) + 0.5))
        bn.eval()
        model = torch.nn.Sequential(bn)
        reg = R.DeepInversion(setup, scale=1.0, first_bn_multiplier=1)
        reg.initialize([model])
        x = torch.tensor(feat, requires_grad=True)
        model(x)
        value = reg(x)
        (g,) = torch.autograd.grad(value, x)
        out[f"bn_{tag}__x"], out[f"bn_{tag}__rm"], out[f"bn_{tag}__rv"] = feat, bn.running_mean.numpy().copy(), bn.running_var.numpy().copy()
        out[f"bn_{tag}__value"], out[f"bn_{tag}__grad"] = value.detach().numpy().reshape(-1), g.numpy()
    # orthogonality regulariser (regularizers.py:156-181) and the PSNR metric (analysis/metrics.py:108-130) -- drawn last so
    # that the vectors above keep their values
    from breaching.analysis import metrics as M

    xo = rng.standard_normal((4, 3, 9, 11)).astype(np.float32)
    out["orth_x"] = xo
    x = torch.tensor(xo, requires_grad=True)
    value = R.OrthogonalityRegularization(setup, scale=0.1)(x)
    (g,) = torch.autograd.grad(value, x)
    out["orth__value"], out["orth__grad"] = value.detach().numpy().reshape(-1), g.numpy()
    mean, std = np.asarray([0.485, 0.456, 0.406], dtype=np.float32), np.asarray([0.229, 0.224, 0.225], dtype=np.float32)
    truth = (rng.random((3, 3, 10, 7)).astype(np.float32) - mean[None, :, None, None]) / std[None, :, None, None]
    recon = truth + rng.standard_normal(truth.shape).astype(np.float32) * np.asarray([0.05, 0.5, 3.0], dtype=np.float32)[:, None, None, None]
    dm, ds = torch.tensor(mean)[None, :, None, None], torch.tensor(std)[None, :, None, None]
    rec_denorm = torch.clamp(torch.tensor(recon) * ds + dm, 0, 1)  # analysis.py:228-229
    gt_denorm = torch.clamp(torch.tensor(truth) * ds + dm, 0, 1)
    avg, best = M.psnr_compute(rec_denorm, gt_denorm, factor=1)
    out.update(psnr_rec=recon, psnr_truth=truth, psnr_mean=mean, psnr_std=std, psnr__avg_max=np.asarray([avg, best], dtype=np.float64))
    np.savez_compressed(os.path.join(GOLDEN, "kernels.npz"), **out)


def golden_schedules():
    import_reference()
    from breaching.attacks.auxiliaries.common import optimizer_lookup

    out = {}
    cases = {"step-lr_0_24000_0.1": ("step-lr", 0, 24000, 0.1), "step-lr_0_100_0.1": ("step-lr", 0, 100, 0.1),
             "cosine-decay_50_20000_0.1": ("cosine-decay", 50, 20000, 0.1), "linear_50_1000_0.05": ("linear", 50, 1000, 0.05),
             "none_0_400_1.0": (None, 0, 400, 1.0), "cosine-decay_0_300_0.1": ("cosine-decay", 0, 300, 0.1),
             "step-lr_5_3_0.1": ("step-lr", 5, 3, 0.1)}
    for key, (sched, warm, max_it, step) in cases.items():
        p = [torch.zeros(1, requires_grad=True)]
        p[0].grad = torch.zeros(1)
        opt, sch = optimizer_lookup(p, "adam", step, scheduler=sched, warmup=warm, max_iterations=max_it)
        lrs = []
        for _ in range(max_it):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[key] = np.asarray(lrs, dtype=np.float64)
    np.savez_compressed(os.path.join(GOLDEN, "schedules.npz"), **out)


def _run_reference_attack(cfg, case, x0, dryrun=False, seed=7, record_candidates=None, record_at=None):
    """Run the unmodified reference attacker.  ``record_candidates`` (a list) receives the candidate the reference
    evaluates at every iteration -- captured by wrapping the objective's bound ``forward`` from outside (test-harness
    instrumentation; no reference file is touched).  With ``record_at`` (a set of iteration indices) ``record_candidates``
    is a dict iteration -> candidate instead, holding only those iterations."""
    breaching = import_reference()
    setup = dict(device=torch.device("cpu"), dtype=torch.float)
    attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, setup)
    if record_candidates is not None:
        inner = attacker.objective.forward

        calls = [0]

        def spy(model, gradient_data, candidate, labels):
            if record_at is None:
                record_candidates.append(candidate.detach().clone())
            elif calls[0] in record_at:
                record_candidates[calls[0]] = candidate.detach().clone()
            calls[0] += 1
            return inner(model, gradient_data, candidate, labels)

        attacker.objective.forward = spy
    torch.manual_seed(seed)
    shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
    rec, stats = attacker.reconstruct(case.server_payload, shared, {}, initial_data=x0, dryrun=dryrun)
    return rec, stats


def _ulp_perturb(x, ulps, gen):
    """x moved by a random integer in [-ulps, ulps] units in the last place, element-wise."""
    from breaching_amd.cases import ulp_perturb

    return ulp_perturb(x, ulps, gen)


def _kink_sensitivity(case, cfg, candidates, ulps=16, trials=3, seed=0):
    """Relative change of the reference objective when the candidate it was evaluated at moves by a few ulp.

    ReLU / max-pool kinks make the gradient-matching objective discontinuous; when a pre-activation sits within rounding
    noise of zero, *any* other arithmetic (another BLAS, a GPU) lands on the other side and the sign-Adam trajectory
    forks.  The per-iteration value recorded here tells the parity tests where the reference's own trajectory stops
    being reproducible."""
    from oracle import restate

    gen = torch.Generator().manual_seed(seed)
    labels = case.shared_data[0]["metadata"]["labels"]
    regs = cfg.regularization if cfg.regularization is not None else {}

    def total(x):
        x = x.detach().clone().requires_grad_(True)
        loss = case.loss_fn(case.model(x), labels)
        g = torch.autograd.grad(loss, tuple(case.model.parameters()), create_graph=True)
        value = restate.gradient_objective(cfg.objective.type, g, case.shared_data[0]["gradients"], cfg.objective)
        if "total_variation" in regs:
            value = value + restate.total_variation(x, **regs["total_variation"])
        return float(value.detach())

    out = []
    for x in candidates:
        base = total(x)
        out.append(max(abs(total(_ulp_perturb(x, ulps, gen)) - base) / abs(base) for _ in range(trials)))
    return np.asarray(out, dtype=np.float64)


def _twin_runs(cfg, case, x0, n_twins, ulps=16, seed=123):
    """The reference attacked from starting points a few ulp away from x0: its own irreproducibility envelope."""
    from breaching_amd.cases import psnr

    gen = torch.Generator().manual_seed(seed)
    hists, psnrs, opts = [], [], []
    for _ in range(n_twins):
        rec, stats = _run_reference_attack(cfg, case, _ulp_perturb(x0, ulps, gen))
        hists.append(np.asarray(stats["Trial_0_Val"], dtype=np.float64))
        psnrs.append(psnr(rec["data"], case.true_user_data["data"], case.data_cfg))
        opts.append(stats["opt_value"])
    return np.stack(hists), np.asarray(psnrs), np.asarray(opts)


def _attack_record(cfg, case, x0, rec, stats, crop=None):
    from breaching_amd.cases import parameter_checksum, psnr

    data = rec["data"].detach()
    out = dict(history=np.asarray(stats["Trial_0_Val"], dtype=np.float64), opt_value=np.float64(stats["opt_value"]),
               psnr=np.float64(psnr(data, case.true_user_data["data"], case.data_cfg)),
               model_checksum=np.float64(parameter_checksum(case.model)),
               labels=rec["labels"].numpy(), rec_mean=np.float64(data.double().mean()), rec_std=np.float64(data.double().std()),
               grad0_checksum=np.float64(case.shared_data[0]["gradients"][0].double().sum()))
    out["rec"] = data.numpy() if crop is None else data[..., :crop, :crop].numpy()
    return out


def golden_convnet():
    from breaching_amd.cases import build_case, initial_candidate

    torch.set_num_threads(8)
    case = build_case("convnet", "CIFAR10", 1)
    cfg = _cfg("invertinggradients", ["optim.max_iterations=100", "optim.callback=50"])
    # choose the starting point whose reference trajectory stays clear of ReLU kinks the longest
    best = None
    for x0_seed in (6,):
        x0 = initial_candidate(case.data_cfg, 1, seed=x0_seed)
        cands = []
        rec, stats = _run_reference_attack(cfg, case, x0, record_candidates=cands)
        sens = _kink_sensitivity(case, cfg, cands[:100])
        unstable = np.nonzero(sens > 1e-5)[0]
        prefix = int(unstable[0]) if len(unstable) else 100
        print(f"  x0 seed {x0_seed}: reproducible prefix {prefix} iterations", flush=True)
        if best is None or prefix > best[0]:
            best = (prefix, x0_seed, x0, rec, stats, sens, cands)
        if prefix >= 60:
            break
    prefix, x0_seed, x0, rec, stats, sens, cands = best
    out = _attack_record(cfg, case, x0, rec, stats)
    out.update(x0_seed=np.int64(x0_seed), stable_prefix=np.int64(prefix), kink_sensitivity=sens)
    # teacher-forcing points: the reference's own iterates (loss at x_k is history[k])
    forced = [k for k in (0, 3, 10, 30, 60, 99) if sens[k] <= 1e-6]
    out.update(forced_k=np.asarray(forced, dtype=np.int64), forced_x=np.stack([cands[k].numpy() for k in forced]))
    twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 3)
    out.update(twin_history=twins, twin_psnr=twin_psnr, twin_opt_value=twin_opt)

    cfg = _cfg("invertinggradients", ["optim.max_iterations=100"])
    rec, stats = _run_reference_attack(cfg, case, x0, dryrun=True)
    out.update({f"dryrun_{k}": v for k, v in _attack_record(cfg, case, x0, rec, stats).items()})
    # a second objective family through the full loop: euclidean + norm prior, soft sign, cosine decay with warm-up
    cfg = _cfg("invertinggradients", ["objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft",
                                      "optim.step_size_decay=cosine-decay", "optim.warmup=5", "optim.max_iterations=40",
                                      "restarts.scoring=euclidean", "regularization.norm.scale=0.01",
                                      "regularization.norm.pnorm=2", "optim.callback=20"])
    cands = []
    rec, stats = _run_reference_attack(cfg, case, x0, record_candidates=cands)
    out.update({f"l2soft_{k}": v for k, v in _attack_record(cfg, case, x0, rec, stats).items()})
    twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 2)
    out.update(l2soft_twin_history=twins, l2soft_twin_psnr=twin_psnr, l2soft_twin_opt_value=twin_opt)
    np.savez_compressed(os.path.join(GOLDEN, "attack_convnet.npz"), **out)


def golden_multiquery():
    """Two server queries answered by one user (two model states, two gradient lists): the objective is summed over the
    (model, gradient) pairs (optimization_based_attack.py:152-155), one packed plan per list on the HIP side.  Smooth
    configuration (euclidean, soft sign) so that the whole trajectory is comparable at 1e-4."""
    from breaching_amd.cases import build_multi_query_case, initial_candidate

    torch.set_num_threads(8)
    case = build_multi_query_case(2)
    cfg = _cfg("invertinggradients", ["objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft",
                                      "optim.max_iterations=16", "restarts.scoring=euclidean", "optim.callback=8"])
    x0 = initial_candidate(case.data_cfg, 2, seed=6)
    rec, stats = _run_reference_attack(cfg, case, x0)
    out = _attack_record(cfg, case, x0, rec, stats)
    twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 2)
    out.update(twin_history=twins, twin_psnr=twin_psnr, twin_opt_value=twin_opt,
               grad1_checksum=np.float64(case.shared_data[1]["gradients"][0].double().sum()))
    # one query alone must give a different trajectory: the fixture really exercises the sum over queries
    single = type(case)(case)
    single.server_payload, single.shared_data = case.server_payload[:1], case.shared_data[:1]
    _, stats1 = _run_reference_attack(cfg, single, x0)
    out["single_query_history"] = np.asarray(stats1["Trial_0_Val"], dtype=np.float64)
    np.savez_compressed(os.path.join(GOLDEN, "attack_multiquery.npz"), **out)


def golden_variants():
    """Further configurations through the full loop on ConvNet/CIFAR-10: the `legacy` family (soft sign, double-opponent
    TV with p=2 q=0.5, feature regulariser, DeepInversion) in the fused loop and the `wei` family (euclidean + task
    regularisation, L-BFGS) in the generic torch.optim loop."""
    from breaching_amd.cases import build_case, initial_candidate

    torch.set_num_threads(8)
    case = build_case("convnet", "CIFAR10", 2)
    x0 = initial_candidate(case.data_cfg, 2, seed=6)
    out = {}
    cfg = _cfg("legacy", ["optim.max_iterations=30", "optim.callback=10", "regularization.deep_inversion.scale=0.001"])
    rec, stats = _run_reference_attack(cfg, case, x0)
    out.update({f"legacy_{k}": v for k, v in _attack_record(cfg, case, x0, rec, stats).items()})
    twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 2)
    out.update(legacy_twin_history=twins, legacy_twin_psnr=twin_psnr, legacy_twin_opt_value=twin_opt)
    cfg = _cfg("wei", ["optim.max_iterations=4", "optim.callback=2"])
    rec, stats = _run_reference_attack(cfg, case, x0)
    out.update({f"wei_{k}": v for k, v in _attack_record(cfg, case, x0, rec, stats).items()})
    twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 2)
    out.update(wei_twin_history=twins, wei_twin_psnr=twin_psnr, wei_twin_opt_value=twin_opt)
    np.savez_compressed(os.path.join(GOLDEN, "attack_variants.npz"), **out)


def golden_fedavg():
    """FedAvg multi-step objective (objectives.py:48-72): 2 local SGD steps of 2 images each, soft-sign cosine attack."""
    from breaching_amd.cases import build_fedavg_case, initial_candidate

    torch.set_num_threads(8)
    case = build_fedavg_case()
    x0 = initial_candidate(case.data_cfg, 4, seed=6)
    cfg = _cfg("invertinggradients", ["optim.max_iterations=20", "optim.callback=10", "optim.signed=soft"])
    rec, stats = _run_reference_attack(cfg, case, x0)
    out = _attack_record(cfg, case, x0, rec, stats)
    twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 2)
    out.update(twin_history=twins, twin_psnr=twin_psnr, twin_opt_value=twin_opt)
    # plain Adam (no sign at all), smaller step: a smooth trajectory, strict tolerance over the whole horizon
    cfg = _cfg("invertinggradients", ["optim.max_iterations=30", "optim.callback=10", "optim.signed=null", "optim.step_size=0.01"])
    rec, stats = _run_reference_attack(cfg, case, x0)
    out.update({f"plain_{k}": v for k, v in _attack_record(cfg, case, x0, rec, stats).items()})
    twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 2)
    out.update(plain_twin_history=twins, plain_twin_psnr=twin_psnr, plain_twin_opt_value=twin_opt)
    np.savez_compressed(os.path.join(GOLDEN, "attack_fedavg.npz"), **out)


def golden_labels():
    """Label recovery strategies (base_attack.py:305-475) of the reference on a 6-image ConvNet update with repeated labels."""
    breaching = import_reference()
    from breaching_amd.cases import build_case

    torch.set_num_threads(8)
    out = {}
    for tag, n in (("b6", 6), ("b1", 1)):
        case = build_case("convnet", "CIFAR10", n, seed_data=5)
        out[f"{tag}_true"] = case.true_user_data["labels"].numpy()
        for strategy in ("iDLG", "analytic", "yin", "wainakh-simple", "bias-corrected"):
            cfg = _cfg("invertinggradients", [f"label_strategy={strategy}"])
            attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
            shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"], labels=None)) for d in case.shared_data]
            torch.manual_seed(0)
            _, labels, _ = attacker.prepare_attack(case.server_payload, shared)
            out[f"{tag}_{strategy}"] = labels.numpy()
    np.savez_compressed(os.path.join(GOLDEN, "labels.npz"), **out)


def golden_dlg():
    """Deep Leakage from Gradients (deepleakage.yaml): joint data+label optimisation with L-BFGS on ConvNet -- the generic
    torch.optim loop of the joint attacker.  Labels withheld; initial data and labels drawn from the seeded CPU generator.

    L-BFGS amplifies rounding differences within its first step (20 closure evaluations, curvature pairs from differences
    of nearly equal gradients), so besides the free-running history the fixture holds TEACHER-FORCING points: the
    (candidate, softmaxed label candidate) pairs the reference's closure evaluated at a few of its ~60 objective calls, with
    the reference objective's value there and its gradient with respect to both (recomputed with the reference's own
    objective module at exactly those points)."""
    breaching = import_reference()
    from breaching_amd.cases import build_case, parameter_checksum, psnr

    torch.set_num_threads(8)
    case = build_case("convnet", "CIFAR10", 1, provide_labels=False)
    cfg = _cfg("deepleakage", ["optim.max_iterations=3", "optim.callback=1"])
    attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
    calls = []
    inner = attacker.objective.forward
    seen = {}

    def spy(model, gradient_data, candidate, labels):
        value, task_loss = inner(model, gradient_data, candidate, labels)
        calls.append((candidate.detach().clone(), labels.detach().clone(), float(value.detach())))
        seen["model"], seen["gradient_data"] = model, gradient_data
        return value, task_loss

    attacker.objective.forward = spy
    torch.manual_seed(5)
    rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {})
    attacker.objective.forward = inner
    n_loop = len(calls) - 1  # the last call is `_score_trial` on a fresh objective? (no: scoring builds its own module) -- keep all
    picks = sorted({0, 1, 7, 20, 21, 40, len(calls) - 1} & set(range(len(calls))))
    forced = dict(x=[], p=[], value=[], gx=[], gp=[], call=[], sens_x=[], sens_p=[], sens_v=[])

    def evaluate(x, p):
        xq, pq = x.clone().requires_grad_(True), p.clone().requires_grad_(True)
        v, _ = attacker.objective(seen["model"], seen["gradient_data"], xq, pq)
        gx, gp = torch.autograd.grad(v, [xq, pq])
        return float(v.detach()), gx, gp

    gen = torch.Generator().manual_seed(77)
    for k in picks:
        x, p, value = calls[k]
        v, gx, gp = evaluate(x, p)
        assert abs(v - value) <= 1e-6 * abs(value)
        # kink sensitivity of the reference's OWN closure at this point: the iterates L-BFGS visits late in the run sit on
        # ReLU / max-pool kinks, where moving the candidate by 16 ulp flips units and changes the gradient by a finite amount
        sens_v = sens_x = sens_p = 0.0
        for _ in range(4):
            v2, gx2, gp2 = evaluate(_ulp_perturb(x, 16, gen), p)
            sens_v = max(sens_v, abs(v2 - v) / abs(v))
            sens_x = max(sens_x, float((gx2 - gx).abs().max() / gx.abs().max()))
            sens_p = max(sens_p, float((gp2 - gp).abs().max() / gp.abs().max()))
        for key, item in zip(("x", "p", "value", "gx", "gp", "call", "sens_x", "sens_p", "sens_v"),
                             (x.numpy(), p.numpy(), value, gx.numpy(), gp.numpy(), k, sens_x, sens_p, sens_v)):
            forced[key].append(item)
    out = dict(history=np.asarray(stats["Trial_0_Val"], dtype=np.float64), opt_value=np.float64(stats["opt_value"]),
               rec=rec["data"].numpy(), labels=rec["labels"].numpy(), seed=np.int64(5),
               psnr=np.float64(psnr(rec["data"], case.true_user_data["data"], case.data_cfg)),
               model_checksum=np.float64(parameter_checksum(case.model)), n_objective_calls=np.int64(len(calls)),
               forced_call=np.asarray(forced["call"], dtype=np.int64), forced_x=np.stack(forced["x"]), forced_p=np.stack(forced["p"]),
               forced_value=np.asarray(forced["value"], dtype=np.float64), forced_gx=np.stack(forced["gx"]),
               forced_gp=np.stack(forced["gp"]), forced_sensitivity_x=np.asarray(forced["sens_x"]),
               forced_sensitivity_p=np.asarray(forced["sens_p"]), forced_sensitivity_value=np.asarray(forced["sens_v"]))
    np.savez_compressed(os.path.join(GOLDEN, "attack_dlg.npz"), **out)


def golden_resnet18():
    from breaching_amd.cases import build_case, initial_candidate

    torch.set_num_threads(8)
    case = build_case("resnet18", "ImageNet", 1)
    x0 = initial_candidate(case.data_cfg, 1)
    # 24k iterations are out of reach on CPU.  The first step-lr milestone of the full run sits at iteration 8998, so a
    # 20-iteration run at constant step size reproduces the first 20 iterations of the real schedule exactly.
    cfg = _cfg("invertinggradients", ["optim.max_iterations=20", "optim.step_size_decay=null", "optim.callback=5"])
    cands = []
    rec, stats = _run_reference_attack(cfg, case, x0, record_candidates=cands)
    out = _attack_record(cfg, case, x0, rec, stats, crop=32)
    sens = _kink_sensitivity(case, cfg, cands[:20], trials=2)
    unstable = np.nonzero(sens > 1e-5)[0]
    out.update(kink_sensitivity=sens, stable_prefix=np.int64(int(unstable[0]) if len(unstable) else 20))
    forced = [k for k in (19, 18, 17) if sens[k] <= 1e-6][:1]
    out.update(forced_k=np.asarray(forced, dtype=np.int64), forced_x=np.stack([cands[k].numpy() for k in forced]))
    twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 2)
    out.update(twin_history=twins, twin_psnr=twin_psnr, twin_opt_value=twin_opt)
    np.savez_compressed(os.path.join(GOLDEN, "attack_resnet18.npz"), **out)


# ---- BASELINE configs[1] at a long horizon -------------------------------------------------------------------------
LONG_ITERS = 1000  # step-lr milestones of the reference fall at 374 / 625 / 875 (common.py:22-27: max_it // 2.667, 1.6, 1.142)
LONG_TWINS = 7
LONG_FORCED = (100, 250, 380, 500, 630, 750, 880, 990)  # teacher-forcing targets: two after each milestone
LONG_SEED = 123


def long_start(data_cfg, idx):
    """Starting point of long run `idx`: 0 = the nominal x0, idx > 0 = x0 moved by <= 16 ulp (seeded per twin)."""
    from breaching_amd.cases import initial_candidate, ulp_perturb

    x0 = initial_candidate(data_cfg, 1)
    if idx == 0:
        return x0
    return ulp_perturb(x0, 16, torch.Generator().manual_seed(LONG_SEED + idx))


def _resnet18_long_worker(idx, out_path, threads=2):
    """One 1000-iteration run of the unmodified reference on ResNet-18 / 224 x 224 with the real (scaled) step-lr schedule."""
    from breaching_amd.cases import build_case, psnr

    torch.set_num_threads(threads)
    case = build_case("resnet18", "ImageNet", 1)
    cfg = _cfg("invertinggradients", [f"optim.max_iterations={LONG_ITERS}", "optim.callback=100"])
    x0 = long_start(case.data_cfg, idx)
    want = set()
    if idx == 0:
        for k in LONG_FORCED:
            want.update(range(k, k + 4))  # a few neighbours: the one clear of ReLU kinks is kept
    cands = {}
    rec, stats = _run_reference_attack(cfg, case, x0, record_candidates=cands, record_at=want)
    data = rec["data"].detach()
    out = dict(history=np.asarray(stats["Trial_0_Val"], dtype=np.float64), opt_value=np.float64(stats["opt_value"]),
               psnr=np.float64(psnr(data, case.true_user_data["data"], case.data_cfg)),
               rec_mean=np.float64(data.double().mean()), rec_std=np.float64(data.double().std()))
    if idx == 0:
        from breaching_amd.cases import parameter_checksum

        out.update(rec=data[..., :32, :32].numpy(), model_checksum=np.float64(parameter_checksum(case.model)),
                   labels=rec["labels"].numpy())
        ks, xs, sens_kept = [], [], []
        for k in LONG_FORCED:
            group = [j for j in range(k, k + 4) if j in cands]
            sens = _kink_sensitivity(case, cfg, [cands[j] for j in group], trials=2)
            j = int(np.argmin(sens))
            print(f"  forced target {k}: sensitivities {sens} -> keep {group[j]}", flush=True)
            # kept whatever the sensitivity: late iterates sit on ReLU kinks (16 ulp on x move the reference's own objective
            # by up to 4e-4 there), and the parity test widens its tolerance to 10x the recorded sensitivity for them
            ks.append(group[j]), xs.append(cands[group[j]].numpy()), sens_kept.append(sens[j])
        out.update(forced_k=np.asarray(ks, dtype=np.int64), forced_x=np.stack(xs), forced_sensitivity=np.asarray(sens_kept))
    np.savez(out_path, **out)


def golden_resnet18_long(parallel=4):
    """BASELINE configs[1] pinned at a long horizon: 1 + LONG_TWINS runs of the unmodified reference, 1000 iterations each
    on the step-lr schedule (three milestones inside).  The twins (starting points <= 16 ulp apart) give the reference's
    OWN end-of-run distribution of final loss / opt_value / PSNR -- hard-sign Adam on a ReLU net is chaotic, so that
    distribution, not one trajectory, is what a second implementation can be held to; single points of the nominal
    trajectory are additionally pinned by teacher forcing at iterates after every milestone."""
    import subprocess
    import tempfile

    tmp = os.environ.get("GOLDEN_LONG_DIR") or tempfile.mkdtemp(prefix="golden_long_")  # finished runs found there are kept
    pending = [i for i in range(LONG_TWINS + 1) if not os.path.exists(os.path.join(tmp, f"run{i}.npz"))]
    running = {}
    while pending or running:
        while pending and len(running) < parallel:
            idx = pending.pop(0)
            path = os.path.join(tmp, f"run{idx}.npz")
            running[idx] = (subprocess.Popen([sys.executable, "-m", "oracle.make_golden", "--long-worker", str(idx), path], cwd=ROOT), path)
            print(f"  started long run {idx}", flush=True)
        for idx, (proc, path) in list(running.items()):
            if proc.poll() is not None:
                if proc.returncode != 0:
                    raise RuntimeError(f"long run {idx} failed")
                del running[idx]
                print(f"  long run {idx} finished", flush=True)
        import time

        time.sleep(5)
    main = dict(np.load(os.path.join(tmp, "run0.npz")))
    twins = [np.load(os.path.join(tmp, f"run{i}.npz")) for i in range(1, LONG_TWINS + 1)]
    main.update(twin_history=np.stack([t["history"] for t in twins]), twin_psnr=np.asarray([t["psnr"] for t in twins]),
                twin_opt_value=np.asarray([t["opt_value"] for t in twins]),
                twin_rec_mean=np.asarray([t["rec_mean"] for t in twins]), twin_rec_std=np.asarray([t["rec_std"] for t in twins]),
                iterations=np.int64(LONG_ITERS), twin_seed=np.int64(LONG_SEED))
    main["forced_x"] = main["forced_x"].astype(np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "attack_resnet18_long.npz"), **main)


def reference_more_starts_report(first=8, last=31, out_name="r5_reference_cpu_1000its_more_starts.json"):
    """Round 5: a LARGER sample of the reference's own end-of-run distribution at 1000 iterations (starts idx 8 ... 31, each <= 16 ulp
    from x0, seeded like the eight of attack_resnet18_long.npz), produced by `--long-worker IDX oracle/_long/longIDX.npz`; together
    with the fixture's eight: what the HIP distribution over 64 starts (profiles/r5_hip_64starts_1000its.json) is compared with.
    Written under profiles/ (evidence, not a test fixture)."""
    import json

    gold = np.load(os.path.join(GOLDEN, "attack_resnet18_long.npz"))
    hists = [gold["history"]] + list(gold["twin_history"])
    opts = [float(gold["opt_value"])] + [float(v) for v in gold["twin_opt_value"]]
    psnrs = [float(gold["psnr"])] + [float(v) for v in gold["twin_psnr"]]
    used = list(range(8))
    for idx in range(first, last + 1):
        path = os.path.join(FULL_DIR, f"long{idx}.npz")
        if os.path.exists(path):
            run = np.load(path)
            hists.append(run["history"]), opts.append(float(run["opt_value"])), psnrs.append(float(run["psnr"])), used.append(idx)
    marks = (373, 380, 624, 630, 874, 880, 999)
    cols = {f"loss@{m}": np.asarray([h[m] for h in hists], dtype=np.float64) for m in marks}
    cols["opt_value"], cols["psnr"] = np.asarray(opts), np.asarray(psnrs)
    report = dict(n=len(used), starts=used, iterations=1000,
                  source="unmodified reference on CPU (oracle/make_golden.py --long-worker), starts <= 16 ulp from x0",
                  quantities={k: dict(mean=float(v.mean()), sd=float(v.std(ddof=1)), values=[round(float(x), 6) for x in v]) for k, v in cols.items()})
    with open(os.path.join(ROOT, "profiles", out_name), "w") as f:
        json.dump(report, f, indent=1)
    for k, v in cols.items():
        print(f"  {k:10s} reference {v.mean():.6f} +- {v.std(ddof=1):.6f} (n = {len(v)}, standard error {v.std(ddof=1) / np.sqrt(len(v)):.6f})")
    return report


# ---- BASELINE configs[1] at its STATED horizon: 24 000 iterations --------------------------------------------------
FULL_ITERS = 24000  # invertinggradients.yaml:19; step-lr milestones at 8998 / 15000 / 21015 (common.py:22-27)
FULL_TWINS = 7      # + the nominal start: eight unmodified-reference runs (round 4: three; 2.5 h each at 2 threads, 4.7 h at 1)
FULL_FORCED = (100, 1000, 5000, 9100, 15100, 21100, 23990)  # teacher-forcing targets: early, mid, after each milestone, end
FULL_DIR = os.path.join(HERE, "_long")  # scratch (git-ignored): partial and finished runs live here


def _input_gradient(case, cfg, x):
    """d total / dx of the reference's objective (restated: objective + TV) at x, fp32 on CPU -- used only to measure how
    far the REFERENCE's own step direction moves when x moves by a few ulp."""
    from oracle import restate

    labels = case.shared_data[0]["metadata"]["labels"]
    regs = cfg.regularization if cfg.regularization is not None else {}
    x = x.detach().clone().requires_grad_(True)
    loss = case.loss_fn(case.model(x), labels)
    g = torch.autograd.grad(loss, tuple(case.model.parameters()), create_graph=True)
    value = restate.gradient_objective(cfg.objective.type, g, case.shared_data[0]["gradients"], cfg.objective)
    if "total_variation" in regs:
        value = value + restate.total_variation(x, **regs["total_variation"])
    (gx,) = torch.autograd.grad(value, x)
    return float(value.detach()), gx


def _resnet18_full_worker(idx, out_path, threads=2, iters=FULL_ITERS, forced=FULL_FORCED):
    """One full-horizon run of the unmodified reference (optimization_based_attack.py:110-143) on ResNet-18 / 224 x 224.

    The nominal run (idx 0) additionally records, at the teacher-forcing targets, the candidate x_k the closure was
    evaluated at, the raw d total/dx that reached ``candidate.grad`` (tensor hook on the leaf) and the map after
    ``candidate.grad.sign_()`` (:181-182) -- by wrapping the bound ``_compute_objective`` of the attacker INSTANCE from
    outside; no reference file is touched.  A partial file is rewritten every 500 iterations."""
    from breaching_amd.cases import build_case, parameter_checksum, psnr

    torch.set_num_threads(threads)
    breaching = import_reference()
    case = build_case("resnet18", "ImageNet", 1)
    cfg = _cfg("invertinggradients", [f"optim.max_iterations={iters}", "optim.callback=1000"])
    x0 = long_start(case.data_cfg, idx)
    want = set()
    if idx == 0:
        for k in forced:
            want.update(range(k, min(k + 3, iters)))
    setup = dict(device=torch.device("cpu"), dtype=torch.float)
    attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, setup)
    inner_compute = attacker._compute_objective
    rec_x, rec_raw, rec_sign = {}, {}, {}
    state = dict(hooked=None, it=-1, stats=None)
    import time as _time

    t_start = _time.time()

    def wrapped_compute(candidate, labels, rec_model, optimizer, shared_data, iteration):
        closure = inner_compute(candidate, labels, rec_model, optimizer, shared_data, iteration)
        if state["hooked"] is not candidate:
            state["hooked"] = candidate

            def grab(grad):
                if state["it"] in want:
                    rec_raw[state["it"]] = grad.detach().clone()

            candidate.register_hook(grab)

        def spy_closure():
            state["it"] = iteration
            if iteration in want:
                rec_x[iteration] = candidate.detach().clone()
            value = closure()
            if iteration in want:
                rec_sign[iteration] = candidate.grad.detach().clone()
            return value

        if iteration % 500 == 0 and iteration > 0:
            print(f"  run {idx}: iteration {iteration}, {(_time.time() - t_start) / iteration:.3f} s/it", flush=True)
        return spy_closure

    attacker._compute_objective = wrapped_compute
    torch.manual_seed(7)
    shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
    rec, stats = attacker.reconstruct(case.server_payload, shared, {}, initial_data=x0)
    data = rec["data"].detach()
    out = dict(history=np.asarray(stats["Trial_0_Val"], dtype=np.float64), opt_value=np.float64(stats["opt_value"]),
               psnr=np.float64(psnr(data, case.true_user_data["data"], case.data_cfg)),
               rec_mean=np.float64(data.double().mean()), rec_std=np.float64(data.double().std()),
               seconds=np.float64(_time.time() - t_start), threads=np.int64(threads))
    if idx == 0:
        out.update(rec=data[..., :32, :32].numpy(), model_checksum=np.float64(parameter_checksum(case.model)),
                   labels=rec["labels"].numpy())
        np.savez(out_path + ".raw.npz", **out, **{f"x_{k}": v.numpy() for k, v in rec_x.items()},
                 **{f"g_{k}": v.numpy() for k, v in rec_raw.items()}, **{f"s_{k}": v.numpy() for k, v in rec_sign.items()})
        ks, xs, gs, ss, sens_kept, agree_kept, wagree_kept = [], [], [], [], [], [], []
        gen = torch.Generator().manual_seed(99)
        for k in forced:
            group = [j for j in range(k, k + 3) if j in rec_x]
            sens = _kink_sensitivity(case, cfg, [rec_x[j] for j in group], trials=2)
            j = group[int(np.argmin(sens))]
            # the reference's own step-direction reproducibility at x_j: sign agreement of d total/dx between x_j and
            # x_j moved by <= 16 ulp (plain fraction of pixels, and weighted by |g|)
            base_val, base_g = _input_gradient(case, cfg, rec_x[j])
            agree, wagree = [], []
            for _ in range(3):
                _, g2 = _input_gradient(case, cfg, _ulp_perturb(rec_x[j], 16, gen))
                same = (torch.sign(g2) == torch.sign(base_g)).double()
                agree.append(float(same.mean()))
                wagree.append(float((same * base_g.abs().double()).sum() / base_g.abs().double().sum()))
            print(f"  forced target {k}: sensitivities {sens} -> keep {j}; twin sign agreement {agree} weighted {wagree}; "
                  f"restated-vs-hook sign agreement {float((torch.sign(base_g) == rec_sign[j]).double().mean()):.6f}", flush=True)
            ks.append(j), xs.append(rec_x[j].numpy()), gs.append(rec_raw[j].numpy()), ss.append(rec_sign[j].numpy().astype(np.int8))
            sens_kept.append(float(np.min(sens))), agree_kept.append(min(agree)), wagree_kept.append(min(wagree))
        out.update(forced_k=np.asarray(ks, dtype=np.int64), forced_x=np.stack(xs).astype(np.float32),
                   forced_grad_bf16=np.stack([_bf16_bits(torch.as_tensor(g)) for g in gs]), forced_sign=np.stack(ss),
                   forced_sensitivity=np.asarray(sens_kept), forced_twin_sign_agreement=np.asarray(agree_kept),
                   forced_twin_weighted_sign_agreement=np.asarray(wagree_kept))
    np.savez(out_path, **out)


def golden_resnet18_24k(parallel=3):
    """BASELINE configs[1] at the horizon the config states (24 000 iterations): the nominal start + FULL_TWINS starts
    <= 16 ulp away, each the unmodified reference on CPU.  Finished runs found in oracle/_long/ are kept, so the step can
    be re-entered; `--full-worker IDX OUT` runs one of them in the foreground."""
    import subprocess
    import time

    os.makedirs(FULL_DIR, exist_ok=True)
    pending = [i for i in range(FULL_TWINS + 1) if not os.path.exists(os.path.join(FULL_DIR, f"run{i}.npz"))]
    running = {}
    while pending or running:
        while pending and len(running) < parallel:
            idx = pending.pop(0)
            path = os.path.join(FULL_DIR, f"run{idx}.npz")
            log = open(os.path.join(FULL_DIR, f"run{idx}.log"), "w")
            running[idx] = subprocess.Popen([sys.executable, "-m", "oracle.make_golden", "--full-worker", str(idx), path],
                                            cwd=ROOT, stdout=log, stderr=subprocess.STDOUT)
            print(f"  started 24k run {idx}", flush=True)
        for idx, proc in list(running.items()):
            if proc.poll() is not None:
                if proc.returncode != 0:
                    raise RuntimeError(f"24k run {idx} failed")
                del running[idx]
                print(f"  24k run {idx} finished", flush=True)
        time.sleep(10)
    assemble_resnet18_24k()


def assemble_resnet18_24k():
    main = dict(np.load(os.path.join(FULL_DIR, "run0.npz")))
    twins = [np.load(os.path.join(FULL_DIR, f"run{i}.npz")) for i in range(1, FULL_TWINS + 1)
             if os.path.exists(os.path.join(FULL_DIR, f"run{i}.npz"))]
    main["history"] = main["history"].astype(np.float32)  # the reference's values ARE fp32 (.item() of an fp32 scalar)
    if "forced_grad" in main:  # a worker started before the bf16 packing existed wrote fp32: pack here
        main["forced_grad_bf16"] = np.stack([_bf16_bits(torch.as_tensor(g)) for g in main.pop("forced_grad")])
    raw_path = os.path.join(FULL_DIR, "run0.npz.raw.npz")
    if os.path.exists(raw_path):  # the fp32 gradients the hook recorded: re-pack with the current (round-to-nearest) packing
        raw = np.load(raw_path)
        main["forced_grad_bf16"] = np.stack([_bf16_bits(torch.as_tensor(raw[f"g_{int(k)}"])) for k in main["forced_k"]])
    main.update(twin_history=np.stack([t["history"] for t in twins]).astype(np.float32),
                twin_psnr=np.asarray([t["psnr"] for t in twins]), twin_opt_value=np.asarray([t["opt_value"] for t in twins]),
                twin_rec_mean=np.asarray([t["rec_mean"] for t in twins]), twin_rec_std=np.asarray([t["rec_std"] for t in twins]),
                iterations=np.int64(len(main["history"])), twin_seed=np.int64(LONG_SEED))
    np.savez_compressed(os.path.join(GOLDEN, "attack_resnet18_24k.npz"), **main)


def _bf16_bits(t):
    """fp32 tensor -> bfloat16 bit patterns (uint16), ROUND TO NEAREST EVEN: every element within 2^-9 and, unlike the truncation
    of round 4 (which shrank every |g| by 0.3 % on average and so forced a +- 2 % band on the norm ratio),
    without a bias -- the norm of a stored gradient is the reference's to ~1e-5, which is what lets the tests hold `d total/dx` to
    +- 2e-3 in SIZE as well as in direction (a scale error is invisible to hard sign but fatal for see-through / TAG)."""
    bits = t.detach().contiguous().view(torch.int32).numpy().astype(np.int64) & 0xFFFFFFFF
    rounded = bits + 0x7FFF + ((bits >> 16) & 1)
    return ((rounded >> 16) & 0xFFFF).astype(np.uint16)


def _reference_step_direction(case, cfg, x):
    """(loss, raw d total/dx, candidate.grad after the closure) of the UNMODIFIED reference evaluated at x: a `dryrun`
    reconstruct from initial_data = x with the closure spied on from outside (optimization_based_attack.py:145-189)."""
    breaching = import_reference()
    attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
    inner_compute = attacker._compute_objective
    got = {}

    def wrapped(candidate, labels, rec_model, optimizer, shared_data, iteration):
        closure = inner_compute(candidate, labels, rec_model, optimizer, shared_data, iteration)
        candidate.register_hook(lambda grad: got.__setitem__("raw", grad.detach().clone()))

        def spy():
            value = closure()
            got["sign"] = candidate.grad.detach().clone()
            return value

        return spy

    attacker._compute_objective = wrapped
    torch.manual_seed(7)
    shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
    _, stats = attacker.reconstruct(case.server_payload, shared, {}, initial_data=x, dryrun=True)
    return stats["Trial_0_Val"][0], got["raw"], got["sign"]


def golden_resnet18_long_signs():
    """Step direction of the reference at the teacher-forcing iterates of attack_resnet18_long.npz (1000-iteration run): the
    quantity hard-sign Adam consumes is sign(d total/dx) (optimization_based_attack.py:165,181-182), so a loss that agrees at
    x_k is not enough.  For every stored x_k: the unmodified reference's sign map and raw gradient at x_k, and the
    reference's OWN reproducibility of that map -- its agreement with the map at x_k moved by <= 16 ulp (3 draws; plain
    pixel fraction and |g|-weighted) -- which is the yardstick a second implementation is held to."""
    from breaching_amd.cases import build_case

    torch.set_num_threads(2)
    gold = np.load(os.path.join(GOLDEN, "attack_resnet18_long.npz"))
    case = build_case("resnet18", "ImageNet", 1)
    cfg = _cfg("invertinggradients", [f"optim.max_iterations={int(gold['iterations'])}"])
    gen = torch.Generator().manual_seed(99)
    out = dict(forced_k=gold["forced_k"])
    signs, grads, agree, wagree, losses = [], [], [], [], []
    for k, x in zip(gold["forced_k"], gold["forced_x"]):
        x = torch.as_tensor(x)
        loss, raw, sign = _reference_step_direction(case, cfg, x)
        a, w = [], []
        for _ in range(3):
            _, raw2, sign2 = _reference_step_direction(case, cfg, _ulp_perturb(x, 16, gen))
            same = (sign2 == sign).double()
            a.append(float(same.mean())), w.append(float((same * raw.abs().double()).sum() / raw.abs().double().sum()))
        print(f"  k={int(k)}: loss {loss:.6f} (history {float(gold['history'][k]):.6f}); twin sign agreement {a}, weighted {w}", flush=True)
        signs.append(sign.numpy().astype(np.int8)), grads.append(_bf16_bits(raw)), agree.append(min(a)), wagree.append(min(w)), losses.append(loss)
    out.update(forced_sign=np.stack(signs), forced_grad_bf16=np.stack(grads), forced_twin_sign_agreement=np.asarray(agree),
               forced_twin_weighted_sign_agreement=np.asarray(wagree), forced_loss=np.asarray(losses))
    np.savez_compressed(os.path.join(GOLDEN, "attack_resnet18_long_signs.npz"), **out)


def golden_seethrough():
    from breaching_amd.cases import build_case, initial_candidate

    torch.set_num_threads(8)
    case = build_case("resnet50", "ImageNet", 2, provide_buffers=True)
    x0 = initial_candidate(case.data_cfg, 2)
    # Langevin noise off: comparable across RNG implementations (GPU parity tests use this record)
    cfg = _cfg("seethroughgradients", ["optim.max_iterations=6", "optim.warmup=2", "optim.callback=2", "optim.langevin_noise=0.0"])
    rec, stats = _run_reference_attack(cfg, case, x0, seed=11)
    out = _attack_record(cfg, case, x0, rec, stats, crop=32)
    twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 1)
    out.update(twin_history=twins, twin_psnr=twin_psnr, twin_opt_value=twin_opt)
    # Langevin noise on (the shipped default 0.01): CPU-only pin of the restatement, same torch CPU generator stream
    cfg = _cfg("seethroughgradients", ["optim.max_iterations=4", "optim.warmup=2", "optim.callback=2"])
    rec, stats = _run_reference_attack(cfg, case, x0, seed=11)
    out.update({f"noise_{k}": v for k, v in _attack_record(cfg, case, x0, rec, stats, crop=32).items()})
    np.savez_compressed(os.path.join(GOLDEN, "attack_seethrough.npz"), **out)


NOISE_SEEDS = (11, 12, 13, 14, 15)
NOISE_ITERS = 8


def golden_seethrough_noise():
    """The reference's own run-to-run envelope under Langevin noise: see-through-gradients on ResNet-50, 2 images, the shipped
    noise 0.01, from ONE starting point with five different noise streams (torch.manual_seed before `reconstruct`; the CPU
    run draws one noise tensor per iteration, optimization_based_attack.py:167-170).  A second implementation that draws
    its noise on the device (another generator, hence another stream) can only be held to this envelope, not to one of its
    members.  8 iterations with a 2-iteration warm-up."""
    from breaching_amd.cases import build_case, initial_candidate, psnr

    torch.set_num_threads(2)
    case = build_case("resnet50", "ImageNet", 2, provide_buffers=True)
    x0 = initial_candidate(case.data_cfg, 2)
    cfg = _cfg("seethroughgradients", [f"optim.max_iterations={NOISE_ITERS}", "optim.warmup=2", "optim.callback=4"])
    assert cfg.optim.langevin_noise == 0.01
    hists, psnrs, opts, means, stds = [], [], [], [], []
    for seed in NOISE_SEEDS:
        rec, stats = _run_reference_attack(cfg, case, x0, seed=seed)
        data = rec["data"].detach()
        hists.append(np.asarray(stats["Trial_0_Val"], dtype=np.float64))
        psnrs.append(psnr(data, case.true_user_data["data"], case.data_cfg))
        opts.append(stats["opt_value"]), means.append(float(data.double().mean())), stds.append(float(data.double().std()))
        print(f"  noise seed {seed}: history {hists[-1]}, psnr {psnrs[-1]:.4f}", flush=True)
    from breaching_amd.cases import parameter_checksum

    np.savez_compressed(os.path.join(GOLDEN, "attack_seethrough_noise.npz"), history=np.stack(hists), psnr=np.asarray(psnrs),
                        opt_value=np.asarray(opts), rec_mean=np.asarray(means), rec_std=np.asarray(stds), seeds=np.asarray(NOISE_SEEDS),
                        model_checksum=np.float64(parameter_checksum(case.model)), labels=rec["labels"].numpy())


def golden_seethrough_b8():
    """BASELINE configs[2] at its real batch size: ResNet-50, 8 images, see-through-gradients (euclidean 1e-4 + TV + L2 norm +
    DeepInversion 0.1), Langevin noise 0.01 ON, labels recovered with `yin`, user-supplied BN buffers; 12 iterations with a
    3-iteration warm-up so the cosine schedule moves.  The noise is what torch's seeded default CPU generator produces
    (the reference on CPU draws it there, optimization_based_attack.py:167-170); the HIP test re-creates the same stream
    (`impl.langevin_noise=host`), so the two sides see identical noise tensors."""
    from breaching_amd.cases import build_case, initial_candidate

    torch.set_num_threads(8)
    case = build_case("resnet50", "ImageNet", 8, provide_buffers=True, provide_labels=False)
    x0 = initial_candidate(case.data_cfg, 8)
    cfg = _cfg("seethroughgradients", ["optim.max_iterations=12", "optim.warmup=3", "optim.callback=4"])
    assert cfg.optim.langevin_noise == 0.01 and cfg.label_strategy == "yin"
    rec, stats = _run_reference_attack(cfg, case, x0, seed=11)
    out = _attack_record(cfg, case, x0, rec, stats, crop=32)
    out.update(seed=np.int64(11), true_labels=case.true_user_data["labels"].numpy())
    twins, twin_psnr, twin_opt = [], [], []
    from breaching_amd.cases import psnr

    gen = torch.Generator().manual_seed(123)
    rec_t, stats_t = _run_reference_attack(cfg, case, _ulp_perturb(x0, 16, gen), seed=11)  # same noise, start 16 ulp away
    out.update(twin_history=np.asarray([stats_t["Trial_0_Val"]], dtype=np.float64),
               twin_psnr=np.asarray([psnr(rec_t["data"], case.true_user_data["data"], case.data_cfg)]),
               twin_opt_value=np.asarray([stats_t["opt_value"]]))
    # Pixel level: where the gradient of a pixel is smaller than the noise, Adam's normalised step follows rounding, so even
    # the reference's own twin agrees with the nominal run on a fraction of the pixels only -- recorded as the yardstick,
    # together with the (much lower) agreement of a run that used ANOTHER noise stream.
    def close_fraction(rec_other):
        return float(np.isclose(rec_other["data"].numpy()[..., :32, :32], out["rec"], rtol=2e-3, atol=2e-3).mean())

    rec_s, _ = _run_reference_attack(cfg, case, x0, seed=12)
    out.update(twin_close_fraction=np.float64(close_fraction(rec_t)), other_noise_close_fraction=np.float64(close_fraction(rec_s)))
    np.savez_compressed(os.path.join(GOLDEN, "attack_seethrough_b8.npz"), **out)


def golden_tag_bert_base():
    """BASELINE configs[4] at its real size: BERT-base masked-LM (BertConfig() defaults, 109.5 M parameters), sequence
    length 32, TAG joint attack (tag-euclidean over 201 of the 202 tensors: the word-embedding gradient is cut off, the
    tied decoder weight makes the reconstructed list one longer than the observed one), AdamW, clipping, warm-up + linear
    decay; 12 iterations.  Run by the reference's OptimizationJointAttacker on CPU."""
    breaching = import_reference(preload_transformers=True)
    from breaching_amd.cases import build_text_case, parameter_checksum

    torch.set_num_threads(8)
    out = {}
    over = ["optim.max_iterations=12", "optim.callback=4", "optim.warmup=3"]
    for tag, seed in (("", 3), ("twin_", 4)):
        case = build_text_case(full_size=True, seq_len=32)
        cfg = _cfg("tag", over)
        attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
        torch.manual_seed(seed)
        rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {})
        out[f"{tag}history"] = np.asarray(stats["Trial_0_Val"], dtype=np.float64)
        out[f"{tag}opt_value"] = np.float64(stats["opt_value"])
        out[f"{tag}tokens"] = rec["data"].numpy()
        out[f"{tag}labels"] = rec["labels"].numpy()
        out[f"{tag}raw_embeddings"] = rec["raw_embeddings"].numpy()
        out[f"{tag}seed"] = np.int64(seed)
        if tag == "":
            out["n_observed"] = np.int64(len(case.shared_data[0]["gradients"]))  # after the embedding gradient was popped
            out["n_parameters"] = np.int64(sum(1 for _ in case.model.parameters()))
            out["model_checksum"] = np.float64(parameter_checksum(case.model))
            out["true_tokens"] = case.true_user_data["data"].numpy()
            out["grad_checksum"] = np.float64(sum(float(g.double().sum()) for g in case.shared_data[0]["gradients"]))
    np.savez_compressed(os.path.join(GOLDEN, "attack_tag_bert_base.npz"), **out)


# ---- BASELINE configs[4] at its STATED schedule: tag.yaml untouched (1 000 iterations, warm-up 50, linear decay, clip 1.0) ----
TAG_LONG_DIR = os.path.join(HERE, "_long")


def _tag_bert_base_1000_worker(idx, out_path, threads=3, iters=None):
    """One run of the reference's OptimizationJointAttacker (optimization_with_label_attack.py:89-205) on BERT-base, sequence
    length 32, with tag.yaml exactly as shipped.  idx 0 = the nominal run (torch.manual_seed(3) before `reconstruct`, as in
    golden_tag_bert_base); idx 1 = the same seed with the drawn embedding start moved by <= 16 ulp (the bound
    `_initialize_data` of the attacker INSTANCE is wrapped from outside; no reference file is touched): the reference's own
    reproducibility envelope for this smooth (AdamW, no sign) trajectory."""
    import time as _time

    breaching = import_reference(preload_transformers=True)
    from breaching_amd.cases import build_text_case, parameter_checksum

    torch.set_num_threads(threads)
    case = build_text_case(full_size=True, seq_len=32)
    over = ["optim.callback=100"] if iters is None else [f"optim.max_iterations={iters}", "optim.callback=100"]
    cfg = _cfg("tag", over)
    if iters is None:
        assert cfg.optim.max_iterations == 1000 and cfg.optim.warmup == 50 and cfg.optim.step_size_decay == "linear"
        assert cfg.optim.grad_clip == 1.0 and cfg.optim.optimizer == "bert-adam" and cfg.optim.step_size == 0.05
    attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
    if idx > 0:
        inner_init = attacker._initialize_data
        calls = [0]
        gen = torch.Generator().manual_seed(LONG_SEED + idx)

        def perturbed_init(shape):
            out = inner_init(shape)
            calls[0] += 1
            if calls[0] == 2:  # 1 = label template (:42-49), 2 = the trial's embedding candidate (:98), 3 = its labels (:99)
                with torch.no_grad():
                    out.copy_(_ulp_perturb(out.detach(), 16, gen))
            return out

        attacker._initialize_data = perturbed_init
    inner_compute = attacker._compute_objective
    t0 = _time.time()

    def timed_compute(candidate, labels, rec_model, optimizer, shared_data, iteration):
        if iteration % 50 == 0 and iteration > 0:
            print(f"  tag run {idx}: iteration {iteration}, {(_time.time() - t0) / iteration:.3f} s/it", flush=True)
        return inner_compute(candidate, labels, rec_model, optimizer, shared_data, iteration)

    attacker._compute_objective = timed_compute
    torch.manual_seed(3)
    rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {})
    out = dict(history=np.asarray(stats["Trial_0_Val"], dtype=np.float32), opt_value=np.float64(stats["opt_value"]),
               tokens=rec["data"].numpy(), labels=rec["labels"].numpy(), raw_embeddings=rec["raw_embeddings"].numpy(),
               seconds=np.float64(_time.time() - t0), threads=np.int64(threads))
    if idx == 0:
        out.update(n_observed=np.int64(len(case.shared_data[0]["gradients"])), n_parameters=np.int64(sum(1 for _ in case.model.parameters())),
                   model_checksum=np.float64(parameter_checksum(case.model)), true_tokens=case.true_user_data["data"].numpy(),
                   seed=np.int64(3), grad_checksum=np.float64(sum(float(g.double().sum()) for g in case.shared_data[0]["gradients"])))
    np.savez(out_path, **out)


def assemble_tag_bert_base_1000():
    main = dict(np.load(os.path.join(TAG_LONG_DIR, "tag0.npz")))
    twin = np.load(os.path.join(TAG_LONG_DIR, "tag1.npz"))
    main.update({f"twin_{k}": twin[k] for k in ("history", "opt_value", "tokens", "labels", "raw_embeddings")})
    main.update(iterations=np.int64(len(main["history"])), twin_seed=np.int64(LONG_SEED + 1), twin_ulps=np.int64(16))
    np.savez_compressed(os.path.join(GOLDEN, "attack_tag_bert_base_1000.npz"), **main)


def golden_tag_bert_base_1000():
    """BASELINE configs[4] with tag.yaml untouched: nominal + one 16-ulp twin, run one after the other (finished runs found in
    oracle/_long/ are kept)."""
    os.makedirs(TAG_LONG_DIR, exist_ok=True)
    for idx in (0, 1):
        path = os.path.join(TAG_LONG_DIR, f"tag{idx}.npz")
        if not os.path.exists(path):
            _tag_bert_base_1000_worker(idx, path, threads=int(os.environ.get("GOLDEN_THREADS", "3")))
    assemble_tag_bert_base_1000()


# ---- BASELINE configs[2] on its real schedule: warm-up 50 + cosine decay, Langevin noise, 1 000 iterations (round 5: 300) -----
# Three runs of ~1.5 h each (5.5 s per iteration on two threads, the three side by side):
#   for i in 0 1 2; do python oracle/make_golden.py --seethrough-worker $i oracle/_long/see$i.npz --threads 2 & done; wait
#   python oracle/make_golden.py --only assemble_seethrough_long
SEE_LONG_ITERS = 1000
SEE_LONG_RUNS = {0: dict(seed=11, ulps=0), 1: dict(seed=11, ulps=16), 2: dict(seed=12, ulps=0)}  # nominal, same-noise twin, other noise


def _seethrough_b8_long_worker(idx, out_path, threads=1, iters=SEE_LONG_ITERS):
    """One run of the unmodified reference on ResNet-50, 8 images, seethroughgradients.yaml as shipped except for the horizon
    (max_iterations 1 000 instead of 20 000: the cosine period follows it, warm-up stays 50, seethroughgradients.yaml:20-23):
    euclidean 1e-4 + TV + L2 norm + DeepInversion 0.1, Langevin noise 0.01 from torch's seeded CPU generator
    (optimization_based_attack.py:167-170), labels recovered with `yin`, user BN buffers.  idx 0 nominal; idx 1 the same noise
    stream from a start <= 16 ulp away (the reference's own reproducibility envelope); idx 2 another noise stream (what "not the
    same run" looks like).  The history so far is rewritten to `<out>.partial.npy` every 10 iterations."""
    import time as _time

    from breaching_amd.cases import build_case, initial_candidate, parameter_checksum, psnr

    torch.set_num_threads(threads)
    spec = SEE_LONG_RUNS[idx]
    case = build_case("resnet50", "ImageNet", 8, provide_buffers=True, provide_labels=False)
    x0 = initial_candidate(case.data_cfg, 8)
    if spec["ulps"]:
        x0 = _ulp_perturb(x0, spec["ulps"], torch.Generator().manual_seed(LONG_SEED))
    cfg = _cfg("seethroughgradients", [f"optim.max_iterations={iters}", "optim.callback=50"])
    assert cfg.optim.langevin_noise == 0.01 and cfg.label_strategy == "yin" and cfg.optim.warmup == 50
    assert cfg.optim.step_size_decay == "cosine-decay" and cfg.optim.signed is False
    breaching = import_reference()
    attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
    inner_trial = attacker._run_trial
    seen = {}

    def spy_trial(rec_model, shared_data, labels, stats, trial, initial_data=None, dryrun=False):
        seen["stats"] = stats
        return inner_trial(rec_model, shared_data, labels, stats, trial, initial_data, dryrun)

    attacker._run_trial = spy_trial
    inner_compute = attacker._compute_objective
    t0 = _time.time()

    def timed_compute(candidate, labels, rec_model, optimizer, shared_data, iteration):
        if iteration % 10 == 0 and iteration > 0:
            print(f"  see-through run {idx}: iteration {iteration}, {(_time.time() - t0) / iteration:.2f} s/it, "
                  f"loss {seen['stats']['Trial_0_Val'][-1]:.6f}", flush=True)
            np.save(out_path + ".partial.npy", np.asarray(seen["stats"]["Trial_0_Val"], dtype=np.float32))
        return inner_compute(candidate, labels, rec_model, optimizer, shared_data, iteration)

    attacker._compute_objective = timed_compute
    torch.manual_seed(spec["seed"])
    shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
    rec, stats = attacker.reconstruct(case.server_payload, shared, {}, initial_data=x0)
    data = rec["data"].detach()
    out = dict(history=np.asarray(stats["Trial_0_Val"], dtype=np.float32), opt_value=np.float64(stats["opt_value"]),
               psnr=np.float64(psnr(data, case.true_user_data["data"], case.data_cfg)), labels=rec["labels"].numpy(),
               rec_mean=np.float64(data.double().mean()), rec_std=np.float64(data.double().std()), rec=data[..., :32, :32].numpy(),
               seed=np.int64(spec["seed"]), ulps=np.int64(spec["ulps"]), seconds=np.float64(_time.time() - t0), threads=np.int64(threads))
    if idx == 0:
        out.update(model_checksum=np.float64(parameter_checksum(case.model)), true_labels=case.true_user_data["labels"].numpy(),
                   grad0_checksum=np.float64(case.shared_data[0]["gradients"][0].double().sum()))
    np.savez(out_path, **out)


def assemble_seethrough_b8_long():
    main = dict(np.load(os.path.join(TAG_LONG_DIR, "see0.npz")))
    twin, other = np.load(os.path.join(TAG_LONG_DIR, "see1.npz")), np.load(os.path.join(TAG_LONG_DIR, "see2.npz"))
    for tag, run in (("twin", twin), ("other_noise", other)):
        main.update({f"{tag}_{k}": run[k] for k in ("history", "opt_value", "psnr", "rec", "rec_mean", "rec_std", "seed", "ulps")})
    main.update(iterations=np.int64(len(main["history"])), twin_start_seed=np.int64(LONG_SEED))
    np.savez_compressed(os.path.join(GOLDEN, "attack_seethrough_b8_long.npz"), **main)


def golden_seethrough_b8_long():
    os.makedirs(TAG_LONG_DIR, exist_ok=True)
    for idx in SEE_LONG_RUNS:
        path = os.path.join(TAG_LONG_DIR, f"see{idx}.npz")
        if not os.path.exists(path):
            _seethrough_b8_long_worker(idx, path, threads=int(os.environ.get("GOLDEN_THREADS", "8")))
    assemble_seethrough_b8_long()


class _LegacyTorchSemantics:
    """Harness-side shim (no reference file is touched) that lets the reference's Pearlmutter objectives run under torch 2.x:
    they were written for torch 1.10, where (a) `torch._foreach_add_/_foreach_sub_` on parameters that require grad did not
    raise "a leaf Variable that requires grad is being used in an in-place operation" and (b) `optimizer.zero_grad()` zeroed
    the gradients instead of setting them to None (objectives.py:354 does `candidate.grad += ...` right after it).  Inside
    this context both behave as they did then: the in-place list ops run under no_grad, zero_grad keeps the tensors."""

    def __enter__(self):
        self._add, self._sub, self._zero = torch._foreach_add_, torch._foreach_sub_, torch.optim.Optimizer.zero_grad

        def no_grad(fn):
            def wrapped(*args, **kwargs):
                with torch.no_grad():
                    return fn(*args, **kwargs)

            return wrapped

        torch._foreach_add_, torch._foreach_sub_ = no_grad(self._add), no_grad(self._sub)
        original_zero = self._zero

        def zero_grad(opt, set_to_none=False):
            return original_zero(opt, set_to_none=False)

        torch.optim.Optimizer.zero_grad = zero_grad
        return self

    def __exit__(self, *exc):
        torch._foreach_add_, torch._foreach_sub_, torch.optim.Optimizer.zero_grad = self._add, self._sub, self._zero
        return False


def golden_pearlmutter():
    """Pearlmutter finite-difference objectives (objectives.py:279-493) of the unmodified reference, run through
    `_LegacyTorchSemantics`: (1) one evaluation per variant -- objective value, the estimate it accumulates into
    candidate.grad, and the exact double-backward gradient of the corresponding plain objective for comparison;
    (2) a short soft-sign attack per family through the reference attacker."""
    breaching = import_reference()
    from breaching.attacks.auxiliaries import objectives as O
    from breaching_amd.cases import build_case, initial_candidate

    torch.set_num_threads(8)
    # (1) on a kink-free model: on ReLU / max-pool nets the finite difference is dominated by activation kinks (measured on
    # ConvNet in fp64: the response to the parameter offset plateaus at 2.5e-7 for eps -> 0 instead of vanishing)
    case = build_case("smoothnet", "CIFAR10", 2)
    x0 = initial_candidate(case.data_cfg, 2, seed=6)
    labels = case.shared_data[0]["metadata"]["labels"]
    impl = type("Impl", (), dict(mixed_precision=False, dtype="float", JIT=None))()
    from breaching_amd.cases import parameter_checksum

    out = dict(x0_seed=np.int64(6), smooth_model_checksum=np.float64(parameter_checksum(case.model)))
    with _LegacyTorchSemantics():
        for name, plain in (("pearlmutter-loss", "euclidean"), ("pearlmutter-cosine", "cosine-similarity")):
            for implementation in ("forward", "backward", "central", "upwind"):
                objective = O.objective_lookup[name](scale=0.7, eps=1e-3, task_regularization=0.05, implementation=implementation)
                objective.initialize(case.loss_fn, impl, None)
                candidate = x0.clone().requires_grad_(True)
                candidate.grad = torch.zeros_like(candidate)
                before = [p.detach().clone() for p in case.model.parameters()]
                value, task_loss = objective(case.model, case.shared_data[0]["gradients"], candidate, labels)
                assert all(torch.equal(a, b) for a, b in zip(before, case.model.parameters()))  # parameters restored (:326-328)
                key = f"{name}_{implementation}"
                out[f"{key}__value"] = np.float64(value)
                out[f"{key}__task_loss"] = np.float64(task_loss.detach())
                out[f"{key}__grad"] = candidate.grad.numpy().copy()
            exact = O.objective_lookup[plain](scale=0.7, task_regularization=0.05)
            exact.initialize(case.loss_fn, impl, None)
            candidate = x0.clone().requires_grad_(True)
            value, _ = exact(case.model, case.shared_data[0]["gradients"], candidate, labels)
            (g,) = torch.autograd.grad(value, candidate)
            out[f"{name}__exact_value"], out[f"{name}__exact_grad"] = np.float64(value), g.numpy()
        case = build_case("convnet", "CIFAR10", 2)  # (2) the attack on the usual ConvNet
        for name, scoring in (("pearlmutter-loss", "euclidean"), ("pearlmutter-cosine", "cosine-similarity")):
            cfg = _cfg("invertinggradients", [f"objective.type={name}", "optim.signed=soft", "optim.max_iterations=12",
                                              "optim.callback=6", f"restarts.scoring={scoring}"])
            rec, stats = _run_reference_attack(cfg, case, x0)
            key = name.replace("-", "_")
            out.update({f"{key}_{k}": v for k, v in _attack_record(cfg, case, x0, rec, stats).items()})
            twins, twin_psnr, twin_opt = _twin_runs(cfg, case, x0, 2)
            out.update({f"{key}_twin_history": twins, f"{key}_twin_psnr": twin_psnr, f"{key}_twin_opt_value": twin_opt})
    np.savez_compressed(os.path.join(GOLDEN, "pearlmutter.npz"), **out)


def golden_tag():
    """BASELINE config 5 family: TAG joint data+label attack (tag-euclidean, AdamW, clipping, warm-up + linear decay) on a
    tiny random-init BERT masked-LM, run by the reference's OptimizationJointAttacker."""
    breaching = import_reference(preload_transformers=True)
    from breaching_amd.cases import build_text_case, parameter_checksum

    torch.set_num_threads(8)
    out = {}
    for tag, seed in (("", 3), ("twin_", 4)):
        case = build_text_case()
        cfg = _cfg("tag", ["optim.max_iterations=30", "optim.callback=10", "optim.warmup=5"])
        attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
        torch.manual_seed(seed)  # draws: label template, then per trial data + labels (optimization_with_label_attack.py:42-49, :98-99)
        rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {})
        out[f"{tag}history"] = np.asarray(stats["Trial_0_Val"], dtype=np.float64)
        out[f"{tag}opt_value"] = np.float64(stats["opt_value"])
        out[f"{tag}tokens"] = rec["data"].numpy()
        out[f"{tag}labels"] = rec["labels"].numpy()
        out[f"{tag}raw_embeddings"] = rec["raw_embeddings"].numpy()
        out[f"{tag}seed"] = np.int64(seed)
    out["model_checksum"] = np.float64(parameter_checksum(build_text_case().model))
    out["true_tokens"] = build_text_case().true_user_data["data"].numpy()
    np.savez_compressed(os.path.join(GOLDEN, "attack_tag.npz"), **out)


STEPS = dict(configs=golden_configs, kernels=golden_kernels, schedules=golden_schedules, convnet=golden_convnet,
             resnet18=golden_resnet18, seethrough=golden_seethrough, tag=golden_tag,
             variants=golden_variants, fedavg=golden_fedavg, labels=golden_labels, dlg=golden_dlg, multiquery=golden_multiquery,
             resnet18_long=golden_resnet18_long, seethrough_b8=golden_seethrough_b8, tag_bert_base=golden_tag_bert_base,
             pearlmutter=golden_pearlmutter, resnet18_24k=golden_resnet18_24k, seethrough_noise=golden_seethrough_noise,
             resnet18_long_signs=golden_resnet18_long_signs, tag_bert_base_1000=golden_tag_bert_base_1000,
             seethrough_b8_long=golden_seethrough_b8_long, reference_more_starts=reference_more_starts_report,
             assemble_resnet18_24k=lambda: assemble_resnet18_24k(), assemble_tag_1000=lambda: assemble_tag_bert_base_1000(),
             assemble_seethrough_long=lambda: assemble_seethrough_b8_long())
SLOW_STEPS = ("resnet18_long", "seethrough_b8", "tag_bert_base", "resnet18_24k", "seethrough_noise", "resnet18_long_signs",
              "tag_bert_base_1000", "seethrough_b8_long", "reference_more_starts", "assemble_resnet18_24k", "assemble_tag_1000",
              "assemble_seethrough_long")  # hours of CPU: only run when asked for by name

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--only", default=None)
    parser.add_argument("--long-worker", nargs=2, default=None, metavar=("IDX", "OUT"))
    parser.add_argument("--full-worker", nargs=2, default=None, metavar=("IDX", "OUT"))
    parser.add_argument("--full-iters", type=int, default=None, help="(testing the generator) shorter horizon")
    parser.add_argument("--seethrough-worker", nargs=2, default=None, metavar=("IDX", "OUT"))
    parser.add_argument("--tag-worker", nargs=2, default=None, metavar=("IDX", "OUT"))
    parser.add_argument("--threads", type=int, default=2, help="torch CPU threads of a --long-worker / --full-worker run")
    args = parser.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    if args.long_worker is not None:
        _resnet18_long_worker(int(args.long_worker[0]), args.long_worker[1], threads=args.threads)
        sys.exit(0)
    if args.seethrough_worker is not None:
        _seethrough_b8_long_worker(int(args.seethrough_worker[0]), args.seethrough_worker[1], threads=args.threads,
                                   iters=args.full_iters or SEE_LONG_ITERS)
        sys.exit(0)
    if args.tag_worker is not None:
        _tag_bert_base_1000_worker(int(args.tag_worker[0]), args.tag_worker[1], threads=args.threads, iters=args.full_iters)
        sys.exit(0)
    if args.full_worker is not None:
        if args.full_iters:
            n = args.full_iters
            _resnet18_full_worker(int(args.full_worker[0]), args.full_worker[1], threads=args.threads, iters=n, forced=(2, n // 2, n - 4))
        else:
            _resnet18_full_worker(int(args.full_worker[0]), args.full_worker[1], threads=args.threads)
        sys.exit(0)
    for name, fn in STEPS.items():
        if (args.only is None and name not in SLOW_STEPS) or args.only == name:
            print(f"[golden] {name}", flush=True)
            fn()
