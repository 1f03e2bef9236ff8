"""End-to-end parity of the HIP attacker against the reference (golden fixtures) and the CPU restatement.

Tolerances from BASELINE.json north_star: final gradient-matching loss within 1e-4 relative, PSNR within 0.1 dB.
The golden files were produced by the unmodified reference on CPU (oracle/make_golden.py); the observed gradient is
recomputed on CPU here so both sides attack the same target.
"""

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4
PSNR_TOL_DB = 0.1


def _attack(case, cfg, x0, dryrun=False, seed=7):
    import breaching_amd

    setup = dict(device=torch.device("cuda:0"), dtype=torch.float)
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
    torch.manual_seed(seed)
    shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
    rec, stats = attacker.reconstruct(case.server_payload, shared, {}, initial_data=x0, dryrun=dryrun)
    return rec, stats, attacker


def _check_against_golden(prefix, gold, rec, stats, case, crop=None, check_rec=True):
    from breaching_amd.cases import parameter_checksum, psnr

    assert parameter_checksum(case.model) == pytest.approx(float(gold[f"{prefix}model_checksum"]), rel=1e-12)
    hist_ref = gold[f"{prefix}history"]
    hist = np.asarray(stats["Trial_0_Val"])
    assert len(hist) == len(hist_ref)
    # the whole loss trajectory, not only its end
    np.testing.assert_allclose(hist, hist_ref, rtol=LOSS_RTOL, atol=1e-7)
    assert stats["opt_value"] == pytest.approx(float(gold[f"{prefix}opt_value"]), rel=LOSS_RTOL)
    got_psnr = psnr(rec["data"], case.true_user_data["data"], case.data_cfg)
    assert abs(got_psnr - float(gold[f"{prefix}psnr"])) <= PSNR_TOL_DB
    if check_rec:
        data = rec["data"].detach().cpu().numpy()
        if crop is not None:
            data = data[..., :crop, :crop]
        ref = gold[f"{prefix}rec"]
        # sign-Adam moves every pixel by ~lr per step; allow a small fraction of pixels to have taken another branch
        close = np.isclose(data, ref, rtol=1e-3, atol=1e-3)
        assert close.mean() > 0.99, f"only {close.mean():.4f} of the reconstruction matches the reference"


def test_convnet_invertinggradients_100_iterations(golden_dir):
    """BASELINE config 1: ConvNet CIFAR-10, invertinggradients, 1 image, 100 iterations."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_convnet.npz"))
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 1)
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=100", "optim.callback=50"])
    rec, stats, _ = _attack(case, cfg, x0)
    assert rec["data"].is_cuda and rec["data"].shape == (1, 3, 32, 32)
    _check_against_golden("", gold, rec, stats, case)


def test_convnet_dryrun_one_iteration(golden_dir):
    """`dryrun=True` runs exactly one iteration (README.md:24 smoke test; optimization_based_attack.py:137-138)."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_convnet.npz"))
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 1)
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=100"])
    rec, stats, _ = _attack(case, cfg, x0, dryrun=True)
    assert len(stats["Trial_0_Val"]) == 1
    _check_against_golden("dryrun_", gold, rec, stats, case)


def test_convnet_euclidean_softsign_warmup(golden_dir):
    """Euclidean objective + norm prior + soft sign + warm-up/cosine schedule through the fused loop."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_convnet.npz"))
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 1)
    cfg = get_attack_config("invertinggradients", [
        "objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft", "optim.step_size_decay=cosine-decay",
        "optim.warmup=5", "optim.max_iterations=40", "restarts.scoring=euclidean", "regularization.norm.scale=0.01",
        "regularization.norm.pnorm=2", "optim.callback=20"])
    rec, stats, _ = _attack(case, cfg, x0)
    _check_against_golden("l2soft_", gold, rec, stats, case)


def test_resnet18_imagenet_first_iterations(golden_dir):
    """BASELINE config 2 (ResNet-18, 224x224, cosine + TV, hard-sign Adam): first 20 iterations of the schedule."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_resnet18.npz"))
    case = build_case("resnet18", "ImageNet", 1, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 1)
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=20", "optim.step_size_decay=null", "optim.callback=5"])
    rec, stats, _ = _attack(case, cfg, x0)
    _check_against_golden("", gold, rec, stats, case, crop=32)


def test_resnet50_seethrough_deepinversion(golden_dir):
    """BASELINE config 3 family: ResNet-50, see-through-gradients objective (Euclid + TV + L2 + DeepInversion prior),
    user-provided BN buffers, Langevin noise off for determinism across RNG implementations."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    path = os.path.join(golden_dir, "attack_seethrough.npz")
    gold = np.load(path)
    case = build_case("resnet50", "ImageNet", 2, device="cuda:0", provide_buffers=True)
    x0 = initial_candidate(case.data_cfg, 2)
    cfg = get_attack_config("seethroughgradients", ["optim.max_iterations=6", "optim.warmup=2", "optim.callback=2",
                                                    "optim.langevin_noise=0.0"])
    rec, stats, _ = _attack(case, cfg, x0)
    _check_against_golden("", gold, rec, stats, case, crop=32, check_rec=False)


def test_attacker_vs_restatement_with_restarts():
    """num_trials > 1 on one GPU against the CPU restatement: same winner, same score."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case
    from oracle import restate

    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=12", "restarts.num_trials=3", "init=zeros",
                                                   "optim.callback=6"])
    rec, stats, _ = _attack(case, cfg, None)
    cpu_case = build_case("convnet", "CIFAR10", 1, device="cpu")
    rec_o, stats_o = restate.run_attack(cpu_case.model, cpu_case.loss_fn, cfg, cpu_case.server_payload, cpu_case.shared_data)
    for t in range(3):
        np.testing.assert_allclose(stats[f"Trial_{t}_Val"], stats_o[f"Trial_{t}_Val"], rtol=LOSS_RTOL)
    assert stats["opt_value"] == pytest.approx(stats_o["opt_value"], rel=LOSS_RTOL)


def test_nonfinite_objective_returns_zeros():
    """Non-finite loss ends the trial quietly and an all-non-finite run returns zeros (:131-133, :213-218)."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    case.shared_data[0]["gradients"][0] = case.shared_data[0]["gradients"][0] * float("nan")
    x0 = initial_candidate(case.data_cfg, 1)
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=5", "optim.callback=2"])
    rec, stats, _ = _attack(case, cfg, x0)
    assert len(stats["Trial_0_Val"]) == 0
    assert stats["opt_value"] == float("inf")
    assert float(rec["data"].abs().max()) == 0.0


def test_cpu_device_is_refused():
    import breaching_amd
    from breaching_amd.cases import build_case

    case = build_case("convnet", "CIFAR10", 1)
    cfg = breaching_amd.get_attack_config("invertinggradients")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))


def test_invalid_config_strings_raise_value_error():
    import breaching_amd
    from breaching_amd.cases import build_case

    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float)
    with pytest.raises(ValueError):
        breaching_amd.prepare_attack(case.model, case.loss_fn, breaching_amd.get_attack_config("invertinggradients", ["objective.type=nope"]), setup)
    with pytest.raises(ValueError):
        breaching_amd.prepare_attack(case.model, case.loss_fn, breaching_amd.get_attack_config("invertinggradients", ["attack_type=nope"]), setup)
    att = breaching_amd.prepare_attack(case.model, case.loss_fn, breaching_amd.get_attack_config("invertinggradients", ["init=nope", "optim.max_iterations=1"]), setup)
    with pytest.raises(ValueError):
        att.reconstruct(case.server_payload, case.shared_data, {})
