"""End-to-end parity of the HIP attacker against the reference (golden fixtures) and the CPU restatement.

Tolerances from BASELINE.json north_star: final gradient-matching loss within 1e-4 relative, PSNR within 0.1 dB.

What can and cannot be compared (measured on the reference itself, see oracle/make_golden.py and DESIGN.md "Parity"):
hard-sign Adam on a ReLU network is chaotic at the ulp level -- the *unmodified reference*, restarted from a starting
point moved by 16 ulp, leaves its own trajectory after a handful of iterations (ConvNet: 9e-4 relative at iteration 10,
2-6 % at the end; ResNet-18: 5e-4 at iteration 4).  No second implementation, and no second run of the reference on
other hardware, can stay within 1e-4 of one particular trajectory for long.  The tests therefore assert
  (1) 1e-4 on the loss at the reference's OWN iterates (teacher forcing: our objective + priors evaluated at x_k taken
      from the reference run, early and late in the optimisation),
  (2) 1e-4 on the free-running loss trajectory for as long as the reference's twin runs themselves agree to 1e-5 (a
      tenth of the tolerance: our own GPU runs differ from each other as well -- MIOpen's backward kernels use atomics -- and
      an iteration where the twins are 2.6e-5 apart was seen to put two of our runs 2.5e-4 apart),
  (3) afterwards, agreement within a band derived from the reference's own twin runs (10x their deviation, floor 3e-4
      right after the fork, 2 % once a twin has left by 1e-3),
  (4) PSNR within 0.1 dB (or the twin spread if that is larger).
Configurations without the sign (soft sign / plain Adam) are not chaotic over the tested horizon and are held to 1e-4
over the whole trajectory.  The observed gradient is recomputed on CPU so both sides attack the same target.
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


def _reproducible_horizon(hist_ref, twins, tol=1e-5):
    """First iteration at which any twin run of the reference deviates from the reference by more than `tol`."""
    if twins is None:
        return len(hist_ref)
    dev = np.abs(twins - hist_ref[None, :]) / np.abs(hist_ref[None, :])
    bad = np.nonzero(dev.max(axis=0) > tol)[0]
    return int(bad[0]) if len(bad) else len(hist_ref)


def _check_against_golden(prefix, gold, rec, stats, case, crop=None, checksum_rel=1e-12):
    from breaching_amd.cases import parameter_checksum, psnr

    assert parameter_checksum(case.model) == pytest.approx(float(gold[f"{prefix}model_checksum"]), rel=checksum_rel)
    hist_ref = gold[f"{prefix}history"]
    hist = np.asarray(stats["Trial_0_Val"])
    assert len(hist) == len(hist_ref)
    twins = gold[f"{prefix}twin_history"] if f"{prefix}twin_history" in gold.files else None
    horizon = _reproducible_horizon(hist_ref, twins)
    # (2) strict tolerance while the reference itself is reproducible
    np.testing.assert_allclose(hist[:horizon], hist_ref[:horizon], rtol=LOSS_RTOL, atol=1e-7)
    ref_psnr = float(gold[f"{prefix}psnr"])
    got_psnr = psnr(rec["data"], case.true_user_data["data"], case.data_cfg)
    if horizon == len(hist_ref):
        assert stats["opt_value"] == pytest.approx(float(gold[f"{prefix}opt_value"]), rel=LOSS_RTOL)
        assert abs(got_psnr - ref_psnr) <= PSNR_TOL_DB
        data = rec["data"].detach().cpu().numpy()
        if crop is not None:
            data = data[..., :crop, :crop]
        assert np.isclose(data, gold[f"{prefix}rec"], rtol=2e-3, atol=2e-3).mean() > 0.99
        return horizon
    # (3) beyond the horizon: inside a band derived from the reference's own twin runs.  Two or three twins only sample
    # the spread of a chaotic trajectory (and MIOpen's atomics make our own GPU runs differ from each other), so the band
    # is 10x the largest twin deviation seen so far, at least 3e-4 right after the fork and at least 2 % once any twin has
    # left by 1e-3.  Systematic errors are caught by (1), (2) and by the non-chaotic configurations, which stay at 1e-4.
    twin_dev = np.abs(twins - hist_ref[None, :]).max(axis=0)
    running = np.maximum.accumulate(twin_dev)  # a fork, once taken, is never undone
    forked = np.maximum.accumulate(twin_dev / np.abs(hist_ref)) > 1e-3
    floor = np.where(forked, 0.02, 3.0 * LOSS_RTOL) * np.abs(hist_ref)
    allowed = np.maximum(10.0 * running, floor)
    excess = np.abs(hist - hist_ref) - allowed
    assert (excess[horizon:] <= 0).all(), f"outside the reference's twin envelope by {excess.max():.3e} at {int(excess.argmax())}"
    twin_opt = gold[f"{prefix}twin_opt_value"]
    ref_opt = float(gold[f"{prefix}opt_value"])
    assert abs(stats["opt_value"] - ref_opt) <= max(10.0 * np.abs(twin_opt - ref_opt).max(), (0.05 if forked[-1] else 3 * LOSS_RTOL) * ref_opt)
    twin_psnr = gold[f"{prefix}twin_psnr"]
    assert abs(got_psnr - ref_psnr) <= max(PSNR_TOL_DB, 10.0 * np.abs(twin_psnr - ref_psnr).max())
    return horizon


def _teacher_forced_losses(case, cfg_overrides, xs):
    """Our objective + priors evaluated at given candidates: first history entry of a one-iteration run from x_k."""
    from breaching_amd import get_attack_config

    out = []
    for x in xs:
        cfg = get_attack_config("invertinggradients", cfg_overrides)
        _, stats, _ = _attack(case, cfg, torch.as_tensor(x), dryrun=True)
        out.append(stats["Trial_0_Val"][0])
    return np.asarray(out)


def test_convnet_invertinggradients_100_iterations(golden_dir):
    """BASELINE config 1: ConvNet CIFAR-10, invertinggradients, 1 image, 100 iterations."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_convnet.npz"))
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 1, seed=int(gold["x0_seed"]))
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=100", "optim.callback=50"])
    rec, stats, _ = _attack(case, cfg, x0)
    assert rec["data"].is_cuda and rec["data"].shape == (1, 3, 32, 32)
    horizon = _check_against_golden("", gold, rec, stats, case)
    assert horizon >= 3  # the fixture must leave a non-trivial strictly compared prefix


def test_convnet_teacher_forced_losses_along_reference_trajectory(golden_dir):
    """(1): loss at the reference's own iterates x_k, k up to 99, within 1e-4 of the reference's history[k]."""
    from breaching_amd.cases import build_case

    gold = np.load(os.path.join(golden_dir, "attack_convnet.npz"))
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    ks = gold["forced_k"]
    assert len(ks) >= 3 and ks.max() >= 30
    got = _teacher_forced_losses(case, ["optim.max_iterations=100"], gold["forced_x"])
    np.testing.assert_allclose(got, gold["history"][ks], rtol=LOSS_RTOL)


def test_convnet_dryrun_one_iteration(golden_dir):
    """`dryrun=True` runs exactly one iteration (README.md:24 smoke test; optimization_based_attack.py:137-138)."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_convnet.npz"))
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 1, seed=int(gold["x0_seed"]))
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
    x0 = initial_candidate(case.data_cfg, 1, seed=int(gold["x0_seed"]))
    cfg = get_attack_config("invertinggradients", [
        "objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft", "optim.step_size_decay=cosine-decay",
        "optim.warmup=5", "optim.max_iterations=40", "restarts.scoring=euclidean", "regularization.norm.scale=0.01",
        "regularization.norm.pnorm=2", "optim.callback=20"])
    rec, stats, _ = _attack(case, cfg, x0)
    # smooth configuration: the STRICT branch (whole history 1e-4, opt_value 1e-4, PSNR 0.1 dB, no twin band) must be the one taken
    assert _check_against_golden("l2soft_", gold, rec, stats, case) == len(gold["l2soft_history"]) == 40


def test_resnet18_imagenet_first_iterations(golden_dir):
    """BASELINE config 2 (ResNet-18, 224x224, cosine + TV, hard-sign Adam): first 20 iterations of the schedule."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_resnet18.npz"))
    case = build_case("resnet18", "ImageNet", 1, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 1)
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=20", "optim.step_size_decay=null", "optim.callback=5"])
    rec, stats, attacker = _attack(case, cfg, x0)
    assert attacker.last_trial_execution == "hipGraph replay"
    _check_against_golden("", gold, rec, stats, case, crop=32)
    got = _teacher_forced_losses(case, ["optim.max_iterations=20", "optim.step_size_decay=null"], gold["forced_x"])
    np.testing.assert_allclose(got, gold["history"][gold["forced_k"]], rtol=LOSS_RTOL)


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
    rec, stats, attacker = _attack(case, cfg, x0)
    assert attacker.last_trial_execution == "hipGraph replay"  # ResNet-50 + DeepInversion iteration captured, not eager
    # the user's BN buffers come from a train-mode forward on this host's CPU: equal to the fixture's up to rounding
    _check_against_golden("", gold, rec, stats, case, crop=32, checksum_rel=1e-8)


def test_attacker_vs_restatement_with_restarts():
    """restarts.num_trials > 1 through the trial loop, scoring and selection, against the CPU restatement.

    Uses the non-chaotic soft-sign / euclidean configuration with `initial_data` (the reference overwrites every trial's
    random start with it, optimization_based_attack.py:100-101; GPU and CPU random streams differ anyway), so the strict
    tolerance applies to every trial.  (A constant start such as `init=zeros` is useless here: every max-pool window
    ties exactly and the two devices break the ties differently.)"""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate
    from oracle import restate

    over = ["objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft", "optim.max_iterations=12",
            "restarts.num_trials=2", "restarts.scoring=euclidean", "optim.callback=6"]
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    cfg = get_attack_config("invertinggradients", over)
    x0 = initial_candidate(case.data_cfg, 1, seed=6)
    rec, stats, _ = _attack(case, cfg, x0)
    cpu_case = build_case("convnet", "CIFAR10", 1, device="cpu")
    rec_o, stats_o = restate.run_attack(cpu_case.model, cpu_case.loss_fn, cfg, cpu_case.server_payload, cpu_case.shared_data,
                                        initial_data=x0)
    for t in range(2):
        np.testing.assert_allclose(stats[f"Trial_{t}_Val"], stats_o[f"Trial_{t}_Val"], rtol=LOSS_RTOL)
    assert stats["opt_value"] == pytest.approx(stats_o["opt_value"], rel=LOSS_RTOL)
    torch.testing.assert_close(rec["data"].cpu(), rec_o["data"], rtol=1e-3, atol=1e-3)


def test_two_server_queries_sum_the_objective_over_models(golden_dir):
    """`num_queries = 2` (two model states, two observed gradient lists of the same private batch): one packed plan per list,
    the objective summed over the pairs (optimization_based_attack.py:152-155), against the unmodified reference's run."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_multi_query_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_multiquery.npz"))
    case = build_multi_query_case(2, device="cuda:0")
    cfg = get_attack_config("invertinggradients", ["objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft",
                                                   "optim.max_iterations=16", "restarts.scoring=euclidean", "optim.callback=8"])
    x0 = initial_candidate(case.data_cfg, 2, seed=6)
    rec, stats, attacker = _attack(case, cfg, x0)
    assert len(attacker.objective._plans) == 2 and stats["execution"]["trials"] == {0: "hipGraph replay"}
    _check_against_golden("", gold, rec, stats, case)
    assert not np.allclose(stats["Trial_0_Val"], gold["single_query_history"], rtol=1e-4)


def test_random_restarts_select_the_best_trial():
    """Random initialisation, 3 trials: the returned candidate is the one with the smallest rescored objective."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case

    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=8", "restarts.num_trials=3", "optim.callback=4"])
    rec, stats, attacker = _attack(case, cfg, None)
    assert sorted(k for k in stats if k.startswith("Trial_")) == ["Trial_0_Val", "Trial_1_Val", "Trial_2_Val"]
    assert all(len(stats[f"Trial_{t}_Val"]) == 8 for t in range(3))
    assert len({tuple(stats[f"Trial_{t}_Val"]) for t in range(3)}) == 3  # different starting points
    shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
    rec_models, labels, _ = attacker.prepare_attack(case.server_payload, shared)
    rescored = float(attacker._score_trial(rec["data"], labels, rec_models, shared))
    assert rescored == pytest.approx(stats["opt_value"], rel=1e-5)


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


def _draw_on_cpu(attacker):
    """Make the attacker draw its random initialisations from torch's CPU generator (then move them to the GPU), so a
    seeded run starts exactly where the seeded CPU run of the reference started."""
    original = attacker._initialize_data

    def cpu_init(shape):
        device = attacker.setup["device"]
        attacker.setup["device"] = torch.device("cpu")
        try:
            t = original(shape)
        finally:
            attacker.setup["device"] = device
        t = t.detach().to(device).requires_grad_(True)
        t.grad = torch.zeros_like(t)
        return t

    attacker._initialize_data = cpu_init


def test_tag_joint_attack_on_bert(golden_dir):
    """BASELINE config 5 family: TAG (tag-euclidean objective with per-tensor weights, AdamW, gradient clipping, warm-up +
    linear decay) optimising embeddings and labels jointly on a random-init BERT masked-LM; reference =
    OptimizationJointAttacker run on CPU (oracle/make_golden.py::golden_tag)."""
    import breaching_amd
    from breaching_amd.cases import build_text_case, parameter_checksum

    gold = np.load(os.path.join(golden_dir, "attack_tag.npz"))
    case = build_text_case(device="cuda:0")
    assert parameter_checksum(case.model) == pytest.approx(float(gold["model_checksum"]), rel=1e-12)
    cfg = breaching_amd.get_attack_config("tag", ["optim.max_iterations=30", "optim.callback=10", "optim.warmup=5"])
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cuda:0"), dtype=torch.float))
    assert type(attacker).__name__ == "HipOptimizationJointAttacker"
    _draw_on_cpu(attacker)
    torch.manual_seed(int(gold["seed"]))
    rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {})
    np.testing.assert_allclose(stats["Trial_0_Val"], gold["history"], rtol=LOSS_RTOL)
    assert stats["opt_value"] == pytest.approx(float(gold["opt_value"]), rel=LOSS_RTOL)
    assert set(rec) == {"data", "labels", "raw_embeddings"}
    np.testing.assert_array_equal(rec["data"].cpu().numpy(), gold["tokens"])
    np.testing.assert_array_equal(rec["labels"].cpu().numpy(), gold["labels"])
    np.testing.assert_allclose(rec["raw_embeddings"].cpu().numpy(), gold["raw_embeddings"], rtol=2e-3, atol=2e-4)
    # labels must be withheld for the joint attack (optimization_with_label_attack.py:54-58)
    case.shared_data[0]["metadata"]["labels"] = torch.zeros(1, 8, dtype=torch.long, device="cuda:0")
    with pytest.raises(ValueError, match="Joint optimization"):
        attacker.reconstruct(case.server_payload, case.shared_data, {})


def test_legacy_family_softsign_opponent_tv_features_deepinversion(golden_dir):
    """`legacy.yaml` family on a 2-image ConvNet case: soft sign, double-opponent TV with p=2 / q=0.5 (general stencil
    path), feature regulariser (kernel A's euclidean reduction on the last linear layer's input), DeepInversion prior
    (kernel D), cosine decay."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_variants.npz"))
    case = build_case("convnet", "CIFAR10", 2, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 2, seed=6)
    cfg = get_attack_config("legacy", ["optim.max_iterations=30", "optim.callback=10", "regularization.deep_inversion.scale=0.001"])
    rec, stats, attacker = _attack(case, cfg, x0)
    assert [type(r).__name__ for r in attacker.regularizers] == ["HipTotalVariation", "HipFeatureRegularization", "HipDeepInversion"]
    # smooth configuration: strict branch, no twin band (whole history / opt_value 1e-4, PSNR 0.1 dB)
    assert _check_against_golden("legacy_", gold, rec, stats, case) == len(gold["legacy_history"]) == 30


def test_wei_family_lbfgs_generic_loop(golden_dir):
    """`wei.yaml` family: euclidean objective with task regularisation under L-BFGS -- the generic torch.optim loop with
    the HIP objective as an autograd node (closure evaluated many times per step)."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_variants.npz"))
    case = build_case("convnet", "CIFAR10", 2, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 2, seed=6)
    cfg = get_attack_config("wei", ["optim.max_iterations=4", "optim.callback=2"])
    rec, stats, attacker = _attack(case, cfg, x0)
    assert not attacker._fused_loop_supported()
    _check_against_golden("wei_", gold, rec, stats, case)


def test_fedavg_multi_step_objective(golden_dir):
    """FedAvg user update (2 local SGD steps x 2 images): `_grad_fn_multi_step` (objectives.py:48-72); every local step's
    parameter update and the final `p_local - p_server` are one multi-tensor launch each (bh_mt_axpy), matched against the
    reference's make_functional implementation."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_fedavg_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "attack_fedavg.npz"))
    case = build_fedavg_case(device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 4, seed=6)
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=20", "optim.callback=10", "optim.signed=soft"])
    rec, stats, _ = _attack(case, cfg, x0)
    assert rec["data"].shape == (4, 3, 32, 32)
    _check_against_golden("", gold, rec, stats, case)
    # plain Adam, no sign: the smoothest setting the unroll can be driven in (the reference's own twins still part by 5e-4
    # within five iterations -- max-pool / ReLU kinks inside two chained local steps)
    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=30", "optim.callback=10", "optim.signed=null",
                                                   "optim.step_size=0.01"])
    rec, stats, _ = _attack(case, cfg, x0)
    _check_against_golden("plain_", gold, rec, stats, case)


def test_trials_in_flight_match_sequential_trials():
    """Running a rank's trials concurrently on separate streams gives every trial exactly the sequential result
    (non-chaotic soft-sign configuration; each trial starts from its own random draw, identical in both runs)."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case

    over = ["objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft", "optim.max_iterations=10",
            "restarts.num_trials=3", "restarts.scoring=euclidean", "optim.callback=5"]
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    rec_seq, stats_seq, _ = _attack(case, get_attack_config("invertinggradients", over + ["impl.trials_in_flight=1"]), None, seed=3)
    rec_ctl, stats_ctl, _ = _attack(case, get_attack_config("invertinggradients", over + ["impl.trials_in_flight=1"]), None, seed=3)
    rec_par, stats_par, _ = _attack(case, get_attack_config("invertinggradients", over + ["impl.trials_in_flight=3"]), None, seed=3)
    from conftest import assert_same_attack

    assert sorted(k for k in stats_par if k.startswith("Trial_")) == [f"Trial_{t}_Val" for t in range(3)]
    assert_same_attack((rec_par["data"], stats_par), (rec_seq["data"], stats_seq), control=(rec_ctl["data"], stats_ctl))
    assert stats_par["execution"]["trials"] == {t: "hipGraph replay" for t in range(3)} == stats_seq["execution"]["trials"]


def test_trials_in_flight_share_priors_with_per_trial_state():
    """What concurrent trials actually share: the `legacy` family -- soft sign, DeepInversion
    (producer-side statistics, the backward term riding in kernel E's launch, per-trial sums buffers and tickets, per-pass
    coefficient records hanging off the SHARED BatchNorm modules), feature hooks on the shared last linear layer, double-
    opponent TV with general exponents -- four restarts, four in flight vs strictly one after the other.  Every trial's
    history, the winner and its candidate agree (run-vs-run limits of conftest); all four iterations ran as graph replays.
    reference: regularizers.py:23-60,203-230; optimization_based_attack.py:70-78."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case
    from conftest import assert_same_attack

    over = ["optim.max_iterations=14", "optim.callback=7", "restarts.num_trials=4", "init=randn",
            "regularization.deep_inversion.scale=0.001", "regularization.features.scale=0.1"]
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0", provide_buffers=True)  # one image: bit-reproducible on every box measured
    runs = {}
    for width in (1, "control", 4):
        rec, stats, attacker = _attack(case, get_attack_config("legacy", over + [f"impl.trials_in_flight={1 if width == 'control' else width}"]), None, seed=3)
        assert sorted(type(r).__name__ for r in attacker.regularizers) == ["HipDeepInversion", "HipFeatureRegularization", "HipTotalVariation"]
        assert stats["execution"]["trials"] == {t: "hipGraph replay" for t in range(4)}
        taps = [h for r in attacker.regularizers if type(r).__name__ == "HipDeepInversion" for h in r.losses[0]]
        assert taps and all(h.in_producer and h.fed for h in taps)  # statistics and backward term inside kernel E's launches
        runs[width] = (rec["data"], stats)
    histories = [runs[4][1][f"Trial_{t}_Val"] for t in range(4)]
    assert all(len(h) == 14 for h in histories)
    assert len({round(h[0], 6) for h in histories}) == 4  # four different starting points, not one trial four times
    assert_same_attack(runs[4], runs[1], control=runs["control"])


@pytest.mark.parametrize("family", ["invertinggradients", "legacy"])
def test_later_trial_groups_rearm_the_captured_graphs_and_match_sequential_trials(family):
    """Ten restarts, four in flight: groups 2 and 3 RE-ARM the first group's trials (new starting point copied into the captured
    candidate tensors; moments, best copy, history and state record reset) instead of building and capturing new ones -- every trial's
    history, the winner and its candidate equal the strictly sequential run (each trial a fresh object, three eager iterations, its own
    capture).  `legacy`: DeepInversion + feature hooks + double-opponent TV, whose device buffers the captured graphs keep writing to
    (a re-armed group must not re-initialise them).  optimization_based_attack.py:70-78: the loop being reorganised."""
    from breaching_amd import attacker as attacker_module
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case
    from conftest import assert_same_attack

    if family == "legacy":
        over = ["optim.max_iterations=12", "optim.callback=6", "restarts.num_trials=10", "init=randn",
                "regularization.deep_inversion.scale=0.001", "regularization.features.scale=0.1"]
        case = build_case("convnet", "CIFAR10", 1, device="cuda:0", provide_buffers=True)
    else:
        over = ["objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft", "optim.max_iterations=12",
                "restarts.num_trials=10", "restarts.scoring=euclidean", "optim.callback=6"]
        case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    built = []
    original = attacker_module.FusedTrial.__init__

    def counting(self, *args, **kwargs):
        built.append(1)
        return original(self, *args, **kwargs)

    runs = {}
    attacker_module.FusedTrial.__init__ = counting
    try:
        for width in (1, "control", 4):
            built.clear()
            rec, stats, _ = _attack(case, get_attack_config(family, over + [f"impl.trials_in_flight={1 if width == 'control' else width}"]), None, seed=3)
            assert stats["execution"]["trials"] == {t: "hipGraph replay" for t in range(10)}
            assert len(built) == (4 if width == 4 else 10)  # four objects serve ten trials (groups of 4 + 4 + 2)
            runs[width] = (rec["data"], stats)
    finally:
        attacker_module.FusedTrial.__init__ = original
    assert len({round(runs[4][1][f"Trial_{t}_Val"][0], 6) for t in range(10)}) == 10  # ten different starting points
    assert all(len(runs[4][1][f"Trial_{t}_Val"]) == 12 for t in range(10))
    assert_same_attack(runs[4], runs[1], control=runs["control"])


def test_device_langevin_noise_under_graph_replay(golden_dir):
    """see-through-gradients with the shipped Langevin noise drawn ON THE DEVICE inside the replayed iteration
    (`torch.randn_like` captured into the hipGraph; optimization_based_attack.py:167-170).
    (1) ConvNet, one image: the trial runs as graph replays; the same seed reproduces every loss of the run bit for bit (and on
        boxes whose vendor kernels are reproducible every pixel), another seed gives another run -- the captured generator
        offsets advance with every replay and follow the seed.
    (2) ResNet-50, 2 images: six device-noise runs sit inside the reference's OWN envelope over five noise streams (fixture
        attack_seethrough_noise.npz, unmodified reference on CPU): per iteration within the reference's range widened by 3x
        its spread (floor: north_star's 1e-4, which is what the noise-free first two iterations are held to), mean PSNR within
        0.1 dB of the reference's."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate, parameter_checksum, psnr

    small = build_case("convnet", "CIFAR10", 1, device="cuda:0", provide_buffers=True)  # one image: bit-reproducible on every box measured
    small_cfg = get_attack_config("seethroughgradients", ["optim.max_iterations=12", "optim.warmup=2", "optim.callback=4"])
    assert small_cfg.optim.langevin_noise == 0.01
    xs = initial_candidate(small.data_cfg, 1, seed=6)

    def run_small(seed):
        rec, stats, attacker = _attack(small, small_cfg, xs, seed=seed)
        assert attacker.last_trial_execution == "hipGraph replay"
        return np.asarray(stats["Trial_0_Val"]), rec["data"].detach().clone()

    a, a_again, b = run_small(21), run_small(21), run_small(22)
    assert np.array_equal(a[0], a_again[0])  # same seed: same captured noise stream, every loss bit for bit
    same, other = float((a[1] == a_again[1]).float().mean()), float((a[1] == b[1]).float().mean())
    print(f"  pixels bit-identical: same seed {same:.4f}, other seed {other:.4f}")
    # ... and the candidate with it: bit for bit on most boxes; on a box whose MIOpen picks are not run-to-run reproducible the
    # pixels with near-zero gradient follow rounding (seen: 0.69 identical with every loss identical) -- still far above what
    # another noise stream leaves identical (0.21: the pixels sitting on the box constraint)
    assert same >= 0.3 and same >= 1.5 * other
    assert not np.array_equal(a[0][3:], b[0][3:])   # another seed: another stream
    assert np.array_equal(a[0][:1], b[0][:1])                                   # (the first loss is computed before any noise)

    gold = np.load(os.path.join(golden_dir, "attack_seethrough_noise.npz"))
    case = build_case("resnet50", "ImageNet", 2, device="cuda:0", provide_buffers=True)
    assert parameter_checksum(case.model) == pytest.approx(float(gold["model_checksum"]), rel=1e-8)
    x0 = initial_candidate(case.data_cfg, 2)
    its = gold["history"].shape[1]
    cfg = get_attack_config("seethroughgradients", [f"optim.max_iterations={its}", "optim.warmup=2", "optim.callback=4"])

    def run(seed):
        rec, stats, attacker = _attack(case, cfg, x0, seed=seed)
        assert attacker.last_trial_execution == "hipGraph replay"
        return np.asarray(stats["Trial_0_Val"]), psnr(rec["data"], case.true_user_data["data"], case.data_cfg)

    runs = [run(seed) for seed in (21, 22, 23, 24, 25, 26)]
    ref = gold["history"]
    lo, hi, spread = ref.min(axis=0), ref.max(axis=0), ref.max(axis=0) - ref.min(axis=0)
    slack = np.maximum(3.0 * spread, LOSS_RTOL * np.abs(ref.mean(axis=0)))
    hist = np.stack([r[0] for r in runs])
    print("  reference range", lo, hi, "\n  hip range      ", hist.min(axis=0), hist.max(axis=0))
    assert hist.shape == (6, its)
    assert (hist >= lo - slack).all() and (hist <= hi + slack).all()
    assert np.ptp(hist[:, -1]) > 0  # six device streams do not collapse onto one trajectory
    mean_psnr = float(np.mean([r[1] for r in runs]))
    print(f"  PSNR: hip mean {mean_psnr:.4f} dB, reference mean {gold['psnr'].mean():.4f} dB (range {gold['psnr'].min():.4f} .. {gold['psnr'].max():.4f})")
    assert abs(mean_psnr - float(gold["psnr"].mean())) <= PSNR_TOL_DB


def test_label_recovery_strategies_match_reference(golden_dir):
    """`_recover_label_information` (base_attack.py:305-475) when the user withholds labels.  Strategies that pad with
    random labels (iDLG / analytic on batches with repeated labels) are compared on their deterministic part only."""
    import breaching_amd
    from breaching_amd.cases import build_case

    gold = np.load(os.path.join(golden_dir, "labels.npz"))
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float)
    for tag, n in (("b6", 6), ("b1", 1)):
        case = build_case("convnet", "CIFAR10", n, device="cuda:0", seed_data=5)
        assert case.true_user_data["labels"].tolist() == gold[f"{tag}_true"].tolist()
        for strategy in ("iDLG", "analytic", "yin", "wainakh-simple", "bias-corrected"):
            cfg = breaching_amd.get_attack_config("invertinggradients", [f"label_strategy={strategy}"])
            attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
            shared = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"], labels=None))
                      for d in case.shared_data]
            _, labels, _ = attacker.prepare_attack(case.server_payload, shared)
            want = gold[f"{tag}_{strategy}"]
            assert labels.shape == want.shape and labels.device.type == "cuda"
            if n == 1 or strategy in ("wainakh-simple", "bias-corrected"):
                assert labels.tolist() == want.tolist(), (tag, strategy)
            else:  # deterministic part: every label the strategy can infer is present; the rest is random padding
                # (iDLG, analytic) or an argsort over exactly tied zeros whose order is device specific (yin)
                inferred = {7} if strategy == "iDLG" else {4, 6, 7}
                assert inferred <= set(labels.tolist()), (tag, strategy)
                assert labels.tolist() == sorted(labels.tolist())
    with pytest.raises(ValueError):
        cfg = breaching_amd.get_attack_config("invertinggradients", ["label_strategy=nonsense"])
        breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup).prepare_attack(case.server_payload, shared)


def test_class_attack_secrets_scatter_the_reconstruction():
    """`server_secrets["ClassAttack"]` (optimization_based_attack.py:82-87): the reconstruction is scattered into a zero
    tensor of the true batch size and the labels are replaced; `server_secrets=None` raises TypeError as in the reference."""
    import breaching_amd
    from breaching_amd.cases import build_case, initial_candidate

    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    cfg = breaching_amd.get_attack_config("invertinggradients", ["optim.max_iterations=2", "optim.callback=1"])
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cuda:0"), dtype=torch.float))
    secrets = dict(ClassAttack=dict(true_num_data=3, target_indx=[1], all_labels=torch.tensor([0, 5, 9], device="cuda:0")))
    x0 = initial_candidate(case.data_cfg, 1, seed=6)
    rec, _ = attacker.reconstruct(case.server_payload, case.shared_data, secrets, initial_data=x0)
    assert rec["data"].shape == (3, 3, 32, 32) and rec["labels"].tolist() == [0, 5, 9]
    assert float(rec["data"][0].abs().max()) == 0.0 and float(rec["data"][2].abs().max()) == 0.0 and float(rec["data"][1].abs().max()) > 0
    with pytest.raises(TypeError):
        attacker.reconstruct(case.server_payload, case.shared_data, None, initial_data=x0)


def test_graph_replay_and_eager_launches_give_the_same_trajectory():
    """hipGraph replay is an execution-mode change only: same kernels, same order, same results."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 1, seed=6)
    over = ["objective.type=euclidean", "objective.scale=0.01", "optim.signed=soft", "optim.max_iterations=15",
            "restarts.scoring=euclidean", "optim.callback=5"]
    from conftest import assert_same_attack

    runs = {}
    for flag in (True, False, "control"):  # control = a second eager run
        rec, stats, _ = _attack(case, get_attack_config("invertinggradients", over + [f"impl.hip_graph={flag is True}"]), x0)
        runs[flag] = (rec["data"], stats)
    assert runs[True][1]["execution"]["trials"] == {0: "hipGraph replay"}
    assert runs[False][1]["execution"]["trials"] == {0: "eager launches (graph replay switched off)"} == runs["control"][1]["execution"]["trials"]
    assert len(runs[True][1]["Trial_0_Val"]) == 15
    # strict when a second eager run reproduces the first bit for bit (it did on every box measured), RUN_VS_RUN otherwise
    assert_same_attack(runs[True], runs[False], control=runs["control"])


def test_deep_leakage_joint_lbfgs(golden_dir):
    """`deepleakage.yaml`: joint data+label optimisation (classification labels, softmax'ed) under L-BFGS -- the joint
    attacker on the generic torch.optim loop with the HIP euclidean objective evaluated ~20 times per step."""
    import breaching_amd
    from breaching_amd.cases import build_case, psnr

    gold = np.load(os.path.join(golden_dir, "attack_dlg.npz"))
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0", provide_labels=False)
    cfg = breaching_amd.get_attack_config("deepleakage", ["optim.max_iterations=3", "optim.callback=1"])
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cuda:0"), dtype=torch.float))
    assert type(attacker).__name__ == "HipOptimizationJointAttacker" and not attacker._fused_loop_supported()
    _draw_on_cpu(attacker)
    torch.manual_seed(int(gold["seed"]))
    rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {})
    assert rec["labels"].cpu().tolist() == gold["labels"].tolist()
    assert len(stats["Trial_0_Val"]) == 3
    # L-BFGS (20 closure evaluations per step, curvature pairs from differences of nearly equal gradients) amplifies
    # rounding differences within its very first step -- that part is chaotic in the reference itself, so the free-running
    # comparison only asks for the same starting objective (strict) and a run of the same kind.  What is NOT chaotic is
    # the closure: `test_deep_leakage_closure_at_the_reference_iterates` holds objective and gradient to the strict
    # tolerance at the points the reference's own L-BFGS visited.
    assert stats["Trial_0_Val"][0] == pytest.approx(float(gold["history"][0]), rel=LOSS_RTOL)
    np.testing.assert_allclose(stats["Trial_0_Val"], gold["history"], rtol=0.5)
    assert stats["opt_value"] == pytest.approx(float(gold["opt_value"]), rel=0.5)
    assert abs(psnr(rec["data"], case.true_user_data["data"], case.data_cfg) - float(gold["psnr"])) <= 1.0
    assert stats["execution"]["trials"] == {0: "torch.optim loop (L-BFGS)"}


def test_deep_leakage_closure_at_the_reference_iterates(golden_dir):
    """Teacher forcing for the joint L-BFGS attack: at (candidate, softmaxed label candidate) pairs the reference's closure
    evaluated during its run (calls 0, 1, 7, 20, 21, 40, 59 of 60 -- start, inside the first line search, after each L-BFGS
    step, the end), the HIP euclidean objective and its gradient with respect to BOTH optimised tensors match the
    reference objective module (optimization_with_label_attack.py:168-174 -> objectives.py:26-46, :89-95): value to 1e-4,
    gradients to 1e-4 of their peak."""
    import breaching_amd
    from breaching_amd.cases import build_case, parameter_checksum

    gold = np.load(os.path.join(golden_dir, "attack_dlg.npz"))
    case = build_case("convnet", "CIFAR10", 1, device="cuda:0", provide_labels=False)
    assert parameter_checksum(case.model) == pytest.approx(float(gold["model_checksum"]), rel=1e-12)
    cfg = breaching_amd.get_attack_config("deepleakage", ["optim.max_iterations=3"])
    attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cuda:0"), dtype=torch.float))
    rec_models, _, _ = attacker.prepare_attack(case.server_payload, case.shared_data)
    attacker.objective.initialize(attacker.loss_fn, cfg.impl, case.shared_data[0]["metadata"]["local_hyperparams"])
    assert int(gold["n_objective_calls"]) == 60 and len(gold["forced_call"]) >= 6
    strict = 0
    for idx, (k, x, p, want, gx_ref, gp_ref) in enumerate(zip(gold["forced_call"], gold["forced_x"], gold["forced_p"],
                                                            gold["forced_value"], gold["forced_gx"], gold["forced_gp"])):
        xq = torch.as_tensor(x, device="cuda:0").requires_grad_(True)
        pq = torch.as_tensor(p, device="cuda:0").requires_grad_(True)
        value, _ = attacker.objective(rec_models[0], case.shared_data[0]["gradients"], xq, pq)
        gx, gp = torch.autograd.grad(value, [xq, pq])
        value = float(value.detach())
        err_x = float(np.abs(gx.cpu().numpy() - gx_ref).max() / np.abs(gx_ref).max())
        err_p = float(np.abs(gp.cpu().numpy() - gp_ref).max() / np.abs(gp_ref).max())
        # The iterates L-BFGS visits sit on ReLU / max-pool kinks: the fixture records how much the REFERENCE's own closure
        # moves when such a point is moved by 16 ulp (a flipped unit changes the gradient by a finite amount -- up to 1.5 % of
        # its peak late in the run, 1e-6 at the clean points).  Tolerance: 1e-4, or 3x that recorded sensitivity.
        sx, sp, sv = (float(gold[f"forced_sensitivity_{n}"][idx]) for n in ("x", "p", "value"))
        print(f"  call {int(k):2d}: reference {want:.6e}  hip {value:.6e}  rel {abs(value - want) / want:.1e} (kink {sv:.1e})  "
              f"grad x {err_x:.1e} (kink {sx:.1e})  grad labels {err_p:.1e} (kink {sp:.1e}) of peak")
        assert value == pytest.approx(float(want), rel=max(LOSS_RTOL, 3 * sv))
        assert err_x <= max(1e-4, 3 * sx) and err_p <= max(1e-4, 3 * sp)
        strict += int(abs(value - want) <= LOSS_RTOL * want and err_x <= 1e-4 and err_p <= 1e-4)
    assert strict >= 3  # the kink-free points (calls 1, 7, 40 in the fixture) are held to the strict 1e-4


@pytest.mark.parametrize("name,plain", [("pearlmutter-loss", "euclidean"), ("pearlmutter-cosine", "cosine-similarity")])
def test_pearlmutter_objectives_match_reference_on_a_smooth_model(name, plain, golden_dir):
    """Pearlmutter finite-difference objectives (objectives.py:279-493) against the unmodified reference (run under torch 2.10
    through the harness-side shim of oracle/make_golden.py) on a kink-free model: objective value to 1e-4; the estimate of
    d objective / d candidate agrees with the reference's estimate AND with the exact double-backward gradient to the
    accuracy fp32 finite differences have at eps = 1e-3 (the reference's own estimate is 1-7 % off the exact gradient)."""
    from breaching_amd.cases import build_case, initial_candidate, parameter_checksum
    from breaching_amd.gm import objective_lookup

    gold = np.load(os.path.join(golden_dir, "pearlmutter.npz"))
    case = build_case("smoothnet", "CIFAR10", 2, device="cuda:0")
    assert parameter_checksum(case.model) == pytest.approx(float(gold["smooth_model_checksum"]), rel=1e-12)
    x0 = initial_candidate(case.data_cfg, 2, seed=int(gold["x0_seed"])).to("cuda:0")
    labels = case.shared_data[0]["metadata"]["labels"]
    impl = type("Impl", (), dict(mixed_precision=False))()
    exact = gold[f"{name}__exact_grad"]
    peak = float(np.abs(exact).max())
    noise = 0.08 if name == "pearlmutter-loss" else 0.25  # fp32 finite-difference noise relative to the peak, generous x4
    for implementation in ("forward", "backward", "central", "upwind"):
        objective = objective_lookup[name](scale=0.7, eps=1e-3, task_regularization=0.05, implementation=implementation)
        objective.initialize(case.loss_fn, impl, None)
        candidate = x0.clone().requires_grad_(True)
        before = [p.detach().clone() for p in case.model.parameters()]
        value, task_loss = objective(case.model, case.shared_data[0]["gradients"], candidate, labels)
        (estimate,) = torch.autograd.grad(value, candidate)
        assert all(torch.equal(a, b) for a, b in zip(before, case.model.parameters()))  # the live parameters are never touched
        key = f"{name}_{implementation}"
        # the reference forms 1 - cos in fp32 (cos = 1 - 3e-5 here: 2e-3 relative rounding noise); ours sums in fp64
        assert value.item() == pytest.approx(float(gold[f"{key}__value"]), rel=LOSS_RTOL, abs=2e-7)
        assert float(task_loss) == pytest.approx(float(gold[f"{key}__task_loss"]), rel=1e-5)
        got = estimate.cpu().numpy()
        ref = gold[f"{key}__grad"]
        err_ref = float(np.abs(got - ref).max()) / float(np.abs(ref).max())
        print(f"  {key}: vs reference estimate {err_ref:.3e} of peak")
        assert err_ref <= noise
        if implementation != "upwind":  # upwind weights the differences with max / min of dL/dx along dim 0 (:444): another quantity
            err_exact = float(np.abs(got - exact).max()) / peak
            print(f"  {key}: vs exact double-backward gradient {err_exact:.3e} of peak")
            assert err_exact <= noise
    with pytest.raises(ValueError, match="finite difference"):
        bad = objective_lookup[name](implementation="sideways")
        bad.initialize(case.loss_fn, impl, None)
    with pytest.raises(ValueError, match="local gradients"):
        objective_lookup[name]().initialize(case.loss_fn, impl, dict(steps=2))


@pytest.mark.parametrize("name,scoring", [("pearlmutter-loss", "euclidean"), ("pearlmutter-cosine", "cosine-similarity")])
def test_pearlmutter_attack_through_the_fused_loop(name, scoring, golden_dir):
    """A short soft-sign attack with a Pearlmutter objective through `reconstruct` (fused loop, hipGraph) against the
    reference attacker's run; beyond the reference's own reproducible horizon the twin envelope applies."""
    from breaching_amd import get_attack_config
    from breaching_amd.cases import build_case, initial_candidate

    gold = np.load(os.path.join(golden_dir, "pearlmutter.npz"))
    case = build_case("convnet", "CIFAR10", 2, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 2, seed=int(gold["x0_seed"]))
    cfg = get_attack_config("invertinggradients", [f"objective.type={name}", "optim.signed=soft", "optim.max_iterations=12",
                                                   "optim.callback=6", f"restarts.scoring={scoring}"])
    rec, stats, attacker = _attack(case, cfg, x0)
    assert type(attacker.objective).__name__.startswith("HipPearlmutter")
    _check_against_golden(name.replace("-", "_") + "_", gold, rec, stats, case)


def test_batchnorm_epilogue_fusion_fails_open_on_a_model_it_cannot_serve():
    """cfg.impl.fuse_bn_relu = "auto" (the default): a victim model whose block hands `bn(conv(x))` straight to a THIRD-PARTY
    autograd.Function -- the one consumer the deferred BatchNorm launch (`_PendingBatchNorm`) cannot serve, since the wrapper carries
    no autograd edge -- is attacked anyway: at the first evaluation the fusion is switched off on that model copy, the iteration is
    evaluated again, `stats["execution"]["fused_epilogue_fallback"]` says so, and the trajectory is the one of
    fuse_bn_relu=False.  "required" keeps the error.  The reference attacks arbitrary models (objectives.py:36-46)."""
    import copy

    import breaching_amd
    from breaching_amd import get_attack_config
    from breaching_amd.cases import AttrDict, get_data_config, honest_payload, single_step_update, synthetic_user_data

    class Twice(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t * 2

        @staticmethod
        def backward(ctx, g):
            return g * 2

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1, self.b1 = torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.BatchNorm2d(8)
            self.c2, self.b2 = torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.BatchNorm2d(8)
            self.head = torch.nn.Linear(8 * 32 * 32, 10)

        def forward(self, x):
            out = torch.relu(self.b1(self.c1(x)))       # the pattern the fusion is for
            out = Twice.apply(self.b2(self.c2(out)))    # ... and the one it cannot serve
            return self.head(torch.tanh(out).flatten(1))

    torch.manual_seed(0)
    data_cfg = get_data_config("CIFAR10")
    model = Net().eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    loss_fn = torch.nn.CrossEntropyLoss()
    x_true, labels = synthetic_user_data(data_cfg, 2, 1)
    shared = single_step_update(model, loss_fn, x_true, labels)
    dev = torch.device("cuda:0")
    model = model.to(dev)
    for entry in shared:
        entry["gradients"] = [g.to(dev) for g in entry["gradients"]]
        entry["metadata"]["labels"] = entry["metadata"]["labels"].to(dev)
    case = AttrDict(model=model, loss_fn=loss_fn, server_payload=honest_payload(model, data_cfg), shared_data=shared,
                    true_user_data=dict(data=x_true, labels=labels), data_cfg=data_cfg)
    x0 = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(5))
    over = ["optim.max_iterations=8", "optim.callback=4", "optim.signed=soft"]

    def run(extra):
        cfg = get_attack_config("invertinggradients", over + extra)
        attacker = breaching_amd.prepare_attack(copy.deepcopy(case.model), case.loss_fn, cfg, dict(device=dev, dtype=torch.float))
        shared_copy = [dict(gradients=list(d["gradients"]), buffers=d["buffers"], metadata=dict(d["metadata"])) for d in case.shared_data]
        return attacker.reconstruct(case.server_payload, shared_copy, {}, initial_data=x0)

    rec_auto, stats_auto = run([])
    note = stats_auto["execution"]["fused_epilogue_fallback"]
    assert note is not None and "switched off on 2 layers" in note and "custom autograd.Function" in note
    assert set(stats_auto["execution"]["trials"].values()) == {"hipGraph replay"}  # the fall-back happened before the capture
    rec_off, stats_off = run(["impl.fuse_bn_relu=False"])
    assert stats_off["execution"]["fused_epilogue_fallback"] is None
    np.testing.assert_allclose(stats_auto["Trial_0_Val"], stats_off["Trial_0_Val"], rtol=1e-6)
    torch.testing.assert_close(rec_auto["data"], rec_off["data"], rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError, match="custom autograd.Function"):
        run(["impl.fuse_bn_relu=required"])
