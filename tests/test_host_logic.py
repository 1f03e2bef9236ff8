"""Host-side logic that needs no GPU: score keys, trial sharding over gloo (world size 2), config plumbing."""

import math
import os
import random
import socket
import struct

import pytest
import torch


def test_score_key_orders_like_floats():
    from breaching_amd.trials import score_key, unpack_key

    rng = random.Random(0)
    scores = [0.0, 1e-30, 1e-8, 0.007, 0.5, 1.0, 1.9999, 2.0, 649.24, 3e38, float("inf")] + [rng.random() * 10 for _ in range(200)]
    items = [(s, t) for t, s in enumerate(scores)]
    by_key = sorted(items, key=lambda it: score_key(it[0], it[1]))
    by_float = sorted(items, key=lambda it: (struct.unpack("<f", struct.pack("<f", it[0]))[0], it[1]))
    assert by_key == by_float
    s, t = unpack_key(score_key(0.0987, 31))
    assert t == 31 and abs(s - 0.0987) < 1e-8
    # NaN is treated as +inf, exactly like `score if score.isfinite() else inf` (optimization_based_attack.py:204)
    assert score_key(float("nan"), 3) == score_key(float("inf"), 3)
    assert score_key(0.5, 7) < score_key(0.5, 8)  # ties resolve to the lower trial like torch.min's first index
    assert score_key(1.0, 0) < 2**63


def test_trial_shard_partition():
    from breaching_amd.trials import TrialShard

    seen = []
    for rank in range(8):
        seen += list(TrialShard(32, rank, 8).local_trials())
        assert len(list(TrialShard(32, rank, 8).local_trials())) == 4
    assert sorted(seen) == list(range(32))
    assert list(TrialShard(3, 2, 8).local_trials()) == [2]
    assert list(TrialShard(3, 5, 8).local_trials()) == []
    assert list(TrialShard(5).local_trials()) == [0, 1, 2, 3, 4]


def test_single_process_selection_matches_reference_argmin():
    from breaching_amd.trials import TrialShard

    sols = {t: torch.full((2, 2), float(t)) for t in range(4)}
    scores = {0: torch.tensor([0.4]), 1: 0.3, 2: float("inf"), 3: torch.tensor(0.3)}
    value, sol = TrialShard(4).select(sols, scores, {}, torch.device("cpu"))
    assert value == pytest.approx(0.3) and sol is sols[1]
    value, sol = TrialShard(2).select({0: sols[0], 1: sols[1]}, {0: float("nan"), 1: float("inf")}, {}, torch.device("cpu"))
    assert math.isinf(value)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker_ragged(rank, world, port, out_dir):
    """Rank 1 finished no trial although it owns some (interrupted before its first trial ended); the solution parts are
    fp64 data and int64 labels -- neither may be rounded through an fp32 buffer."""
    import torch.distributed as dist

    from breaching_amd.trials import TrialShard

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = TrialShard.current(4)
        sols, scores, stats = {}, {}, {}
        if rank == 0:
            for t in shard.local_trials():
                sols[t] = (torch.full((1, 3), 1.0 + 2.0 ** -40 * (t + 1), dtype=torch.float64),
                           torch.full((1, 2), 2 ** 40 + t, dtype=torch.int64))
                scores[t] = [0.75, None, 0.5][t]
                stats[f"Trial_{t}_Val"] = [float(t)]
        value, sol = shard.select(sols, scores, stats, torch.device("cpu"))
        torch.save(dict(value=value, data=sol[0], labels=sol[1], stats=stats), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_selection_when_a_rank_finished_no_trial_and_parts_are_not_fp32(tmp_path):
    import torch.multiprocessing as mp

    mp.spawn(_worker_ragged, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        out = torch.load(tmp_path / f"rank{r}.pt", weights_only=False)
        assert out["value"] == 0.5
        assert out["data"].dtype == torch.float64 and out["data"].tolist() == [[1.0 + 2.0 ** -40 * 3] * 3]
        assert out["labels"].dtype == torch.int64 and out["labels"].tolist() == [[2 ** 40 + 2] * 2]
        assert sorted(out["stats"]) == ["Trial_0_Val", "Trial_2_Val"]


def _worker(rank, world, port, num_trials, out_dir):
    import torch.distributed as dist

    from breaching_amd.trials import TrialShard

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = TrialShard.current(num_trials)
        assert (shard.rank, shard.world) == (rank, world)
        gen = torch.Generator().manual_seed(0)
        all_scores = torch.rand(num_trials, generator=gen)
        if num_trials > 3:
            all_scores[3] = float("inf")
        sols, scores, stats = {}, {}, {}
        for t in shard.local_trials():
            sols[t] = (torch.full((1, 3, 4, 4), float(t)), torch.full((1, 5), float(-t)))  # joint attacker: tuple
            scores[t] = all_scores[t]
            stats[f"Trial_{t}_Val"] = [float(t)] * (t + 1)
        value, sol = shard.select(sols, scores, stats, torch.device("cpu"))
        torch.save(dict(value=value, data=sol[0], labels=sol[1], stats=stats, expect=int(all_scores.argmin()),
                        expect_value=float(all_scores.min())), os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_trials", [5, 2, 1])
def test_two_rank_gloo_selection(tmp_path, num_trials):
    """World size 2 on CPU (gloo): every rank ends with the same winner, the winner's tensors and all trial stats."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_worker, args=(2, port, num_trials, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt", weights_only=False) for r in range(2))
    for r in (r0, r1):
        assert r["value"] == pytest.approx(r["expect_value"])
        assert float(r["data"].flatten()[0]) == float(r["expect"])
        assert float(r["labels"].flatten()[0]) == -float(r["expect"])
        assert sorted(r["stats"]) == [f"Trial_{t}_Val" for t in range(num_trials)]
        for t in range(num_trials):
            assert r["stats"][f"Trial_{t}_Val"] == [float(t)] * (t + 1)
    assert r0["value"] == r1["value"]


def test_dry_collective_one_rank_gloo():
    """The one-rank run of the selection collectives (what bench.py and the GPU suite execute with "nccl" on the GPU box),
    here over gloo on CPU, in a subprocess like there."""
    import subprocess
    import sys

    from breaching_amd.trials import parse_dry_collective

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, "-m", "breaching_amd.trials", "--dry-collective", "gloo", "cpu"], cwd=root,
                          capture_output=True, text=True, timeout=240)
    assert proc.returncode == 0, proc.stderr[-2000:]
    record = parse_dry_collective(proc.stdout)
    assert record is not None and record["ok"] and record["backend"] == "gloo" and record["world"] == 1 and record["value"] == 0.25


def test_config_overrides_and_attrdict():
    from breaching_amd.config import AttrDict, get_attack_config

    cfg = get_attack_config("invertinggradients", ["optim.max_iterations=7", "regularization.total_variation.scale=0.5",
                                                   "optim.step_size_decay=null", "restarts.num_trials=4"])
    assert cfg.optim.max_iterations == 7 and cfg["optim"]["step_size_decay"] is None
    assert cfg.regularization.total_variation.scale == 0.5 and cfg.restarts.num_trials == 4
    assert dict(**cfg.objective) == dict(type="cosine-similarity", scale=1.0, task_regularization=0.0)
    assert list(cfg.regularization.keys()) == ["total_variation"]
    assert isinstance(cfg.optim, AttrDict)
    with pytest.raises(ValueError):
        get_attack_config("does-not-exist")
    untouched = get_attack_config("invertinggradients")
    assert untouched.optim.max_iterations == 24_000  # overrides never leak into the templates


def test_optimizer_lookup_names():
    from breaching_amd.schedules import optimizer_hparams

    assert optimizer_hparams("Adam")["eps"] == 1e-8
    assert optimizer_hparams("bert-adam") == dict(betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, decoupled=True)
    assert optimizer_hparams("L-BFGS") is None and optimizer_hparams("momGD") is None
    with pytest.raises(ValueError):
        optimizer_hparams("adagrad")


def test_adam_schedule_table_matches_torch_scalars():
    from breaching_amd.schedules import adam_schedule_table

    table = adam_schedule_table([0.1, 0.05, 0.0], 0.9, 0.999, 0.01)
    for k, lr in enumerate([0.1, 0.05, 0.0]):
        step = k + 1
        assert table[k, 0] == lr / (1 - 0.9**step)
        assert table[k, 1] == (1 - 0.999**step) ** 0.5
        assert table[k, 2] == 1 - lr * 0.01 and table[k, 3] == lr


def test_install_rebinds_the_reference_factory():
    """`breaching_amd.install()` makes the reference's own `prepare_attack` build our attackers (drop-in seam used by
    simulate_breach.py:38).  Needs the reference checkout, i.e. runs in the build container only."""
    from oracle.ref_shim import have_reference, import_reference

    if not have_reference():
        pytest.skip("reference checkout not present")
    breaching = import_reference()
    import breaching_amd
    from breaching_amd.attacker import HipOptimizationAttacker, HipOptimizationJointAttacker
    from breaching_amd.cases import build_case

    original = (breaching.attacks.OptimizationBasedAttacker, breaching.attacks.OptimizationJointAttacker)
    try:
        breaching_amd.install()
        assert breaching.attacks.OptimizationBasedAttacker is HipOptimizationAttacker
        assert breaching.attacks.OptimizationJointAttacker is HipOptimizationJointAttacker
        case = build_case("convnet", "CIFAR10", 1)
        cfg = breaching_amd.get_attack_config("invertinggradients")
        # the reference factory now dispatches to the HIP attacker, which refuses a CPU device instead of falling back
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
        cfg.attack_type = "analytic"  # every other attack type keeps the reference implementation
        assert type(breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg)).__name__ == "AnalyticAttacker"
    finally:
        breaching.attacks.OptimizationBasedAttacker, breaching.attacks.OptimizationJointAttacker = original


def test_initialisation_schemes_match_reference_bitwise():
    """Candidate initialisation (base_attack.py:222-285) restated: same seed, same draws, same tensors for every scheme.
    Runs the method unbound on CPU against the reference's (build container only)."""
    from oracle.ref_shim import have_reference, import_reference

    if not have_reference():
        pytest.skip("reference checkout not present")
    import_reference()
    from breaching.attacks.base_attack import _BaseAttacker

    from breaching_amd.attacker import HipOptimizationAttacker
    from breaching_amd.config import AttrDict

    class Holder:
        pass

    schemes = ["randn", "randn-trunc", "rand", "zeros", "red", "green-true", "blue", "dark", "light-true", "rand-patterned-4",
               "randn-patterned-8", "patterned-4", "rand-wei-4", "wei-8", "randn-wei-4"]
    for scheme in schemes:
        drawn = []
        for cls in (_BaseAttacker, HipOptimizationAttacker):
            holder = Holder()
            holder.cfg = AttrDict(init=scheme)
            holder.setup = dict(device=torch.device("cpu"), dtype=torch.float32)
            holder.memory_format = torch.contiguous_format
            holder.dm = torch.tensor([0.4, 0.5, 0.6])[None, :, None, None]
            holder.ds = torch.tensor([0.2, 0.3, 0.25])[None, :, None, None]
            torch.manual_seed(0)
            drawn.append(cls._initialize_data(holder, [2, 3, 10, 12]).detach())
        assert torch.equal(drawn[0], drawn[1]), scheme
    holder.cfg = AttrDict(init="nonsense")
    with pytest.raises(ValueError):
        HipOptimizationAttacker._initialize_data(holder, [1, 3, 4, 4])


def test_label_recovery_matches_reference_on_cpu():
    """Every label strategy, three batch sizes, same seed: identical labels to the reference's
    `_recover_label_information` (base_attack.py:305-475), random padding included (same CPU draws)."""
    from oracle.ref_shim import have_reference, import_reference

    if not have_reference():
        pytest.skip("reference checkout not present")
    import_reference()
    from breaching.attacks.base_attack import _BaseAttacker

    from breaching_amd.attacker import HipOptimizationAttacker
    from breaching_amd.cases import build_case
    from breaching_amd.config import AttrDict

    class Holder:
        pass

    helpers = ("_labels_idlg", "_labels_analytic", "_labels_yin", "_labels_wainakh_simple", "_labels_bias_corrected")
    for n in (6, 1, 3):
        case = build_case("convnet", "CIFAR10", n, seed_data=5)
        for strategy in ("iDLG", "analytic", "yin", "wainakh-simple", "bias-corrected", "random"):
            got = []
            for cls in (_BaseAttacker, HipOptimizationAttacker):
                holder = Holder()
                holder.cfg = AttrDict(label_strategy=strategy)
                holder.setup = dict(device=torch.device("cpu"), dtype=torch.float32)
                if cls is HipOptimizationAttacker:
                    for name in helpers:
                        fn = getattr(HipOptimizationAttacker, name)
                        setattr(holder, name, fn.__get__(holder) if name == "_labels_wainakh_simple" else fn)
                shared = [dict(gradients=[g.clone() for g in d["gradients"]], buffers=None, metadata=dict(d["metadata"], labels=None))
                          for d in case.shared_data]
                torch.manual_seed(0)
                got.append(cls._recover_label_information(holder, shared, case.server_payload, [case.model]))
            assert torch.equal(got[0], got[1]), (n, strategy)


def _bare_attacker(cls, model, loss_fn, cfg):
    """An attacker object without running __init__'s device check: the preparation code is device agnostic."""
    import copy

    obj = object.__new__(cls)
    obj.cfg = cfg
    obj.memory_format = torch.contiguous_format
    obj.setup = dict(device=torch.device("cpu"), dtype=torch.float32)
    obj.model_template = copy.deepcopy(model)
    obj.loss_fn = copy.deepcopy(loss_fn)
    return obj


def test_prepare_attack_matches_reference_on_cpu():
    """`prepare_attack` (base_attack.py:43-74, 169-220, 298-303): model rebuild from payload / user buffers / no buffers,
    gradient cast and normalisation, box constants, labels -- compared field by field with the reference on CPU."""
    from oracle.ref_shim import have_reference, import_reference

    if not have_reference():
        pytest.skip("reference checkout not present")
    breaching = import_reference()
    from breaching_amd import get_attack_config
    from breaching_amd.attacker import HipOptimizationAttacker
    from breaching_amd.cases import build_case

    for scenario in ("server-buffers", "user-buffers", "no-buffers"):
        case = build_case("convnet", "CIFAR10", 2, provide_buffers=(scenario == "user-buffers"))
        if scenario == "no-buffers":
            case.server_payload[0]["buffers"] = None
        cfg = get_attack_config("invertinggradients", ["normalize_gradients=True"])
        ref = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
        ours = _bare_attacker(HipOptimizationAttacker, case.model, case.loss_fn, cfg)
        results = []
        for attacker in (ref, ours):
            shared = [dict(gradients=[g.clone() for g in d["gradients"]], buffers=d["buffers"], metadata=dict(d["metadata"]))
                      for d in case.shared_data]
            models, labels, stats = attacker.prepare_attack(case.server_payload, shared)
            results.append((models, labels, shared, attacker.dm, attacker.ds, attacker.data_shape))
        (m_r, l_r, s_r, dm_r, ds_r, shape_r), (m_o, l_o, s_o, dm_o, ds_o, shape_o) = results
        assert torch.equal(l_r, l_o) and torch.equal(dm_r, dm_o) and torch.equal(ds_r, ds_o) and list(shape_r) == list(shape_o)
        assert m_r[0].training == m_o[0].training == (scenario == "no-buffers")
        for p, q in zip(m_r[0].parameters(), m_o[0].parameters()):
            assert torch.equal(p, q)
        for p, q in zip(m_r[0].buffers(), m_o[0].buffers()):
            assert torch.equal(p, q)
        flags_r = [getattr(m, "track_running_stats", None) for m in m_r[0].modules()]
        flags_o = [getattr(m, "track_running_stats", None) for m in m_o[0].modules()]
        assert flags_r == flags_o
        for g, h in zip(s_r[0]["gradients"], s_o[0]["gradients"]):  # cast + normalised in place, like the reference
            assert torch.equal(g, h)
        norm = torch.stack([g.pow(2).sum() for g in s_o[0]["gradients"]]).sum().sqrt()
        assert abs(float(norm) - 1.0) < 1e-5


def test_text_preparation_and_token_recovery_match_reference_on_cpu():
    """Text path of the joint attacker on a tiny BERT: embedding cut-off (base_attack.py:76-122), label-candidate draw
    (optimization_with_label_attack.py:42-49) and token recovery (base_attack.py:124-167), field by field on CPU."""
    from oracle.ref_shim import have_reference, import_reference

    if not have_reference():
        pytest.skip("reference checkout not present")
    breaching = import_reference(preload_transformers=True)
    from breaching_amd import get_attack_config
    from breaching_amd.attacker import HipOptimizationJointAttacker
    from breaching_amd.cases import build_text_case

    for recovery in ("from-embedding", "from-labels", "from-limited-embedding"):
        cfg = get_attack_config("tag", [f"token_recovery={recovery}"])
        out = []
        for which in ("ref", "ours"):
            case = build_text_case()
            if which == "ref":
                attacker = breaching.attacks.prepare_attack(case.model, case.loss_fn, cfg, dict(device=torch.device("cpu"), dtype=torch.float))
            else:
                attacker = _bare_attacker(HipOptimizationJointAttacker, case.model, case.loss_fn, cfg)
            torch.manual_seed(2)
            models, labels, _ = attacker.prepare_attack(case.server_payload, case.shared_data)
            gen = torch.Generator().manual_seed(9)
            embeddings = torch.randn(1, 8, 64, generator=gen)
            token_labels = torch.randint(0, 300, (1, 8), generator=gen)
            rec = attacker._postprocess_text_data(dict(data=embeddings.clone(), labels=token_labels.clone()))
            out.append(dict(labels=labels.detach(), shape=list(attacker.data_shape), n_grads=len(case.shared_data[0]["gradients"]),
                            n_params=len(list(models[0].parameters())), tokens=rec["data"],
                            identity=type(models[0].model.bert.embeddings.word_embeddings).__name__))
        ref, ours = out
        assert torch.equal(ref["labels"], ours["labels"]) and ref["shape"] == ours["shape"] == [8, 64]
        assert ref["n_grads"] == ours["n_grads"] == 41 and ref["n_params"] == ours["n_params"] == 42
        assert ref["identity"] == ours["identity"] == "Identity"
        assert torch.equal(ref["tokens"], ours["tokens"]), recovery


def test_generic_loop_optimizers_and_schedule_match_reference_on_cpu():
    """`_init_optimizer` (base_attack.py:287-296 -> common.py:5-40) for every optimiser name: same torch.optim class, same
    hyper-parameters, and the same learning rate before every step as the reference's scheduler produces."""
    from oracle.ref_shim import have_reference, import_reference

    if not have_reference():
        pytest.skip("reference checkout not present")
    import_reference()
    from breaching.attacks.auxiliaries.common import optimizer_lookup

    from breaching_amd import get_attack_config
    from breaching_amd.attacker import HipOptimizationAttacker

    for name, sched, warm in (("adam", "step-lr", 0), ("adam-safe", "cosine-decay", 3), ("bert-adam", "linear", 5),
                              ("momgd", None, 0), ("gd", "step-lr", 2), ("l-bfgs", None, 0)):
        cfg = get_attack_config("invertinggradients", [f"optim.optimizer={name}", f"optim.step_size_decay={sched or 'null'}",
                                                       f"optim.warmup={warm}", "optim.max_iterations=40", "optim.step_size=0.3"])
        ours = object.__new__(HipOptimizationAttacker)
        ours.cfg = cfg
        p_ref, p_ours = [torch.zeros(3, requires_grad=True)], [torch.zeros(3, requires_grad=True)]
        opt_r, sch_r = optimizer_lookup(p_ref, name, 0.3, scheduler=sched, warmup=warm, max_iterations=40)
        opt_o, sch_o = ours._init_optimizer(p_ours)
        assert type(opt_r) is type(opt_o)
        skip = {"params", "lr", "initial_lr"}
        hyper_r = {k: v for k, v in opt_r.param_groups[0].items() if k not in skip}
        hyper_o = {k: v for k, v in opt_o.param_groups[0].items() if k not in skip}
        assert hyper_r == hyper_o
        for _ in range(40):
            assert opt_r.param_groups[0]["lr"] == opt_o.param_groups[0]["lr"]
            if name != "l-bfgs":
                for p in (p_ref[0], p_ours[0]):
                    p.grad = torch.ones(3)
                opt_r.step()
                opt_o.step()
            sch_r.step()
            sch_o.step()
    bad = object.__new__(HipOptimizationAttacker)
    bad.cfg = get_attack_config("invertinggradients", ["optim.optimizer=adagrad"])
    with pytest.raises(ValueError):
        bad._init_optimizer([torch.zeros(1, requires_grad=True)])


def test_implementation_switches(monkeypatch):
    """cfg.impl / environment switches of the host side: graph capture policy, restarts in flight (clamped to 4), how the
    eval-mode BatchNorm and LayerNorm layers of the model copy run."""
    from breaching_amd import attacker, get_attack_config

    for name in ("BREACH_HIP_GRAPH", "BREACH_HIP_GRAPH_STRICT", "BREACH_HIP_TRIALS_IN_FLIGHT", "BREACH_HIP_FAST_BN", "BREACH_HIP_FAST_LN"):
        monkeypatch.delenv(name, raising=False)
    cfg = get_attack_config("invertinggradients")
    assert attacker.graph_replay_policy(cfg) == "auto" and attacker.graph_replay_enabled(cfg)  # never a crash by default
    assert attacker.trials_in_flight(cfg) == attacker.DEFAULT_TRIALS_IN_FLIGHT == 4
    assert attacker.fast_eval_bn_mode(cfg) == "hip" and attacker.fast_layer_norm_enabled(cfg)
    cfg = get_attack_config("invertinggradients", ["impl.hip_graph=required", "impl.trials_in_flight=9", "impl.fast_eval_bn=addcmul",
                                                   "impl.fast_layer_norm=False"])
    assert attacker.graph_replay_policy(cfg) == "required" and attacker.graph_replay_enabled(cfg)
    assert attacker.trials_in_flight(cfg) == attacker.MAX_TRIALS_IN_FLIGHT  # more than four concurrent replays run slower
    assert attacker.fast_eval_bn_mode(cfg) == "addcmul" and not attacker.fast_layer_norm_enabled(cfg)
    cfg = get_attack_config("invertinggradients", ["impl.hip_graph=False", "impl.fast_eval_bn=False", "impl.trials_in_flight=2"])
    assert attacker.graph_replay_policy(cfg) == "off" and not attacker.graph_replay_enabled(cfg)
    assert attacker.fast_eval_bn_mode(cfg) is None and not attacker.fast_eval_bn_enabled(cfg) and attacker.trials_in_flight(cfg) == 2
    monkeypatch.setenv("BREACH_HIP_GRAPH", "required")
    assert attacker.graph_replay_policy(cfg) == "required"
    monkeypatch.setenv("BREACH_HIP_GRAPH", "auto")
    monkeypatch.setenv("BREACH_HIP_GRAPH_STRICT", "1")
    assert attacker.graph_replay_policy(cfg) == "required"
    monkeypatch.delenv("BREACH_HIP_GRAPH")
    assert attacker.graph_replay_policy(cfg) == "off"  # cfg says hip_graph=False: strictness never switches capture on
    monkeypatch.delenv("BREACH_HIP_GRAPH_STRICT")
    monkeypatch.setenv("BREACH_HIP_GRAPH", "auto")
    monkeypatch.setenv("BREACH_HIP_FAST_BN", "hip")
    monkeypatch.setenv("BREACH_HIP_FAST_LN", "1")
    monkeypatch.setenv("BREACH_HIP_TRIALS_IN_FLIGHT", "1")
    assert attacker.graph_replay_policy(cfg) == "auto" and attacker.fast_eval_bn_mode(cfg) == "hip"
    assert attacker.fast_layer_norm_enabled(cfg) and attacker.trials_in_flight(cfg) == 1


def test_model_copy_modules_are_swapped_in_place():
    """The eval-BatchNorm / LayerNorm replacements are class swaps: same parameters, buffers, state_dict keys and isinstance."""
    from breaching_amd.attacker import _EvalAffineBatchNorm2d, _HipLayerNorm, use_affine_eval_batchnorm, use_hip_layernorm

    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Flatten(), torch.nn.LayerNorm(4 * 6 * 6))
    keys = list(model.state_dict())
    params = [id(p) for p in model.parameters()]
    use_affine_eval_batchnorm(model, "addcmul")
    use_hip_layernorm(model)
    assert type(model[1]) is _EvalAffineBatchNorm2d and model[1].eval_mode == "addcmul" and isinstance(model[1], torch.nn.BatchNorm2d)
    assert type(model[3]) is _HipLayerNorm and isinstance(model[3], torch.nn.LayerNorm)
    assert list(model.state_dict()) == keys and [id(p) for p in model.parameters()] == params
    # on CPU both fall through to torch's own arithmetic (the HIP functions need a ROCm tensor): same numbers as the stock modules
    reference = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Flatten(), torch.nn.LayerNorm(4 * 6 * 6))
    reference.load_state_dict(model.state_dict())
    x = torch.randn(2, 3, 8, 8)
    torch.testing.assert_close(model.eval()(x), reference.eval()(x), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(model.train()(x), reference.train()(x), rtol=1e-5, atol=1e-6)
    # torch.func transforms: the custom autograd Functions have no functorch rules, so the swapped modules must notice a
    # transform (and route to the torch formulation); vmap(grad) over the swapped model works and agrees with the stock one
    from breaching_amd.attacker import _under_functorch

    seen = []

    def probe(v):
        seen.append(_under_functorch(v))
        return (v * v).sum()

    assert not _under_functorch(x)
    torch.func.vmap(torch.func.grad(probe))(x)
    assert seen == [True]
    model.eval(), reference.eval()
    per_sample = lambda m: torch.func.vmap(torch.func.grad(lambda v: (m(v[None]) ** 2).sum()))(x)  # noqa: E731
    torch.testing.assert_close(per_sample(model), per_sample(reference), rtol=1e-3, atol=1e-4)


def test_pending_batchnorm_absorbs_residual_and_relu_without_changing_the_model(monkeypatch):
    """`_PendingBatchNorm`: the eval-BatchNorm output waits for its first consumer so that `+ identity` and `relu` ride in the
    BatchNorm's launch.  The peephole logic is host code: here the kernel launch is replaced by its torch formula (and the
    modules are told they run on the GPU), and ResNet-style blocks written in every common way -- `out += identity` then an
    in-place ReLU module, `torch.relu(out + identity)`, `out.add_(identity); self.relu(out)` without rebinding, `identity + out`
    then `.relu()`, and a BatchNorm output that is used twice by non-ReLU consumers -- give the values and input gradients of
    the stock modules, with exactly the launches expected (fused where the pattern is there, plain otherwise, none for a
    BatchNorm output nobody uses)."""
    import copy

    import breaching_amd.victim_layers as A

    calls = []

    def fake_launch(module, x, sink, tap, residual, relu):
        calls.append((residual is not None, bool(relu)))
        inv_std, mean_inv = module._frozen_statistics()
        z = x * (module.weight * inv_std).view(1, -1, 1, 1) + (module.bias - module.weight * mean_inv).view(1, -1, 1, 1)
        if residual is not None:
            z = z + residual
        return torch.relu(z) if relu else z

    monkeypatch.setattr(A, "_launch_eval_bn", fake_launch)
    monkeypatch.delenv("BREACH_HIP_FUSE_BN_RELU", raising=False)

    class Block(torch.nn.Module):
        def __init__(self, inplace, style):
            super().__init__()
            self.c1, self.b1 = torch.nn.Conv2d(4, 4, 3, padding=1), torch.nn.BatchNorm2d(4)
            self.c2, self.b2 = torch.nn.Conv2d(4, 4, 3, padding=1), torch.nn.BatchNorm2d(4)
            self.relu = torch.nn.ReLU(inplace=inplace)
            self.down = torch.nn.Sequential(torch.nn.Conv2d(4, 4, 1), torch.nn.BatchNorm2d(4))
            self.style = style

        def forward(self, x):
            identity = self.down(x)
            out = self.relu(self.b1(self.c1(x)))
            out = self.b2(self.c2(out))
            if self.style == 0:
                out += identity
                out = self.relu(out)
            elif self.style == 1:
                out = torch.relu(out + identity)
            elif self.style == 2:
                out.add_(identity)
                self.relu(out)  # in-place module, result not rebound
            elif self.style == 3:
                out = identity + out
                out = out.relu()
            else:  # used twice, never through a ReLU first: the plain values must survive; `identity` is never consumed
                out = torch.tanh(out) + torch.relu(out) + out.shape[1]
            return out

    fused = [(False, True), (False, False), (True, True)]  # bn1 + relu; downsample bn (plain, it becomes the residual); bn2 + identity + relu
    expected = {0: fused, 1: fused, 2: fused, 3: fused, 4: [(False, True), (False, False)]}
    for style in range(5):
        for inplace in (True, False):
            if style == 2 and not inplace:
                continue
            torch.manual_seed(style)
            ref = Block(inplace, style).eval()
            for m in ref.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.normal_()
                    m.running_var.uniform_(0.5, 2)
                    m.weight.data.normal_(1, 0.2)
                    m.bias.data.normal_()
            hip = A.use_affine_eval_batchnorm(copy.deepcopy(ref), "hip")
            for m in hip.modules():
                if isinstance(m, A._EvalAffineBatchNorm2d):
                    m._runs_on_hip = lambda x: True  # the launch itself is the torch formula above
            x = torch.randn(2, 4, 8, 8)
            calls.clear()
            y_hip = hip(x.clone())
            assert calls == expected[style], (style, inplace, calls)
            if style != 2:
                assert type(y_hip) is torch.Tensor
            torch.testing.assert_close(y_hip + 0, ref(x), rtol=1e-5, atol=1e-5)
            xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            (ga,) = torch.autograd.grad((hip(xa) ** 2).sum(), xa)
            (gb,) = torch.autograd.grad((ref(xb) ** 2).sum(), xb)
            torch.testing.assert_close(ga, gb, rtol=1e-4, atol=1e-5)
    # a BatchNorm fed DIRECTLY by a BatchNorm: the waiting output is resolved at the module boundary (an autograd.Function must
    # never receive the metadata-only wrapper as an input -- it carries no autograd edge, the gradient would silently stop)
    chain = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.BatchNorm2d(4), torch.nn.ReLU()).eval()
    for m in chain:
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2)
    chain_hip = A.use_affine_eval_batchnorm(copy.deepcopy(chain), "hip")
    for m in chain_hip:
        if isinstance(m, A._EvalAffineBatchNorm2d):
            m._runs_on_hip = lambda x: True
    xc = torch.randn(2, 3, 8, 8)
    xa, xb = xc.clone().requires_grad_(True), xc.clone().requires_grad_(True)
    calls.clear()
    (ga,) = torch.autograd.grad((chain_hip(xa) ** 2).sum(), xa)
    (gb,) = torch.autograd.grad((chain(xb) ** 2).sum(), xb)
    assert calls == [(False, False), (False, True)]
    torch.testing.assert_close(ga, gb, rtol=1e-4, atol=1e-5)
    # ... and a THIRD-PARTY autograd.Function fed directly by a BatchNorm fails loudly instead of dropping its gradient
    class Twice(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t * 2

        @staticmethod
        def backward(ctx, g):
            return g * 2

    with pytest.raises(RuntimeError, match="custom autograd.Function"):
        Twice.apply(chain_hip[:2](xc.clone().requires_grad_(True)))
    with torch.no_grad():  # whole forward without autograd: nothing to lose, no error
        torch.testing.assert_close(chain_hip(xc), chain(xc), rtol=1e-5, atol=1e-5)
    # switched off: the modules launch immediately, nothing is deferred
    monkeypatch.setenv("BREACH_HIP_FUSE_BN_RELU", "0")
    calls.clear()
    hip(x)
    assert calls == [(False, False)] * 3
    monkeypatch.delenv("BREACH_HIP_FUSE_BN_RELU")
    A.use_affine_eval_batchnorm(hip, "hip", fuse_epilogue=False)
    calls.clear()
    hip(x)
    assert calls == [(False, False)] * 3


def test_design_tables_are_generated_from_the_committed_profiles():
    """DESIGN.md section 6's round-6 block and profiles/TABLES.md (the round-5 tables in full) are the output of
    scripts/design_tables.py over the files under profiles/: a number in the text cannot drift from the committed evidence (round 3
    had a quoted 40.2 / 47.6 us that the summary file no longer showed).  Regenerate with `python scripts/design_tables.py --write`."""
    import importlib.util
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("design_tables", os.path.join(root, "scripts", "design_tables.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    text = open(os.path.join(root, "DESIGN.md")).read()
    assert module.BEGIN in text and module.END in text
    block = text[text.index(module.BEGIN): text.index(module.END) + len(module.END)]
    assert block == module.build(), "DESIGN.md is stale: run `python scripts/design_tables.py --write`"
    detail = open(module.DETAIL_PATH).read()
    assert detail == module.DETAIL_HEAD + module.build_detail() + "\n", "profiles/TABLES.md is stale: run `python scripts/design_tables.py --write`"
    # the files the two blocks quote exist, and the round-6 block covers what the round-5 review asked the line to carry
    quoted = set(re.findall(r"`(r[56]_[A-Za-z0-9_.]+\.(?:json|jsonl|txt|csv))`", block + detail))
    assert len([q for q in quoted if q.startswith("r6_")]) >= 8 and len([q for q in quoted if q.startswith("r5_")]) >= 12, quoted
    for name in quoted | {"r6_bench_driver_style.json", "r6_bench_kernel_summary.txt", "r6_affine_layers.json", "r5_read_ceiling_probe.jsonl",
                          "r5_hip_64starts_1000its.json"}:
        assert os.path.exists(os.path.join(root, "profiles", name)), name
    for needle in ("BASELINE configs[3] through `attacker.reconstruct`", "Kernel E at BASELINE configs[2]'s batch size", "on the SAME inputs",
                   "Trial batching, decided with a number", "Joint data + label attack"):
        assert needle in block, needle
    assert len(text) < 36_000  # the current-state document stays readable; history lives in HISTORY.md


def test_bench_hbm_resident_list_is_bert_base_without_the_word_embedding():
    """bench.py's `roofline.hbm_resident` leg times kernel A on a synthetic list with the shapes a TAG attack on BERT-base matches
    (SURVEY.md section 8 size table: 201 of 202 tensors, 86 073 402 elements): checked against the real model's parameter list."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    shapes = bench.bert_base_gradient_shapes()
    assert len(shapes) == 201 and sum(math.prod(s) for s in shapes) == 86_073_402
    from breaching_amd.cases import build_text_case

    case = build_text_case(device="cpu", full_size=True, seq_len=32)
    real = [tuple(p.shape) for name, p in case.model.named_parameters() if "word_embeddings" not in name]
    assert sorted(real) == sorted(shapes)
    # the two figures the line takes from COMMITTED profiles (bench.py cannot collect them itself): PMC traffic per launch of kernel A at
    # the headline list size -- within 1 % of the algorithmic bytes -- and the kernel's dispatch duration inside graph replay
    n = 11_689_512
    sources = []
    fwd, bwd = bench.pmc_traffic_bytes("gm_fwd_kernel", sources), bench.pmc_traffic_bytes("gm_bwd_kernel")
    assert sources and all(name.startswith("r6_pmc_") for name in sources)
    assert 1.0 <= fwd / (2 * n * 4) <= 1.01 and 1.0 <= bwd / (3 * n * 4) <= 1.01
    us, name = bench.committed_replay_duration("gm_fwd_kernel<0")
    assert name == "r6_bench_kernel_summary.txt" and 12.0 < us < 25.0


def test_bench_cpu_baseline_is_the_unmodified_reference_when_a_checkout_is_importable(monkeypatch):
    """bench.py's `cpu_baseline` leg (no GPU needed): with the reference checkout present (this container) it times the
    UNMODIFIED reference (`kind: "reference"`); pointed at a directory without one (the GPU box) it times the port and reports
    the committed port / reference anchor ratio."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    threads = torch.get_num_threads()
    try:
        if os.path.isdir(os.path.join(os.environ.get("BREACHING_REFERENCE", "/root/reference"), "breaching")):
            leg = bench.cpu_baseline_leg("resnet18", 2, cpu_threads=4)
            assert leg["kind"] == "reference" and leg["value"] > 0 and leg["cores"] == 4 and "UNMODIFIED reference" in leg["sample"]
            assert "anchor" not in leg
        monkeypatch.setenv("BREACHING_REFERENCE", "/nonexistent")
        leg = bench.cpu_baseline_leg("resnet18", 2, cpu_threads=4)
        assert leg["kind"] == "port" and leg["value"] > 0 and "oracle/restate.py" in leg["sample"]
        assert leg["anchor"]["port_over_reference"] == pytest.approx(1.0, abs=0.1) and leg["anchor"]["file"].endswith("cpu_baseline_anchor.json")
    finally:
        torch.set_num_threads(threads)


def test_gap_census_on_a_synthetic_trace(tmp_path):
    """scripts/gap_census.py (the instrument behind DESIGN.md section 6's dispatch / gap tables) on a trace whose answer is known:
    two queues, ten iterations of 50 dispatches each ending in a candidate step, 2 us kernels, 1 us gaps, one 40 us gap per
    iteration behind an Im2Col."""
    import csv
    import importlib.util
    import json
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = []
    for queue in ("1", "2"):
        t = 1_000_000 + int(queue) * 300
        for _ in range(10):
            for k in range(50):
                name = "candidate_step_kernel(Word)" if k == 49 else ("Im2d2Col_v2" if k == 10 else "void at::native::vectorized_elementwise_kernel<4, at::native::CUDAFunctor_add<float>>")
                gap = 40_000 if k == 11 else 1_000
                rows.append(dict(Kind="KERNEL_DISPATCH", Queue_Id=queue, Stream_Id=queue, Kernel_Name=name, Start_Timestamp=t + gap, End_Timestamp=t + gap + 2_000,
                                 Grid_Size_X=256, Grid_Size_Y=1, Grid_Size_Z=1, Workgroup_Size_X=256, Workgroup_Size_Y=1, Workgroup_Size_Z=1, Scratch_Size=0))
                t += gap + 2_000
    trace = tmp_path / "1_kernel_trace.csv"
    with open(trace, "w") as f:
        writer = csv.DictWriter(f, fieldnames=list(rows[0]))
        writer.writeheader()
        writer.writerows(rows)
    spec = importlib.util.spec_from_file_location("gap_census", os.path.join(root, "scripts", "gap_census.py"))
    census = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(census)
    import sys

    argv, sys.argv = sys.argv, ["gap_census.py", str(trace), str(tmp_path / "out"), "--iters", "5"]
    try:
        census.main()
    finally:
        sys.argv = argv
    result = json.load(open(tmp_path / "out.json"))
    assert sorted(result["per_queue"]) == ["1", "2"]
    q = result["per_queue"]["1"]
    assert q["iterations"] == 5 and q["dispatches_per_iter"] == 50 and q["kernel_us_per_iter"] == 100.0
    assert q["gap_us_per_iter"] == pytest.approx(49 * 1.0 + 40.0) and q["wall_us_per_iter"] == pytest.approx(189.0)
    families = {f["family"]: f for f in q["families"]}
    assert families["MIOpen: Im2Col / Col2Im"]["gap_after_us_per_iter"] == pytest.approx(40.0)  # the long gap is charged to its predecessor ...
    assert families["ATen: add (accumulation, residual)"]["gap_before_us_per_iter"] == pytest.approx(40.0 + 47.0)  # ... and to its successor
    assert families["ours: kernel B/C, commit (step, tv, loss)"]["calls_per_iter"] == 1.0
    assert os.path.exists(tmp_path / "out_one_iteration.csv")


def test_fail_open_policies_parse_config_and_environment(monkeypatch):
    """The three policies that decide whether an optimisation of ours may stop an attack -- hipGraph capture, the BatchNorm -> ReLU
    fusion -- default to "auto" (fail open: fall back, say so), turn into "required" only when asked to, and the environment
    overrides the config (what the test suite uses to make every silent fall-back an error)."""
    from breaching_amd import get_attack_config
    from breaching_amd.attacker import graph_replay_policy
    from breaching_amd.victim_layers import fuse_bn_relu_enabled, fuse_bn_relu_policy

    for var in ("BREACH_HIP_FUSE_BN_RELU", "BREACH_HIP_GRAPH", "BREACH_HIP_GRAPH_STRICT"):
        monkeypatch.delenv(var, raising=False)
    base = get_attack_config("invertinggradients")
    assert fuse_bn_relu_policy(base) == "auto" and fuse_bn_relu_policy(None) == "auto" and graph_replay_policy(base) == "auto"
    for value, want in ((True, "auto"), ("auto", "auto"), ("required", "required"), (False, "off"), ("0", "off"), (None, "auto")):
        cfg = get_attack_config("invertinggradients", [f"impl.fuse_bn_relu={value}"])
        assert fuse_bn_relu_policy(cfg) == want, (value, fuse_bn_relu_policy(cfg))
        assert fuse_bn_relu_enabled(cfg) == (want != "off")
        cfg = get_attack_config("invertinggradients", [f"impl.hip_graph={value}"])
        assert graph_replay_policy(cfg) == want, (value, graph_replay_policy(cfg))
    monkeypatch.setenv("BREACH_HIP_FUSE_BN_RELU", "required")
    assert fuse_bn_relu_policy(get_attack_config("invertinggradients", ["impl.fuse_bn_relu=False"])) == "required"
    monkeypatch.setenv("BREACH_HIP_FUSE_BN_RELU", "0")
    assert fuse_bn_relu_policy(base) == "off"
    monkeypatch.setenv("BREACH_HIP_GRAPH_STRICT", "1")  # tests/conftest.py: every "auto" becomes "required", "off" stays off
    assert graph_replay_policy(base) == "required"
    assert graph_replay_policy(get_attack_config("invertinggradients", ["impl.hip_graph=False"])) == "off"
