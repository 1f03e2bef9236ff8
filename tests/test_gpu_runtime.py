"""Runtime behaviour around the kernels on the GPU box: the RCCL code path of the trial selection on one rank, hipGraph
capture policy, one process running attack after attack (the reference's benchmark_breaches.py:60-70 pattern)."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_selection_collectives_run_through_rccl_on_one_rank():
    """`all_reduce(MIN)` on the packed int64 key + the flat winner broadcast + the stats gather, on device tensors, through a
    ONE-rank "nccl" (= RCCL) process group: communicator creation, both collectives and teardown are the code the 2 / 4 / 8
    GPU runs execute.  In a subprocess: the pytest process keeps no process group, and a hung
    communicator start cannot hang the suite."""
    proc = subprocess.run([sys.executable, "-m", "breaching_amd.trials", "--dry-collective", "nccl", "cuda:0"], cwd=ROOT,
                          capture_output=True, text=True, timeout=240)
    assert proc.returncode == 0, proc.stderr[-2000:]
    from breaching_amd.trials import parse_dry_collective

    record = parse_dry_collective(proc.stdout)
    assert record is not None, (proc.stdout[-1000:], proc.stderr[-1000:])
    print(" ", record)
    assert record["backend"] == "nccl" and record["world"] == 1 and record["device"] == "cuda:0"
    assert record["ok"] and record["value"] == 0.25
    assert record["ship_ok"] is True  # the broadcast `TrialWorkerPool.ship` sends job inputs with, device flats of mixed dtypes, through RCCL


def test_capture_failure_falls_back_visibly_and_raises_when_required(monkeypatch):
    """impl.hip_graph=True / "auto" (default): a failed capture continues with eager launches and says so in
    stats["execution"] (the reference attacks arbitrary models; one that cannot be captured must not crash the attack);
    "required" (this suite's and bench.py's mode): it raises; False: never captured."""
    import breaching_amd
    from breaching_amd.cases import build_case, initial_candidate

    case = build_case("convnet", "CIFAR10", 1, device="cuda:0")
    x0 = initial_candidate(case.data_cfg, 1, seed=6)
    setup = dict(device=torch.device("cuda:0"), dtype=torch.float)

    class _Broken:
        def __init__(self, *a, **k):
            raise RuntimeError("capture refused")

    def run(flag):
        cfg = breaching_amd.get_attack_config("invertinggradients", ["optim.max_iterations=6", "optim.callback=3", f"impl.hip_graph={flag}"])
        att = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, setup)
        return att.reconstruct(case.server_payload, case.shared_data, {}, initial_data=x0)

    monkeypatch.delenv("BREACH_HIP_GRAPH_STRICT", raising=False)  # the suite's own strictness would turn "auto" into "required"
    _, healthy = run("True")
    assert healthy["execution"]["trials"] == {0: "hipGraph replay"}
    monkeypatch.setattr(torch.cuda, "CUDAGraph", _Broken)
    with pytest.raises(RuntimeError, match="hipGraph capture of the attack iteration failed"):
        run("required")
    for flag in ("True", "auto"):
        _, stats = run(flag)
        assert stats["execution"]["trials"][0].startswith("eager launches (capture failed: RuntimeError('capture refused')")
        np.testing.assert_allclose(stats["Trial_0_Val"], healthy["Trial_0_Val"], rtol=1e-4)
    _, stats = run("False")
    assert stats["execution"]["trials"] == {0: "eager launches (graph replay switched off)"}


def _bisect(*flags):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stall_bisect.py"), *flags], cwd=ROOT, capture_output=True,
                          text=True, timeout=400)
    assert proc.returncode == 0, proc.stderr[-2000:]
    return json.loads(proc.stdout.strip().splitlines()[-1])


def test_restarts_in_flight_keep_their_rate_after_an_earlier_attack_in_the_same_process():
    """benchmark_breaches.py:60-70 runs attack after attack in ONE process.  Round 2 left a defect on exactly that pattern: the
    4-in-flight ResNet-18 restarts of configs[3] ran 3.6x slower (~120 instead of ~430 iterations/s) once any earlier attack
    of the process had replayed a hipGraph on the caller's stream -- one trial of the group used to share that stream
    (profiles/r3_stall_bisect.jsonl).  Two fresh processes, second pass of each (first pass = MIOpen warm-up): 4 restarts in
    flight alone, and the same after a ConvNet attack; both in hipGraph replay mode, rates within 25 % of each other."""
    alone = _bisect("--first", "none", "--its", "250", "--repeat", "2")
    after = _bisect("--first", "convnet", "--its", "250", "--repeat", "2")
    assert after["first"]["execution"] == "hipGraph replay"
    for record in (alone, after):
        assert all(p["execution"] == "hipGraph replay" and p["iterations"] == 1000 for p in record["passes"])
    rate_alone, rate_after = alone["passes"][1]["it_per_s"], after["passes"][1]["it_per_s"]
    print(f"  4 restarts in flight: {rate_alone} it/s alone, {rate_after} it/s after an earlier attack in the process")
    assert rate_after >= 0.75 * rate_alone


def test_side_streams_are_chosen_on_different_hardware_pipes():
    """Trials in flight need streams that do not share one of the four hardware compute pipes (two busy streams on one pipe
    run slower than one after the other: profiles/r4_inflight_pipes_probe.jsonl).  Fresh processes: (a) whatever streams the
    process created and used before, `side_streams` returns four streams that are pairwise collision-free by its own
    measurement, caches the choice and hands the same streams to the next group; (b) fed the 1st, 5th, 2nd, 6th, 3rd, 7th, 4th
    stream of the process in that order -- every second candidate on the pipe of its predecessor -- the measurement finds
    exactly those three collisions and keeps streams 1, 2, 3, 4."""
    code = r"""
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from breaching_amd import streams
dev = torch.device("cuda", 0)
mode, decoys = sys.argv[1], int(sys.argv[2])
pool = [torch.cuda.Stream(dev) for _ in range(8 if mode == "scrambled" else decoys)]
for s in pool:
    with torch.cuda.stream(s):
        torch.zeros(1, device=dev).add_(1)
torch.cuda.synchronize()
if mode == "scrambled":
    order = [pool[i] for i in (0, 4, 1, 5, 2, 6, 3, 7)]
    chosen, report = streams.calibrate(dev, 4, candidates=order)
    picked = [pool.index(s) for s in chosen]
    again = chosen[:3]
else:
    chosen = streams.side_streams(dev, 4)
    report, picked = streams.calibration_report(dev), None
    again = streams.side_streams(dev, 3)
(ga, _), (gb, _) = streams._probe_graphs(dev, chosen[0])
solo = min(streams._timed(dev, [(chosen[0], ga)]) for _ in range(3))
pairs = {}
for i in range(4):
    for j in range(i + 1, 4):
        pairs[f"{i}-{j}"] = min(streams._timed(dev, [(chosen[i], ga), (chosen[j], gb)]) for _ in range(2))
print(json.dumps(dict(report=report, solo_ms=solo, pairs=pairs, distinct=len({s.cuda_stream for s in chosen}), picked=picked,
                      cached=[a.cuda_stream == b.cuda_stream for a, b in zip(again, chosen)])))
"""
    for mode, decoys in (("plain", 0), ("plain", 1), ("plain", 3), ("scrambled", 0)):
        proc = subprocess.run([sys.executable, "-c", code, mode, str(decoys)], cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert proc.returncode == 0, proc.stderr[-2000:]
        rec = json.loads(proc.stdout.strip().splitlines()[-1])
        print(f"  {mode}, {decoys} earlier streams: candidates {rec['report']['candidates']}, collisions skipped {rec['report']['collisions']}, "
              f"solo {rec['solo_ms']:.3f} ms, worst pair {max(rec['pairs'].values()):.3f} ms, picked {rec['picked']}")
        assert rec["distinct"] == 4 and rec["cached"] == [True, True, True] and not rec["report"].get("incomplete")
        assert max(rec["pairs"].values()) <= 1.9 * rec["solo_ms"], rec  # no two of the chosen streams collide (clean ~1.2x, colliding ~3.1x)
        if mode == "scrambled":
            # measured on every box so far: streams i and i + 4 share a pipe, so candidates 2, 4, 6 of the scrambled order are skipped
            # (against chosen 0, 1, 2) and streams 1, 2, 3, 4 of the process are picked; what must hold on ANY box is that at least one
            # of the deliberately interleaved candidates was found colliding and none of the kept ones collide (asserted above)
            print(f"    scrambled order: picked process streams {rec['picked']}, skipped {[(c['candidate'], c['with_chosen']) for c in rec['report']['collisions']]}")
            assert len(rec["report"]["collisions"]) >= 1 and len(set(rec["picked"])) == 4


def test_side_streams_fall_back_to_pool_streams_when_the_calibration_cannot_run(monkeypatch):
    """The stream calibration captures and replays two probe hipGraphs; where that fails (capture unsupported, a profiler that
    refuses it) or comes back incomplete (a busy GPU, ranks sharing a device), `side_streams` hands out plain pool streams --
    round 3's behaviour -- and `calibration_report` carries the reason, instead of aborting the attack before a trial starts."""
    import torch

    from breaching_amd import streams

    dev = torch.device("cuda", 0)
    for failure in ("raises", "incomplete"):
        streams._CHOSEN.clear()

        def broken(device, n=streams.PIPES, candidates=None):
            if failure == "raises":
                raise RuntimeError("operation not permitted when stream is capturing")
            return [torch.cuda.Stream(device) for _ in range(n)], dict(method="probe", candidates=12, collisions=[], incomplete=True)

        monkeypatch.setattr(streams, "calibrate", broken)
        chosen = streams.side_streams(dev, 4)
        report = streams.calibration_report(dev)
        assert len(chosen) == 4 and len({s.cuda_stream for s in chosen}) == 4
        assert report["method"].startswith("fallback: next streams of torch's pool")
        assert ("failed" in report) == (failure == "raises")
        assert [a.cuda_stream for a in streams.side_streams(dev, 3)] == [s.cuda_stream for s in chosen[:3]]  # cached like a measured choice
    streams._CHOSEN.clear()


def test_bench_line_contract():
    """`python bench.py` (the driver's entry point) as a subprocess with small counts: ONE JSON line on stdout with the contract's
    fields -- BASELINE.json's metric, value = K steps / measured time, n_gpus, dtype of the arithmetic, a workload-naming config --
    and the legs this repository adds to it: `roofline` (HBM bound, frac = achieved / peak, algorithmic bytes of kernel A's forward,
    a measured read `ceiling`), `parity` (teacher-forced evaluation against the reference fixture, ok), `gpu_torch_baseline` (the port
    on the same GPU, slower than the HIP path), `cpu_baseline`."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--cpu-baseline-iters", "2",
           "--gpu-torch-baseline-iters", "4", "--roofline-steps", "6", "--no-live-pmc", "--no-hbm-resident", "--no-dry-collective",
           "--restarts32-iters", "12", "--restarts32-trials", "8"]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, proc.stdout[-1000:]
    line = json.loads(lines[0])
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert line["metric"] == json.load(f)["metric"]
    assert line["unit"] == "attack iterations/s" and line["n_gpus"] == 1 and line["steps"] == 6 and line["higher_is_better"] is True
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "resnet18" in line["config"]["workload"] and "model" not in line["config"]
    assert line["value"] == pytest.approx(6 / (line["ms_per_step"] * 6e-3), rel=1e-3) and line["launch_mode"] == "hipGraph replay"
    roof = line["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0 and roof["algorithmic_bytes"] == 2 * 11_689_512 * 4
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], abs=1e-3) and 0.3 < roof["frac"] < 1.0
    assert roof["traffic"] is None or 0.99 < roof["traffic"] / roof["algorithmic_bytes"] < 1.02
    assert 5.0 < roof["ceiling"]["us"] < 40.0 and 0.5 < roof["frac_of_ceiling"] <= 1.1
    parity = line["parity"]
    assert parity["ok"] is True and parity["iterate"] > 23_000 and parity["loss_rel_err"] <= parity["loss_tolerance"]
    # the gate is tied to the same-arithmetic control of the same run (PyTorch-ROCm ops on this GPU), and the HIP path sits with it
    # what the attack-side kernels add, on the SAME inputs as torch ops (no victim pass in between, nothing vendor-chosen): 1e-4 outright
    same = parity["kernels_on_the_same_inputs"]
    assert same["ok"] is True and max(same["gm_value_rel"], same["gm_gradient_max_dev_over_peak"], same["tv_value_rel"],
                                      same["tv_gradient_max_dev_over_peak"]) <= 1e-4
    # every full evaluation on this GPU -- the PyTorch-ROCm control included -- within 3 x the fixture's own 16-ulp sensitivity of the CPU run
    control, kernels = parity["control"], parity["kernels_A_C"]
    assert parity["loss_tolerance"] == pytest.approx(max(1e-4, 3.0 * parity["kink_sensitivity_recorded"]))
    assert max(control["loss_rel_err"], kernels["loss_rel_err"], parity["loss_rel_err"]) <= parity["loss_tolerance"]
    assert control["sign_agreement"] > 0.99 and kernels["sign_agreement_vs_control"] > 0.99
    assert 0 < line["gpu_torch_baseline"]["value"] < line["value"]
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["kind"] in ("reference", "port")
    _check_restarts_leg(line["restarts32"], trials=8, iters=12, ranks=1)
    assert line["restarts32"]["pool"] is None and line["restarts32"]["entry"] == "one process, one GPU"


def _check_restarts_leg(leg, trials, iters, ranks):
    """Contract of bench.py's `restarts32` record (BASELINE configs[3] through `attacker.reconstruct`): the whole call is timed --
    shipping, preparation, the trial loops, rescoring every trial, selection -- and accounted for."""
    assert "error" not in leg, leg
    assert leg["num_trials"] == trials and leg["iterations_per_trial"] == iters and leg["ranks"] == ranks
    assert leg["trials_per_rank"] == -(-trials // ranks) and leg["trials_in_flight_per_rank"] == min(4, -(-trials // ranks))
    assert leg["histories_complete"] is True and leg["launch_modes"] == ["hipGraph replay"]
    assert leg["trial_iterations_per_s"] == pytest.approx(trials * iters / leg["wall_s"], rel=1e-2) and leg["wall_s"] > 0
    timing = leg["timing_rank0"]
    parts = timing["prepare_s"] + timing["trials_s"] + timing["score_s"] + timing["select_s"]
    assert timing["trials_s"] > 0 and timing["score_s"] > 0 and parts <= timing["total_s"] * 1.001 <= leg["wall_s"] * 1.01
    assert parts >= 0.9 * timing["total_s"] or ranks > 1  # rank 0 of a pool also waits for the slowest worker (pool.trials_wait_s)
    assert np.isfinite(leg["opt_value"]) and 0 < leg["opt_value"] < 2 and 3.0 < leg["psnr_db_selected"] < 60.0
    assert leg["warmup_call_s"] > 0


@pytest.mark.trial_pool
def test_bench_restarts_leg_through_the_worker_pool():
    """`bench.py --restarts32-pool 2`: the leg's multi-GPU shape -- the single-process entry with a TrialWorkerPool -- as the N > 1
    bench runs it from rank 0 in a bounded subprocess; here two ranks share cuda:0 (gloo, `oversubscribed`).  The inputs of the
    ResNet-18 call (2 x 46.8 MB of parameters and gradients) reach the worker by broadcast over the group: the pipe carries
    kilobytes."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--restarts32-pool", "2", "--restarts32-iters", "12", "--restarts32-trials", "8"]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    leg = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    print(" ", {k: leg[k] for k in ("wall_s", "trial_iterations_per_s", "warmup_call_s", "pool")})
    _check_restarts_leg(leg, trials=8, iters=12, ranks=2)
    pool = leg["pool"]
    assert pool["backend"] == ("gloo" if leg["oversubscribed"] else "nccl") and pool["world"] == 2
    assert pool["job_ship_bytes"] >= 2 * 11_689_512 * 4 and pool["job_pipe_bytes"] < 256 * 1024
    assert pool["job_ship_s"] > 0 and pool["trials_wait_s"] >= 0 and pool["select_s"] > 0 and leg["warmup_pool"]["pool_start_s"] > 0


@pytest.mark.trial_pool
def test_bench_restarts_leg_with_eight_pool_ranks_on_one_gpu():
    """BASELINE configs[3]'s full shape -- 32 restarts over EIGHT ranks, four in flight each -- through the product's single-process
    entry, as rank 0 of `bench.py --gpus 8` runs it on an 8-GPU node; here the eight ranks share cuda:0 (gloo): start-up of seven
    workers, both shipments by broadcast, 32 graph-replayed trials, one selection.  Contract only (the trajectories of this shape are
    compared in test_configs3_shape_32_restarts_over_four_ranks_two_groups_of_four_each)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--restarts32-pool", "8", "--restarts32-iters", "8", "--restarts32-trials", "32"]
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stderr[-3000:]
    leg = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    print(" ", {k: leg[k] for k in ("wall_s", "trial_iterations_per_s", "warmup_call_s", "pool")})
    _check_restarts_leg(leg, trials=32, iters=8, ranks=8)
    assert leg["pool"]["world"] == 8 and leg["trials_per_rank"] == 4 and leg["trials_in_flight_per_rank"] == 4


def test_bench_multi_rank_launch_on_one_gpu():
    """`python bench.py --gpus 2` end to end, as the driver's scaling run launches it (torch.distributed.run, one rank per "GPU"; here
    both ranks on cuda:0 over gloo, `oversubscribed`): the staged start through rank 0's file flag (bounded -- no collective holds the
    other ranks), the weak-scaling line, the selection collective, and then -- after the ranks have left -- BASELINE configs[3]
    through the product's TrialWorkerPool in a bounded process of its own (`restarts32`, entry = TrialWorkerPool, 2 ranks)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--restarts32-iters", "8",
           "--restarts32-trials", "8"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    proc = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1200, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip().startswith("{\"metric\"")]
    assert len(lines) == 1, proc.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["oversubscribed"] is True and line["collective_backend"] == "gloo" and line["scaling"] == "weak"
    assert line["value"] == pytest.approx(2 * 4 / (line["ms_per_step"] * 4e-3), rel=1e-3) and len(line["per_rank_ms_per_step"]) == 2
    assert line["select_ms"] > 0 and line["score_ms"] > 0
    staged = line["staged_start"]["ranks"]
    assert [r["rank"] for r in staged] == [1] and staged[0]["released"] is True
    _check_restarts_leg(line["restarts32"], trials=8, iters=8, ranks=2)
    assert line["restarts32"]["entry"].startswith("TrialWorkerPool") and line["restarts32"]["pool"]["world"] == 2


def test_graph_capture_with_a_live_rccl_process_group():
    """What every rank of the multi-GPU path does and a 1-GPU box had never done IN ONE PROCESS: capture the attack iteration into a
    hipGraph while an RCCL ("nccl") process group is alive -- its watchdog thread polls events in the background, which a capture in
    the default `global` error mode can take for an illegal call and abort (the run would then fall back to eager launches, 2.4x
    slower, on every rank of the driver's scaling run).  One-rank communicator on cuda:0, one all-reduce to start the machinery, then
    a ConvNet attack with hip_graph="required" and a collective after it."""
    code = r"""
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
import torch.distributed as dist
import breaching_amd
from breaching_amd.cases import build_case, initial_candidate
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[1])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
case = build_case("convnet", "CIFAR10", 1, device=dev)
cfg = breaching_amd.get_attack_config("invertinggradients", ["optim.max_iterations=12", "optim.callback=6", "impl.hip_graph=required"])
attacker = breaching_amd.prepare_attack(case.model, case.loss_fn, cfg, dict(device=dev, dtype=torch.float))
os.environ["BREACH_HIP_TRIAL_DEVICES"] = "0"
rec, stats = attacker.reconstruct(case.server_payload, case.shared_data, {}, initial_data=initial_candidate(case.data_cfg, 1, seed=6))
dist.all_reduce(t); torch.cuda.synchronize()
print(json.dumps(dict(execution=sorted(set(stats["execution"]["trials"].values())), losses=len(stats["Trial_0_Val"]), t=float(t[0]))))
dist.destroy_process_group()
"""
    from breaching_amd.workers import free_port

    proc = subprocess.run([sys.executable, "-c", code, str(free_port())], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    rec = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["execution"] == ["hipGraph replay"] and rec["losses"] == 12 and rec["t"] == 1.0
