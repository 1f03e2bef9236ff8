"""Trial worker pool (breaching_amd/workers.py) on CPU: spawn, gloo rendezvous, job protocol, selection collective, errors.

The attacker itself needs a GPU; here the pool runs a stand-in runner that plays `reconstruct`'s multi-rank part: every rank
holds the solutions of its trials {t : t mod W == rank}, reports `trials_done`, waits for `go`, then joins
`TrialShard.select` -- exactly the sequence the attacker drives (attacker.py, reconstruct)."""

import pytest
import torch


def _runner_factory(rank, world, device_index, conn, num_trials):
    from breaching_amd import trials

    def run(job):
        if job.get("explode") == rank:
            raise ValueError("boom")
        shard = trials.TrialShard.current(num_trials)
        assert (shard.rank, shard.world) == (rank, world)
        solutions = {t: torch.full((2, 3), float(t)) for t in shard.local_trials()}
        scores = {t: job["scores"][t] for t in shard.local_trials()}
        stats = {f"Trial_{t}_Val": [float(t)] * 3 for t in shard.local_trials()}
        from breaching_amd.workers import rendezvous

        rendezvous(conn)  # trials_done -> go (or JobAborted)
        shard.select(solutions, scores, stats, torch.device("cpu"))

    return run


def _unpickle_bomb():
    raise ImportError("victim model class is not importable in the worker")


class _Bomb:
    """Pickles fine in the parent, fails to unpickle in the spawned child -- what an unimportable model class does."""

    def __reduce__(self):
        return (_unpickle_bomb, ())


def test_pool_runs_jobs_selects_and_survives_errors():
    from breaching_amd import trials
    from breaching_amd.workers import TrialWorkerPool

    from breaching_amd import workers

    num_trials = 5
    assert workers.active_pool() is None
    pool = TrialWorkerPool([None, None, None], _runner_factory, (num_trials,))
    try:
        assert pool.backend == "gloo" and pool.world == 3 and workers.active_pool() is pool
        # start-up has its own, longer limit than the collectives (cold `import torch` in every worker); both are reported
        assert pool.start_timeout == workers.DEFAULT_START_TIMEOUT > pool.collective_timeout == workers.DEFAULT_COLLECTIVE_TIMEOUT
        assert pool.describe()["pool_start_s"] > 0
        for scores in ([3.0, 0.25, 7.0, float("nan"), 0.5], [9.0, 8.0, 7.0, 6.0, 0.125]):
            pool.submit([dict(scores=scores)] * 2)
            shard = trials.TrialShard.current(num_trials)
            assert (shard.rank, shard.world) == (0, 3) and list(shard.local_trials()) == [0, 3]
            solutions = {t: torch.full((2, 3), float(t)) for t in shard.local_trials()}
            stats = {f"Trial_{t}_Val": [float(t)] * 3 for t in shard.local_trials()}
            assert pool.describe()["job_ship_s"] >= 0
            pool.expect("trials_done")
            pool.broadcast(("go",))
            value, solution = shard.select(solutions, {t: scores[t] for t in shard.local_trials()}, stats, torch.device("cpu"))
            pool.expect("ok")
            finite = [(s, t) for t, s in enumerate(scores) if s == s]
            want_value, want_trial = min(finite)
            assert value == want_value and float(solution[0, 0]) == float(want_trial)
            assert sorted(stats) == [f"Trial_{t}_Val" for t in range(num_trials)]  # every rank's histories merged
        # a failing worker reports its traceback; the parent raises instead of waiting in the collective ...
        pool.submit([dict(scores=[1.0] * 5, explode=2)] * 2)
        with pytest.raises(RuntimeError, match="boom"):
            pool.expect("trials_done")
        # ... and cancels the job: rank 1 (healthy, waiting for `go`) leaves it without entering the collective, rank 2's
        # error report is discarded, and the SAME pool runs the next job cleanly (what `reconstruct` does on any failure)
        assert pool.busy
        pool.abort()
        assert not pool.closed and not pool.busy
        scores = [4.0, 2.0, 0.5, 1.0, 3.0]
        pool.submit([dict(scores=scores)] * 2)
        shard = trials.TrialShard.current(num_trials)
        solutions = {t: torch.full((2, 3), float(t)) for t in shard.local_trials()}
        pool.expect("trials_done")
        pool.broadcast(("go",))
        value, solution = shard.select(solutions, {t: scores[t] for t in shard.local_trials()}, {}, torch.device("cpu"))
        pool.finish()
        assert value == 0.5 and float(solution[0, 0]) == 2.0 and not pool.busy
        # rank 0's own failure before any worker finished: abort drains a pool whose workers are all waiting for `go`
        pool.submit([dict(scores=scores)] * 2)
        pool.abort()
        assert not pool.closed
    finally:
        pool.close(force=True)
    import torch.distributed as dist

    assert workers.active_pool() is None and not dist.is_initialized()  # the group never outlives its pool


def test_pool_start_fails_fast_when_a_child_cannot_boot_and_pool_dies_with_its_owner(monkeypatch):
    """A child whose arguments do not unpickle (an unimportable victim model class) is noticed through `is_alive` within
    seconds, not after the rendezvous timeout; the module-level handle on the active pool is weak."""
    import gc
    import time

    import torch.distributed as dist

    from breaching_amd import workers
    from breaching_amd.workers import TrialWorkerPool

    t0 = time.time()
    with pytest.raises(RuntimeError, match="died"):
        TrialWorkerPool([None, None], _runner_factory, (_Bomb(),), collective_timeout=120.0)
    assert time.time() - t0 < 60.0
    assert workers.active_pool() is None and not dist.is_initialized()

    # a process started by torch.distributed.run still carries the elastic agent's store flag: the pool hosts its OWN store
    # (round 6: the pool leg of `bench.py --gpus N`, started from a rank that had left its group, timed out in the rendezvous)
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    pool = TrialWorkerPool([None, None], _runner_factory, (2,))
    assert workers.active_pool() is pool
    procs = [proc for proc, _ in pool.workers]
    del pool
    gc.collect()
    assert workers.active_pool() is None and not dist.is_initialized()
    assert all(not proc.is_alive() for proc in procs)


def test_requested_devices_parsing(monkeypatch):
    from breaching_amd import AttrDict
    from breaching_amd.workers import requested_devices

    cfg = AttrDict(impl=dict(trial_devices=[0, 0]))
    dev = torch.device("cuda", 0)
    monkeypatch.delenv("BREACH_HIP_TRIAL_DEVICES", raising=False)
    assert requested_devices(cfg, dev) == [0, 0]
    monkeypatch.setenv("BREACH_HIP_TRIAL_DEVICES", "0, 0,0")
    assert requested_devices(cfg, dev) == [0, 0, 0]
    monkeypatch.setenv("BREACH_HIP_TRIAL_DEVICES", "1,0")
    with pytest.raises(ValueError, match="must start with"):
        requested_devices(cfg, dev)


def test_each_rank_gets_its_own_miopen_user_db(monkeypatch, tmp_path):
    """Eight first processes on a fresh box must not share one MIOpen find-db file (lock contention, solver timings taken
    against each other): every rank points MIOPEN_USER_DB_PATH at its own directory unless the user already chose one or switched
    the isolation off; the directory starts as a copy of the user's existing find-db (an already tuned database is kept, ranks
    that start from the same database pick the same solvers), and bench.py's staged start copies rank 0's into the others."""
    import os

    from breaching_amd import workers

    monkeypatch.setenv("HOME", str(tmp_path))
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("BREACH_HIP_MIOPEN_ISOLATE", raising=False)
    tuned = tmp_path / ".config" / "miopen"
    tuned.mkdir(parents=True)
    (tuned / "gfx950_256.HIP.3_5_1.ufdb.txt").write_text("tuned on this box\n")
    first = workers.isolate_miopen_user_db(3)
    assert first.endswith("breach_hip_rank3") and first.startswith(str(tmp_path)) and os.path.isdir(first)
    assert os.listdir(first) == ["gfx950_256.HIP.3_5_1.ufdb.txt"]  # seeded from the existing database
    assert workers.isolate_miopen_user_db(5) == first  # already set in this process (by the call above): left alone
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    assert workers.isolate_miopen_user_db(1) == "/somewhere/else"
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("BREACH_HIP_MIOPEN_ISOLATE", "0")
    assert workers.isolate_miopen_user_db(2) is None and "MIOPEN_USER_DB_PATH" not in os.environ  # opt-out: the shared default database
    monkeypatch.delenv("BREACH_HIP_MIOPEN_ISOLATE")
    # the staged start: rank 0's results copied over whatever the other rank's directory holds
    zero = workers.isolate_miopen_user_db(0)
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    with open(os.path.join(zero, "found_by_rank0.udb"), "w") as f:
        f.write("x")
    other = workers.isolate_miopen_user_db(6)
    assert workers.seed_miopen_user_db(zero, other) == 2 and sorted(os.listdir(other)) == ["found_by_rank0.udb", "gfx950_256.HIP.3_5_1.ufdb.txt"]


def test_eight_ranks_share_32_restarts_like_an_eight_gpu_node():
    """BASELINE configs[3] in its multi-rank shape, on CPU over gloo: the caller + seven worker processes = eight ranks, 32 restarts,
    rank r holds trials {t : t mod 8 = r} (four each).  One all-reduce(MIN) on the packed keys + one broadcast make every rank agree
    on the winner; the 32 loss histories reach rank 0; NaN scores lose; a tie goes to the lower trial index (torch.min's choice in
    the reference's sequential loop, optimization_based_attack.py:206-218).  The 8-rank rendezvous, the job protocol and the
    selection have otherwise only run with 2-4 ranks."""
    from breaching_amd import trials, workers
    from breaching_amd.workers import TrialWorkerPool

    num_trials = 32
    pool = TrialWorkerPool([None] * 8, _runner_factory, (num_trials,))
    try:
        assert pool.world == 8 and pool.backend == "gloo"
        rng = torch.Generator().manual_seed(0)
        for round_ in range(2):
            scores = (torch.rand(num_trials, generator=rng) + 0.5).tolist()
            scores[5] = float("nan")
            scores[29] = 0.125 if round_ == 0 else scores[29]  # a clear winner on the last rank's share ...
            if round_ == 1:
                scores[11] = scores[19] = 0.0625                # ... and a tie between trials of rank 3
            pool.submit([dict(scores=scores)] * 7)
            shard = trials.TrialShard.current(num_trials)
            assert (shard.rank, shard.world) == (0, 8) and list(shard.local_trials()) == [0, 8, 16, 24]
            solutions = {t: torch.full((2, 3), float(t)) for t in shard.local_trials()}
            stats = {f"Trial_{t}_Val": [float(t)] * 3 for t in shard.local_trials()}
            pool.expect("trials_done")
            pool.broadcast(("go",))
            value, solution = shard.select(solutions, {t: scores[t] for t in shard.local_trials()}, stats, torch.device("cpu"))
            pool.finish()
            want_trial = 29 if round_ == 0 else 11
            assert value == pytest.approx(scores[want_trial]) and float(solution[0, 0]) == float(want_trial)
            assert sorted(stats) == sorted(f"Trial_{t}_Val" for t in range(num_trials))
            assert all(stats[f"Trial_{t}_Val"] == [float(t)] * 3 for t in range(num_trials))
        described = pool.describe()
        assert described["world"] == 8 and described["pool_start_s"] > 0 and described["job_ship_s"] >= 0
    finally:
        pool.close(force=True)
    assert workers.active_pool() is None


def _shipping_runner_factory(rank, world, device_index, conn):
    """Stand-in for `attacker_runner_factory`: checks what `TrialWorkerPool.ship` delivered and reports a checksum back through the
    selection collective (score = what this rank saw)."""
    from breaching_amd import trials
    from breaching_amd.workers import rendezvous

    def run(job):
        inputs, starts = job["inputs"], job["starts"]
        grads = inputs["shared_data"][0]["gradients"]
        assert inputs["shared_data"][0]["metadata"]["name"] == "meta" and inputs["server_secrets"] == {} and inputs["initial_data"] is None
        assert [tuple(g.shape) for g in grads] == [(3, 5), (7,), (0,), (2, 2, 2)] and grads[3].dtype == torch.float64
        assert inputs["server_payload"][0]["buffers"][0] is inputs["shared_data"][0]["buffers"][0]  # one object shipped once
        assert inputs["server_payload"][0]["buffers"][0].dtype == torch.int64
        assert all(g.data_ptr() % 64 == 0 for g in grads if g.numel() > 0)  # 512-byte grain inside the flat buffer (the host allocator aligns to 64)
        mine = {t: s for t, s in starts["inits"].items() if t % world == rank}
        assert sorted(mine) == list(range(rank, job["num_trials"], world)) and 0 not in starts["inits"]
        checksum = float(sum(g.double().sum() for g in grads) + inputs["server_payload"][0]["buffers"][0].sum()
                         + sum(float(s[0].sum()) for s in mine.values()) + float(starts["labels"].sum()))
        shard = trials.TrialShard.current(job["num_trials"])
        rendezvous(conn)
        shard.select({t: s[0] for t, s in mine.items()}, {t: checksum + t for t in mine}, {}, torch.device("cpu"))

    return run


def test_job_inputs_travel_over_the_process_group_not_the_pipes():
    """Round 6: `reconstruct` hands its inputs to the workers through `pool.ship` -- the pipes carry shapes, dtypes and metadata, the
    tensor contents one broadcast per dtype over the pool's own group (gloo here, RCCL when every rank has a GPU).  Mixed dtypes,
    empty and shared tensors, two shipments per job (inputs before prepare_attack, starting points after), pipe bytes independent
    of the tensor sizes, and a failure between the two shipments aborts cleanly."""
    from breaching_amd import trials, workers
    from breaching_amd.workers import TrialWorkerPool

    num_trials = 6
    pool = TrialWorkerPool([None, None, None], _shipping_runner_factory, ())
    try:
        pipe_bytes = []
        for scale in (1, 64):
            gen = torch.Generator().manual_seed(scale)
            grads = [torch.randn(3, 5, generator=gen), torch.randn(7, generator=gen), torch.zeros(0), torch.randn(2, 2, 2, generator=gen, dtype=torch.float64)]
            big = torch.randn(1000 * scale, generator=gen)  # rides along in the metadata-free part: only its size changes
            buffers = [torch.arange(4)]
            payload = [dict(parameters=[big], buffers=buffers, metadata="cfg")]
            shared = [dict(gradients=grads, buffers=buffers, metadata=dict(name="meta", labels=None))]
            inits = {t: (torch.full((1, 3, 4, 4), float(t)).requires_grad_(True),) for t in range(num_trials)}
            labels = torch.tensor([3, 1])
            pool.begin_job()
            pool.ship("inputs", dict(server_payload=payload, shared_data=shared, server_secrets={}, initial_data=None))
            pool.ship("starts", dict(labels=labels, inits={t: inits[t] for t in range(num_trials) if t % 3 != 0}))
            pool.submit([dict(num_trials=num_trials)] * 2)
            shard = trials.TrialShard.current(num_trials)
            pool.expect("trials_done")
            pool.broadcast(("go",))
            base = float(sum(g.double().sum() for g in grads) + 6 + 4)
            own = base + sum(float(inits[t][0].sum()) for t in (0, 3))
            value, solution = shard.select({t: inits[t][0].detach() for t in (0, 3)}, {0: own, 3: own + 3}, {}, torch.device("cpu"))
            pool.finish()
            per_rank = {r: base + sum(float(inits[t][0].sum()) for t in range(r, num_trials, 3)) + r for r in range(3)}  # trial r is rank r's best
            winner = min(per_rank, key=per_rank.get)
            assert value == pytest.approx(per_rank[winner], rel=1e-6) and float(solution.flatten()[0]) == float(winner)
            timing = pool.describe()
            assert timing["job_ship_s"] > 0 and timing["job_ship_bytes"] >= big.numel() * 4
            pipe_bytes.append(timing["job_pipe_bytes"])
        assert pipe_bytes[1] - pipe_bytes[0] < 64  # 64x the tensor bytes, the same skeleton on the pipe (digits of the shapes aside)
        # a failure between the shipments (rank 0's prepare_attack raising): the workers hold a shipment and wait for more
        pool.begin_job()
        pool.ship("inputs", dict(server_payload=payload, shared_data=shared, server_secrets={}, initial_data=None))
        assert pool.busy
        pool.abort()
        assert not pool.closed and not pool.busy
        pool.begin_job()
        pool.ship("inputs", dict(server_payload=payload, shared_data=shared, server_secrets={}, initial_data=None))
        pool.ship("starts", dict(labels=labels, inits={t: inits[t] for t in range(num_trials) if t % 3 != 0}))
        pool.submit([dict(num_trials=num_trials)] * 2)
        pool.expect("trials_done")
        pool.broadcast(("go",))
        shard.select({t: inits[t][0].detach() for t in (0, 3)}, {0: 1e9, 3: 1e9}, {}, torch.device("cpu"))
        pool.finish()
    finally:
        pool.close(force=True)
    assert workers.active_pool() is None


def test_split_and_join_tensors_round_trip():
    from breaching_amd.workers import join_tensors, shipment_layout, split_tensors

    a, b = torch.arange(6.0).view(2, 3), torch.arange(3)
    tree = dict(x=[a, (b, a)], y=None, z={"k": 4})
    skeleton, tensors = split_tensors(tree)
    assert len(tensors) == 2 and tensors[0] is a and tensors[1] is b
    back = join_tensors(skeleton, tensors)
    assert back["x"][0] is a and back["x"][1][1] is a and back["x"][1][0] is b and back["y"] is None and back["z"] == {"k": 4}
    layout = shipment_layout([((2, 3), "float32"), ((3,), "int64"), ((5,), "float32")])
    assert layout["float32"] == (256, [(0, 0, 6), (2, 128, 5)]) and layout["int64"] == (64, [(1, 0, 3)])
