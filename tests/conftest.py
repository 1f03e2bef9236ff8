import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run with `-m gpu` on the MI355X box)")
    config.addinivalue_line("markers", "trial_pool: the test chooses the trial worker devices itself (cfg.impl.trial_devices)")


@pytest.fixture(autouse=True)
def _restarts_stay_on_one_gpu(request, monkeypatch):
    """On a multi-GPU box `reconstruct` with several restarts starts one worker process per visible GPU by default.  The
    parity tests compare single-device trajectories: pin them (and the scripts they spawn) to the attacker's own GPU, whatever
    the box looks like.  Tests marked `trial_pool` pick their devices themselves."""
    # The product's default lets a failed hipGraph capture continue with eager launches (visible in stats["execution"]);
    # in this suite a silent 2.4x slower path must fail the test instead.
    monkeypatch.setenv("BREACH_HIP_GRAPH_STRICT", "1")  # "auto" -> "required"; cfg.impl.hip_graph=False still means eager
    if request.node.get_closest_marker("trial_pool") is None:
        monkeypatch.setenv("BREACH_HIP_TRIAL_DEVICES", "0")
    else:
        monkeypatch.delenv("BREACH_HIP_TRIAL_DEVICES", raising=False)


# Collection order: the cheap, deterministic, oracle-anchored tests run FIRST, the end-to-end
# trajectories of chaotic configurations LAST -- under `-x` a wobble in a long trajectory must not blank kernel parity.
_MODULE_ORDER = ["test_abi", "test_oracle_pinning", "test_host_logic", "test_workers", "test_gpu_kernels", "test_gpu_runtime",
                 "test_gpu_attack", "test_gpu_baseline_configs"]


def _module_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return _MODULE_ORDER.index(name) if name in _MODULE_ORDER else len(_MODULE_ORDER)


# Run-vs-run comparisons of OUR OWN attack (hipGraph replay vs eager launches, trials in flight vs sequential, worker pool
# vs one rank).  On the boxes measured in round 3 such runs are bit-identical (profiles/r3_mode_spread_convnet.json: 3 eager
# and 3 graph runs, default and deterministic MIOpen mode, every iterate equal), but the vendor kernels of the victim's double
# backward give no such guarantee: the driver's round-2 box produced an eager run whose 15-iteration opt_value sat 4.5e-5
# (relative) away from the graph run's.  A replay / scheduling defect (stale scalar, missed best copy, wrong iteration count)
# shows at the percent level, so these limits -- north_star's 1e-4 on the trajectory, 20x the observed wobble on the rescored
# optimum, 1e-3 on >= 99.9 % of the pixels -- still catch it.
RUN_VS_RUN = dict(history_rtol=1e-4, opt_value_rel=1e-3, pixel_tol=1e-3, pixel_fraction=0.999)


def runs_identical(run_a, run_b):
    """Bit for bit: every loss history, opt_value and every pixel of the reconstruction."""
    import torch

    (rec_a, stats_a), (rec_b, stats_b) = run_a, run_b
    keys = sorted(k for k in stats_a if k.startswith("Trial_"))
    return (keys == sorted(k for k in stats_b if k.startswith("Trial_")) and all(list(stats_a[k]) == list(stats_b[k]) for k in keys)
            and stats_a["opt_value"] == stats_b["opt_value"] and torch.equal(rec_a.detach().cpu(), rec_b.detach().cpu()))


def assert_same_attack(run_a, run_b, stats_keys=None, control=None):
    """`run_x = (reconstruction tensor, stats)` of two executions of the same attack from the same start.

    `control`: a SECOND execution of run_b's own configuration.  When the control reproduces run_b bit for bit -- the vendor
    kernels are deterministic on this box for this workload, which is what was measured on every box so far -- run_a has to be
    bit-identical as well: a replay / staleness defect of any size fails.  Only when the control itself wobbles do the
    RUN_VS_RUN limits below apply (and the test says so)."""
    import numpy as np

    pixel_fraction = RUN_VS_RUN["pixel_fraction"]
    if control is not None:
        if runs_identical(control, run_b):
            assert runs_identical(run_a, run_b), "the control run reproduced bit for bit, the run under test did not"
            return
        # The box does not reproduce the baseline itself (ulp-level nondeterminism of vendor kernels; pixels whose gradient is
        # near zero then follow rounding through Adam's normalisation while the losses stay put).  The control's own agreement
        # with the baseline is the yardstick for the pixels; losses and opt_value keep the RUN_VS_RUN limits.
        import numpy as np

        ctl, base = control[0].detach().cpu().numpy(), run_b[0].detach().cpu().numpy()
        ctl_close = float(np.isclose(ctl, base, rtol=RUN_VS_RUN["pixel_tol"], atol=RUN_VS_RUN["pixel_tol"]).mean())
        pixel_fraction = min(pixel_fraction, ctl_close - 0.03)
        print(f"  [run-vs-run] the control run of the same configuration was NOT bit-identical on this box (it agrees with the baseline on "
              f"{ctl_close:.4f} of the pixels): comparing within RUN_VS_RUN limits, pixels against that yardstick")
    (rec_a, stats_a), (rec_b, stats_b) = run_a, run_b
    keys = stats_keys if stats_keys is not None else sorted(k for k in stats_a if k.startswith("Trial_"))
    assert keys and sorted(k for k in stats_b if k.startswith("Trial_")) == sorted(k for k in stats_a if k.startswith("Trial_"))
    for key in keys:
        np.testing.assert_allclose(stats_a[key], stats_b[key], rtol=RUN_VS_RUN["history_rtol"], err_msg=key)
    assert stats_a["opt_value"] == pytest.approx(stats_b["opt_value"], rel=RUN_VS_RUN["opt_value_rel"])
    a, b = rec_a.detach().cpu().numpy(), rec_b.detach().cpu().numpy()
    close = np.isclose(a, b, rtol=RUN_VS_RUN["pixel_tol"], atol=RUN_VS_RUN["pixel_tol"]).mean()
    assert close >= pixel_fraction, (close, pixel_fraction)


def pytest_collection_modifyitems(config, items):
    import torch

    items.sort(key=_module_rank)  # stable: the order inside a module stays as written
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


@pytest.fixture(scope="session")
def kernels_oracle():
    """ctypes handle of the plain-C kernel oracle (built on demand with gcc)."""
    from oracle.kernels_ref import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def hip_lib():
    from breaching_amd import _lib, build

    build.build_library()
    return _lib.load()
