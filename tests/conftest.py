import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run with `-m gpu` on the MI355X box)")


def pytest_collection_modifyitems(config, items):
    import torch

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
