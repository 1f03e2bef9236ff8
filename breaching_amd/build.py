"""Build libbreach_hip.so for gfx950 with hipcc, in-tree (the built .so travels to the GPU box with the snapshot).

Usage: ``python -m breaching_amd.build [--force]``.  hipcc cross-compiles without a GPU.
No torch extension machinery is involved: torch's ROCm extension path runs hipify, which this project must not use.
"""

import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(REPO_ROOT, "include")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libbreach_hip.so")
STAMP_PATH = os.path.join(LIB_DIR, "libbreach_hip.stamp")

SOURCES = ["gm_kernels.hip", "prior_kernels.hip", "step_kernels.hip", "mt_kernels.hip", "affine_kernels.hip", "layernorm_kernels.hip"]
HEADERS = [os.path.join(CSRC, "bh_common.h"), os.path.join(INCLUDE, "breach_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; libbreach_hip.so cannot be built (set HIPCC=/path/to/hipcc)")


def source_digest():
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_current():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP_PATH)):
        return False
    with open(STAMP_PATH) as f:
        return f.read().strip() == source_digest()


def build_library(force=False, verbose=False):
    """Compile every HIP source into one shared library.  Returns the library path."""
    if not force and is_current():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), *FLAGS, f"-I{INCLUDE}", f"-I{CSRC}", *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed ({proc.returncode}):\n{proc.stdout}\n{proc.stderr}")
    if verbose and proc.stderr.strip():
        print(proc.stderr, file=sys.stderr)
    with open(STAMP_PATH, "w") as f:
        f.write(source_digest())
    return LIB_PATH


if __name__ == "__main__":
    path = build_library(force="--force" in sys.argv, verbose=True)
    print(path)
