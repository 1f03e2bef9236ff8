"""TEST INFRASTRUCTURE ONLY -- compile oracle/kernels_oracle.c with gcc into oracle/_build/libkernels_oracle.so."""

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "kernels_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libkernels_oracle.so")


def build_oracle(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    cmd = ["gcc", "-O2", "-std=c99", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"gcc failed:\n{proc.stdout}\n{proc.stderr}")
    return LIB


if __name__ == "__main__":
    print(build_oracle(force=True))
