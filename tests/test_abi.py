"""The C-ABI library loads and exports exactly what include/breach_hip.h declares (no compute calls: no GPU here)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "breach_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bh_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for required in ("bh_gm_fwd", "bh_gm_bwd", "bh_gm_finalize", "bh_gm_pack", "bh_prior_tv_norm", "bh_bnstat_sums",
                     "bh_bnstat_finalize", "bh_bnstat_bwd", "bh_loss_commit", "bh_candidate_step", "bh_grad_norm",
                     "bh_state_reset", "bh_abi_version"):
        assert required in names


def test_library_exports_every_declared_symbol(hip_lib):
    from breaching_amd import _lib

    raw = ctypes.CDLL(_lib.library_path())
    for name in _declared_functions():
        assert hasattr(raw, name), f"{name} declared in include/breach_hip.h but not exported"
    assert set(_declared_functions()) == set(_lib.EXPORTED_SYMBOLS)
    assert hip_lib.bh_abi_version() == _lib.BH_ABI_VERSION
    assert hip_lib.bh_build_arch() == b"gfx950"


def test_binding_constants_match_header():
    from breaching_amd import _lib

    text = open(HEADER).read()
    for name in ("BH_ABI_VERSION", "BH_GM_CHUNK", "BH_GM_MAX_PTRS", "BH_GM_PARTIAL_STRIDE", "BH_PRIOR_MAX_GRID",
                 "BH_PRIOR_PARTIAL_STRIDE", "BH_SCHED_STRIDE"):
        value = int(re.search(rf"#define {name} (\d+)", text).group(1))
        assert getattr(_lib, name) == value, name
    assert int(re.search(r"BH_GM_STAT_WORDS = (\d+)", text).group(1)) == _lib.BH_GM_STAT_WORDS
    assert int(re.search(r"BH_STATE_WORDS = (\d+)", text).group(1)) == _lib.BH_STATE_WORDS
    assert ctypes.sizeof(_lib.GmChunk) == 24


def test_chunk_table_host_helpers(hip_lib):
    """Pure host arithmetic of the ABI: ragged, empty and exact-multiple tensors."""
    from ctypes import byref, c_int32, c_int64

    from breaching_amd import _lib

    numel = [0, 1, 4095, 4096, 4097, 3 * 4096, 7]
    arr = (c_int64 * len(numel))(*numel)
    n_chunks, flat = c_int64(), c_int64()
    assert hip_lib.bh_gm_table_size(len(numel), arr, byref(n_chunks), byref(flat)) == 0
    assert n_chunks.value == 0 + 1 + 1 + 1 + 2 + 3 + 1
    assert flat.value == sum((n + 3) // 4 * 4 for n in numel)
    chunks = (_lib.GmChunk * n_chunks.value)()
    offs = (c_int64 * len(numel))()
    assert hip_lib.bh_gm_build_table(len(numel), arr, chunks, n_chunks.value, offs) == 0
    covered = {}
    for c in chunks:
        assert 1 <= c.len <= _lib.BH_GM_CHUNK and c.tensor_off % _lib.BH_GM_CHUNK == 0 and c.flat_off % 4 == 0
        assert c.flat_off == offs[c.tensor] + c.tensor_off
        covered[c.tensor] = covered.get(c.tensor, 0) + c.len
    assert covered == {i: n for i, n in enumerate(numel) if n}
    bounds = (c_int32 * (hip_lib.bh_gm_num_groups(len(numel)) + 1))()
    assert hip_lib.bh_gm_group_bounds(len(numel), chunks, n_chunks.value, bounds) == 0
    assert list(bounds) == [0, n_chunks.value]
    # more tensors than fit one kernel-argument block -> several launch groups
    many = [5] * (_lib.BH_GM_MAX_PTRS + 3)
    arr = (c_int64 * len(many))(*many)
    assert hip_lib.bh_gm_table_size(len(many), arr, byref(n_chunks), byref(flat)) == 0
    chunks = (_lib.GmChunk * n_chunks.value)()
    offs = (c_int64 * len(many))()
    assert hip_lib.bh_gm_build_table(len(many), arr, chunks, n_chunks.value, offs) == 0
    assert hip_lib.bh_gm_num_groups(len(many)) == 2
    bounds = (c_int32 * 3)()
    assert hip_lib.bh_gm_group_bounds(len(many), chunks, n_chunks.value, bounds) == 0
    assert list(bounds) == [0, _lib.BH_GM_MAX_PTRS, len(many)]
    # invalid arguments are reported, not crashed on
    assert hip_lib.bh_gm_table_size(-1, arr, byref(n_chunks), byref(flat)) == -1
    assert hip_lib.bh_gm_fwd(99, 1, None, None, None, 1, None, None, 0.0, None, None, None, None) == -1
    assert hip_lib.bh_candidate_step(None, None, None, None, None, None, None, None, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from breaching_amd import _lib

    monkeypatch.setenv("BREACH_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(_lib.BreachHipError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under breaching_amd/ may import it."""
    pkg = os.path.join(ROOT, "breaching_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_header_is_valid_c99(tmp_path):
    """The boundary is a C ABI: include/breach_hip.h must compile as plain C (no C++-isms, no HIP / torch types)."""
    import subprocess

    src = tmp_path / "use_header.c"
    src.write_text('#include "breach_hip.h"\nint main(void) { bh_step_params p; bh_gm_chunk c; (void)p; (void)c; return (int)sizeof(p) * 0; }\n')
    proc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only",
                           f"-I{os.path.join(ROOT, 'include')}", str(src)], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr


def test_step_params_layout_matches_ctypes(tmp_path):
    """sizeof / offsetof of bh_step_params as the C compiler sees them == the ctypes mirror in _lib.StepParams."""
    import subprocess

    from breaching_amd import _lib

    fields = [name for name, _ in _lib.StepParams._fields_]
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "breach_hip.h"\nint main(void) {\n printf("%zu", sizeof(bh_step_params));\n'
    prog += "".join(f' printf(" %zu", offsetof(bh_step_params, {f}));\n' for f in fields)
    prog += " return 0; }\n"
    src, exe = tmp_path / "layout.c", tmp_path / "layout"
    src.write_text(prog)
    subprocess.run(["gcc", "-std=c99", f"-I{os.path.join(ROOT, 'include')}", str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(_lib.StepParams)
    for name, off in zip(fields, out[1:]):
        assert int(off) == getattr(_lib.StepParams, name).offset, name


def test_no_compatibility_layers_in_the_product():
    """CDNA4 code written directly: no Triton, no hipify output, no CUDA/HIP dual paths, no tracing compiler."""
    banned = ("import triton", "torch.compile(", "hipify", "__HIP_PLATFORM_AMD__", "__CUDA_ARCH__", "cuda_runtime.h",
              "cpp_extension")
    pkg = os.path.join(ROOT, "breaching_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in banned:
                    if needle == "hipify" and f == "build.py":
                        continue  # mentioned in a comment explaining why torch's extension builder is not used
                    assert needle not in text, (f, needle)
