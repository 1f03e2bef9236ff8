"""ctypes binding of libbreach_hip.so (C ABI declared in include/breach_hip.h).

There is deliberately NO fallback: if the shared library is missing or a kernel launch fails, the attack raises.
"""

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

from . import build as _build

# ---- constants mirrored from include/breach_hip.h (checked against the library in tests) -------------------------
BH_ABI_VERSION = 7
BH_GM_CHUNK = 4096
BH_GM_MAX_PTRS = 448
BH_GM_PARTIAL_STRIDE = 4
BH_GM_MAX_ROWS = 2048
BH_GM_DEFAULT_ROWS = 512
BH_GM_STAT_WORDS = 12
BH_PRIOR_MAX_GRID = 1024
BH_BN_MAX_LAYERS = 448
BH_MT_MAX_PTRS = 112
BH_BN_TILE = 4096
BH_PRIOR_PARTIAL_STRIDE = 2
BH_STATE_WORDS = 16
BH_SCHED_STRIDE = 4
GM_CACHE_AUTO, GM_CACHE_KEEP, GM_CACHE_STREAM, GM_CACHE_STREAM_ALL = 0, 1, 2, 3

GM_KINDS = {
    "cosine-similarity": 0,
    "masked-cosine-similarity": 1,
    "fast-cosine-similarity": 2,
    "angular": 3,
    "euclidean": 4,
    "l1": 5,
    "tag-euclidean": 6,
    "pearlmutter-loss": 7,
}
GM_STAT_PATCH_D, GM_STAT_PATCH_R, GM_STAT_FD_STEP, GM_STAT_FD_SCALE = 8, 9, 10, 11
STATE_IT, STATE_DEAD, STATE_FIRST_BAD, STATE_IMPROVED, STATE_MIN, STATE_TOTAL, STATE_GNORM = range(7)
SIGN_NONE, SIGN_HARD, SIGN_SOFT = 0, 1, 2


class GmChunk(Structure):
    _fields_ = [("flat_off", c_int64), ("tensor_off", c_int64), ("tensor", c_int32), ("len", c_int32)]


class BnLayer(Structure):
    _fields_ = [("flat_off", c_int64), ("sums_off", c_int64), ("chan_off", c_int32), ("B", c_int32), ("C", c_int32),
                ("HW", c_int32), ("S", c_int32), ("narrow", c_int32), ("weight", c_float), ("div_unit_mul", c_uint32),
                ("div_unit_shr", c_uint32), ("div_c_mul", c_uint32), ("div_c_shr", c_uint32), ("fwd_items", c_int32)]


class BnItem(Structure):
    _fields_ = [("layer", c_int32), ("a", c_int32), ("b", c_int32), ("c", c_int32)]


class PsnrParams(Structure):
    _fields_ = [("mean", c_float * 4), ("std", c_float * 4), ("factor", c_float), ("clip", c_int32)]


class StepParams(Structure):
    _fields_ = [
        ("n", c_int64),
        ("plane", c_int64),
        ("channels", c_int32),
        ("boxed", c_int32),
        ("sign_mode", c_int32),
        ("max_iterations", c_int32),
        ("lo", c_float * 4),
        ("hi", c_float * 4),
        ("beta1", c_double),
        ("beta2", c_double),
        ("eps", c_double),
        ("decoupled_wd", c_int32),
        ("langevin", c_float),
        ("grad_clip", c_float),
    ]


BH_STEP_MAX_SLOTS = 4


class StepSlot(Structure):  # bh_step_slot: one optimised tensor of a list launch (include/breach_hip.h)
    _fields_ = [("params", StepParams), ("x", c_void_p), ("g", c_void_p), ("g_reg", c_void_p), ("noise", c_void_p), ("m", c_void_p),
                ("v", c_void_p), ("best", c_void_p)]


_PROTOTYPES = {
    # name: (restype, argtypes)
    "bh_abi_version": (c_int32, []),
    "bh_build_arch": (c_char_p, []),
    "bh_gm_num_groups": (c_int32, [c_int32]),
    "bh_gm_table_size": (c_int, [c_int32, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "bh_gm_build_table": (c_int, [c_int32, POINTER(c_int64), POINTER(GmChunk), c_int64, POINTER(c_int64)]),
    "bh_gm_group_bounds": (c_int, [c_int32, POINTER(GmChunk), c_int64, POINTER(c_int32)]),
    "bh_gm_fwd": (
        c_int,
        [c_int32, c_int32, POINTER(c_void_p), c_void_p, c_void_p, c_int64, POINTER(c_int32), c_void_p, c_float, c_void_p,
         c_int32, c_int32, c_void_p, c_void_p, c_void_p],
    ),
    "bh_gm_fwd_rows": (c_int32, [c_int32, POINTER(c_int32), c_int32]),
    "bh_gm_finalize": (
        c_int,
        [c_int32, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "bh_wall_clock_khz": (c_int32, []),
    "bh_gm_bwd": (
        c_int,
        [c_int32, c_int32, POINTER(c_void_p), c_void_p, c_void_p, c_int64, POINTER(c_int32), c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
         c_void_p, c_void_p, c_void_p],
    ),
    "bh_gm_pack": (c_int, [c_int32, POINTER(c_void_p), c_void_p, c_int64, POINTER(c_int32), c_void_p, c_void_p]),
    "bh_prior_tv_norm": (
        c_int,
        [c_void_p, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_float, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p],
    ),
    "bh_bn_plan_size": (
        c_int,
        [c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
         POINTER(c_int64), POINTER(c_int64)],
    ),
    "bh_bn_plan_build": (
        c_int,
        [c_int32, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_float), POINTER(BnLayer), POINTER(BnItem),
         c_int64, POINTER(BnItem), c_int64],
    ),
    "bh_bn_bwd_accumulate": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "bh_bn_eval_fwd": (c_int, [c_void_p] * 8 + [c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "bh_bn_eval_slabs": (c_int32, [c_int32, c_int32, c_int32]),
    "bh_bn_eval_bwd": (c_int, [c_void_p] * 14 + [c_int32, c_int32, c_int32, c_void_p]),
    "bh_bn_eval_bwd_bwd": (c_int, [c_void_p] * 14 + [c_int32, c_int32, c_int32, c_void_p]),
    "bh_ln_fwd": (c_int, [c_void_p] * 6 + [c_int32, c_int32, ctypes.c_float, c_void_p]),
    "bh_ln_bwd": (c_int, [c_void_p] * 8 + [c_int32, c_int32, c_void_p]),
    "bh_ln_bwd_bwd": (c_int, [c_void_p] * 12 + [c_int32, c_int32, c_void_p]),
    "bh_bn_sums": (c_int, [c_int32, POINTER(c_void_p), POINTER(c_int32), c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p]),
    "bh_bn_finalize": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "bh_bn_bwd": (
        c_int,
        [c_int32, POINTER(c_void_p), POINTER(c_int32), c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
    "bh_mt_num_groups": (c_int32, [c_int32]),
    "bh_mt_group_bounds": (c_int, [c_int32, POINTER(GmChunk), c_int64, POINTER(c_int32)]),
    "bh_mt_axpy": (
        c_int,
        [c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_float, c_void_p, c_int64, POINTER(c_int32), c_void_p,
         c_void_p],
    ),
    "bh_mt_scale": (c_int, [c_int32, POINTER(c_void_p), c_float, c_void_p, c_int64, POINTER(c_int32), c_void_p, c_void_p]),
    "bh_mt_patch": (
        c_int,
        [c_int32, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p, c_float, c_void_p, c_int64, POINTER(c_int32), c_void_p,
         c_void_p],
    ),
    "bh_prior_orthogonality": (c_int, [c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_void_p]),
    "bh_metric_psnr": (
        c_int,
        [c_void_p, c_void_p, c_int32, c_int64, c_int64, c_int32, POINTER(PsnrParams), c_void_p, c_void_p, c_void_p],
    ),
    "bh_event_create": (c_int, [POINTER(c_void_p)]),
    "bh_event_destroy": (c_int, [c_void_p]),
    "bh_event_record": (c_int, [c_void_p, c_void_p]),
    "bh_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "bh_state_reset": (c_int, [c_void_p, c_void_p]),
    "bh_loss_commit": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "bh_grad_norm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_void_p, c_void_p]),
    "bh_trial_key": (c_int64, [c_float, c_int32]),
    "bh_trial_key_unpack": (c_int, [c_int64, POINTER(c_float), POINTER(c_int32)]),
    "bh_step_list_norm_rows": (c_int32, [c_int32, POINTER(StepSlot)]),
    "bh_grad_norm_list": (c_int, [c_void_p, c_int32, POINTER(StepSlot), c_void_p, c_void_p, c_void_p]),
    "bh_candidate_step_list": (c_int, [c_void_p, c_void_p, c_int32, POINTER(StepSlot), c_void_p, c_void_p]),
    "bh_candidate_step": (
        c_int,
        [c_void_p, c_void_p, POINTER(StepParams), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p],
    ),
}

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)

_LIB = None


class BreachHipError(RuntimeError):
    """A C-ABI call returned a non-zero status."""


def library_path():
    return os.environ.get("BREACH_HIP_LIB", _build.LIB_PATH)


def load():
    """Load (once) and return the ctypes handle.  Raises if the library is absent or has the wrong ABI."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch first: it bundles its own libamdhip64 / libhsa-runtime64.  Loading libbreach_hip.so before torch would pull in
    # /opt/rocm's runtime instead and leave two HIP runtimes in one process (launches on torch's streams then fail with
    # hipErrorNoDevice).  With torch loaded, the library's NEEDED libamdhip64.so.* resolves to torch's copy.
    import torch  # noqa: F401

    path = library_path()
    if not os.path.exists(path):
        raise BreachHipError(
            f"{path} not found. Build it with `python -m breaching_amd.build` (needs hipcc); "
            "there is no CPU fallback for the HIP hot path."
        )
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in _PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.bh_abi_version() != BH_ABI_VERSION:
        raise BreachHipError(f"ABI mismatch: library {lib.bh_abi_version()} vs binding {BH_ABI_VERSION}")
    _LIB = lib
    return lib


def check(status, what):
    """Raise on a negative status; return the (possibly positive) status otherwise."""
    if status < 0:
        if status <= -1000:
            raise BreachHipError(f"{what}: HIP runtime error {-status - 1000}")
        raise BreachHipError(f"{what}: invalid argument (status {status})")
    return status


def ptr(tensor):
    """Device (or host) address of a tensor as c_void_p; None -> NULL."""
    if tensor is None:
        return c_void_p(0)
    return c_void_p(tensor.data_ptr())


def current_stream_handle(device=None):
    import torch

    return c_void_p(torch.cuda.current_stream(device).cuda_stream)
