"""TEST INFRASTRUCTURE ONLY -- numpy-facing wrappers around oracle/kernels_oracle.c (fp64 kernel-level oracle)."""

import ctypes
from ctypes import POINTER, c_double, c_float, c_int, c_int64

import numpy as np

from .build import build_oracle

KINDS = {"cosine-similarity": 0, "masked-cosine-similarity": 1, "fast-cosine-similarity": 2, "angular": 3,
         "euclidean": 4, "l1": 5, "tag-euclidean": 6}
_LIB = None


def load_oracle():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_oracle())
        lib.oracle_gm.restype = c_double
        lib.oracle_gm.argtypes = [c_int, c_int, POINTER(c_int64), POINTER(c_float), POINTER(c_float), POINTER(c_double),
                                  c_double, c_double, c_double, POINTER(c_double)]
        lib.oracle_tv_norm.restype = None
        lib.oracle_tv_norm.argtypes = [POINTER(c_float), c_int, c_int, c_int, c_double, c_double, c_double, c_double, c_int,
                                       c_double, c_double, POINTER(c_double), POINTER(c_double)]
        lib.oracle_bnstat.restype = c_double
        lib.oracle_bnstat.argtypes = [POINTER(c_float), c_int, c_int, c_int64, POINTER(c_float), POINTER(c_float),
                                      POINTER(c_double), POINTER(c_double), POINTER(c_double)]
        lib.oracle_candidate_step.restype = None
        lib.oracle_candidate_step.argtypes = [c_int64, c_int64, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double),
                                              POINTER(c_double), POINTER(c_double), c_double, c_double, c_double, c_double,
                                              c_double, c_int, c_int, c_int, c_int, c_int, c_double, c_double, c_int,
                                              POINTER(c_double), POINTER(c_double)]
        lib.oracle_orthogonality.restype = c_double
        lib.oracle_orthogonality.argtypes = [POINTER(c_float), c_int, c_int64, POINTER(c_double)]
        lib.oracle_psnr.restype = None
        lib.oracle_psnr.argtypes = [POINTER(c_float), POINTER(c_float), c_int, c_int64, c_int64, c_int, POINTER(c_double),
                                    POINTER(c_double), c_double, c_int, POINTER(c_double)]
        _LIB = lib
    return _LIB


def _fp(a):
    return a.ctypes.data_as(POINTER(c_float))


def _dp(a):
    return a.ctypes.data_as(POINTER(c_double)) if a is not None else None


def tag_weights(n, scheme):
    """objectives.py:115-125 in fp64."""
    if scheme == "linear":
        return np.arange(n, 0, -1, dtype=np.float64) / n
    if scheme == "exp":
        w = np.arange(n, 0, -1, dtype=np.float64)
        w = np.exp(w - w.max())
        w = w / w.sum()
        return w / w[0]
    return np.ones(n, dtype=np.float64)


def gm(kind, rec_list, data_list, scale=1.0, tag_scale=0.1, fudge=1e-7, weights=None, want_grad=True):
    """Value and per-tensor gradients (fp64) of a gradient-matching objective over lists of fp32 arrays."""
    lib = load_oracle()
    n = min(len(rec_list), len(data_list))
    rec_list, data_list = rec_list[:n], data_list[:n]
    sizes = [int(r.size) for r in rec_list]
    off = np.zeros(n + 1, dtype=np.int64)
    off[1:] = np.cumsum(sizes)
    rec = np.concatenate([np.ascontiguousarray(r, dtype=np.float32).reshape(-1) for r in rec_list]) if n else np.zeros(0, np.float32)
    data = np.concatenate([np.ascontiguousarray(d, dtype=np.float32).reshape(-1) for d in data_list])
    grad = np.zeros(rec.size, dtype=np.float64) if want_grad else None
    w = np.ascontiguousarray(weights, dtype=np.float64) if weights is not None else np.ones(n, dtype=np.float64)
    value = lib.oracle_gm(KINDS[kind], n, off.ctypes.data_as(POINTER(c_int64)), _fp(rec), _fp(data), _dp(w), float(scale),
                          float(tag_scale), float(fudge), _dp(grad))
    grads = [grad[off[i]:off[i + 1]].reshape(rec_list[i].shape) for i in range(n)] if want_grad else None
    return value, grads


def tv_norm(x, tv_scale=0.0, inner_exp=1.0, outer_exp=1.0, eps=1e-8, double_opponents=False, norm_scale=0.0, norm_p=2.0):
    lib = load_oracle()
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, C, H, W = x.shape
    assert C == 3
    out = np.zeros(2, dtype=np.float64)
    grad = np.zeros(x.shape, dtype=np.float64)
    lib.oracle_tv_norm(_fp(x), B, H, W, float(tv_scale), float(inner_exp), float(outer_exp), float(eps), int(double_opponents),
                       float(norm_scale), float(norm_p), _dp(out), _dp(grad))
    return out[0], out[1], grad


def bnstat(x, running_mean, running_var):
    lib = load_oracle()
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, C = x.shape[0], x.shape[1]
    HW = x.size // (B * C)
    rm = np.ascontiguousarray(running_mean, dtype=np.float32)
    rv = np.ascontiguousarray(running_var, dtype=np.float32)
    grad = np.zeros(x.shape, dtype=np.float64)
    mean, var = np.zeros(C, dtype=np.float64), np.zeros(C, dtype=np.float64)
    value = lib.oracle_bnstat(_fp(x), B, C, HW, _fp(rm), _fp(rv), _dp(grad), _dp(mean), _dp(var))
    return value, grad, mean, var


def candidate_step(x, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False, sign_mode=0,
                   iteration=0, max_iterations=1, noise=None, langevin=0.0, clip=-1.0, boxed=False, lo=None, hi=None,
                   plane=1, channels=1):
    """Returns updated fp64 copies (x, m, v)."""
    lib = load_oracle()
    x, m, v = (np.array(a, dtype=np.float64).reshape(-1).copy() for a in (x, m, v))
    g = np.ascontiguousarray(g, dtype=np.float64).reshape(-1)
    nz = np.ascontiguousarray(noise, dtype=np.float64).reshape(-1) if noise is not None else None
    lo_a = np.ascontiguousarray(lo if lo is not None else [0.0] * 4, dtype=np.float64)
    hi_a = np.ascontiguousarray(hi if hi is not None else [0.0] * 4, dtype=np.float64)
    lib.oracle_candidate_step(x.size, int(plane), int(channels), _dp(x), _dp(g), _dp(nz), _dp(m), _dp(v), float(lr), beta1, beta2,
                              eps, weight_decay, int(decoupled), int(step), int(sign_mode), int(iteration), int(max_iterations),
                              float(langevin), float(clip), int(boxed), _dp(lo_a), _dp(hi_a))
    return x, m, v


def orthogonality(x):
    """Value and gradient (fp64) of OrthogonalityRegularization on x[B, ...] (regularizers.py:170-178)."""
    lib = load_oracle()
    x = np.ascontiguousarray(x, dtype=np.float32)
    B = x.shape[0]
    D = x.size // B
    grad = np.zeros(x.size, dtype=np.float64)
    value = lib.oracle_orthogonality(_fp(x), B, D, _dp(grad))
    return value, grad.reshape(x.shape)


def psnr(rec, ref, mean=None, std=None, factor=1.0, clip=True):
    """[mean, max, per-example...] PSNR (analysis/metrics.py:117-130) of normalised batches rec / ref [B, C, ...]."""
    lib = load_oracle()
    rec = np.ascontiguousarray(rec, dtype=np.float32)
    ref = np.ascontiguousarray(ref, dtype=np.float32)
    B = rec.shape[0]
    per_example = rec.size // B
    channels = 1 if mean is None else len(mean)
    m = np.zeros(4, dtype=np.float64)
    s = np.ones(4, dtype=np.float64)
    if mean is not None:
        m[:channels], s[:channels] = np.asarray(mean, dtype=np.float32), np.asarray(std, dtype=np.float32)
    out = np.zeros(2 + B, dtype=np.float64)
    lib.oracle_psnr(_fp(rec), _fp(ref), B, per_example, per_example // channels, channels, _dp(m), _dp(s), float(factor), int(clip), _dp(out))
    return out
