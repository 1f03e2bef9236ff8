// Multi-tensor elementwise kernels over per-parameter lists, and two small batch kernels.
//
//  * FedAvg unroll (breaching/attacks/auxiliaries/objectives.py:48-72): the reference rebuilds the whole parameter list
//    with a Python list comprehension per local step (`param - lr * grad`, T small launches) and once more for the final
//    difference (`p_local - p_server`).  Here: one launch per step writes all T updated tensors into one packed buffer
//    (views of it are handed to the functional model), the last step also subtracts the server state, and the backward
//    of a step (d/d grad = -lr) is one launch as well.
//  * Pearlmutter finite-difference objectives (objectives.py:279-493): the offset parameters
//    `theta + eps_n * v(grad, data)` are written out of place by one launch that evaluates the first-order direction v on
//    the fly from the coefficient pair the kernel-A finalize left on the device (no host round trip, nothing to restore).
//  * OrthogonalityRegularization (regularizers.py:156-181): value and analytic gradient in one pass.
//  * PSNR of a reconstruction batch (analysis/metrics.py:108-130) on the device.
//
// All HBM/latency bound: 16-byte loads and stores, chunk table shared with kernel A (chunks never straddle a tensor).
// No float contraction where the reference rounds twice (`a + alpha * b` is mul then add, like torch's two ops).

#include "bh_common.h"

namespace {

using bh::kBlock;

enum MtOp { kAxpy = 0, kAxpyMinus = 1, kScale = 2, kPatch = 3 };

// Pointer lists travel in the kernel-argument segment (as kernel A's do): three lists of BH_MT_MAX_PTRS (112) pointers for the
// one three-list form, (a + alpha b) - c; every other form reads at most two lists, and then TWO adjacent base groups ride in one
// launch (2 x 224 pointers = 3584 bytes, the size of kernel A's block).  The gradient lists of ResNet-50 (161 tensors) and
// BERT-base (201) are one launch per call that way instead of two (round 4: 128 per list and launch for every form) -- the launch
// boundary in the middle of a 50 us call was most of what separated this kernel from kernel A's backward on the same bytes
// (profiles/r5_mt_kernel_probe.jsonl).
struct MtPtrs3 {
  static constexpr int kGroups = 1;
  const float* a[BH_MT_MAX_PTRS];
  const float* b[BH_MT_MAX_PTRS];
  const float* c[BH_MT_MAX_PTRS];
};
struct MtPtrs2 {
  static constexpr int kGroups = 2;
  const float* a[2 * BH_MT_MAX_PTRS];
  const float* b[2 * BH_MT_MAX_PTRS];
};
template <int OP>
struct MtPtrsFor {
  using type = MtPtrs2;
};
template <>
struct MtPtrsFor<kAxpyMinus> {
  using type = MtPtrs3;
};

// op(a, b, c) per element.  k0 / k1: alpha (axpy, scale) or the two direction coefficients (patch).
template <int OP>
__device__ __forceinline__ float mt_elem(float a, float b, float c, float k0, float k1) {
#pragma clang fp contract(off)  // torch rounds `lr * grad` and `param - (.)` separately: no fused multiply-add here
  if constexpr (OP == kAxpy) {
    const float t = k0 * b;
    return a + t;  // a + alpha*b
  }
  if constexpr (OP == kAxpyMinus) {
    const float t = k0 * b;
    const float u = a + t;
    return u - c;  // (a + alpha*b) - c
  }
  if constexpr (OP == kScale) return k0 * a;  // alpha*a
  return a + (k0 * c + k1 * b);               // patch: theta + (k0 * data + k1 * grad); b = grad, c = packed data
}

template <int OP>
__device__ __forceinline__ float4 mt_elem4(const float4& a, const float4& b, const float4& c, float k0, float k1) {
  return make_float4(mt_elem<OP>(a.x, b.x, c.x, k0, k1), mt_elem<OP>(a.y, b.y, c.y, k0, k1), mt_elem<OP>(a.z, b.z, c.z, k0, k1),
                     mt_elem<OP>(a.w, b.w, c.w, k0, k1));
}

// One chunk, one workgroup.  On a full chunk every thread issues all of its 16-byte loads (BH_GM_CHUNK / 4 / kBlock per list = 4)
// before the first use, then computes and stores: 12 loads in flight per thread for the three-operand forms.  HAS_A is resolved
// per workgroup by the caller -- with the null test inside the loop (`a4 ? a4[i] : zero`, rounds 2-3) the compiler
// scalarised the float4 into four branch-guarded dword loads per operand (profiles/r4_kernel_isa_census.txt).
constexpr int kMtIters = BH_GM_CHUNK / 4 / kBlock;
static_assert(kMtIters * 4 * kBlock == BH_GM_CHUNK, "a chunk is a whole number of 16-byte accesses per thread");

// NT: the operands are read with non-temporal loads (`global_load_dwordx4 ... nt`), as kernel A does for lists that cannot stay
// in the 256 MiB Infinity Cache (gm_kernels.hip, BH_GM_CACHE_*): every operand is read once per launch, and the lines of the
// previous launch's output are still draining.  The output keeps plain stores -- the model's next forward pass reads it.
typedef float bh_mt_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 mt_load(const float4* p) {
  if constexpr (NT) {
    const bh_mt_v4f v = __builtin_nontemporal_load(reinterpret_cast<const bh_mt_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}

// NTS: the output list itself is larger than the Infinity Cache (BERT-base: 344 MB), so none of it can wait there for the model's
// next forward pass; non-temporal stores keep it from evicting the operands still to be read (kernel A's backward, same policy:
// 170.6 -> 163.2 us on that list, gm_kernels.hip BH_GM_CACHE_STREAM_ALL).
template <bool NTS>
__device__ __forceinline__ void mt_store(float4* p, const float4& v) {
  if constexpr (NTS) {
    bh_mt_v4f w;
    w.x = v.x, w.y = v.y, w.z = v.z, w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<bh_mt_v4f*>(p));
  } else {
    *p = v;
  }
}

template <int OP, bool HAS_A, bool NT, bool NTS>
__device__ __forceinline__ void mt_chunk(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                         float* __restrict__ o, int len, float k0, float k1) {
  constexpr bool needs_b = OP != kScale, needs_c = OP == kAxpyMinus || OP == kPatch;
  const int tid = threadIdx.x;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* __restrict__ a4 = reinterpret_cast<const float4*>(a);
  const float4* __restrict__ b4 = reinterpret_cast<const float4*>(b);
  const float4* __restrict__ c4 = reinterpret_cast<const float4*>(c);
  float4* __restrict__ o4 = reinterpret_cast<float4*>(o);
  if (len == BH_GM_CHUNK) {
    // Full chunk (all but the last chunk of a tensor), uniform over the workgroup: no per-access guard, every 16-byte load of
    // every operand issued before the first use -- the shape of kernel A's backward (gm_kernels.hip, gm_bwd_kernel).  With the
    // `i < n4` guard around each load (round 4) the same bytes took 1.5x as long as that kernel.
    float4 av[kMtIters], bv[kMtIters], cv[kMtIters];
#pragma unroll
    for (int j = 0; j < kMtIters; ++j) av[j] = HAS_A ? mt_load<NT>(a4 + tid + j * kBlock) : zero;
#pragma unroll
    for (int j = 0; j < kMtIters; ++j) bv[j] = needs_b ? mt_load<NT>(b4 + tid + j * kBlock) : zero;
#pragma unroll
    for (int j = 0; j < kMtIters; ++j) cv[j] = needs_c ? mt_load<NT>(c4 + tid + j * kBlock) : zero;
#pragma unroll
    for (int j = 0; j < kMtIters; ++j) mt_store<NTS>(o4 + tid + j * kBlock, mt_elem4<OP>(av[j], bv[j], cv[j], k0, k1));
    return;
  }
  const int n4 = len >> 2;
  for (int i = tid; i < n4; i += kBlock) {
    const float4 av = HAS_A ? mt_load<NT>(a4 + i) : zero;
    const float4 bv = needs_b ? mt_load<NT>(b4 + i) : zero;
    const float4 cv = needs_c ? mt_load<NT>(c4 + i) : zero;
    mt_store<NTS>(o4 + i, mt_elem4<OP>(av, bv, cv, k0, k1));
  }
  const int tail = len & 3;
  if (tid < tail) {
    const int i = (n4 << 2) + tid;
    const float a_i = HAS_A ? a[i] : 0.f;
    const float b_i = needs_b ? b[i] : 0.f;
    const float c_i = needs_c ? c[i] : 0.f;
    o[i] = mt_elem<OP>(a_i, b_i, c_i, k0, k1);
  }
}

// One workgroup per chunk.  `c_flat`: the third operand comes from a packed buffer (patch) instead of a pointer list.
// `coef`: device pair overriding (k0, k1) when non-NULL (patch).  A NULL `a` pointer reads as zeros (scale of a missing
// upstream gradient).
template <int OP, bool NT, bool NTS = false>
__global__ __launch_bounds__(kBlock) void mt_kernel(typename MtPtrsFor<OP>::type ptrs, int tensor_base, const float* __restrict__ c_flat,
                                                    const bh_gm_chunk* __restrict__ chunks, int chunk_base, float k0,
                                                    float k1, const float* __restrict__ coef, float* __restrict__ out_flat) {
  const bh_gm_chunk ch = chunks[chunk_base + blockIdx.x];
  const int t = ch.tensor - tensor_base;
  const float* __restrict__ a = ptrs.a[t];
  const float* __restrict__ b = ptrs.b[t];
  const float* __restrict__ c;
  if constexpr (OP == kAxpyMinus) c = ptrs.c[t];
  else c = c_flat ? c_flat + ch.flat_off : nullptr;
  float* __restrict__ o = out_flat + ch.flat_off;
  if (coef) {  // patch: the host value k0 is the multiplier of the device coefficient pair
    k1 = k0 * coef[1];
    k0 = k0 * coef[0];
  }
  constexpr bool needs_b = OP != kScale, needs_c = OP == kAxpyMinus || OP == kPatch;
  const float* __restrict__ bp = needs_b ? b + ch.tensor_off : nullptr;
  const float* __restrict__ cp = needs_c ? (OP == kAxpyMinus ? c + ch.tensor_off : c) : nullptr;
  if (a)  // uniform over the workgroup
    mt_chunk<OP, true, NT, NTS>(a + ch.tensor_off, bp, cp, o, ch.len, k0, k1);
  else
    mt_chunk<OP, false, NT, NTS>(nullptr, bp, cp, o, ch.len, k0, k1);
}

bool ok_ptr(const void* p, bool allow_null) {
  if (p == nullptr) return allow_null;
  return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// Pointers of launch `g` (P::kGroups base groups of BH_MT_MAX_PTRS tensors each); false on a misaligned / unexpected null pointer.
template <class P>
bool fill(P& out, const void* const* a, const void* const* b, const void* const* c, int n_tensors, int g, bool a_nullable) {
  constexpr int kPer = P::kGroups * BH_MT_MAX_PTRS;
  const int base = g * kPer;
  const int cnt = (n_tensors - base) < kPer ? (n_tensors - base) : kPer;
  for (int i = 0; i < kPer; ++i) {
    out.a[i] = out.b[i] = nullptr;
    if constexpr (P::kGroups == 1) out.c[i] = nullptr;
  }
  for (int i = 0; i < cnt; ++i) {
    if (!ok_ptr(a[base + i], a_nullable)) return false;
    out.a[i] = static_cast<const float*>(a[base + i]);
    if (b) {
      if (!ok_ptr(b[base + i], false)) return false;
      out.b[i] = static_cast<const float*>(b[base + i]);
    }
    if constexpr (P::kGroups == 1) {
      if (c) {
        if (!ok_ptr(c[base + i], false)) return false;
        out.c[i] = static_cast<const float*>(c[base + i]);
      }
    }
  }
  return true;
}

template <int OP>
int run_mt(int32_t n_tensors, const void* const* a, const void* const* b, const void* const* c, const float* c_flat,
           float k0, float k1, const float* coef, const bh_gm_chunk* chunks_dev, int64_t n_chunks,
           const int32_t* group_chunk_begin, float* out_flat, void* stream, bool a_nullable) {
  if (n_tensors <= 0 || a == nullptr || chunks_dev == nullptr || n_chunks <= 0 || group_chunk_begin == nullptr ||
      out_flat == nullptr || (reinterpret_cast<uintptr_t>(out_flat) & 15u) != 0)
    return BH_EINVAL;
  if (c_flat != nullptr && (reinterpret_cast<uintptr_t>(c_flat) & 15u) != 0) return BH_EINVAL;
  using P = typename MtPtrsFor<OP>::type;
  const int base_groups = bh_mt_num_groups(n_tensors);  // `group_chunk_begin` is laid out for these (BH_MT_MAX_PTRS tensors each)
  const int launches = (base_groups + P::kGroups - 1) / P::kGroups;
  for (int g = 0; g < launches; ++g) {  // validate everything before anything is enqueued
    P probe;
    if (!fill(probe, a, b, c, n_tensors, g, a_nullable)) return BH_EINVAL;
  }
  hipStream_t st = bh::as_stream(stream);
  // Footprint of one call = operands read + the list written.  While it fits the 256 MiB Infinity Cache (BH_GM_CACHE_AUTO_BYTES)
  // plain loads win -- the next call, or the model's forward pass, finds its operands there (ResNet-18, (a + alpha b) - c:
  // 30.9 us plain vs 33.6 us non-temporal); beyond it the operands are read once and only displace the output: non-temporal
  // loads (ResNet-50: 71.6 -> 68.4 us and 98.8 -> 89.3 us; BERT-base 197.1 -> 190.8 and 262.0 -> 254.2 us;
  // profiles/r4_mt_kernel_probe_nt.jsonl).
  constexpr int kOperands = OP == kScale ? 1 : (OP == kAxpy ? 2 : 3);
  const bool stream_loads = n_chunks * (int64_t)BH_GM_CHUNK * 4 * (kOperands + 1) > (int64_t)BH_GM_CACHE_AUTO_BYTES;
  const bool stream_stores = n_chunks * (int64_t)BH_GM_CHUNK * 4 > (int64_t)BH_GM_CACHE_AUTO_BYTES;  // the output alone does not fit
  for (int g = 0; g < launches; ++g) {
    const int first = g * P::kGroups, last = (first + P::kGroups) < base_groups ? (first + P::kGroups) : base_groups;
    const int begin = group_chunk_begin[first], n = group_chunk_begin[last] - begin;
    if (n <= 0) continue;
    P ptrs;
    fill(ptrs, a, b, c, n_tensors, g, a_nullable);
    const int tensor_base = g * P::kGroups * BH_MT_MAX_PTRS;
    if (stream_stores)
      hipLaunchKernelGGL((mt_kernel<OP, true, true>), dim3(n), dim3(kBlock), 0, st, ptrs, tensor_base, c_flat, chunks_dev, begin, k0, k1,
                         coef, out_flat);
    else if (stream_loads)
      hipLaunchKernelGGL((mt_kernel<OP, true>), dim3(n), dim3(kBlock), 0, st, ptrs, tensor_base, c_flat, chunks_dev, begin, k0, k1, coef,
                         out_flat);
    else
      hipLaunchKernelGGL((mt_kernel<OP, false>), dim3(n), dim3(kBlock), 0, st, ptrs, tensor_base, c_flat, chunks_dev, begin, k0, k1, coef,
                         out_flat);
    const int rc = bh::launch_status();
    if (rc != 0) return rc;
  }
  return 0;
}

// ---- OrthogonalityRegularization -------------------------------------------------------------------------------------
// value = (1/D) sum_k ( (sum_i x_ik^2)^2 - sum_i x_ik^4 )   (all ordered pairs i != j of mean_k (x_ik x_jk)^2)
// grad_ik = (4/D) x_ik ( sum_j x_jk^2 - x_ik^2 )
__global__ __launch_bounds__(kBlock) void orthogonality_kernel(const float* __restrict__ x, int B, int64_t D,
                                                               float* __restrict__ grad, double* __restrict__ partials) {
  __shared__ double lds[bh::kWavesPerBlock];
  const float inv_d = (float)(1.0 / (double)D);
  double acc = 0.0;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < D; k += (int64_t)gridDim.x * kBlock) {
    float s2 = 0.f, s4 = 0.f;
    for (int i = 0; i < B; ++i) {
      const float v = x[(int64_t)i * D + k];
      const float q = v * v;
      s2 += q;
      s4 = fmaf(q, q, s4);
    }
    acc += (double)(s2 * s2 - s4);
    for (int i = 0; i < B; ++i) {
      const float v = x[(int64_t)i * D + k];  // second read comes from L2
      grad[(int64_t)i * D + k] = 4.f * inv_d * v * (s2 - v * v);
    }
  }
  double v[1] = {acc};
  bh::block_sum<1>(v, lds);
  if (threadIdx.x == 0) partials[blockIdx.x] = v[0] * (double)inv_d;
}

// ---- PSNR ------------------------------------------------------------------------------------------------------------
// One 1024-thread workgroup per example: mean squared error between the de-normalised, clamped images.
constexpr int kPsnrBlock = 1024;

__global__ __launch_bounds__(kPsnrBlock) void psnr_mse_kernel(const float* __restrict__ rec, const float* __restrict__ ref,
                                                              int64_t per_example, int64_t plane, int channels,
                                                              bh_psnr_params P, double* __restrict__ mse) {
  __shared__ double lds[kPsnrBlock / bh::kWave];
  const float* __restrict__ a = rec + (int64_t)blockIdx.x * per_example;
  const float* __restrict__ b = ref + (int64_t)blockIdx.x * per_example;
  double acc = 0.0;
  float part = 0.f;
  int cnt = 0;
  for (int64_t i = threadIdx.x; i < per_example; i += kPsnrBlock) {
    const int c = channels > 1 ? (int)((i / plane) % channels) : 0;
    float u = fmaf(a[i], P.std[c], P.mean[c]), v = fmaf(b[i], P.std[c], P.mean[c]);
    if (P.clip) {
      u = fminf(fmaxf(u, 0.f), 1.f);
      v = fminf(fmaxf(v, 0.f), 1.f);
    }
    const float e = u - v;
    part = fmaf(e, e, part);
    if (++cnt == 16) {
      acc += (double)part;
      part = 0.f;
      cnt = 0;
    }
  }
  double v[1] = {acc + (double)part};
  bh::block_sum<1>(v, lds);
  if (threadIdx.x == 0) mse[blockIdx.x] = v[0] / (double)per_example;
}

// out[0] = mean PSNR, out[1] = max PSNR, out[2 + b] = PSNR of example b; +inf if any example matches exactly, NaN if any
// MSE is not finite (metrics.py:122-130).
__global__ void psnr_finalize_kernel(const double* __restrict__ mse, int B, float factor, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  bool any_zero = false, any_bad = false;
  double sum = 0.0, best = -__builtin_inf();
  for (int b = 0; b < B; ++b) {
    const double m = mse[b];
    any_zero |= m == 0.0;
    any_bad |= !isfinite(m);
    const double p = 10.0 * log10((double)factor * (double)factor / m);
    out[2 + b] = (float)p;
    sum += p;
    best = p > best ? p : best;
  }
  if (any_zero) {
    out[0] = out[1] = __builtin_inff();
  } else if (any_bad) {
    out[0] = out[1] = __builtin_nanf("");
  } else {
    out[0] = (float)(sum / B);
    out[1] = (float)best;
  }
}

}  // namespace

extern "C" {

int32_t bh_mt_num_groups(int32_t n_tensors) {
  return n_tensors <= 0 ? 0 : (n_tensors + BH_MT_MAX_PTRS - 1) / BH_MT_MAX_PTRS;
}

int bh_mt_group_bounds(int32_t n_tensors, const bh_gm_chunk* chunks_host, int64_t n_chunks, int32_t* group_chunk_begin) {
  if (n_tensors < 0 || n_chunks < 0 || group_chunk_begin == nullptr || (n_chunks > 0 && chunks_host == nullptr))
    return BH_EINVAL;
  const int groups = bh_mt_num_groups(n_tensors);
  int64_t c = 0;
  for (int g = 0; g < groups; ++g) {
    while (c < n_chunks && chunks_host[c].tensor < g * BH_MT_MAX_PTRS) ++c;
    group_chunk_begin[g] = (int32_t)c;
  }
  group_chunk_begin[groups] = (int32_t)n_chunks;
  return 0;
}

int bh_mt_axpy(int32_t n_tensors, const void* const* a_ptrs, const void* const* b_ptrs, const void* const* c_ptrs,
               float alpha, const bh_gm_chunk* chunks_dev, int64_t n_chunks, const int32_t* group_chunk_begin,
               float* out_flat, void* stream) {
  if (b_ptrs == nullptr) return BH_EINVAL;
  if (c_ptrs != nullptr)
    return run_mt<kAxpyMinus>(n_tensors, a_ptrs, b_ptrs, c_ptrs, nullptr, alpha, 0.f, nullptr, chunks_dev, n_chunks,
                              group_chunk_begin, out_flat, stream, false);
  return run_mt<kAxpy>(n_tensors, a_ptrs, b_ptrs, nullptr, nullptr, alpha, 0.f, nullptr, chunks_dev, n_chunks,
                       group_chunk_begin, out_flat, stream, false);
}

int bh_mt_scale(int32_t n_tensors, const void* const* a_ptrs, float alpha, const bh_gm_chunk* chunks_dev, int64_t n_chunks,
                const int32_t* group_chunk_begin, float* out_flat, void* stream) {
  return run_mt<kScale>(n_tensors, a_ptrs, nullptr, nullptr, nullptr, alpha, 0.f, nullptr, chunks_dev, n_chunks,
                        group_chunk_begin, out_flat, stream, true);
}

int bh_mt_patch(int32_t n_tensors, const void* const* theta_ptrs, const void* const* grad_ptrs, const float* data_flat,
                const float* coef_dev, float mult, const bh_gm_chunk* chunks_dev, int64_t n_chunks,
                const int32_t* group_chunk_begin, float* out_flat, void* stream) {
  if (grad_ptrs == nullptr || data_flat == nullptr || coef_dev == nullptr) return BH_EINVAL;
  return run_mt<kPatch>(n_tensors, theta_ptrs, grad_ptrs, nullptr, data_flat, mult, 0.f, coef_dev, chunks_dev, n_chunks,
                        group_chunk_begin, out_flat, stream, false);
}

int bh_prior_orthogonality(const float* x, int32_t B, int64_t D, float* grad_out, double* partials_dev, void* stream) {
  if (x == nullptr || grad_out == nullptr || partials_dev == nullptr || B <= 0 || D <= 0) return BH_EINVAL;
  int64_t blocks = (D + kBlock - 1) / kBlock;
  const int grid = (int)(blocks < BH_PRIOR_MAX_GRID ? blocks : BH_PRIOR_MAX_GRID);
  hipLaunchKernelGGL(orthogonality_kernel, dim3(grid), dim3(kBlock), 0, bh::as_stream(stream), x, B, D, grad_out,
                     partials_dev);
  const int rc = bh::launch_status();
  return rc != 0 ? rc : grid;
}

int bh_metric_psnr(const float* rec, const float* ref, int32_t B, int64_t per_example, int64_t plane, int32_t channels,
                   const bh_psnr_params* params, double* mse_dev, float* out_dev, void* stream) {
  if (rec == nullptr || ref == nullptr || B <= 0 || per_example <= 0 || params == nullptr || mse_dev == nullptr ||
      out_dev == nullptr || channels < 1 || channels > 4 || plane <= 0)
    return BH_EINVAL;
  hipStream_t st = bh::as_stream(stream);
  hipLaunchKernelGGL(psnr_mse_kernel, dim3(B), dim3(kPsnrBlock), 0, st, rec, ref, per_example, plane, channels, *params,
                     mse_dev);
  int rc = bh::launch_status();
  if (rc != 0) return rc;
  hipLaunchKernelGGL(psnr_finalize_kernel, dim3(1), dim3(64), 0, st, mse_dev, B, params->factor, out_dev);
  return bh::launch_status();
}

}  // extern "C"
