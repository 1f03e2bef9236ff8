// Kernel A: gradient-matching reductions over a per-parameter gradient list, forward and backward.
//
// Replaces the TorchScript list loops of breaching/attacks/auxiliaries/objectives.py
// (:89-95 Euclidean, :133-141 EuclideanTag, :158-166 L1, :183-196 Cosine, :233-244 masked, :259-273 fast) -- about
// 1.5k tiny ATen launches per attack iteration for ResNet-18 -- by ONE multi-tensor launch per direction.
//
// Layout: the autograd gradients arrive as T separately allocated tensors whose addresses may change every
// iteration, so their base pointers travel in the kernel-argument segment (no device table to refresh, no copy,
// graph-capture friendly).  The observed gradient and the output gradient live in packed flat buffers.  Work is
// cut into chunks of BH_GM_CHUNK elements that never straddle a tensor.  The forward launch is a PERSISTENT grid of at
// most BH_GM_MAX_ROWS (default cap 512 = two per CU) 256-thread workgroups, all resident at once, no tail wave:
// workgroup w streams chunks w, w+G, w+2G, ... with 16-byte loads (4 per thread per operand in flight), keeps fp32 sums
// per chunk and fp64 sums across chunks, reduces once with wave64 shuffles + LDS and writes ONE row of fp64 partial
// sums.  A one-workgroup finalize kernel combines the <= 512 rows (two per thread) in a fixed order and runs the
// objective epilogue.  (Measured and rejected, profiles/r2_kernel_bench_fused_vs_twolaunch.json: letting the last
// workgroup to finish do the combine in the same launch -- the agent-scope release fence + ticket costs ~30 ns per
// workgroup, serialised: +13 us at 483 workgroups, +44 us at 1447.)
//
// Roofline: HBM-bound, 2*N*4 bytes forward, 3*N*4 bytes backward (N = total elements).  No MFMA: <= 2 flop/byte.

#include "bh_common.h"

#include <hip/hip_ext.h>

namespace {

using bh::kBlock;

struct GmPtrs {
  const float* p[BH_GM_MAX_PTRS];
};

constexpr int kVecPerThread = BH_GM_CHUNK / 4 / kBlock;  // float4 loads per thread per operand for a full chunk

// Cache policy of kernel A's streaming accesses (round 4, profiles/r4_nt_loads_probe.jsonl).  Both lists are read exactly once
// per launch, so nothing is gained by keeping their lines -- and in the attack loop the forward runs right behind a producer
// (autograd, or our own backward) whose dirty lines are still draining from L2 / the Infinity Cache: with plain loads the
// BERT-base forward (688.6 MB) takes 131.5 us behind a writer against 106.7 us alone; with non-temporal loads
// (`global_load_dwordx4 ... nt`) 94.6 us behind the same writer = 0.91 of the 8 TB/s peak (0.65 before).  A list that fits the
// 256 MiB Infinity Cache wants the opposite: the backward re-reads what the forward pulled in (ResNet-18: forward 15.2 vs 15.6 us
// but backward 22.3 vs 21.2 us with non-temporal forward loads).  Hence a per-launch policy, chosen by size unless the caller says
// otherwise: BH_GM_CACHE_AUTO / _KEEP / _STREAM / _STREAM_ALL (include/breach_hip.h; the AUTO rule is at resolve_cache_policy).
typedef float bh_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 stream_load(const float4* p) {
  if constexpr (NT) {
    const bh_v4f v = __builtin_nontemporal_load(reinterpret_cast<const bh_v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  } else {
    return *p;
  }
}
static_assert(BH_GM_CHUNK % (4 * kBlock) == 0, "chunk must be a whole number of float4 sweeps");

constexpr bool is_cosine_family(int kind) { return kind <= BH_GM_ANGULAR; }

template <int KIND>
__device__ __forceinline__ void accumulate(float r, float d, float& a0, float& a1, float& a2) {
  if constexpr (KIND == BH_GM_COSINE_MASKED) {
    const bool keep = fabsf(d) > 1e-6f;  // objectives.py:236
    r = keep ? r : 0.f;
    d = keep ? d : 0.f;
  }
  if constexpr (is_cosine_family(KIND)) {
    a0 = fmaf(r, d, a0);
    a1 = fmaf(r, r, a1);
    a2 = fmaf(d, d, a2);
  } else if constexpr (KIND == BH_GM_PEARL_L2) {
    const float e = r - d;
    a0 = fmaf(e, e, a0);  // residual norm (the objective)
    a1 = fmaf(r, r, a1);  // |grad|^2 (the finite-difference step adapts to it, objectives.py:346)
  } else {
    const float e = r - d;
    if constexpr (KIND != BH_GM_L1) a0 = fmaf(e, e, a0);
    if constexpr (KIND != BH_GM_L2) a1 += fabsf(e);
  }
}

template <int KIND>
__device__ __forceinline__ void accumulate4(const float4& r, const float4& d, float& a0, float& a1, float& a2) {
  accumulate<KIND>(r.x, d.x, a0, a1, a2);
  accumulate<KIND>(r.y, d.y, a0, a1, a2);
  accumulate<KIND>(r.z, d.z, a0, a1, a2);
  accumulate<KIND>(r.w, d.w, a0, a1, a2);
}

// Objective epilogue on the three combined sums (thread 0 of one workgroup).
__device__ void gm_epilogue(int kind, const double (&v)[3], float scale, float tag_scale, float fudge, float fd_eps,
                            float span_ticks, float* __restrict__ stats) {
  const double s = (double)scale;
  double loss = 0.0, c1 = 0.0, c2 = 0.0;
  // Pearlmutter finite differences (fd_eps > 0): first-order direction v = kd * data + kr * grad of the objective at
  // scale 1, and the step eps_n = eps / |grad| (objectives.py:343-346, :468-486)
  double kd = 0.0, kr = 0.0;
  if (kind <= BH_GM_ANGULAR) {
    const double dot = v[0], rr = v[1], dd = v[2];
    const double rn = sqrt(rr), dn = sqrt(dd);
    const double inv = 1.0 / (rn * dn);
    const double cosv = dot * inv;
    // d cos / d r = d * inv - r * dot / (rr * rn * dn)
    const double dcos_d = inv;
    const double dcos_r = -dot * inv / rr;
    kd = -dcos_d;  // first_order_cosine = data / (-|g| |d|) + grad * <g,d> / (|g|^3 |d|)
    kr = -dcos_r;
    if (kind == BH_GM_ANGULAR) {
      // objectives.py:210-214: acos(clamp(cos, -1+f, 1-f)) / pi * scale
      const double lo = -1.0 + (double)fudge, hi = 1.0 - (double)fudge;
      const bool clamped = !(cosv > lo && cosv < hi);
      const double cc = cosv < lo ? lo : (cosv > hi ? hi : cosv);
      const double pi = 3.14159265358979323846;
      loss = s * acos(cc) / pi;
      const double dl = clamped ? 0.0 : -s / (pi * sqrt(1.0 - cc * cc));
      c1 = dl * dcos_d;
      c2 = dl * dcos_r;
    } else {
      loss = s * (1.0 - cosv);  // objectives.py:195
      c1 = -s * dcos_d;
      c2 = (kind == BH_GM_COSINE_FAST) ? 0.0 : -s * dcos_r;  // fast variant: norms detached (:268-269)
    }
  } else if (kind == BH_GM_L2 || kind == BH_GM_PEARL_L2) {
    loss = s * 0.5 * v[0];  // objectives.py:95, :459
    c1 = s;
    kd = -1.0;  // residuals = grad - data (:455)
    kr = 1.0;
  } else if (kind == BH_GM_L1) {
    loss = s * 0.5 * v[1];  // objectives.py:166
    c2 = 0.5 * s;
  } else {  // BH_GM_TAG
    loss = s * 0.5 * v[0];  // objectives.py:141
    c1 = s;
    c2 = 0.5 * s * (double)tag_scale;
  }
  stats[BH_GM_STAT_LOSS] = (float)loss;
  stats[BH_GM_STAT_C1] = (float)c1;
  stats[BH_GM_STAT_C2] = (float)c2;
  stats[BH_GM_STAT_S0] = (float)v[0];
  stats[BH_GM_STAT_S1] = (float)v[1];
  stats[BH_GM_STAT_S2] = (float)v[2];
  stats[BH_GM_STAT_SPAN_TICKS] = span_ticks;
  stats[7] = 0.f;
  // |grad|^2 is the second sum for every kind the finite differences apply to (cosine family, BH_GM_PEARL_L2).  Read it
  // straight from v[1]: carrying it through the branch chain above as a third variable was miscompiled by hipcc 7.2 for
  // kind == BH_GM_PEARL_L2 (the value arrived as 0, eps_n = inf; seen in the ISA and on the GPU).
  const double eps_n = fd_eps > 0.f ? (double)fd_eps / sqrt(v[1]) : 0.0;
  stats[BH_GM_STAT_PATCH_D] = (float)(eps_n * kd);
  stats[BH_GM_STAT_PATCH_R] = (float)(eps_n * kr);
  stats[BH_GM_STAT_FD_STEP] = (float)eps_n;
  stats[BH_GM_STAT_FD_SCALE] = fd_eps > 0.f ? (float)(s / eps_n) : 0.f;
}

// Fixed-order combine of `n_rows` partial rows by ONE 256-thread workgroup, then the epilogue: thread t sums rows
// t, t+256, ... (two for a single launch group at the default cap), wave64 shuffles, one LDS slot per wave, thread 0 finishes.  Each
// row also carries the wall-clock stamps of its workgroup; their envelope is the span of the forward launch.
__device__ void gm_combine_rows(int kind, const double* __restrict__ partials, int n_rows, float scale, float tag_scale,
                                float fudge, float fd_eps, float* __restrict__ stats, double* __restrict__ span_accum,
                                double* lds, int* span_lds) {
  double v[3] = {0.0, 0.0, 0.0};
  const bool want_span = span_accum != nullptr;  // bench-only bookkeeping: off the product path (two dependent loads less)
  const unsigned int base_tick =
      want_span ? (unsigned int)((unsigned long long)__double_as_longlong(partials[BH_GM_PARTIAL_STRIDE - 1]) >> 32) : 0u;
  int lo = 0x7fffffff, hi = -0x7fffffff;
  const double4* __restrict__ rows = reinterpret_cast<const double4*>(partials);
  for (int row = threadIdx.x; row < n_rows; row += kBlock) {
    const double4 p = rows[row];
    v[0] += p.x;
    v[1] += p.y;
    v[2] += p.z;
    if (want_span) {
      const unsigned long long packed = (unsigned long long)__double_as_longlong(p.w);
      const int t0 = (int)((unsigned int)(packed >> 32) - base_tick), t1 = (int)((unsigned int)packed - base_tick);
      lo = t0 < lo ? t0 : lo;  // wrap safe: 32-bit deltas relative to row 0
      hi = t1 > hi ? t1 : hi;
    }
  }
  const int lane = threadIdx.x & (bh::kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 3; ++k) v[k] = bh::wave_sum(v[k]);
  if (want_span) {
#pragma unroll
    for (int off = bh::kWave / 2; off > 0; off >>= 1) {
      const int olo = __shfl_down(lo, off, bh::kWave), ohi = __shfl_down(hi, off, bh::kWave);
      lo = olo < lo ? olo : lo;
      hi = ohi > hi ? ohi : hi;
    }
  }
  if (lane == 0) {
    lds[wave * 3 + 0] = v[0];
    lds[wave * 3 + 1] = v[1];
    lds[wave * 3 + 2] = v[2];
    span_lds[wave * 2 + 0] = lo;
    span_lds[wave * 2 + 1] = hi;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int w = 1; w < bh::kWavesPerBlock; ++w) {
    v[0] += lds[w * 3 + 0];
    v[1] += lds[w * 3 + 1];
    v[2] += lds[w * 3 + 2];
    lo = span_lds[w * 2] < lo ? span_lds[w * 2] : lo;
    hi = span_lds[w * 2 + 1] > hi ? span_lds[w * 2 + 1] : hi;
  }
  float span_ticks = 0.f;
  if (want_span) {  // running sum / count for bench.py (works under hipGraph replay, no host involvement)
    span_ticks = (float)(hi - lo);
    span_accum[0] += (double)span_ticks;
    span_accum[1] += 1.0;
  }
  gm_epilogue(kind, v, scale, tag_scale, fudge, fd_eps, span_ticks, stats);
}

// Measured and rejected in round 3 (commit 03d1fef, profiles/r3_kernel_bench_pipeline_sweep.json): software pipelining of
// the chunk loop -- next descriptor prefetched (mode 1), descriptors two ahead plus the next full chunk's loads issued
// before the current chunk is summed (mode 2).  ResNet-18 16.2 / 16.0 / 16.8 us, ResNet-50 31.1 / 31.1 / 32.6 us, BERT-base
// 111 / 110 / 115 us for modes 0 / 1 / 2: the eight waves per CU already overlap each other's descriptor and data latency, the
// extra registers of mode 2 (112 VGPRs) buy nothing.  The plain loop stays.
template <int KIND, bool NT>
__global__ __launch_bounds__(kBlock) void gm_fwd_kernel(GmPtrs ptrs, int tensor_base, const float* __restrict__ data_flat,
                                                        const bh_gm_chunk* __restrict__ chunks, int chunk_begin,
                                                        int chunk_end, const float* __restrict__ weights, float tag_scale,
                                                        double* __restrict__ partials, int row_base) {
  __shared__ double lds[bh::kWavesPerBlock * 3];
  const int tid = threadIdx.x;
  // constant-rate wall clock at block entry (thread 0 only): lets the combine step report the launch's true span
  const unsigned int tick0 = tid == 0 ? (unsigned int)wall_clock64() : 0u;
  double acc[3] = {0.0, 0.0, 0.0};  // per-thread sums across this workgroup's chunks
  for (int c = chunk_begin + blockIdx.x; c < chunk_end; c += gridDim.x) {
    const bh_gm_chunk ch = chunks[c];
    const float* __restrict__ r = ptrs.p[ch.tensor - tensor_base] + ch.tensor_off;
    const float* __restrict__ d = data_flat + ch.flat_off;
    // two independent fp32 accumulator sets keep the fma chains short; at most 16 values each before going to fp64
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
    const float4* __restrict__ r4 = reinterpret_cast<const float4*>(r);
    const float4* __restrict__ d4 = reinterpret_cast<const float4*>(d);
    if (ch.len == BH_GM_CHUNK) {
      float4 rv[kVecPerThread], dv[kVecPerThread];
#pragma unroll
      for (int k = 0; k < kVecPerThread; ++k) rv[k] = stream_load<NT>(r4 + tid + k * kBlock);
#pragma unroll
      for (int k = 0; k < kVecPerThread; ++k) dv[k] = stream_load<NT>(d4 + tid + k * kBlock);
#pragma unroll
      for (int k = 0; k < kVecPerThread; k += 2) {
        accumulate4<KIND>(rv[k], dv[k], a0, a1, a2);
        if (k + 1 < kVecPerThread) accumulate4<KIND>(rv[k + 1], dv[k + 1], b0, b1, b2);
      }
    } else {
      const int n4 = ch.len >> 2;
      for (int i = tid; i < n4; i += kBlock) accumulate4<KIND>(stream_load<NT>(r4 + i), stream_load<NT>(d4 + i), a0, a1, a2);
      const int tail = ch.len & 3;
      if (tid < tail) accumulate<KIND>(r[(n4 << 2) + tid], d[(n4 << 2) + tid], b0, b1, b2);
    }
    const double s0 = (double)a0 + (double)b0, s1 = (double)a1 + (double)b1, s2 = (double)a2 + (double)b2;
    if constexpr (KIND == BH_GM_TAG) {
      // objectives.py:139-140: (rec-data).pow(2).sum() + tag_scale * weight * (rec-data).abs().sum(), weight per tensor
      acc[0] += s0 + (double)tag_scale * (double)weights[ch.tensor] * s1;
    } else {
      acc[0] += s0;
    }
    acc[1] += s1;
    acc[2] += s2;
  }
  bh::block_sum<3>(acc, lds);
  if (tid == 0) {
    double* row = partials + (int64_t)(row_base + blockIdx.x) * BH_GM_PARTIAL_STRIDE;
    row[0] = acc[0];
    row[1] = acc[1];
    row[2] = acc[2];
    const unsigned long long packed = ((unsigned long long)tick0 << 32) | (unsigned int)wall_clock64();
    row[3] = __longlong_as_double((long long)packed);
  }
}

// Combine + epilogue (one workgroup).
__global__ __launch_bounds__(kBlock) void gm_finalize_kernel(int kind, const double* __restrict__ partials, int n_rows,
                                                             float scale, float tag_scale, float fudge, float fd_eps,
                                                             float* __restrict__ stats, double* __restrict__ span_accum) {
  __shared__ double lds[bh::kWavesPerBlock * 3];
  __shared__ int span_lds[bh::kWavesPerBlock * 2];
  gm_combine_rows(kind, partials, n_rows, scale, tag_scale, fudge, fd_eps, stats, span_accum, lds, span_lds);
}

template <int KIND>
__device__ __forceinline__ float bwd_elem(float r, float d, float k1, float k2) {
  if constexpr (is_cosine_family(KIND)) {
    float o = fmaf(k1, d, k2 * r);
    if constexpr (KIND == BH_GM_COSINE_MASKED) o = (fabsf(d) > 1e-6f) ? o : 0.f;
    return o;
  } else {
    const float e = r - d;
    if constexpr (KIND == BH_GM_L2) return k1 * e;
    if constexpr (KIND == BH_GM_L1) return k2 * bh::sgnf(e);
    return fmaf(k1, e, k2 * bh::sgnf(e));
  }
}

template <int KIND>
__device__ __forceinline__ float4 bwd_elem4(const float4& r, const float4& d, float k1, float k2) {
  return make_float4(bwd_elem<KIND>(r.x, d.x, k1, k2), bwd_elem<KIND>(r.y, d.y, k1, k2),
                     bwd_elem<KIND>(r.z, d.z, k1, k2), bwd_elem<KIND>(r.w, d.w, k1, k2));
}

template <bool NT>
__device__ __forceinline__ void stream_store(float4* p, const float4& v) {
  if constexpr (NT) {
    bh_v4f w;
    w.x = v.x, w.y = v.y, w.z = v.z, w.w = v.w;
    __builtin_nontemporal_store(w, reinterpret_cast<bh_v4f*>(p));
  } else {
    *p = v;
  }
}

template <int KIND, bool NTL, bool NTS>
__global__ __launch_bounds__(kBlock) void gm_bwd_kernel(GmPtrs ptrs, int tensor_base, const float* __restrict__ data_flat,
                                                        const bh_gm_chunk* __restrict__ chunks, int chunk_base,
                                                        const float* __restrict__ weights,
                                                        const float* __restrict__ stats, const float* __restrict__ gout,
                                                        float* __restrict__ grad_flat) {
  const int c = chunk_base + blockIdx.x;
  const bh_gm_chunk ch = chunks[c];
  const float* __restrict__ r = ptrs.p[ch.tensor - tensor_base] + ch.tensor_off;
  const float* __restrict__ d = data_flat + ch.flat_off;
  float* __restrict__ o = grad_flat + ch.flat_off;
  const float g = gout ? gout[0] : 1.f;
  const float k1 = g * stats[BH_GM_STAT_C1];
  float k2 = g * stats[BH_GM_STAT_C2];
  if constexpr (KIND == BH_GM_TAG) k2 *= weights[ch.tensor];
  const int tid = threadIdx.x;
  const float4* __restrict__ r4 = reinterpret_cast<const float4*>(r);
  const float4* __restrict__ d4 = reinterpret_cast<const float4*>(d);
  float4* __restrict__ o4 = reinterpret_cast<float4*>(o);
  if (ch.len == BH_GM_CHUNK) {
    float4 rv[kVecPerThread], dv[kVecPerThread];
#pragma unroll
    for (int k = 0; k < kVecPerThread; ++k) rv[k] = stream_load<NTL>(r4 + tid + k * kBlock);
#pragma unroll
    for (int k = 0; k < kVecPerThread; ++k) dv[k] = stream_load<NTL>(d4 + tid + k * kBlock);
#pragma unroll
    for (int k = 0; k < kVecPerThread; ++k) stream_store<NTS>(o4 + tid + k * kBlock, bwd_elem4<KIND>(rv[k], dv[k], k1, k2));
  } else {
    const int n4 = ch.len >> 2;
    for (int i = tid; i < n4; i += kBlock) stream_store<NTS>(o4 + i, bwd_elem4<KIND>(stream_load<NTL>(r4 + i), stream_load<NTL>(d4 + i), k1, k2));
    const int tail = ch.len & 3;
    if (tid < tail) {
      const int i = (n4 << 2) + tid;
      o[i] = bwd_elem<KIND>(r[i], d[i], k1, k2);
    }
  }
}

__global__ __launch_bounds__(kBlock) void gm_pack_kernel(GmPtrs ptrs, int tensor_base,
                                                         const bh_gm_chunk* __restrict__ chunks, int chunk_base,
                                                         float* __restrict__ flat_dst) {
  const int c = chunk_base + blockIdx.x;
  const bh_gm_chunk ch = chunks[c];
  const float* __restrict__ s = ptrs.p[ch.tensor - tensor_base] + ch.tensor_off;
  float* __restrict__ o = flat_dst + ch.flat_off;
  const int n4 = ch.len >> 2;
  const float4* __restrict__ s4 = reinterpret_cast<const float4*>(s);
  float4* __restrict__ o4 = reinterpret_cast<float4*>(o);
  for (int i = threadIdx.x; i < n4; i += kBlock) o4[i] = s4[i];
  const int tail = ch.len & 3;
  if ((int)threadIdx.x < tail) o[(n4 << 2) + threadIdx.x] = s[(n4 << 2) + threadIdx.x];
  // zero the alignment padding behind the last chunk of a tensor so the flat buffer is fully defined
  if (ch.len != BH_GM_CHUNK) {
    const int pad = (4 - (ch.len & 3)) & 3;
    if ((int)threadIdx.x < pad) o[ch.len + threadIdx.x] = 0.f;
  }
}

bool valid_kind(int kind) { return kind >= BH_GM_COSINE && kind <= BH_GM_PEARL_L2; }

// Fill the kernarg pointer block for launch group `g`; returns false on a misaligned / null pointer.
bool fill_ptrs(GmPtrs& out, const void* const* ptrs, int n_tensors, int g) {
  const int base = g * BH_GM_MAX_PTRS;
  const int cnt = (n_tensors - base) < BH_GM_MAX_PTRS ? (n_tensors - base) : BH_GM_MAX_PTRS;
  for (int i = 0; i < cnt; ++i) {
    const void* p = ptrs[base + i];
    if (p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15u) != 0) return false;
    out.p[i] = static_cast<const float*>(p);
  }
  for (int i = cnt; i < BH_GM_MAX_PTRS; ++i) out.p[i] = nullptr;
  return true;
}

bool aligned16(const void* p) { return p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Timed launches go through hipExtLaunchKernelGGL: the two events then carry the dispatch's own start / end
// timestamps (the same completion-signal times rocprofv3 reports), free of host latency and marker overhead.
struct LaunchEvents {
  hipEvent_t start = nullptr, stop = nullptr;
};

template <int KIND, bool NT>
void launch_fwd_policy(const GmPtrs& ptrs, int tensor_base, const float* data_flat, const bh_gm_chunk* chunks, int chunk_begin,
                       int chunk_end, int grid, const float* weights, float tag_scale, double* partials, int row_base,
                       hipStream_t st, LaunchEvents ev) {
  if (ev.start || ev.stop)
    hipExtLaunchKernelGGL((gm_fwd_kernel<KIND, NT>), dim3(grid), dim3(kBlock), 0, st, ev.start, ev.stop, 0, ptrs, tensor_base,
                          data_flat, chunks, chunk_begin, chunk_end, weights, tag_scale, partials, row_base);
  else
    hipLaunchKernelGGL((gm_fwd_kernel<KIND, NT>), dim3(grid), dim3(kBlock), 0, st, ptrs, tensor_base, data_flat, chunks,
                       chunk_begin, chunk_end, weights, tag_scale, partials, row_base);
}

template <int KIND>
void launch_fwd(const GmPtrs& ptrs, int tensor_base, const float* data_flat, const bh_gm_chunk* chunks, int chunk_begin,
                int chunk_end, int grid, const float* weights, float tag_scale, double* partials, int row_base,
                hipStream_t st, LaunchEvents ev, bool stream) {
  if (stream) launch_fwd_policy<KIND, true>(ptrs, tensor_base, data_flat, chunks, chunk_begin, chunk_end, grid, weights, tag_scale, partials, row_base, st, ev);
  else launch_fwd_policy<KIND, false>(ptrs, tensor_base, data_flat, chunks, chunk_begin, chunk_end, grid, weights, tag_scale, partials, row_base, st, ev);
}

template <int KIND, bool NTL, bool NTS>
void launch_bwd_policy(const GmPtrs& ptrs, int tensor_base, const float* data_flat, const bh_gm_chunk* chunks, int chunk_base,
                       int n, const float* weights, const float* stats, const float* gout, float* grad_flat, hipStream_t st,
                       LaunchEvents ev) {
  if (ev.start || ev.stop)
    hipExtLaunchKernelGGL((gm_bwd_kernel<KIND, NTL, NTS>), dim3(n), dim3(kBlock), 0, st, ev.start, ev.stop, 0, ptrs, tensor_base,
                          data_flat, chunks, chunk_base, weights, stats, gout, grad_flat);
  else
    hipLaunchKernelGGL((gm_bwd_kernel<KIND, NTL, NTS>), dim3(n), dim3(kBlock), 0, st, ptrs, tensor_base, data_flat, chunks,
                       chunk_base, weights, stats, gout, grad_flat);
}

// policy: BH_GM_CACHE_KEEP plain accesses; BH_GM_CACHE_STREAM non-temporal loads; BH_GM_CACHE_STREAM_ALL non-temporal loads and stores
template <int KIND>
void launch_bwd(const GmPtrs& ptrs, int tensor_base, const float* data_flat, const bh_gm_chunk* chunks, int chunk_base,
                int n, const float* weights, const float* stats, const float* gout, float* grad_flat, hipStream_t st,
                LaunchEvents ev, int policy) {
  if (policy == BH_GM_CACHE_STREAM_ALL) launch_bwd_policy<KIND, true, true>(ptrs, tensor_base, data_flat, chunks, chunk_base, n, weights, stats, gout, grad_flat, st, ev);
  else if (policy == BH_GM_CACHE_STREAM) launch_bwd_policy<KIND, true, false>(ptrs, tensor_base, data_flat, chunks, chunk_base, n, weights, stats, gout, grad_flat, st, ev);
  else launch_bwd_policy<KIND, false, false>(ptrs, tensor_base, data_flat, chunks, chunk_base, n, weights, stats, gout, grad_flat, st, ev);
}

// BH_GM_CACHE_AUTO, by the bytes the forward streams (two lists of n_chunks chunks of 4096 floats), measured INSIDE the attack loop
// (profiles/r4_cache_policy_probe.jsonl, r4_cache_policy_inloop_resnet18.txt, r4_config3_cache_policy_{1,2}_kernel_summary.txt):
//   ResNet-18 (93.5 MB): KEEP both -- non-temporal forward loads save 0.65 us there and cost the backward, which re-reads both lists
//     out of the Infinity Cache, 1.5 us;
//   ResNet-50 (204.5 MB): forward STREAM (43.0 -> 35.6 us behind the victim's double backward at B = 8), backward KEEP;
//   BERT-base (688.6 MB): forward STREAM (131.9 -> 99.5 us behind a writer), backward STREAM_ALL (170.6 -> 163.2 us).
int resolve_cache_policy(int32_t policy, int64_t n_chunks, bool backward) {
  if (policy != BH_GM_CACHE_AUTO) return policy;
  const int64_t forward_bytes = n_chunks * (int64_t)BH_GM_CHUNK * 8;
  if (backward) return forward_bytes > BH_GM_CACHE_AUTO_BYTES ? BH_GM_CACHE_STREAM_ALL : BH_GM_CACHE_KEEP;
  return forward_bytes > BH_GM_CACHE_AUTO_FWD_BYTES ? BH_GM_CACHE_STREAM : BH_GM_CACHE_KEEP;
}

// events of launch group g out of `groups`: start on the first non-empty group, stop on the last
LaunchEvents group_events(void* ev_start, void* ev_stop, bool first, bool last) {
  LaunchEvents ev;
  if (first) ev.start = static_cast<hipEvent_t>(ev_start);
  if (last) ev.stop = static_cast<hipEvent_t>(ev_stop);
  return ev;
}

// Persistent-grid size for a launch group of n chunks: every workgroup gets the same number of chunks (+-1) and the
// whole grid is resident at once.  Cap 512 (two workgroups per CU, 8 x 16 B loads in flight per thread) measured fastest
// or equal at every BASELINE size: BERT-base 109.7 us vs 117.6 us at 2048, ResNet-50 32.0 vs 33.0, ResNet-18 17.5 = 17.5.
// The cap is a launch argument (0 = BH_GM_DEFAULT_ROWS): the library keeps no mutable tuning state.
bool rows_cap_ok(int32_t cap) { return cap >= 0 && cap <= BH_GM_MAX_ROWS; }

int group_rows(int n_chunks_in_group, int32_t rows_cap) {
  if (n_chunks_in_group <= 0) return 0;
  const int cap = rows_cap > 0 ? rows_cap : BH_GM_DEFAULT_ROWS;
  const int rounds = (n_chunks_in_group + cap - 1) / cap;
  return (n_chunks_in_group + rounds - 1) / rounds;
}

}  // namespace

extern "C" {

int32_t bh_gm_num_groups(int32_t n_tensors) {
  return n_tensors <= 0 ? 0 : (n_tensors + BH_GM_MAX_PTRS - 1) / BH_GM_MAX_PTRS;
}

int bh_gm_table_size(int32_t n_tensors, const int64_t* numel, int64_t* n_chunks, int64_t* flat_elems) {
  if (n_tensors < 0 || (n_tensors > 0 && numel == nullptr) || n_chunks == nullptr || flat_elems == nullptr)
    return BH_EINVAL;
  int64_t chunks = 0, flat = 0;
  for (int32_t t = 0; t < n_tensors; ++t) {
    if (numel[t] < 0) return BH_EINVAL;
    chunks += (numel[t] + BH_GM_CHUNK - 1) / BH_GM_CHUNK;
    flat += (numel[t] + 3) & ~int64_t(3);
  }
  if (chunks > INT32_MAX) return BH_EINVAL;
  *n_chunks = chunks;
  *flat_elems = flat;
  return 0;
}

int bh_gm_build_table(int32_t n_tensors, const int64_t* numel, bh_gm_chunk* chunks, int64_t n_chunks,
                      int64_t* tensor_flat_off) {
  int64_t need = 0, flat_total = 0;
  int rc = bh_gm_table_size(n_tensors, numel, &need, &flat_total);
  if (rc != 0) return rc;
  if (need != n_chunks || (n_chunks > 0 && chunks == nullptr) || (n_tensors > 0 && tensor_flat_off == nullptr))
    return BH_EINVAL;
  int64_t c = 0, flat = 0;
  for (int32_t t = 0; t < n_tensors; ++t) {
    tensor_flat_off[t] = flat;
    for (int64_t off = 0; off < numel[t]; off += BH_GM_CHUNK) {
      const int64_t len = (numel[t] - off) < BH_GM_CHUNK ? (numel[t] - off) : BH_GM_CHUNK;
      chunks[c].flat_off = flat + off;
      chunks[c].tensor_off = off;
      chunks[c].tensor = t;
      chunks[c].len = (int32_t)len;
      ++c;
    }
    flat += (numel[t] + 3) & ~int64_t(3);
  }
  return 0;
}

int bh_gm_group_bounds(int32_t n_tensors, const bh_gm_chunk* chunks_host, int64_t n_chunks,
                       int32_t* group_chunk_begin) {
  if (n_tensors < 0 || n_chunks < 0 || group_chunk_begin == nullptr || (n_chunks > 0 && chunks_host == nullptr))
    return BH_EINVAL;
  const int groups = bh_gm_num_groups(n_tensors);
  int64_t c = 0;
  for (int g = 0; g < groups; ++g) {
    while (c < n_chunks && chunks_host[c].tensor < g * BH_GM_MAX_PTRS) ++c;
    group_chunk_begin[g] = (int32_t)c;
  }
  group_chunk_begin[groups] = (int32_t)n_chunks;
  return 0;
}

int32_t bh_gm_fwd_rows(int32_t n_tensors, const int32_t* group_chunk_begin, int32_t rows_cap) {
  if (n_tensors <= 0 || group_chunk_begin == nullptr || !rows_cap_ok(rows_cap)) return BH_EINVAL;
  const int groups = bh_gm_num_groups(n_tensors);
  int rows = 0;
  for (int g = 0; g < groups; ++g) rows += group_rows(group_chunk_begin[g + 1] - group_chunk_begin[g], rows_cap);
  return rows;
}

int bh_gm_fwd(int32_t kind, int32_t n_tensors, const void* const* rec_ptrs, const float* data_flat,
              const bh_gm_chunk* chunks_dev, int64_t n_chunks, const int32_t* group_chunk_begin,
              const float* weights_dev, float tag_scale, double* partials_dev, int32_t rows_cap, int32_t cache_policy,
              void* stream, void* ev_start, void* ev_stop) {
  if (!rows_cap_ok(rows_cap) || cache_policy < BH_GM_CACHE_AUTO || cache_policy > BH_GM_CACHE_STREAM_ALL) return BH_EINVAL;
  const bool stream_loads = resolve_cache_policy(cache_policy, n_chunks, false) != BH_GM_CACHE_KEEP;
  if (!valid_kind(kind) || n_tensors <= 0 || rec_ptrs == nullptr || !aligned16(data_flat) || chunks_dev == nullptr ||
      n_chunks <= 0 || group_chunk_begin == nullptr || partials_dev == nullptr)
    return BH_EINVAL;
  if ((reinterpret_cast<uintptr_t>(partials_dev) & 31u) != 0) return BH_EINVAL;  // rows are read as 32-byte vectors
  if (kind == BH_GM_TAG && weights_dev == nullptr) return BH_EINVAL;
  hipStream_t st = bh::as_stream(stream);
  const int groups = bh_gm_num_groups(n_tensors);
  for (int g = 0; g < groups; ++g) {  // validate every pointer before anything is enqueued
    GmPtrs probe;
    if (!fill_ptrs(probe, rec_ptrs, n_tensors, g)) return BH_EINVAL;
  }
  int row_base = 0;
  for (int g = 0; g < groups; ++g) {
    const int begin = group_chunk_begin[g], end = group_chunk_begin[g + 1];
    const int grid = group_rows(end - begin, rows_cap);
    if (grid <= 0) continue;
    GmPtrs ptrs;
    if (!fill_ptrs(ptrs, rec_ptrs, n_tensors, g)) return BH_EINVAL;
    const int tb = g * BH_GM_MAX_PTRS;
    const LaunchEvents ev = group_events(ev_start, ev_stop, begin == 0, end == n_chunks);
#define BH_FWD(K) \
  launch_fwd<K>(ptrs, tb, data_flat, chunks_dev, begin, end, grid, weights_dev, tag_scale, partials_dev, row_base, st, ev, stream_loads)
    switch (kind) {
      case BH_GM_COSINE:
      case BH_GM_COSINE_FAST:
      case BH_GM_ANGULAR:
        BH_FWD(BH_GM_COSINE);
        break;
      case BH_GM_COSINE_MASKED:
        BH_FWD(BH_GM_COSINE_MASKED);
        break;
      case BH_GM_L2:
        BH_FWD(BH_GM_L2);
        break;
      case BH_GM_L1:
        BH_FWD(BH_GM_L1);
        break;
      case BH_GM_PEARL_L2:
        BH_FWD(BH_GM_PEARL_L2);
        break;
      default:
        BH_FWD(BH_GM_TAG);
        break;
    }
#undef BH_FWD
    const int rc = bh::launch_status();
    if (rc != 0) return rc;
    row_base += grid;
  }
  return 0;
}

int bh_gm_finalize(int32_t kind, const double* partials_dev, int64_t n_rows, float scale, float tag_scale, float fudge,
                   float fd_eps, float* stats_dev, double* span_accum_dev, void* stream, void* ev_start, void* ev_stop) {
  if (!valid_kind(kind) || partials_dev == nullptr || n_rows <= 0 || n_rows > INT32_MAX || stats_dev == nullptr)
    return BH_EINVAL;
  if ((reinterpret_cast<uintptr_t>(partials_dev) & 31u) != 0) return BH_EINVAL;  // rows are read as 32-byte vectors
  if (ev_start || ev_stop)
    hipExtLaunchKernelGGL(gm_finalize_kernel, dim3(1), dim3(kBlock), 0, bh::as_stream(stream),
                          static_cast<hipEvent_t>(ev_start), static_cast<hipEvent_t>(ev_stop), 0, kind, partials_dev,
                          (int)n_rows, scale, tag_scale, fudge, fd_eps, stats_dev, span_accum_dev);
  else
    hipLaunchKernelGGL(gm_finalize_kernel, dim3(1), dim3(kBlock), 0, bh::as_stream(stream), kind, partials_dev, (int)n_rows,
                       scale, tag_scale, fudge, fd_eps, stats_dev, span_accum_dev);
  return bh::launch_status();
}

int32_t bh_wall_clock_khz(void) {
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess) return BH_EINVAL;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return BH_EINVAL;
  return khz;
}

int bh_gm_bwd(int32_t kind, int32_t n_tensors, const void* const* rec_ptrs, const float* data_flat,
              const bh_gm_chunk* chunks_dev, int64_t n_chunks, const int32_t* group_chunk_begin,
              const float* weights_dev, const float* stats_dev, const float* gout_dev, float* grad_flat, int32_t cache_policy,
              void* stream, void* ev_start, void* ev_stop) {
  if (cache_policy < BH_GM_CACHE_AUTO || cache_policy > BH_GM_CACHE_STREAM_ALL) return BH_EINVAL;
  const int policy = resolve_cache_policy(cache_policy, n_chunks, true);
  if (!valid_kind(kind) || n_tensors <= 0 || rec_ptrs == nullptr || !aligned16(data_flat) || chunks_dev == nullptr ||
      n_chunks <= 0 || group_chunk_begin == nullptr || stats_dev == nullptr || !aligned16(grad_flat))
    return BH_EINVAL;
  if (kind == BH_GM_TAG && weights_dev == nullptr) return BH_EINVAL;
  hipStream_t st = bh::as_stream(stream);
  const int groups = bh_gm_num_groups(n_tensors);
  for (int g = 0; g < groups; ++g) {  // validate every pointer before anything is enqueued
    GmPtrs probe;
    if (!fill_ptrs(probe, rec_ptrs, n_tensors, g)) return BH_EINVAL;
  }
  for (int g = 0; g < groups; ++g) {
    const int begin = group_chunk_begin[g], n = group_chunk_begin[g + 1] - begin;
    if (n <= 0) continue;
    GmPtrs ptrs;
    if (!fill_ptrs(ptrs, rec_ptrs, n_tensors, g)) return BH_EINVAL;
    const int tb = g * BH_GM_MAX_PTRS;
    const LaunchEvents ev = group_events(ev_start, ev_stop, begin == 0, group_chunk_begin[g + 1] == n_chunks);
    switch (kind) {
      case BH_GM_COSINE:
      case BH_GM_COSINE_FAST:
      case BH_GM_ANGULAR:
        launch_bwd<BH_GM_COSINE>(ptrs, tb, data_flat, chunks_dev, begin, n, weights_dev, stats_dev, gout_dev, grad_flat,
                                 st, ev, policy);
        break;
      case BH_GM_COSINE_MASKED:
        launch_bwd<BH_GM_COSINE_MASKED>(ptrs, tb, data_flat, chunks_dev, begin, n, weights_dev, stats_dev, gout_dev,
                                        grad_flat, st, ev, policy);
        break;
      case BH_GM_L2:
      case BH_GM_PEARL_L2:  // same derivative as the euclidean objective
        launch_bwd<BH_GM_L2>(ptrs, tb, data_flat, chunks_dev, begin, n, weights_dev, stats_dev, gout_dev, grad_flat, st, ev, policy);
        break;
      case BH_GM_L1:
        launch_bwd<BH_GM_L1>(ptrs, tb, data_flat, chunks_dev, begin, n, weights_dev, stats_dev, gout_dev, grad_flat, st, ev, policy);
        break;
      default:
        launch_bwd<BH_GM_TAG>(ptrs, tb, data_flat, chunks_dev, begin, n, weights_dev, stats_dev, gout_dev, grad_flat,
                              st, ev, policy);
        break;
    }
    const int rc = bh::launch_status();
    if (rc != 0) return rc;
  }
  return 0;
}

int bh_gm_pack(int32_t n_tensors, const void* const* src_ptrs, const bh_gm_chunk* chunks_dev, int64_t n_chunks,
               const int32_t* group_chunk_begin, float* flat_dst, void* stream) {
  if (n_tensors <= 0 || src_ptrs == nullptr || chunks_dev == nullptr || n_chunks <= 0 || group_chunk_begin == nullptr ||
      !aligned16(flat_dst))
    return BH_EINVAL;
  hipStream_t st = bh::as_stream(stream);
  const int groups = bh_gm_num_groups(n_tensors);
  for (int g = 0; g < groups; ++g) {
    const int begin = group_chunk_begin[g], n = group_chunk_begin[g + 1] - begin;
    if (n <= 0) continue;
    GmPtrs ptrs;
    if (!fill_ptrs(ptrs, src_ptrs, n_tensors, g)) return BH_EINVAL;
    hipLaunchKernelGGL(gm_pack_kernel, dim3(n), dim3(kBlock), 0, st, ptrs, g * BH_GM_MAX_PTRS, chunks_dev, begin,
                       flat_dst);
    const int rc = bh::launch_status();
    if (rc != 0) return rc;
  }
  return 0;
}

}  // extern "C"
