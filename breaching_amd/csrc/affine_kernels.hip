// Eval-mode BatchNorm of the attacker's private model copy as ONE kernel per autograd order.
//
// In eval mode (public or user-supplied buffers: base_attack.py:182-188 puts the rebuilt model in eval()) a BatchNorm2d is the
// per-channel affine map  y = x * s_c + t_c,  s_c = weight_c * inv_std_c,  t_c = bias_c - weight_c * mean_c * inv_std_c.
// The attack differentiates it twice per iteration: the first-order pass (autograd.grad(task_loss, params, create_graph=True),
// objectives.py:40-46) needs d/dx, d/dweight, d/dbias, and the pass from the gradient-matching objective back to the candidate
// needs the derivative of THAT.  PyTorch decomposes the three orders into ~35 elementwise / reduction launches per layer
// (measured on ResNet-18: ~700 of the ~1170 launches of an iteration are this decomposition; profiles/r3_bench_kernel_summary.csv)
// -- pure launch latency at B = 1.  Here each order is one launch:
//   forward        y   = x * s_c + t_c
//   backward       gx  = gy * s_c ;  gw_c = inv_c * sum(gy * x) - mi_c * sum(gy) ;  gb_c = sum(gy)          (mi_c = mean_c * inv_c)
//   backward of the backward, for incoming (ggx, ggw, ggb):
//                  d_gy = ggx * s_c + ggw_c * (inv_c * x - mi_c) + ggb_c ;  d_x = ggw_c * inv_c * gy ;
//                  d_w_c = inv_c * sum(ggx * gy)
// Optional epilogue (round 4): the residual add and the ReLU that follow the BatchNorm in ResNet blocks ride in the same launches,
//   forward        z = x * s_c + t_c (+ r) ;  y = relu(z)
//   backward       gz = gy * [y > 0] ;  gx = gz * s_c ;  gr = gz ;  gw_c, gb_c from gz
//   backward^2     d_gy = [y > 0] * (ggx * s_c + ggr + ggw_c * (inv_c * x - mi_c) + ggb_c) ;  d_x = ggw_c * inv_c * gz ;
//                  d_w_c = inv_c * sum(ggx * gz)
// (ReLU's second derivative vanishes almost everywhere: the mask is a constant of all three orders, read back from y.  The mask is
// !(y <= 0), not y > 0: a NaN pre-activation stays NaN in the forward, as in torch, and must then pass its gradient like torch's
// threshold_backward does, instead of being zeroed.)  On
// ResNet-18 that removes ~110 of the ~717 launches of an attack iteration: clamp_min, threshold_backward in both passes, the
// derivative of threshold_backward and its zero fill, and the residual add (profiles/r4_op_attribution.txt).
// The backward can also carry the DeepInversion prior's term of this BatchNorm input: gx += gout * (A_c + B_c * x).
// One workgroup per channel (one wavefront per channel when B * HW is small), fp32 arithmetic with fp64 channel sums, fixed
// reduction order => run-to-run reproducible.  Bandwidth / latency bound elementwise + reduction work: no MFMA.

#include "bh_common.h"

namespace {

using bh::kBlock;

// Geometry shared by the three kernels (bh::channel_geometry, the rule of kernel D's forward): x is [B, C, HW] contiguous; the
// B planes of a channel form one virtual array of B * HW elements.  Narrow (fewer than 2048): one wavefront per channel, four
// channels per workgroup.  Otherwise a channel is cut into S slabs of about 8192 elements, one workgroup each (S = 1 for every
// ResNet-18 layer at B = 1 but the stem); with S > 1 the per-slab channel sums of the backward orders go to a workspace and a
// small second launch adds them in slab order -- large activations (ResNet-50 at B = 8: 100 k elements per channel) would
// otherwise be streamed by C workgroups only.
//
// The forward can also hand kernel D its input: with `stats` given it writes sum(x) and sum(x^2) per (channel, slab) in exactly
// the layout bn_finalize_kernel reads -- the DeepInversion prior's statistics (deepinversion.py:93-96) of a BatchNorm INPUT
// come out of the pass that reads that input anyway, and bn_sums_kernel's 355.6 MB re-read (ResNet-50, B = 8) disappears.

struct ChannelWalk {
  int c, slab, lane, lanes;
  bool active;
};

__device__ __forceinline__ ChannelWalk channel_of(int C, int S, bool narrow) {
  ChannelWalk w;
  if (narrow) {
    w.c = blockIdx.x * bh::kWavesPerBlock + (threadIdx.x >> 6);
    w.slab = 0;
    w.lane = threadIdx.x & (bh::kWave - 1);
    w.lanes = bh::kWave;
  } else {
    w.c = blockIdx.x / S;
    w.slab = blockIdx.x - w.c * S;
    w.lane = threadIdx.x;
    w.lanes = kBlock;
  }
  w.active = w.c < C;
  return w;
}

// the slab's range [v0, v1) of the channel's virtual array, in units (float4 when HW % 4 == 0, float otherwise)
__device__ __forceinline__ void slab_range(int B, int HW, int S, int slab, bool vec, uint32_t& unit, uint32_t& v0, uint32_t& v1) {
  unit = vec ? (uint32_t)(HW >> 2) : (uint32_t)HW;
  const uint32_t total = (uint32_t)B * unit;
  v0 = (uint32_t)((uint64_t)total * (uint32_t)slab / (uint32_t)S);
  v1 = (uint32_t)((uint64_t)total * ((uint32_t)slab + 1u) / (uint32_t)S);
}

// element (float4 or float) `u` of the channel's virtual array -> offset in the [B, C, HW] tensor, in the same units
__device__ __forceinline__ size_t slab_offset(uint32_t u, int B, uint32_t unit, size_t cstride, size_t cbase) {
  const uint32_t b = B == 1 ? 0u : u / unit;
  return (size_t)b * cstride + cbase + (u - b * unit);
}

// 16-byte loads per operand that a lane issues before it uses the first one.  At B = 1 these kernels run 32-128 workgroups, at
// most one wavefront per SIMD, so nothing hides a load's latency but the lane's own next loads: with the plain
// load-use-store loop of rounds 3-4 a ResNet-18 layer-1 launch (784 float4 per channel over 256 lanes) was three to four serial
// memory round trips (instruction census, profiles/r4_kernel_isa_census.txt); staged, it is one.  The order in which a lane
// accumulates its channel sums is unchanged, so every result is bit-identical to the unstaged kernels.
constexpr int kStage = 4;

// per-channel(-slab) sums of K doubles: wide = block_sum, narrow = wave_sum; result valid in lane 0 of the owner
template <int K>
__device__ __forceinline__ void channel_sum(double (&v)[K], bool narrow, double* lds) {
  if (narrow) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = bh::wave_sum(v[k]);
  } else {
    bh::block_sum<K>(v, lds);
  }
}

__global__ __launch_bounds__(kBlock) void bn_eval_fwd_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                             const float* __restrict__ bias, const float* __restrict__ inv_std,
                                                             const float* __restrict__ mean_inv, float* __restrict__ y,
                                                             double* __restrict__ stats, const float* __restrict__ resid, int relu,
                                                             int B, int C, int HW, int S, int narrow) {
  __shared__ double lds[bh::kWavesPerBlock * 2];
  const ChannelWalk w = channel_of(C, S, narrow != 0);
  double sums[2] = {0.0, 0.0};  // sum x, sum x^2 of this (channel, slab): only with `stats`
  // epilogue: (+ residual), ReLU written as (z < 0 ? 0 : z) so that a NaN stays a NaN like torch's clamp_min
  auto act = [relu](float z, float r) {
    z += r;
    return (relu && z < 0.f) ? 0.f : z;
  };
  if (w.active) {
    const float wc = weight ? weight[w.c] : 1.f;
    const float s = wc * inv_std[w.c];
    const float t = (bias ? bias[w.c] : 0.f) - wc * mean_inv[w.c];
    const bool vec = (HW & 3) == 0;
    uint32_t unit, v0, v1;
    slab_range(B, HW, S, w.slab, vec, unit, v0, v1);
    const size_t cstride = (size_t)C * unit, cbase = (size_t)w.c * unit;
    float a0 = 0.f, a1 = 0.f;
    int cnt = 0;
    auto flush = [&]() {  // at most 32 values per fp32 accumulator, like bn_sums_kernel
      if (cnt >= 32) {
        sums[0] += (double)a0;
        sums[1] += (double)a1;
        a0 = a1 = 0.f;
        cnt = 0;
      }
    };
    if (vec) {
      const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
      const float4* __restrict__ r4 = reinterpret_cast<const float4*>(resid);
      float4* __restrict__ y4 = reinterpret_cast<float4*>(y);
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      const uint32_t step = (uint32_t)w.lanes;
      for (uint32_t v = v0 + (uint32_t)w.lane; v < v1; v += step * kStage) {
        float4 q[kStage], r[kStage];
        size_t at[kStage];
#pragma unroll
        for (int k = 0; k < kStage; ++k) {  // every load of this round is issued before the first use (see kStage)
          const uint32_t u = v + (uint32_t)k * step;
          at[k] = slab_offset(u < v1 ? u : v, B, unit, cstride, cbase);  // past the end: re-read the lane's first element, unused
          q[k] = x4[at[k]];
          r[k] = zero;
          if (resid) r[k] = r4[at[k]];
        }
#pragma unroll
        for (int k = 0; k < kStage; ++k) {
          const uint32_t u = v + (uint32_t)k * step;
          if (u < v1) {
            if (resid == nullptr && !relu)
              y4[at[k]] = make_float4(fmaf(q[k].x, s, t), fmaf(q[k].y, s, t), fmaf(q[k].z, s, t), fmaf(q[k].w, s, t));
            else
              y4[at[k]] = make_float4(act(fmaf(q[k].x, s, t), r[k].x), act(fmaf(q[k].y, s, t), r[k].y), act(fmaf(q[k].z, s, t), r[k].z),
                                      act(fmaf(q[k].w, s, t), r[k].w));
            a0 += (q[k].x + q[k].y) + (q[k].z + q[k].w);
            a1 = fmaf(q[k].x, q[k].x, a1);
            a1 = fmaf(q[k].y, q[k].y, a1);
            a1 = fmaf(q[k].z, q[k].z, a1);
            a1 = fmaf(q[k].w, q[k].w, a1);
            cnt += 4;
            flush();
          }
        }
      }
    } else {
      for (uint32_t v = v0 + (uint32_t)w.lane; v < v1; v += (uint32_t)w.lanes) {
        const size_t at = slab_offset(v, B, unit, cstride, cbase);
        const float q = x[at];
        y[at] = (resid == nullptr && !relu) ? fmaf(q, s, t) : act(fmaf(q, s, t), resid ? resid[at] : 0.f);
        a0 += q;
        a1 = fmaf(q, q, a1);
        cnt += 1;
        flush();
      }
    }
    sums[0] += (double)a0;
    sums[1] += (double)a1;
  }
  if (stats == nullptr) return;  // uniform: no barrier is skipped by part of a workgroup
  channel_sum<2>(sums, narrow != 0, lds);
  if (w.active && w.lane == 0) {
    double* out = stats + 2 * ((size_t)w.c * S + w.slab);
    out[0] = sums[0];
    out[1] = sums[1];
  }
}

// partial layout when S > 1: part[(c * S + slab) * 2 + k]
__global__ __launch_bounds__(kBlock) void bn_eval_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                             const float* __restrict__ weight, const float* __restrict__ inv_std,
                                                             const float* __restrict__ mean_inv, float* __restrict__ gx,
                                                             float* __restrict__ gw, float* __restrict__ gb,
                                                             double* __restrict__ part, const float* __restrict__ tap_coef,
                                                             const float* __restrict__ tap_gout, const float* __restrict__ ymask,
                                                             float* __restrict__ gres, const float* __restrict__ add_in, int B, int C,
                                                             int HW, int S, int narrow) {
  __shared__ double lds[bh::kWavesPerBlock * 2];
  const ChannelWalk w = channel_of(C, S, narrow != 0);
  double v[2] = {0.0, 0.0};  // sum gz, sum gz * x   (gz = gy behind the ReLU mask, = gy without one)
  if (w.active) {
    const float s = (weight ? weight[w.c] : 1.f) * inv_std[w.c];
    // DeepInversion tap (kernel D's backward riding in this launch): gx += g * (A_c + B_c * x), the term rounded exactly as
    // bn_bwd_acc_kernel rounds it (fmaf(g * B_c, x, g * A_c)) and then added
    float ta = 0.f, tb = 0.f;
    if (tap_coef) {
      const float2 ab = reinterpret_cast<const float2*>(tap_coef)[w.c];
      const float g0 = tap_gout ? tap_gout[0] : 1.f;
      ta = g0 * ab.x;
      tb = g0 * ab.y;
    }
    const bool vec = (HW & 3) == 0;
    uint32_t unit, v0, v1;
    slab_range(B, HW, S, w.slab, vec, unit, v0, v1);
    const size_t cstride = (size_t)C * unit, cbase = (size_t)w.c * unit;
    float a0 = 0.f, a1 = 0.f;
    int cnt = 0;
    auto flush = [&]() {  // at most 32 values per fp32 accumulator
      if (cnt >= 32) {
        v[0] += (double)a0;
        v[1] += (double)a1;
        a0 = a1 = 0.f;
        cnt = 0;
      }
    };
    if (vec) {
      const float4* __restrict__ gy4 = reinterpret_cast<const float4*>(gy);
      const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
      // without a mask the mask operand reads gy again (the line is in flight already: no HBM traffic) -- an `if (ymask)` around
      // the load lets the compiler sink the masking into it, and with it a wait for the load, in the middle of the load phase
      const float4* m4 = reinterpret_cast<const float4*>(ymask ? ymask : gy);
      const float4* __restrict__ e4 = reinterpret_cast<const float4*>(add_in);
      float4* __restrict__ gres4 = reinterpret_cast<float4*>(gres);
      float4* __restrict__ gx4 = reinterpret_cast<float4*>(gx);
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      const uint32_t step = (uint32_t)w.lanes;
      for (uint32_t u0 = v0 + (uint32_t)w.lane; u0 < v1; u0 += step * kStage) {
        float4 gv[kStage], qv[kStage], mv[kStage], ev[kStage];
        size_t at[kStage];
#pragma unroll
        for (int k = 0; k < kStage; ++k) {  // every load of this round before the first use
          const uint32_t u = u0 + (uint32_t)k * step;
          at[k] = slab_offset(u < v1 ? u : u0, B, unit, cstride, cbase);  // past the end: re-read the lane's first element, unused
          gv[k] = gy4[at[k]];
          qv[k] = x4[at[k]];
          mv[k] = m4[at[k]];
          ev[k] = zero;
          if (gx && add_in) ev[k] = e4[at[k]];
        }
#pragma unroll
        for (int k = 0; k < kStage; ++k) {
          const uint32_t u = u0 + (uint32_t)k * step;
          if (u < v1) {
            float4 g = gv[k];
            const float4 q = qv[k];
            if (ymask) {
              const float4 m = mv[k];
              g = make_float4(m.x <= 0.f ? 0.f : g.x, m.y <= 0.f ? 0.f : g.y, m.z <= 0.f ? 0.f : g.z, m.w <= 0.f ? 0.f : g.w);  // !(y <= 0): see the header
            }
            if (gres) gres4[at[k]] = g;
            if (gx) {
              float4 o = make_float4(g.x * s, g.y * s, g.z * s, g.w * s);
              if (tap_coef) o = make_float4(o.x + fmaf(tb, q.x, ta), o.y + fmaf(tb, q.y, ta), o.z + fmaf(tb, q.z, ta), o.w + fmaf(tb, q.w, ta));
              if (add_in) {  // the other gradient of this BatchNorm input (from the derivative of this very backward): autograd's add, in here
                const float4 e = ev[k];
                o = make_float4(o.x + e.x, o.y + e.y, o.z + e.z, o.w + e.w);
              }
              gx4[at[k]] = o;
            }
            a0 += (g.x + g.y) + (g.z + g.w);
            a1 = fmaf(g.x, q.x, a1);
            a1 = fmaf(g.y, q.y, a1);
            a1 = fmaf(g.z, q.z, a1);
            a1 = fmaf(g.w, q.w, a1);
            cnt += 4;
            flush();
          }
        }
      }
    } else {
      for (uint32_t u = v0 + (uint32_t)w.lane; u < v1; u += (uint32_t)w.lanes) {
        const size_t at = slab_offset(u, B, unit, cstride, cbase);
        float g = gy[at];
        const float q = x[at];
        if (ymask && ymask[at] <= 0.f) g = 0.f;
        if (gres) gres[at] = g;
        if (gx) gx[at] = (tap_coef ? g * s + fmaf(tb, q, ta) : g * s) + (add_in ? add_in[at] : 0.f);
        a0 += g;
        a1 = fmaf(g, q, a1);
        cnt += 1;
        flush();
      }
    }
    v[0] += (double)a0;
    v[1] += (double)a1;
  }
  channel_sum<2>(v, narrow != 0, lds);
  if (w.active && w.lane == 0) {
    if (S > 1) {
      part[((size_t)w.c * S + w.slab) * 2] = v[0];
      part[((size_t)w.c * S + w.slab) * 2 + 1] = v[1];
    } else {
      if (gw) gw[w.c] = (float)((double)inv_std[w.c] * v[1] - (double)mean_inv[w.c] * v[0]);
      if (gb) gb[w.c] = (float)v[0];
    }
  }
}

__global__ __launch_bounds__(kBlock) void bn_eval_bwd_bwd_kernel(const float* __restrict__ ggx, const float* __restrict__ ggw,
                                                                 const float* __restrict__ ggb, const float* __restrict__ gy,
                                                                 const float* __restrict__ x, const float* __restrict__ weight,
                                                                 const float* __restrict__ inv_std,
                                                                 const float* __restrict__ mean_inv, float* __restrict__ d_gy,
                                                                 float* __restrict__ d_x, float* __restrict__ d_w,
                                                                 double* __restrict__ part, const float* __restrict__ ymask,
                                                                 const float* __restrict__ ggr, int B, int C, int HW, int S,
                                                                 int narrow) {
  __shared__ double lds[bh::kWavesPerBlock];
  const ChannelWalk w = channel_of(C, S, narrow != 0);
  double v[1] = {0.0};  // sum ggx * gz
  if (w.active) {
    const float inv = inv_std[w.c], mi = mean_inv[w.c];
    const float s = (weight ? weight[w.c] : 1.f) * inv;
    const float kw = ggw ? ggw[w.c] : 0.f;             // d objective / d gw_c
    const float kb = ggb ? ggb[w.c] : 0.f;             // d objective / d gb_c
    const float kwi = kw * inv, shift = kb - kw * mi;  // d_gy = ggx * s + kwi * x + shift ;  d_x = kwi * gy
    const bool vec = (HW & 3) == 0;
    uint32_t unit, v0, v1;
    slab_range(B, HW, S, w.slab, vec, unit, v0, v1);
    const size_t cstride = (size_t)C * unit, cbase = (size_t)w.c * unit;
    float a0 = 0.f;
    int cnt = 0;
    auto flush = [&]() {
      if (cnt >= 32) {
        v[0] += (double)a0;
        a0 = 0.f;
        cnt = 0;
      }
    };
    if (vec) {
      const float4* __restrict__ gy4 = reinterpret_cast<const float4*>(gy);
      const float4* __restrict__ ggx4 = reinterpret_cast<const float4*>(ggx);
      const float4* m4 = reinterpret_cast<const float4*>(ymask ? ymask : gy);  // see bn_eval_bwd_kernel
      const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
      const float4* __restrict__ r4 = reinterpret_cast<const float4*>(ggr);
      float4* __restrict__ d_gy4 = reinterpret_cast<float4*>(d_gy);
      float4* __restrict__ d_x4 = reinterpret_cast<float4*>(d_x);
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
      const uint32_t step = (uint32_t)w.lanes;
      for (uint32_t u0 = v0 + (uint32_t)w.lane; u0 < v1; u0 += step * kStage) {
        float4 gv[kStage], qv[kStage], yv[kStage], xv[kStage], rv[kStage];
        size_t at[kStage];
#pragma unroll
        for (int k = 0; k < kStage; ++k) {  // every load of this round before the first use
          const uint32_t u = u0 + (uint32_t)k * step;
          at[k] = slab_offset(u < v1 ? u : u0, B, unit, cstride, cbase);  // past the end: re-read the lane's first element, unused
          gv[k] = gy4[at[k]];
          qv[k] = xv[k] = rv[k] = zero;
          yv[k] = m4[at[k]];
          if (ggx) qv[k] = ggx4[at[k]];
          if (d_gy) xv[k] = x4[at[k]];
          if (d_gy && ggr) rv[k] = r4[at[k]];
        }
#pragma unroll
        for (int k = 0; k < kStage; ++k) {
          const uint32_t u = u0 + (uint32_t)k * step;
          if (u < v1) {
            float4 g = gv[k];
            const float4 q = qv[k];
            float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
            if (ymask) {
              m = make_float4(yv[k].x <= 0.f ? 0.f : 1.f, yv[k].y <= 0.f ? 0.f : 1.f, yv[k].z <= 0.f ? 0.f : 1.f, yv[k].w <= 0.f ? 0.f : 1.f);
              g = make_float4(g.x * m.x, g.y * m.y, g.z * m.z, g.w * m.w);  // gz
            }
            if (d_gy) {
              float4 o = make_float4(fmaf(q.x, s, fmaf(kwi, xv[k].x, shift)), fmaf(q.y, s, fmaf(kwi, xv[k].y, shift)),
                                     fmaf(q.z, s, fmaf(kwi, xv[k].z, shift)), fmaf(q.w, s, fmaf(kwi, xv[k].w, shift)));
              if (ggr) o = make_float4(o.x + rv[k].x, o.y + rv[k].y, o.z + rv[k].z, o.w + rv[k].w);
              if (ymask) o = make_float4(m.x > 0.f ? o.x : 0.f, m.y > 0.f ? o.y : 0.f, m.z > 0.f ? o.z : 0.f, m.w > 0.f ? o.w : 0.f);
              d_gy4[at[k]] = o;
            }
            if (d_x) d_x4[at[k]] = make_float4(kwi * g.x, kwi * g.y, kwi * g.z, kwi * g.w);
            a0 = fmaf(q.x, g.x, a0);
            a0 = fmaf(q.y, g.y, a0);
            a0 = fmaf(q.z, g.z, a0);
            a0 = fmaf(q.w, g.w, a0);
            cnt += 4;
            flush();
          }
        }
      }
    } else {
      for (uint32_t u = v0 + (uint32_t)w.lane; u < v1; u += (uint32_t)w.lanes) {
        const size_t at = slab_offset(u, B, unit, cstride, cbase);
        float g = gy[at];
        const float q = ggx ? ggx[at] : 0.f;
        const bool on = ymask == nullptr || !(ymask[at] <= 0.f);
        if (!on) g = 0.f;  // gz
        if (d_gy) d_gy[at] = on ? fmaf(q, s, fmaf(kwi, x[at], shift)) + (ggr ? ggr[at] : 0.f) : 0.f;
        if (d_x) d_x[at] = kwi * g;
        a0 = fmaf(q, g, a0);
        cnt += 1;
        flush();
      }
    }
    v[0] += (double)a0;
  }
  channel_sum<1>(v, narrow != 0, lds);
  if (w.active && w.lane == 0) {
    if (S > 1) part[(size_t)w.c * S + w.slab] = v[0];
    else if (d_w) d_w[w.c] = (float)((double)inv_std[w.c] * v[0]);
  }
}

// Second stage when a channel was cut into S > 1 slabs: thread c adds its channel's S partial rows in slab order.
// K = 2: (gw, gb) of the backward; K = 1: d_w of its derivative.
template <int K>
__global__ __launch_bounds__(kBlock) void bn_eval_combine_kernel(const double* __restrict__ part, const float* __restrict__ inv_std,
                                                                 const float* __restrict__ mean_inv, float* __restrict__ out0,
                                                                 float* __restrict__ out1, int C, int S) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= C) return;
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = 0.0;
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] += part[((size_t)c * S + s) * K + k];
  }
  if constexpr (K == 2) {
    if (out0) out0[c] = (float)((double)inv_std[c] * v[1] - (double)mean_inv[c] * v[0]);  // gw
    if (out1) out1[c] = (float)v[0];                                                        // gb
  } else {
    if (out0) out0[c] = (float)((double)inv_std[c] * v[0]);  // d_w
  }
}

bool eval_bn_args_ok(const void* x, const void* inv_std, const void* mean_inv, int32_t B, int32_t C, int32_t HW) {
  return x != nullptr && inv_std != nullptr && mean_inv != nullptr && B > 0 && C > 0 && HW > 0 &&
         (int64_t)B * HW < ((int64_t)1 << 31) && (int64_t)B * C * HW < ((int64_t)1 << 40);
}

int eval_bn_grid(int32_t B, int32_t C, int32_t HW, int& S, int& narrow) {
  bh::channel_geometry((int64_t)B * HW, S, narrow);
  return narrow ? (C + bh::kWavesPerBlock - 1) / bh::kWavesPerBlock : C * S;
}

}  // namespace

extern "C" {

int32_t bh_bn_eval_slabs(int32_t B, int32_t C, int32_t HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return BH_EINVAL;
  int S = 1, narrow = 0;
  eval_bn_grid(B, C, HW, S, narrow);
  return S;
}

int bh_bn_eval_fwd(const float* x, const float* weight, const float* bias, const float* inv_std, const float* mean_inv, float* y,
                   double* stats, const float* residual, int32_t relu, int32_t B, int32_t C, int32_t HW, void* stream) {
  if (!eval_bn_args_ok(x, inv_std, mean_inv, B, C, HW) || y == nullptr) return BH_EINVAL;
  if ((reinterpret_cast<uintptr_t>(stats) & 7u) != 0 || (relu != 0 && relu != 1)) return BH_EINVAL;
  if ((HW & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15u) != 0)
    return BH_EINVAL;
  int S = 1, narrow = 0;
  const int grid = eval_bn_grid(B, C, HW, S, narrow);
  hipLaunchKernelGGL(bn_eval_fwd_kernel, dim3(grid), dim3(kBlock), 0, bh::as_stream(stream), x, weight, bias, inv_std, mean_inv,
                     y, stats, residual, relu, B, C, HW, S, narrow);
  return bh::launch_status();
}

int bh_bn_eval_bwd(const float* gy, const float* x, const float* weight, const float* inv_std, const float* mean_inv, float* gx,
                   float* gw, float* gb, double* workspace, const float* tap_coef, const float* tap_gout, const float* y_mask,
                   float* g_residual, const float* gx_add, int32_t B, int32_t C, int32_t HW, void* stream) {
  if (!eval_bn_args_ok(x, inv_std, mean_inv, B, C, HW) || gy == nullptr) return BH_EINVAL;
  if ((reinterpret_cast<uintptr_t>(tap_coef) & 7u) != 0 || ((tap_coef != nullptr || gx_add != nullptr) && gx == nullptr)) return BH_EINVAL;
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx) |
                         reinterpret_cast<uintptr_t>(y_mask) | reinterpret_cast<uintptr_t>(g_residual) | reinterpret_cast<uintptr_t>(gx_add)) & 15u) != 0)
    return BH_EINVAL;
  int S = 1, narrow = 0;
  const int grid = eval_bn_grid(B, C, HW, S, narrow);
  if (S > 1 && workspace == nullptr) return BH_EINVAL;
  hipStream_t st = bh::as_stream(stream);
  hipLaunchKernelGGL(bn_eval_bwd_kernel, dim3(grid), dim3(kBlock), 0, st, gy, x, weight, inv_std, mean_inv, gx, gw, gb, workspace,
                     tap_coef, tap_gout, y_mask, g_residual, gx_add, B, C, HW, S, narrow);
  if (S > 1 && (gw != nullptr || gb != nullptr))
    hipLaunchKernelGGL(bn_eval_combine_kernel<2>, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, st, workspace, inv_std,
                       mean_inv, gw, gb, C, S);
  return bh::launch_status();
}

int bh_bn_eval_bwd_bwd(const float* ggx, const float* ggw, const float* ggb, const float* gy, const float* x, const float* weight,
                       const float* inv_std, const float* mean_inv, float* d_gy, float* d_x, float* d_w, double* workspace,
                       const float* y_mask, const float* gg_residual, int32_t B, int32_t C, int32_t HW, void* stream) {
  if (!eval_bn_args_ok(x, inv_std, mean_inv, B, C, HW) || gy == nullptr) return BH_EINVAL;
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(ggx) |
                         reinterpret_cast<uintptr_t>(d_gy) | reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(y_mask) |
                         reinterpret_cast<uintptr_t>(gg_residual)) & 15u) != 0)
    return BH_EINVAL;
  int S = 1, narrow = 0;
  const int grid = eval_bn_grid(B, C, HW, S, narrow);
  if (S > 1 && workspace == nullptr) return BH_EINVAL;
  hipStream_t st = bh::as_stream(stream);
  hipLaunchKernelGGL(bn_eval_bwd_bwd_kernel, dim3(grid), dim3(kBlock), 0, st, ggx, ggw, ggb, gy, x, weight, inv_std, mean_inv, d_gy,
                     d_x, d_w, workspace, y_mask, gg_residual, B, C, HW, S, narrow);
  if (S > 1 && d_w != nullptr)
    hipLaunchKernelGGL(bn_eval_combine_kernel<1>, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, st, workspace, inv_std,
                       mean_inv, d_w, static_cast<float*>(nullptr), C, S);
  return bh::launch_status();
}

}  // extern "C"
