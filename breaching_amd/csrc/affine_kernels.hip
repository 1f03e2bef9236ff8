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
// One workgroup per channel (one wavefront per channel when B * HW is small), fp32 arithmetic with fp64 channel sums, fixed
// reduction order => run-to-run reproducible.  Bandwidth / latency bound elementwise + reduction work: no MFMA.

#include "bh_common.h"

namespace {

using bh::kBlock;

// Geometry shared by the three kernels: x is [B, C, HW] contiguous.  Wide: workgroup blockIdx.x owns channel blockIdx.x; narrow
// (B * HW <= kNarrow): one wavefront per channel, four channels per workgroup.
constexpr int kNarrow = 512;

struct ChannelWalk {
  int c, lane, lanes;
  bool active;
};

__device__ __forceinline__ ChannelWalk channel_of(int C, bool narrow) {
  ChannelWalk w;
  if (narrow) {
    w.c = blockIdx.x * bh::kWavesPerBlock + (threadIdx.x >> 6);
    w.lane = threadIdx.x & (bh::kWave - 1);
    w.lanes = bh::kWave;
  } else {
    w.c = blockIdx.x;
    w.lane = threadIdx.x;
    w.lanes = kBlock;
  }
  w.active = w.c < C;
  return w;
}

// per-channel sums of K doubles: wide = block_sum, narrow = wave_sum; result valid in lane 0 of the owner
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
                                                             const float* __restrict__ mean_inv, float* __restrict__ y, int B,
                                                             int C, int HW, int narrow) {
  const ChannelWalk w = channel_of(C, narrow != 0);
  if (!w.active) return;
  const float wc = weight ? weight[w.c] : 1.f;
  const float s = wc * inv_std[w.c];
  const float t = (bias ? bias[w.c] : 0.f) - wc * mean_inv[w.c];
  for (int b = 0; b < B; ++b) {
    const size_t base = ((size_t)b * C + w.c) * HW;
    if ((HW & 3) == 0) {
      const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x + base);
      float4* __restrict__ y4 = reinterpret_cast<float4*>(y + base);
      for (int i = w.lane; i < (HW >> 2); i += w.lanes) {
        const float4 q = x4[i];
        y4[i] = make_float4(fmaf(q.x, s, t), fmaf(q.y, s, t), fmaf(q.z, s, t), fmaf(q.w, s, t));
      }
    } else {
      for (int i = w.lane; i < HW; i += w.lanes) y[base + i] = fmaf(x[base + i], s, t);
    }
  }
}

__global__ __launch_bounds__(kBlock) void bn_eval_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                             const float* __restrict__ weight, const float* __restrict__ inv_std,
                                                             const float* __restrict__ mean_inv, float* __restrict__ gx,
                                                             float* __restrict__ gw, float* __restrict__ gb, int B, int C, int HW,
                                                             int narrow) {
  __shared__ double lds[bh::kWavesPerBlock * 2];
  const ChannelWalk w = channel_of(C, narrow != 0);
  double v[2] = {0.0, 0.0};  // sum gy, sum gy * x
  if (w.active) {
    const float s = (weight ? weight[w.c] : 1.f) * inv_std[w.c];
    for (int b = 0; b < B; ++b) {
      const size_t base = ((size_t)b * C + w.c) * HW;
      float a0 = 0.f, a1 = 0.f;
      int cnt = 0;
      if ((HW & 3) == 0) {
        const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gy + base);
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x + base);
        float4* __restrict__ o4 = reinterpret_cast<float4*>(gx + base);
        for (int i = w.lane; i < (HW >> 2); i += w.lanes) {
          const float4 g = g4[i], q = x4[i];
          if (gx) o4[i] = make_float4(g.x * s, g.y * s, g.z * s, g.w * s);
          a0 += (g.x + g.y) + (g.z + g.w);
          a1 = fmaf(g.x, q.x, a1);
          a1 = fmaf(g.y, q.y, a1);
          a1 = fmaf(g.z, q.z, a1);
          a1 = fmaf(g.w, q.w, a1);
          if (++cnt == 8) {  // at most 32 values per fp32 accumulator
            v[0] += (double)a0;
            v[1] += (double)a1;
            a0 = a1 = 0.f;
            cnt = 0;
          }
        }
      } else {
        for (int i = w.lane; i < HW; i += w.lanes) {
          const float g = gy[base + i];
          if (gx) gx[base + i] = g * s;
          a0 += g;
          a1 = fmaf(g, x[base + i], a1);
          if (++cnt == 32) {
            v[0] += (double)a0;
            v[1] += (double)a1;
            a0 = a1 = 0.f;
            cnt = 0;
          }
        }
      }
      v[0] += (double)a0;
      v[1] += (double)a1;
    }
  }
  channel_sum<2>(v, narrow != 0, lds);
  if (w.active && w.lane == 0) {
    if (gw) gw[w.c] = (float)((double)inv_std[w.c] * v[1] - (double)mean_inv[w.c] * v[0]);
    if (gb) gb[w.c] = (float)v[0];
  }
}

__global__ __launch_bounds__(kBlock) void bn_eval_bwd_bwd_kernel(const float* __restrict__ ggx, const float* __restrict__ ggw,
                                                                 const float* __restrict__ ggb, const float* __restrict__ gy,
                                                                 const float* __restrict__ x, const float* __restrict__ weight,
                                                                 const float* __restrict__ inv_std,
                                                                 const float* __restrict__ mean_inv, float* __restrict__ d_gy,
                                                                 float* __restrict__ d_x, float* __restrict__ d_w, int B, int C,
                                                                 int HW, int narrow) {
  __shared__ double lds[bh::kWavesPerBlock];
  const ChannelWalk w = channel_of(C, narrow != 0);
  double v[1] = {0.0};  // sum ggx * gy
  if (w.active) {
    const float inv = inv_std[w.c], mi = mean_inv[w.c];
    const float s = (weight ? weight[w.c] : 1.f) * inv;
    const float kw = ggw ? ggw[w.c] : 0.f;       // d objective / d gw_c
    const float kb = ggb ? ggb[w.c] : 0.f;       // d objective / d gb_c
    const float kwi = kw * inv, shift = kb - kw * mi;  // d_gy = ggx * s + kwi * x + shift ;  d_x = kwi * gy
    for (int b = 0; b < B; ++b) {
      const size_t base = ((size_t)b * C + w.c) * HW;
      float a0 = 0.f;
      int cnt = 0;
      for (int i = w.lane; i < HW; i += w.lanes) {
        const float g = gy[base + i];
        const float q = ggx ? ggx[base + i] : 0.f;
        if (d_gy) d_gy[base + i] = fmaf(q, s, fmaf(kwi, x[base + i], shift));
        if (d_x) d_x[base + i] = kwi * g;
        a0 = fmaf(q, g, a0);
        if (++cnt == 32) {
          v[0] += (double)a0;
          a0 = 0.f;
          cnt = 0;
        }
      }
      v[0] += (double)a0;
    }
  }
  channel_sum<1>(v, narrow != 0, lds);
  if (w.active && w.lane == 0 && d_w) d_w[w.c] = (float)((double)inv_std[w.c] * v[0]);
}

bool eval_bn_args_ok(const void* x, const void* inv_std, const void* mean_inv, int32_t B, int32_t C, int32_t HW) {
  return x != nullptr && inv_std != nullptr && mean_inv != nullptr && B > 0 && C > 0 && HW > 0 &&
         (int64_t)B * C * HW < ((int64_t)1 << 40);
}

int eval_bn_grid(int32_t B, int32_t C, int32_t HW, int& narrow) {
  narrow = ((int64_t)B * HW <= kNarrow) ? 1 : 0;
  return narrow ? (C + bh::kWavesPerBlock - 1) / bh::kWavesPerBlock : C;
}

}  // namespace

extern "C" {

int bh_bn_eval_fwd(const float* x, const float* weight, const float* bias, const float* inv_std, const float* mean_inv, float* y,
                   int32_t B, int32_t C, int32_t HW, void* stream) {
  if (!eval_bn_args_ok(x, inv_std, mean_inv, B, C, HW) || y == nullptr) return BH_EINVAL;
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) != 0) return BH_EINVAL;
  int narrow = 0;
  const int grid = eval_bn_grid(B, C, HW, narrow);
  hipLaunchKernelGGL(bn_eval_fwd_kernel, dim3(grid), dim3(kBlock), 0, bh::as_stream(stream), x, weight, bias, inv_std, mean_inv,
                     y, B, C, HW, narrow);
  return bh::launch_status();
}

int bh_bn_eval_bwd(const float* gy, const float* x, const float* weight, const float* inv_std, const float* mean_inv, float* gx,
                   float* gw, float* gb, int32_t B, int32_t C, int32_t HW, void* stream) {
  if (!eval_bn_args_ok(x, inv_std, mean_inv, B, C, HW) || gy == nullptr) return BH_EINVAL;
  if ((HW & 3) == 0 &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(gx)) & 15u) != 0)
    return BH_EINVAL;
  int narrow = 0;
  const int grid = eval_bn_grid(B, C, HW, narrow);
  hipLaunchKernelGGL(bn_eval_bwd_kernel, dim3(grid), dim3(kBlock), 0, bh::as_stream(stream), gy, x, weight, inv_std, mean_inv, gx,
                     gw, gb, B, C, HW, narrow);
  return bh::launch_status();
}

int bh_bn_eval_bwd_bwd(const float* ggx, const float* ggw, const float* ggb, const float* gy, const float* x, const float* weight,
                       const float* inv_std, const float* mean_inv, float* d_gy, float* d_x, float* d_w, int32_t B, int32_t C,
                       int32_t HW, void* stream) {
  if (!eval_bn_args_ok(x, inv_std, mean_inv, B, C, HW) || gy == nullptr) return BH_EINVAL;
  int narrow = 0;
  const int grid = eval_bn_grid(B, C, HW, narrow);
  hipLaunchKernelGGL(bn_eval_bwd_bwd_kernel, dim3(grid), dim3(kBlock), 0, bh::as_stream(stream), ggx, ggw, ggb, gy, x, weight,
                     inv_std, mean_inv, d_gy, d_x, d_w, B, C, HW, narrow);
  return bh::launch_status();
}

}  // extern "C"
