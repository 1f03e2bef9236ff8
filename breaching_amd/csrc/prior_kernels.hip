// Kernels C and D: image priors of the optimisation attack.
//
// C -- TotalVariation (+ optional L^p norm penalty): value AND analytic gradient in one stencil pass.
//      reference: breaching/attacks/auxiliaries/regularizers.py:103-153 (grouped 3x3 conv of forward differences with
//      zero padding, abs + eps, inner / outer exponents, mean) and :184-200 (scale / p * mean(x^p)).  The reference
//      pays a grouped MIOpen conv + ~5 elementwise launches + their autograd backward on a 0.6-4.8 MB tensor.
// D -- DeepInversion batch-norm statistics prior: per-channel mean / biased variance of a BN input compared with the
//      running statistics.  Restated from the mathematical definition only (the reference file
//      auxiliaries/deepinversion.py is NVIDIA-NC licensed; no code was taken from it):
//          r = || running_var - var ||_2 + || running_mean - mean ||_2            (deepinversion.py:93-101)
//      and d r / d x[b,c,hw] = A_c + B_c * x[b,c,hw].
//
// All of these are HBM/latency bound streaming kernels: coalesced (16-byte where alignment allows) loads, wave64
// shuffle + LDS block reductions, fp64 partial sums combined in a fixed order.

#include "bh_common.h"

namespace {

using bh::kBlock;

// ---------------------------------------------------------------------------------------------------------------
// Kernel C
// ---------------------------------------------------------------------------------------------------------------

struct Plane7 {
  float c, s, e, n, ne, w, sw;  // centre, south (i+1), east (j+1), north (i-1), north-east, west (j-1), south-west
};

__device__ __forceinline__ Plane7 load7(const float* __restrict__ u, int i, int j, int H, int W) {
  Plane7 p;
  const int64_t o = (int64_t)i * W + j;
  const bool has_s = i + 1 < H, has_e = j + 1 < W, has_n = i > 0, has_w = j > 0;
  p.c = u[o];
  p.s = has_s ? u[o + W] : 0.f;  // zero padding of the conv (regularizers.py:142-144, padding=1)
  p.e = has_e ? u[o + 1] : 0.f;
  p.n = has_n ? u[o - W] : 0.f;
  p.ne = (has_n && has_e) ? u[o - W + 1] : 0.f;
  p.w = has_w ? u[o - 1] : 0.f;
  p.sw = (has_w && has_s) ? u[o + W - 1] : 0.f;
  return p;
}

__device__ __forceinline__ Plane7 sub7(const Plane7& a, const Plane7& b) {
  return Plane7{a.c - b.c, a.s - b.s, a.e - b.e, a.n - b.n, a.ne - b.ne, a.w - b.w, a.sw - b.sw};
}

// x^e for x > 0 with the exponents the shipped configs use resolved without powf (wave-uniform branches):
// inner_exp 1 / 2, outer_exp 1 / 0.5 (modern.yaml, legacy.yaml) and the derivative exponents e-1 they induce.
__device__ __forceinline__ float pow_pos(float x, float e) {
  if (e == 1.f) return x;
  if (e == 2.f) return x * x;
  if (e == 0.5f) return sqrtf(x);
  if (e == 0.f) return 1.f;
  if (e == -0.5f) return 1.f / sqrtf(x);
  return powf(x, e);
}

// value term and the two partial derivatives of ((|dv|+eps)^p + (|dh|+eps)^p)^q
template <bool PQ1>
__device__ __forceinline__ void tv_term(float dv, float dh, float p, float q, float eps, float& f, float& fv, float& fh) {
  const float a = fabsf(dv) + eps, b = fabsf(dh) + eps;
  if constexpr (PQ1) {
    f = a + b;
    fv = bh::sgnf(dv);
    fh = bh::sgnf(dh);
  } else {
    const float ap = pow_pos(a, p), bp = pow_pos(b, p);
    const float S = ap + bp;
    f = pow_pos(S, q);
    const float common = q * pow_pos(S, q - 1.f) * p;
    fv = common * pow_pos(a, p - 1.f) * bh::sgnf(dv);
    fh = common * pow_pos(b, p - 1.f) * bh::sgnf(dh);
  }
}

// d/du(i,j) of sum_{i,j} f, plus the local value term f(i,j)
template <bool PQ1>
__device__ __forceinline__ float plane_grad(const Plane7& u, bool has_n, bool has_w, float p, float q, float eps,
                                            float& value) {
  float f, fv, fh, t, gv_n, gh_w, unused;
  tv_term<PQ1>(u.s - u.c, u.e - u.c, p, q, eps, f, fv, fh);  // differences anchored at (i, j)
  value += f;
  float g = -fv - fh;
  if (has_n) {  // (i-1, j): its vertical difference touches u(i,j) with +1
    tv_term<PQ1>(u.c - u.n, u.ne - u.n, p, q, eps, t, gv_n, unused);
    g += gv_n;
  }
  if (has_w) {  // (i, j-1): its horizontal difference touches u(i,j) with +1
    tv_term<PQ1>(u.sw - u.w, u.c - u.w, p, q, eps, t, unused, gh_w);
    g += gh_w;
  }
  return g;
}

template <bool PQ1, bool OPP>
__global__ __launch_bounds__(kBlock) void tv_norm_kernel(const float* __restrict__ x, int B, int H, int W, float tv_coef,
                                                         float p, float q, float eps, float norm_val_coef,
                                                         float norm_grad_coef, float norm_p, float* __restrict__ grad,
                                                         double* __restrict__ partials) {
  __shared__ double lds[bh::kWavesPerBlock * 2];
  const int64_t plane = (int64_t)H * W;
  const int64_t total = (int64_t)B * plane;
  double acc_tv = 0.0, acc_norm = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kBlock) {
    const int b = (int)(idx / plane);
    const int64_t rem = idx - (int64_t)b * plane;
    const int i = (int)(rem / W), j = (int)(rem - (int64_t)i * W);
    const float* __restrict__ xb = x + (int64_t)b * 3 * plane;
    const Plane7 r = load7(xb, i, j, H, W), g = load7(xb + plane, i, j, H, W), bl = load7(xb + 2 * plane, i, j, H, W);
    const bool has_n = i > 0, has_w = j > 0;
    float value = 0.f;
    float gr = plane_grad<PQ1>(r, has_n, has_w, p, q, eps, value);
    float gg = plane_grad<PQ1>(g, has_n, has_w, p, q, eps, value);
    float gb = plane_grad<PQ1>(bl, has_n, has_w, p, q, eps, value);
    if constexpr (OPP) {  // regularizers.py:132-141: planes R-G, R-B, G-B appended
      const float o1 = plane_grad<PQ1>(sub7(r, g), has_n, has_w, p, q, eps, value);
      const float o2 = plane_grad<PQ1>(sub7(r, bl), has_n, has_w, p, q, eps, value);
      const float o3 = plane_grad<PQ1>(sub7(g, bl), has_n, has_w, p, q, eps, value);
      gr += o1 + o2;
      gg += o3 - o1;
      gb -= o2 + o3;
    }
    gr *= tv_coef;
    gg *= tv_coef;
    gb *= tv_coef;
    acc_tv += (double)value;
    if (norm_grad_coef != 0.f) {
      float xp0, xp1, xp2;
      if (norm_p == 2.f) {
        xp0 = r.c * r.c, xp1 = g.c * g.c, xp2 = bl.c * bl.c;
        gr = fmaf(norm_grad_coef, r.c, gr);
        gg = fmaf(norm_grad_coef, g.c, gg);
        gb = fmaf(norm_grad_coef, bl.c, gb);
      } else {
        xp0 = powf(r.c, norm_p), xp1 = powf(g.c, norm_p), xp2 = powf(bl.c, norm_p);
        gr += norm_grad_coef * powf(r.c, norm_p - 1.f);
        gg += norm_grad_coef * powf(g.c, norm_p - 1.f);
        gb += norm_grad_coef * powf(bl.c, norm_p - 1.f);
      }
      acc_norm += (double)xp0 + (double)xp1 + (double)xp2;
    }
    float* __restrict__ gbase = grad + (int64_t)b * 3 * plane + rem;
    gbase[0] = gr;
    gbase[plane] = gg;
    gbase[2 * plane] = gb;
  }
  double v[2] = {acc_tv, acc_norm};
  bh::block_sum<2>(v, lds);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * BH_PRIOR_PARTIAL_STRIDE + 0] = v[0] * (double)tv_coef;
    partials[blockIdx.x * BH_PRIOR_PARTIAL_STRIDE + 1] = v[1] * (double)norm_val_coef;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel D
// ---------------------------------------------------------------------------------------------------------------

constexpr int64_t kBnTile = 4096;  // elements of one (b, c) plane handled per inner step

// grid = (C, S).  Workgroup (c, s) sums slab s of channel c: the (b, tile) work items [w0, w1) of that channel.
__global__ __launch_bounds__(kBlock) void bnstat_sums_kernel(const float* __restrict__ x, int B, int C, int64_t HW, int S,
                                                             double* __restrict__ sums) {
  __shared__ double lds[bh::kWavesPerBlock * 2];
  const int c = blockIdx.x, s = blockIdx.y;
  const int64_t tiles = (HW + kBnTile - 1) / kBnTile;
  const int64_t items = (int64_t)B * tiles;
  const int64_t w0 = items * s / S, w1 = items * (s + 1) / S;
  float a0 = 0.f, a1 = 0.f;
  double d0 = 0.0, d1 = 0.0;
  const bool vec = (HW & 3) == 0;
  if (tiles == 1 && HW < 1024) {
    // small planes (late ResNet stages: 7x7, 14x14): walk (b, hw) as one flat index so all 256 lanes stay busy
    const int64_t n = (w1 - w0) * HW;
    int cnt = 0;
    for (int64_t idx = threadIdx.x; idx < n; idx += kBlock) {
      const int64_t b = w0 + idx / HW, off = idx % HW;
      const float q = x[((int64_t)b * C + c) * HW + off];
      a0 += q;
      a1 = fmaf(q, q, a1);
      if (++cnt == 16) {
        d0 += (double)a0;
        d1 += (double)a1;
        a0 = a1 = 0.f;
        cnt = 0;
      }
    }
    d0 += (double)a0;
    d1 += (double)a1;
  } else {
    for (int64_t w = w0; w < w1; ++w) {
      const int64_t b = w / tiles, t = w - b * tiles;
      const int64_t start = t * kBnTile;
      const int64_t len = (HW - start) < kBnTile ? (HW - start) : kBnTile;
      const float* __restrict__ p = x + ((int64_t)b * C + c) * HW + start;
      if (vec && len == kBnTile) {
        // full tile: four 16-byte loads per thread in flight before the first use
        const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
        float4 q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = p4[threadIdx.x + k * kBlock];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          a0 += (q[k].x + q[k].y) + (q[k].z + q[k].w);
          a1 = fmaf(q[k].x, q[k].x, a1);
          a1 = fmaf(q[k].y, q[k].y, a1);
          a1 = fmaf(q[k].z, q[k].z, a1);
          a1 = fmaf(q[k].w, q[k].w, a1);
        }
      } else if (vec) {
        const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
        const int n4 = (int)(len >> 2);
        for (int i = threadIdx.x; i < n4; i += kBlock) {
          const float4 q = p4[i];
          a0 += (q.x + q.y) + (q.z + q.w);
          a1 = fmaf(q.x, q.x, a1);
          a1 = fmaf(q.y, q.y, a1);
          a1 = fmaf(q.z, q.z, a1);
          a1 = fmaf(q.w, q.w, a1);
        }
      } else {
        for (int i = threadIdx.x; i < (int)len; i += kBlock) {
          const float q = p[i];
          a0 += q;
          a1 = fmaf(q, q, a1);
        }
      }
      // spill the fp32 running sums into fp64 once per tile: bounds the fp32 accumulation length to 16 values
      d0 += (double)a0;
      d1 += (double)a1;
      a0 = 0.f;
      a1 = 0.f;
    }
  }
  double v[2] = {d0, d1};
  bh::block_sum<2>(v, lds);
  if (threadIdx.x == 0) {
    sums[((int64_t)c * S + s) * 2 + 0] = v[0];
    sums[((int64_t)c * S + s) * 2 + 1] = v[1];
  }
}

// Single workgroup; thread t owns channels t, t+256, ...
__global__ __launch_bounds__(kBlock) void bnstat_finalize_kernel(const double* __restrict__ sums, int B, int C, int64_t HW,
                                                                 int S, const float* __restrict__ running_mean,
                                                                 const float* __restrict__ running_var,
                                                                 float* __restrict__ value, float* __restrict__ coef,
                                                                 double* __restrict__ scratch /* [2*C] mean,var */) {
  __shared__ double lds[bh::kWavesPerBlock * 2];
  __shared__ double norms[2];
  const double n = (double)B * (double)HW;
  double v[2] = {0.0, 0.0};  // sum (rv - var)^2, sum (rm - mean)^2
  for (int c = threadIdx.x; c < C; c += kBlock) {
    double s0 = 0.0, s1 = 0.0;
    for (int s = 0; s < S; ++s) {
      s0 += sums[((int64_t)c * S + s) * 2 + 0];
      s1 += sums[((int64_t)c * S + s) * 2 + 1];
    }
    const double mean = s0 / n;
    double var = s1 / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    scratch[2 * c] = mean;
    scratch[2 * c + 1] = var;
    const double dvv = (double)running_var[c] - var, dm = (double)running_mean[c] - mean;
    v[0] += dvv * dvv;
    v[1] += dm * dm;
  }
  bh::block_sum<2>(v, lds);
  if (threadIdx.x == 0) {
    norms[0] = sqrt(v[0]);
    norms[1] = sqrt(v[1]);
    value[0] = (float)(norms[0] + norms[1]);
  }
  __syncthreads();
  const double nv = norms[0], nm = norms[1];
  for (int c = threadIdx.x; c < C; c += kBlock) {
    const double mean = scratch[2 * c], var = scratch[2 * c + 1];
    const double pv = nv > 0.0 ? -((double)running_var[c] - var) / nv : 0.0;   // d r / d var_c
    const double pm = nm > 0.0 ? -((double)running_mean[c] - mean) / nm : 0.0; // d r / d mean_c
    coef[2 * c] = (float)((pm - 2.0 * pv * mean) / n);
    coef[2 * c + 1] = (float)(2.0 * pv / n);
  }
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void bnstat_bwd_kernel(const float* __restrict__ x, int C, int64_t HW, int64_t total,
                                                            const float* __restrict__ coef, const float* __restrict__ gout,
                                                            float* __restrict__ grad) {
  const float g = gout ? gout[0] : 1.f;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  if constexpr (VEC) {
    const int64_t total4 = total >> 2, hw4 = HW >> 2;
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    float4* __restrict__ g4 = reinterpret_cast<float4*>(grad);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total4; i += stride) {
      const int c = (int)((i / hw4) % C);
      const float a = g * coef[2 * c], b = g * coef[2 * c + 1];
      const float4 q = x4[i];
      g4[i] = make_float4(fmaf(b, q.x, a), fmaf(b, q.y, a), fmaf(b, q.z, a), fmaf(b, q.w, a));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += stride) {
      const int c = (int)((i / HW) % C);
      grad[i] = fmaf(g * coef[2 * c + 1], x[i], g * coef[2 * c]);
    }
  }
}

}  // namespace

extern "C" {

int bh_prior_tv_norm(const float* x, int32_t B, int32_t H, int32_t W, float tv_scale, float inner_exp, float outer_exp,
                     float eps, int32_t double_opponents, float norm_scale, float norm_p, float* grad_out,
                     double* partials_dev, void* stream) {
  if (x == nullptr || grad_out == nullptr || partials_dev == nullptr || B <= 0 || H <= 0 || W <= 0) return BH_EINVAL;
  if (norm_scale != 0.f && norm_p == 0.f) return BH_EINVAL;
  const int64_t pixels = (int64_t)B * H * W;
  int64_t blocks = (pixels + kBlock - 1) / kBlock;
  const int grid = (int)(blocks < BH_PRIOR_MAX_GRID ? blocks : BH_PRIOR_MAX_GRID);
  const int groups = double_opponents ? 6 : 3;
  // mean over [B, groups, H, W] (regularizers.py:147) / mean over [B, 3, H, W] (:197)
  const float tv_coef = (float)((double)tv_scale / ((double)pixels * groups));
  const float norm_val_coef = norm_scale != 0.f ? (float)((double)norm_scale / (double)norm_p / ((double)pixels * 3)) : 0.f;
  const float norm_grad_coef = norm_scale != 0.f ? (float)((double)norm_scale / ((double)pixels * 3)) : 0.f;
  const bool pq1 = inner_exp == 1.f && outer_exp == 1.f;
  hipStream_t st = bh::as_stream(stream);
#define BH_TV_LAUNCH(PQ1, OPP)                                                                                         \
  hipLaunchKernelGGL((tv_norm_kernel<PQ1, OPP>), dim3(grid), dim3(kBlock), 0, st, x, B, H, W, tv_coef, inner_exp,     \
                     outer_exp, eps, norm_val_coef, norm_grad_coef, norm_p, grad_out, partials_dev)
  if (pq1 && !double_opponents) BH_TV_LAUNCH(true, false);
  else if (pq1) BH_TV_LAUNCH(true, true);
  else if (!double_opponents) BH_TV_LAUNCH(false, false);
  else BH_TV_LAUNCH(false, true);
#undef BH_TV_LAUNCH
  const int rc = bh::launch_status();
  return rc != 0 ? rc : grid;
}

int32_t bh_bnstat_slabs(int32_t B, int32_t C, int64_t HW) {
  if (B <= 0 || C <= 0 || HW <= 0) return BH_EINVAL;
  const int64_t items = (int64_t)B * ((HW + kBnTile - 1) / kBnTile);
  int64_t want = (2048 + C - 1) / C;  // aim for >= 2048 workgroups in total (8 per CU)
  if (want > items) want = items;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  return (int32_t)want;
}

int bh_bnstat_sums(const float* x, int32_t B, int32_t C, int64_t HW, double* sums_dev, void* stream) {
  if (x == nullptr || sums_dev == nullptr) return BH_EINVAL;
  const int S = bh_bnstat_slabs(B, C, HW);
  if (S <= 0) return BH_EINVAL;
  if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) != 0) return BH_EINVAL;
  hipLaunchKernelGGL(bnstat_sums_kernel, dim3(C, S), dim3(kBlock), 0, bh::as_stream(stream), x, B, C, HW, S, sums_dev);
  const int rc = bh::launch_status();
  return rc != 0 ? rc : S;
}

int bh_bnstat_finalize(const double* sums_dev, int32_t B, int32_t C, int64_t HW, const float* running_mean,
                       const float* running_var, float* value_dev, float* coef_dev, double* scratch_dev, void* stream) {
  if (sums_dev == nullptr || running_mean == nullptr || running_var == nullptr || value_dev == nullptr ||
      coef_dev == nullptr || scratch_dev == nullptr)
    return BH_EINVAL;
  const int S = bh_bnstat_slabs(B, C, HW);
  if (S <= 0) return BH_EINVAL;
  hipLaunchKernelGGL(bnstat_finalize_kernel, dim3(1), dim3(kBlock), 0, bh::as_stream(stream), sums_dev, B, C, HW, S,
                     running_mean, running_var, value_dev, coef_dev, scratch_dev);
  return bh::launch_status();
}

int bh_bnstat_bwd(const float* x, int32_t B, int32_t C, int64_t HW, const float* coef_dev, const float* gout_dev,
                  float* grad_x, void* stream) {
  if (x == nullptr || coef_dev == nullptr || grad_x == nullptr || B <= 0 || C <= 0 || HW <= 0) return BH_EINVAL;
  const int64_t total = (int64_t)B * C * HW;
  const bool vec = (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15u) == 0 &&
                   (reinterpret_cast<uintptr_t>(grad_x) & 15u) == 0;
  const int64_t work = vec ? (total >> 2) : total;
  int64_t blocks = (work + kBlock - 1) / kBlock;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = bh::as_stream(stream);
  if (vec)
    hipLaunchKernelGGL(bnstat_bwd_kernel<true>, dim3((int)blocks), dim3(kBlock), 0, st, x, C, HW, total, coef_dev,
                       gout_dev, grad_x);
  else
    hipLaunchKernelGGL(bnstat_bwd_kernel<false>, dim3((int)blocks), dim3(kBlock), 0, st, x, C, HW, total, coef_dev,
                       gout_dev, grad_x);
  return bh::launch_status();
}

}  // extern "C"
