// LayerNorm (over the last dimension, elementwise affine) of the attacker's private model copy as one or two launches per
// autograd order -- kernel F.  The text attacks (tag.yaml: BERT-base, 26 LayerNorms) differentiate every LayerNorm twice per
// iteration (objectives.py:40-46 under create_graph=True, then optimization_with_label_attack.py:168-174); PyTorch decomposes
// the derivative of native_layer_norm_backward into ~83 elementwise / reduction launches per layer -- more than half of the
// ~4000 launches of a BERT-base iteration, all launch latency at 32 x 768 elements.
//
//   x_hat = (x - mean_r) * rstd_r ;  y = x_hat * gamma + beta                                      (rows r = all leading dims)
//   backward:  g = gy * gamma ;  a_r = M(g) ;  b_r = M(g x_hat) ;  gx = rstd (g - a - x_hat b)     (M = mean over the row)
//              ggamma = sum_r gy x_hat ;  gbeta = sum_r gy
//   backward of the backward, for incoming (u = d/d gx, s = d/d ggamma, t = d/d gbeta), with P v = v - M(v) - x_hat M(v x_hat),
//   w = P g, q = s gy:
//              d_gy    = gamma rstd P u + s x_hat + t
//              d_x     = -rstd^2 ( M(u w) x_hat + b P u + M(u x_hat) w ) + rstd P q
//              d_gamma = sum_r gy rstd P u
//   (checked against torch's fp64 layer_norm through both orders, tests/test_gpu_kernels.py).
// Row kernels: one wavefront per row, row sums in fp64 by xor-shuffles.  Column kernels (sums over rows): 64 columns x 4 row
// phases per workgroup, combined through LDS in a fixed order.  fp32 arithmetic, no atomics: run-to-run reproducible.

#include "bh_common.h"

namespace {

using bh::kBlock;

__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
  for (int off = bh::kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, bh::kWave);
  return v;
}

// Loads per operand that a lane issues before it uses the first one.  A 32 x 768 activation is 32 wavefronts in 8 workgroups:
// nothing hides a load's latency but the lane's own next loads, and the plain load-use loops of round 3 made a row kernel
// 2 x 12 serial memory round trips (instruction census, profiles/r4_kernel_isa_census.txt); staged, 2 x 3.  Each lane still visits
// its elements in ascending order, so every sum is accumulated in the same order and the results are bit-identical.
constexpr int kLnStage = 4;

// i = first, first + stride, ... < end: `load(i)` for kLnStage positions, then `use(i, value)` for the same positions in order
template <typename V, typename Load, typename Use>
__device__ __forceinline__ void staged_walk(int first, int end, int stride, Load load, Use use) {
  for (int i0 = first; i0 < end; i0 += stride * kLnStage) {
    V v[kLnStage];
#pragma unroll
    for (int k = 0; k < kLnStage; ++k) {
      const int i = i0 + k * stride;
      v[k] = load(i < end ? i : i0);  // past the end: re-read the lane's first position, unused
    }
#pragma unroll
    for (int k = 0; k < kLnStage; ++k) {
      const int i = i0 + k * stride;
      if (i < end) use(i, v[k]);
    }
  }
}

struct Ln3 {
  float a, b, c;
};
struct Ln6 {
  float gy, gm, x, u, s, t;
};

__global__ __launch_bounds__(kBlock) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int R, int D,
                                                        float eps) {
  const int r = blockIdx.x * bh::kWavesPerBlock + (threadIdx.x >> 6), lane = threadIdx.x & (bh::kWave - 1);
  if (r >= R) return;
  const float* __restrict__ xr = x + (size_t)r * D;
  double s0 = 0.0, s1 = 0.0;
  staged_walk<float>(lane, D, bh::kWave, [&](int i) { return xr[i]; },
                     [&](int, float xv) {
                       const double v = (double)xv;
                       s0 += v;
                       s1 += v * v;
                     });
  s0 = wave_allsum(s0);
  s1 = wave_allsum(s1);
  const double mu = s0 / D;
  double var = s1 / D - mu * mu;
  var = var < 0.0 ? 0.0 : var;
  const float m = (float)mu, rs = (float)(1.0 / sqrt(var + (double)eps));
  if (lane == 0) {
    mean[r] = m;
    rstd[r] = rs;
  }
  float* __restrict__ yr = y + (size_t)r * D;
  // A missing operand reads the row instead and is ignored after the load (`absent`, below): a branch around an optional load
  // lets the compiler sink the first use into it, and with it a wait for the load, in the middle of the load phase.
  const float* gp = gamma ? gamma : xr;
  const float* bp = beta ? beta : xr;
  staged_walk<Ln3>(lane, D, bh::kWave,
                   [&](int i) {
                     Ln3 v;
                     v.a = xr[i];
                     v.b = gp[i];
                     v.c = bp[i];
                     return v;
                   },
                   [&](int i, const Ln3& v) {
                     const float xh = (v.a - m) * rs;
                     yr[i] = fmaf(xh, gamma ? v.b : 1.f, beta ? v.c : 0.f);
                   });
}

__global__ __launch_bounds__(kBlock) void ln_bwd_row_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ gx, int R, int D) {
  const int r = blockIdx.x * bh::kWavesPerBlock + (threadIdx.x >> 6), lane = threadIdx.x & (bh::kWave - 1);
  if (r >= R) return;
  const size_t base = (size_t)r * D;
  const float m = mean[r], rs = rstd[r];
  double s0 = 0.0, s1 = 0.0;  // sum g, sum g x_hat
  const float* gp = gamma ? gamma : x + base;  // a missing operand reads the row instead and is ignored (see ln_fwd_kernel)
  auto load = [&](int i) {
    Ln3 v;
    v.a = gy[base + i];
    v.b = gp[i];
    v.c = x[base + i];
    return v;
  };
  staged_walk<Ln3>(lane, D, bh::kWave, load, [&](int, const Ln3& v) {
    const float g = v.a * (gamma ? v.b : 1.f);
    const float xh = (v.c - m) * rs;
    s0 += (double)g;
    s1 += (double)g * (double)xh;
  });
  const float a = (float)(wave_allsum(s0) / D), b = (float)(wave_allsum(s1) / D);
  staged_walk<Ln3>(lane, D, bh::kWave, load, [&](int i, const Ln3& v) {
    const float g = v.a * (gamma ? v.b : 1.f);
    const float xh = (v.c - m) * rs;
    gx[base + i] = rs * (g - a - xh * b);
  });
}

// sums over rows for 64 columns per workgroup: thread (col = tid & 63, phase = tid >> 6) walks rows phase, phase + 4, ...
// (`load(r, col)` staged kLnStage rows ahead of `use(r, value, acc)`, rows in ascending order)
template <typename V, typename Load, typename Use>
__device__ __forceinline__ void column_sums(int R, int D, double (&acc)[2], Load load, Use use, float* out0, float* out1) {
  __shared__ double lds[kBlock * 2];
  const int col = blockIdx.x * bh::kWave + (threadIdx.x & (bh::kWave - 1)), phase = threadIdx.x >> 6;
  acc[0] = acc[1] = 0.0;
  if (col < D)
    staged_walk<V>(phase, R, bh::kWavesPerBlock, [&](int r) { return load(r, col); }, [&](int r, const V& v) { use(r, v, acc); });
  lds[threadIdx.x * 2] = acc[0];
  lds[threadIdx.x * 2 + 1] = acc[1];
  __syncthreads();
  if (phase == 0 && col < D) {
    double v0 = 0.0, v1 = 0.0;
    for (int p = 0; p < bh::kWavesPerBlock; ++p) {  // fixed order
      v0 += lds[(p * bh::kWave + (threadIdx.x & (bh::kWave - 1))) * 2];
      v1 += lds[(p * bh::kWave + (threadIdx.x & (bh::kWave - 1))) * 2 + 1];
    }
    if (out0) out0[col] = (float)v0;
    if (out1) out1[col] = (float)v1;
  }
}

struct LnCol {
  float gy, x, mean, rstd, u, m0, m1;
};

__global__ __launch_bounds__(kBlock) void ln_bwd_col_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            float* __restrict__ ggamma, float* __restrict__ gbeta, int R, int D) {
  double acc[2];
  column_sums<LnCol>(R, D, acc,
                     [&](int r, int col) {
                       LnCol v;
                       v.gy = gy[(size_t)r * D + col];
                       v.x = x[(size_t)r * D + col];
                       v.mean = mean[r];
                       v.rstd = rstd[r];
                       v.u = v.m0 = v.m1 = 0.f;
                       return v;
                     },
                     [&](int, const LnCol& v, double (&a)[2]) {
                       const float g = v.gy;
                       const float xh = (v.x - v.mean) * v.rstd;
                       a[0] += (double)g * (double)xh;
                       a[1] += (double)g;
                     },
                     ggamma, gbeta);
}

// row_scalars[r] = (M(u), M(u x_hat)): what the column kernel needs to rebuild P u
__global__ __launch_bounds__(kBlock) void ln_bwd_bwd_row_kernel(const float* __restrict__ u, const float* __restrict__ s,
                                                                const float* __restrict__ t, const float* __restrict__ gy,
                                                                const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                float* __restrict__ d_gy, float* __restrict__ d_x,
                                                                float* __restrict__ row_scalars, int R, int D) {
  const int r = blockIdx.x * bh::kWavesPerBlock + (threadIdx.x >> 6), lane = threadIdx.x & (bh::kWave - 1);
  if (r >= R) return;
  const size_t base = (size_t)r * D;
  const float m = mean[r], rs = rstd[r];
  double su = 0.0, sux = 0.0, sg = 0.0, sgx = 0.0, sug = 0.0, sq = 0.0, sqx = 0.0;
  // a missing operand reads the row of x instead and is replaced by its neutral value after the load (see ln_fwd_kernel)
  const float* gp = gamma ? gamma : x + base;
  const float* up = u ? u + base : x + base;
  const float* sp = s ? s : x + base;
  const float* tp = t ? t : x + base;
  auto neutral = [&](Ln6 v) {
    v.gm = gamma ? v.gm : 1.f;
    v.u = u ? v.u : 0.f;
    v.s = s ? v.s : 0.f;
    v.t = t ? v.t : 0.f;
    return v;
  };
  auto load_sums = [&](int i) {  // first pass: everything but t
    Ln6 v;
    v.gy = gy[base + i];
    v.x = x[base + i];
    v.gm = gp[i];
    v.u = up[i];
    v.s = sp[i];
    v.t = 0.f;
    return v;
  };
  auto load = [&](int i) {
    Ln6 v = load_sums(i);
    v.t = tp[i];
    return v;
  };
  staged_walk<Ln6>(lane, D, bh::kWave, load_sums, [&](int, const Ln6& loaded) {
    const Ln6 v = neutral(loaded);
    const float gyv = v.gy;
    const float g = gyv * v.gm;
    const float xh = (v.x - m) * rs;
    const float uv = v.u;
    const float q = s ? v.s * gyv : 0.f;
    su += (double)uv;
    sux += (double)uv * (double)xh;
    sg += (double)g;
    sgx += (double)g * (double)xh;
    sug += (double)uv * (double)g;
    sq += (double)q;
    sqx += (double)q * (double)xh;
  });
  const double inv = 1.0 / D;
  const double Mu = wave_allsum(su) * inv, Mux = wave_allsum(sux) * inv, a = wave_allsum(sg) * inv, b = wave_allsum(sgx) * inv;
  const double Mug = wave_allsum(sug) * inv, Mq = wave_allsum(sq) * inv, Mqx = wave_allsum(sqx) * inv;
  const float Muw = (float)(Mug - a * Mu - b * Mux);
  const float fMu = (float)Mu, fMux = (float)Mux, fa = (float)a, fb = (float)b, fMq = (float)Mq, fMqx = (float)Mqx;
  if (lane == 0 && row_scalars) {
    row_scalars[2 * r] = fMu;
    row_scalars[2 * r + 1] = fMux;
  }
  const float rs2 = rs * rs;
  staged_walk<Ln6>(lane, D, bh::kWave, load, [&](int i, const Ln6& loaded) {
    const Ln6 v = neutral(loaded);
    const float gyv = v.gy;
    const float gm = v.gm;
    const float g = gyv * gm;
    const float xh = (v.x - m) * rs;
    const float uv = v.u;
    const float sv = v.s;
    const float q = sv * gyv;
    const float Pu = uv - fMu - xh * fMux;
    const float w = g - fa - xh * fb;
    if (d_gy) d_gy[base + i] = gm * rs * Pu + sv * xh + v.t;
    if (d_x) d_x[base + i] = -rs2 * (Muw * xh + fb * Pu + fMux * w) + rs * (q - fMq - xh * fMqx);
  });
}

__global__ __launch_bounds__(kBlock) void ln_bwd_bwd_col_kernel(const float* __restrict__ u, const float* __restrict__ gy,
                                                                const float* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ row_scalars, float* __restrict__ d_gamma,
                                                                int R, int D) {
  double acc[2];
  column_sums<LnCol>(R, D, acc,
                     [&](int r, int col) {
                       LnCol v;
                       v.gy = gy[(size_t)r * D + col];
                       v.x = x[(size_t)r * D + col];
                       v.u = u[(size_t)r * D + col];
                       v.mean = mean[r];
                       v.rstd = rstd[r];
                       v.m0 = row_scalars[2 * r];
                       v.m1 = row_scalars[2 * r + 1];
                       return v;
                     },
                     [&](int, const LnCol& v, double (&a)[2]) {
                       const float rs = v.rstd;
                       const float xh = (v.x - v.mean) * rs;
                       const float Pu = v.u - v.m0 - xh * v.m1;
                       a[0] += (double)v.gy * (double)(rs * Pu);
                     },
                     d_gamma, static_cast<float*>(nullptr));
}

bool ln_args_ok(const void* x, int32_t R, int32_t D) { return x != nullptr && R > 0 && D > 0 && (int64_t)R * D < ((int64_t)1 << 40); }

}  // namespace

extern "C" {

int bh_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int32_t R, int32_t D,
              float eps, void* stream) {
  if (!ln_args_ok(x, R, D) || y == nullptr || mean == nullptr || rstd == nullptr) return BH_EINVAL;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((R + bh::kWavesPerBlock - 1) / bh::kWavesPerBlock), dim3(kBlock), 0, bh::as_stream(stream), x,
                     gamma, beta, y, mean, rstd, R, D, eps);
  return bh::launch_status();
}

int bh_ln_bwd(const float* gy, const float* x, const float* gamma, const float* mean, const float* rstd, float* gx, float* ggamma,
              float* gbeta, int32_t R, int32_t D, void* stream) {
  if (!ln_args_ok(x, R, D) || gy == nullptr || mean == nullptr || rstd == nullptr) return BH_EINVAL;
  hipStream_t st = bh::as_stream(stream);
  if (gx != nullptr)
    hipLaunchKernelGGL(ln_bwd_row_kernel, dim3((R + bh::kWavesPerBlock - 1) / bh::kWavesPerBlock), dim3(kBlock), 0, st, gy, x, gamma,
                       mean, rstd, gx, R, D);
  if (ggamma != nullptr || gbeta != nullptr)
    hipLaunchKernelGGL(ln_bwd_col_kernel, dim3((D + bh::kWave - 1) / bh::kWave), dim3(kBlock), 0, st, gy, x, mean, rstd, ggamma,
                       gbeta, R, D);
  return bh::launch_status();
}

int bh_ln_bwd_bwd(const float* u, const float* s, const float* t, const float* gy, const float* x, const float* gamma,
                  const float* mean, const float* rstd, float* d_gy, float* d_x, float* d_gamma, float* row_scalars, int32_t R,
                  int32_t D, void* stream) {
  if (!ln_args_ok(x, R, D) || gy == nullptr || mean == nullptr || rstd == nullptr) return BH_EINVAL;
  if (d_gamma != nullptr && u != nullptr && row_scalars == nullptr) return BH_EINVAL;
  hipStream_t st = bh::as_stream(stream);
  hipLaunchKernelGGL(ln_bwd_bwd_row_kernel, dim3((R + bh::kWavesPerBlock - 1) / bh::kWavesPerBlock), dim3(kBlock), 0, st, u, s, t,
                     gy, x, gamma, mean, rstd, d_gy, d_x, row_scalars, R, D);
  if (d_gamma != nullptr && u != nullptr)
    hipLaunchKernelGGL(ln_bwd_bwd_col_kernel, dim3((D + bh::kWave - 1) / bh::kWave), dim3(kBlock), 0, st, u, gy, x, mean, rstd,
                       row_scalars, d_gamma, R, D);
  return bh::launch_status();
}

}  // extern "C"
