// Kernels C and D: image priors of the optimisation attack.
//
// C -- TotalVariation (+ optional L^p norm penalty): value AND analytic gradient in one stencil pass.
//      reference: breaching/attacks/auxiliaries/regularizers.py:103-153 (grouped 3x3 conv of forward differences with
//      zero padding, abs + eps, inner / outer exponents, mean) and :184-200 (scale / p * mean(x^p)).  The reference
//      pays a grouped MIOpen conv + ~5 elementwise launches + their autograd backward on a 0.6-4.8 MB tensor.
// D -- DeepInversion batch-norm statistics prior: per-channel mean / biased variance of every BN input of the model
//      compared with the running statistics, all layers per launch.  Restated from the mathematical definition only (the reference file
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

// Four horizontally adjacent pixels per thread (W a multiple of 4, 16-byte aligned planes): per colour plane the centre, south and
// north rows arrive as one 16-byte load each plus the four edge scalars (west, east, south-west, north-east of the quad) -- 3 x
// ld128 + 4 x ld32 instead of 28 x ld32 -- and the three output planes leave as 16-byte stores.  Per pixel the arithmetic is the
// scalar kernel's (same Plane7, same plane_grad), so the gradient is bit-identical; the value differs only in the order of the fp64
// adds.  Instantiated for p = q = 1 only (every shipped TV configuration but modern.yaml / legacy.yaml; the general exponents inline
// powf a dozen times per pixel and stay on the scalar kernel).  At BASELINE configs[2] (8 x 3 x 224 x 224) the scalar kernel moved 9.6 MB in 8.2-8.5 us (profiles/r4_config3_*).
struct Rows {
  float c[6];  // row i: west, the four centres, east
  float s[5];  // row i+1: south-west, the four souths
  float n[5];  // row i-1: the four norths, north-east of the last
};

__device__ __forceinline__ Rows load_rows(const float* __restrict__ u, int i, int j, int H, int W) {
  Rows r;
  const int64_t o = (int64_t)i * W + j;
  const bool has_s = i + 1 < H, has_n = i > 0, has_w = j > 0, has_e = j + 4 < W;
  // Every load is unconditional (an out-of-image neighbour reads a clamped, valid address and is zeroed in registers): a
  // `cond ? *p : zero` on a float4 is scalarised by the compiler into four guarded dword loads (profiles/r4_kernel_isa_census.txt,
  // the round-2/3 multi-tensor kernel), and ten independent loads per plane are in flight together this way.
  const float4 c4 = *reinterpret_cast<const float4*>(u + o);
  const float4 s4 = *reinterpret_cast<const float4*>(u + o + (has_s ? W : 0));
  const float4 n4 = *reinterpret_cast<const float4*>(u + o - (has_n ? W : 0));
  const float w = u[o - (has_w ? 1 : 0)];
  const float e = u[o + (has_e ? 4 : 3)];
  const float sw = u[o + ((has_w && has_s) ? W - 1 : 0)];
  const float ne = u[o - ((has_n && has_e) ? W - 4 : 0)];
  r.c[0] = has_w ? w : 0.f;  // zero padding of the conv (regularizers.py:142-144, padding=1)
  r.c[5] = has_e ? e : 0.f;
  r.s[0] = (has_w && has_s) ? sw : 0.f;
  r.n[4] = (has_n && has_e) ? ne : 0.f;
  r.c[1] = c4.x, r.c[2] = c4.y, r.c[3] = c4.z, r.c[4] = c4.w;
  r.s[1] = has_s ? s4.x : 0.f, r.s[2] = has_s ? s4.y : 0.f, r.s[3] = has_s ? s4.z : 0.f, r.s[4] = has_s ? s4.w : 0.f;
  r.n[0] = has_n ? n4.x : 0.f, r.n[1] = has_n ? n4.y : 0.f, r.n[2] = has_n ? n4.z : 0.f, r.n[3] = has_n ? n4.w : 0.f;
  return r;
}

__device__ __forceinline__ Plane7 pixel_of(const Rows& r, int k) {
  return Plane7{r.c[k + 1], r.s[k + 1], r.c[k + 2], r.n[k], r.n[k + 1], r.c[k], r.s[k]};
}

struct TvCoefs {
  float tv_coef, p, q, eps, norm_grad_coef, norm_p;
};

// Pixel K of a quad: gradient of the three colour planes, value terms into the two fp64 accumulators (same statements as the scalar
// kernel's loop body).
template <bool OPP, int K>
__device__ __forceinline__ void quad_pixel(const Rows& rr, const Rows& rg, const Rows& rb, bool has_n, bool has_w, const TvCoefs& k,
                                           float& gr, float& gg, float& gb, double& acc_tv, double& acc_norm) {
  const Plane7 r = pixel_of(rr, K), g = pixel_of(rg, K), bl = pixel_of(rb, K);
  float value = 0.f;
  gr = plane_grad<true>(r, has_n, has_w, k.p, k.q, k.eps, value);
  gg = plane_grad<true>(g, has_n, has_w, k.p, k.q, k.eps, value);
  gb = plane_grad<true>(bl, has_n, has_w, k.p, k.q, k.eps, value);
  if constexpr (OPP) {
    const float o1 = plane_grad<true>(sub7(r, g), has_n, has_w, k.p, k.q, k.eps, value);
    const float o2 = plane_grad<true>(sub7(r, bl), has_n, has_w, k.p, k.q, k.eps, value);
    const float o3 = plane_grad<true>(sub7(g, bl), has_n, has_w, k.p, k.q, k.eps, value);
    gr += o1 + o2;
    gg += o3 - o1;
    gb -= o2 + o3;
  }
  gr *= k.tv_coef;
  gg *= k.tv_coef;
  gb *= k.tv_coef;
  acc_tv += (double)value;
  if (k.norm_grad_coef != 0.f) {
    float xp0, xp1, xp2;
    if (k.norm_p == 2.f) {
      xp0 = r.c * r.c, xp1 = g.c * g.c, xp2 = bl.c * bl.c;
      gr = fmaf(k.norm_grad_coef, r.c, gr);
      gg = fmaf(k.norm_grad_coef, g.c, gg);
      gb = fmaf(k.norm_grad_coef, bl.c, gb);
    } else {
      xp0 = powf(r.c, k.norm_p), xp1 = powf(g.c, k.norm_p), xp2 = powf(bl.c, k.norm_p);
      gr += k.norm_grad_coef * powf(r.c, k.norm_p - 1.f);
      gg += k.norm_grad_coef * powf(g.c, k.norm_p - 1.f);
      gb += k.norm_grad_coef * powf(bl.c, k.norm_p - 1.f);
    }
    acc_norm += (double)xp0 + (double)xp1 + (double)xp2;
  }
}

template <bool OPP, int THREADS>
__global__ __launch_bounds__(THREADS) void tv_norm_vec4_kernel(const float* __restrict__ x, int B, int H, int W, float tv_coef,
                                                               float p, float q, float eps, float norm_val_coef,
                                                               float norm_grad_coef, float norm_p, float* __restrict__ grad,
                                                               double* __restrict__ partials) {
  __shared__ double lds[(THREADS / bh::kWave) * 2];
  const int64_t plane = (int64_t)H * W;
  const int W4 = W >> 2;
  const int64_t quads_per_image = (int64_t)H * W4;
  const int64_t total = (int64_t)B * quads_per_image;
  const TvCoefs k{tv_coef, p, q, eps, norm_grad_coef, norm_p};
  double acc_tv = 0.0, acc_norm = 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * THREADS + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * THREADS) {
    const int b = (int)(idx / quads_per_image);
    const int64_t rem = idx - (int64_t)b * quads_per_image;
    const int i = (int)(rem / W4), j = (int)(rem - (int64_t)i * W4) << 2;
    const float* __restrict__ xb = x + (int64_t)b * 3 * plane;
    const Rows rr = load_rows(xb, i, j, H, W), rg = load_rows(xb + plane, i, j, H, W), rb = load_rows(xb + 2 * plane, i, j, H, W);
    const bool has_n = i > 0;
    float4 out_r, out_g, out_b;
    quad_pixel<OPP, 0>(rr, rg, rb, has_n, j > 0, k, out_r.x, out_g.x, out_b.x, acc_tv, acc_norm);
    quad_pixel<OPP, 1>(rr, rg, rb, has_n, true, k, out_r.y, out_g.y, out_b.y, acc_tv, acc_norm);
    quad_pixel<OPP, 2>(rr, rg, rb, has_n, true, k, out_r.z, out_g.z, out_b.z, acc_tv, acc_norm);
    quad_pixel<OPP, 3>(rr, rg, rb, has_n, true, k, out_r.w, out_g.w, out_b.w, acc_tv, acc_norm);
    float* __restrict__ gbase = grad + (int64_t)b * 3 * plane + (int64_t)i * W + j;
    *reinterpret_cast<float4*>(gbase) = out_r;
    *reinterpret_cast<float4*>(gbase + plane) = out_g;
    *reinterpret_cast<float4*>(gbase + 2 * plane) = out_b;
  }
  double v[2] = {acc_tv, acc_norm};
  bh::block_sum<2>(v, lds);
  if (threadIdx.x == 0) {
    partials[blockIdx.x * BH_PRIOR_PARTIAL_STRIDE + 0] = v[0] * (double)tv_coef;
    partials[blockIdx.x * BH_PRIOR_PARTIAL_STRIDE + 1] = v[1] * (double)norm_val_coef;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Kernel D: all BatchNorm inputs of a model in ONE launch per stage
// ---------------------------------------------------------------------------------------------------------------
//
// The reference hook fires once per BN layer (53 for ResNet-50) and spends ~10 launches each; a per-layer HIP kernel is
// still 53 x 3 launches whose 2-6 us floor dominates (round 1: 0.58 TB/s effective).  Here the layers travel as a
// pointer list in the kernel arguments plus a device table of layer descriptors, and the three stages are one launch
// each: sums over every (layer, channel, slab), finalize with one workgroup per layer (the last one to finish adds up
// the layers in a fixed order), and one elementwise backward over every layer writing a packed gradient buffer.

struct BnPtrs {
  const float* p[BH_BN_MAX_LAYERS];
};

// n / d and n % d for 0 <= n < 2^31 with a host-computed multiplier (d = 1: mul = shr = 0)
__device__ __forceinline__ void fast_divmod(uint32_t n, uint32_t d, uint32_t mul, uint32_t shr, uint32_t& q, uint32_t& r) {
  q = d != 1u ? (__umulhi(n, mul) >> shr) : n;
  r = n - q * d;
}

// Forward items (layer, channel, slab) are streamed by min(n_items, grid cap) workgroups; workgroup w takes items
// w, w + G, w + 2G, ... (consecutive workgroups work on consecutive items, i.e. on neighbouring memory).  Measured on the
// 355.6 MB of ResNet-50 at B = 8 (profiles/r3_kernel_bench.json): G = 512 / 1024 / 2048 / 4096 / one per item =
// 111 / 74 / 69 / 64 / 62 us -- a persistent grid does not pay here (items are short, 32 KB, and their descriptor fetch
// is hidden by the other resident workgroups), so the default is one workgroup per item; the loop stays for the cap.  The B planes of a channel are treated as one virtual array of B * HW elements; slab s owns an even share of
// it.  Wide layers: the whole workgroup walks the share; narrow layers (B * HW small, late ResNet stages): one wavefront
// per channel, four channels per workgroup.  Up to eight 16-byte loads are in flight per thread; each thread keeps fp32
// sums over at most 32 values before spilling into fp64 (round 2: four loads in flight, 64.5 us in the same burst timing).
__device__ __forceinline__ void bn_accumulate(const float4& q, float& a0, float& a1) {
  a0 += (q.x + q.y) + (q.z + q.w);
  a1 = fmaf(q.x, q.x, a1);
  a1 = fmaf(q.y, q.y, a1);
  a1 = fmaf(q.z, q.z, a1);
  a1 = fmaf(q.w, q.w, a1);
}

template <int N>
__device__ __forceinline__ void bn_load_round(const float4* __restrict__ x4, uint32_t v, uint32_t lanes, uint32_t unit,
                                              uint32_t cstride, uint32_t cbase, uint32_t mul, uint32_t shr, double& d0,
                                              double& d1) {
  float4 q[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    uint32_t b, j;
    fast_divmod(v + (uint32_t)k * lanes, unit, mul, shr, b, j);
    q[k] = x4[(size_t)b * cstride + cbase + j];
  }
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int k = 0; k < N; ++k) bn_accumulate(q[k], a0, a1);
  d0 += (double)a0;
  d1 += (double)a1;
}

// Measured and rejected in round 3 (commit 32ae57f, profiles/r3_kernel_bench_with_fused_bn_forward.json): finalising each layer
// inside this launch through a per-layer last-arriver ticket.  Every one of the 14 400 workgroups then needs an agent-scope
// release fence + ticket atomic: 1157 us instead of 63.5 + 12.3 us for the two launches (~80 ns per workgroup, serialised) --
// the same cost kernel A's fused epilogue showed in round 2.  Results were identical bit for bit; the two launches stay.
template <int DEPTH>  // 16-byte loads in flight per thread in the main loop: 8 (default) or 4
__global__ __launch_bounds__(kBlock) void bn_sums_kernel(BnPtrs ptrs, const bh_bn_layer* __restrict__ layers,
                                                         const bh_bn_item* __restrict__ items, int n_items,
                                                         double* __restrict__ sums) {
  __shared__ double lds[2][bh::kWavesPerBlock * 2];
  const int tid = threadIdx.x;
  int parity = 0;
  int i = blockIdx.x;
  if (i >= n_items) return;
  bh_bn_item it = items[i];
  for (; i < n_items; parity ^= 1) {
    const int next = i + (int)gridDim.x;
    const bh_bn_item it_next = items[next < n_items ? next : i];  // in flight while this item streams
    const bh_bn_layer L = layers[it.layer];
    const float* __restrict__ x = ptrs.p[it.layer];
    const bool narrow = L.narrow != 0;
    const uint32_t lanes = narrow ? (uint32_t)bh::kWave : (uint32_t)kBlock;
    const int lane = narrow ? (tid & (bh::kWave - 1)) : tid;
    const int c = narrow ? it.a + (tid >> 6) : it.a;
    const bool active = c < L.C;
    const bool vec = (L.HW & 3) == 0;
    const uint32_t unit = vec ? (uint32_t)(L.HW >> 2) : (uint32_t)L.HW;  // float4s (or floats) per plane
    const uint32_t total = (uint32_t)L.B * unit;
    const uint32_t v0 = (uint32_t)((uint64_t)total * (uint32_t)it.b / (uint32_t)L.S);
    const uint32_t v1 = (uint32_t)((uint64_t)total * ((uint32_t)it.b + 1u) / (uint32_t)L.S);
    double d0 = 0.0, d1 = 0.0;
    if (active) {
      if (vec) {
        const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
        const uint32_t cstride = (uint32_t)L.C * unit;  // float4s between consecutive planes of one channel
        const uint32_t cbase = (uint32_t)c * unit;
        uint32_t v = v0 + (uint32_t)lane;
        if constexpr (DEPTH == 8) {
          for (; v + 7u * lanes < v1; v += 8u * lanes)
            bn_load_round<8>(x4, v, lanes, unit, cstride, cbase, L.div_unit_mul, L.div_unit_shr, d0, d1);
        }
        for (; v + 3u * lanes < v1; v += 4u * lanes) {  // at most once when DEPTH == 8
          bn_load_round<4>(x4, v, lanes, unit, cstride, cbase, L.div_unit_mul, L.div_unit_shr, d0, d1);
        }
        float a0 = 0.f, a1 = 0.f;
        for (; v < v1; v += lanes) {
          uint32_t b, j;
          fast_divmod(v, unit, L.div_unit_mul, L.div_unit_shr, b, j);
          bn_accumulate(x4[(size_t)b * cstride + cbase + j], a0, a1);
        }
        d0 += (double)a0;
        d1 += (double)a1;
      } else {
        const size_t cstride = (size_t)L.C * unit;
        const size_t cbase = (size_t)c * unit;
        float a0 = 0.f, a1 = 0.f;
        int cnt = 0;
        for (uint32_t v = v0 + (uint32_t)lane; v < v1; v += lanes) {
          uint32_t b, j;
          fast_divmod(v, unit, L.div_unit_mul, L.div_unit_shr, b, j);
          const float q = x[(size_t)b * cstride + cbase + j];
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
      }
    }
    if (narrow) {  // wave-local: no LDS, no barrier
      d0 = bh::wave_sum(d0);
      d1 = bh::wave_sum(d1);
      if (lane == 0 && active) {
        double* out = sums + 2 * (L.sums_off + (int64_t)c * L.S + it.b);
        out[0] = d0;
        out[1] = d1;
      }
    } else {
      double v[2] = {d0, d1};
      bh::block_sum<2>(v, lds[parity]);  // one barrier; the other buffer is the previous item's, possibly still being read
      if (tid == 0) {
        double* out = sums + 2 * (L.sums_off + (int64_t)c * L.S + it.b);
        out[0] = v[0];
        out[1] = v[1];
      }
    }
    it = it_next;
    i = next;
  }
}

// The per-layer arithmetic of stage 2: per-channel mean / biased variance from the slab sums, the two norms, the weighted
// layer value, the backward coefficients.  THREADS threads of one workgroup; `lds`: 2 * waves + 2 doubles.
template <int THREADS>
__device__ __forceinline__ double bn_layer_statistic(const bh_bn_layer& L, const double* sums,
                                                     const float* __restrict__ running_mean,
                                                     const float* __restrict__ running_var, float* __restrict__ coef,
                                                     double* lds) {
  constexpr int kWaves = THREADS / bh::kWave;
  double* norms = lds + kWaves * 2;
  const double n = (double)L.B * (double)L.HW;
  auto channel_stats = [&](int c, double& mean, double& var) {
    double s0 = 0.0, s1 = 0.0;
    const double* row = sums + 2 * (L.sums_off + (int64_t)c * L.S);
    for (int s = 0; s < L.S; ++s) {
      s0 += row[2 * s];
      s1 += row[2 * s + 1];
    }
    mean = s0 / n;
    var = s1 / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
  };
  double v[2] = {0.0, 0.0};  // sum (rv - var)^2, sum (rm - mean)^2
  // the first kKeep channels of a thread (all of them up to C = 2048) keep their statistics in registers between the two
  // passes; round 2 re-read and re-summed the slab sums in the second pass
  constexpr int kKeep = 2048 / THREADS;
  double kmean[kKeep], kvar[kKeep];
  float krm[kKeep], krv[kKeep];
#pragma unroll
  for (int k = 0; k < kKeep; ++k) {
    const int c = threadIdx.x + k * THREADS;
    kmean[k] = kvar[k] = 0.0;
    krm[k] = krv[k] = 0.f;
    if (c < L.C) {
      channel_stats(c, kmean[k], kvar[k]);
      krm[k] = running_mean[L.chan_off + c];
      krv[k] = running_var[L.chan_off + c];
      const double dvv = (double)krv[k] - kvar[k], dm = (double)krm[k] - kmean[k];
      v[0] += dvv * dvv;
      v[1] += dm * dm;
    }
  }
  for (int c = threadIdx.x + kKeep * THREADS; c < L.C; c += THREADS) {
    double mean, var;
    channel_stats(c, mean, var);
    const double dvv = (double)running_var[L.chan_off + c] - var, dm = (double)running_mean[L.chan_off + c] - mean;
    v[0] += dvv * dvv;
    v[1] += dm * dm;
  }
  bh::block_sum<2>(v, lds);
  if (threadIdx.x == 0) {
    norms[0] = sqrt(v[0]);
    norms[1] = sqrt(v[1]);
  }
  __syncthreads();
  const double nv = norms[0], nm = norms[1], w = (double)L.weight;
  auto write_coef = [&](int c, double mean, double var, double rm, double rv) {
    const double pv = nv > 0.0 ? -(rv - var) / nv : 0.0;   // d r / d var_c
    const double pm = nm > 0.0 ? -(rm - mean) / nm : 0.0;  // d r / d mean_c
    reinterpret_cast<float2*>(coef)[L.chan_off + c] = make_float2((float)(w * (pm - 2.0 * pv * mean) / n), (float)(w * 2.0 * pv / n));
  };
#pragma unroll
  for (int k = 0; k < kKeep; ++k) {
    const int c = threadIdx.x + k * THREADS;
    if (c < L.C) write_coef(c, kmean[k], kvar[k], (double)krm[k], (double)krv[k]);
  }
  for (int c = threadIdx.x + kKeep * THREADS; c < L.C; c += THREADS) {
    double mean, var;
    channel_stats(c, mean, var);
    write_coef(c, mean, var, (double)running_mean[L.chan_off + c], (double)running_var[L.chan_off + c]);
  }
  return w * (nv + nm);
}

// Thread 0 of the workgroup that finalised a layer: publish the layer value, sign the model-wide ticket; the last layer to
// arrive adds the layers up in index order (fixed order => reproducible) and re-zeroes `n_reset` ticket words.
__device__ __forceinline__ void bn_publish_layer(int layer, int n_layers, double value, double* layer_values,
                                                 float* __restrict__ total, unsigned int* model_ticket,
                                                 unsigned int* reset, int n_reset) {
  layer_values[layer] = value;
  __threadfence();
  if (atomicAdd(model_ticket, 1u) != (unsigned int)n_layers - 1u) return;
  __threadfence();
  // reference order (regularizers.py:222-227): fp32 adds layer by layer; here fp64, rounded once
  double sum = 0.0;
  for (int l = 0; l < n_layers; ++l) sum += layer_values[l];
  total[0] = (float)sum;
  for (int k = 0; k < n_reset; ++k) reset[k] = 0u;
}

// Stage 2: one workgroup per layer (see bn_layer_statistic), the workgroup that finishes last adds the layers up into
// total[0].  THREADS: 256 / 512 / 1024 (`block_threads` of bh_bn_finalize; the statistics of up to 2048 channels stay in registers).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void bn_finalize_kernel(int n_layers, const bh_bn_layer* __restrict__ layers,
                                                         const double* __restrict__ sums,
                                                         const float* __restrict__ running_mean,
                                                         const float* __restrict__ running_var, float* __restrict__ coef,
                                                         double* layer_values, float* __restrict__ total,
                                                         unsigned int* counter) {
  __shared__ double lds[(THREADS / bh::kWave) * 2 + 2];
  const bh_bn_layer L = layers[blockIdx.x];
  const double value = bn_layer_statistic<THREADS>(L, sums, running_mean, running_var, coef, lds);
  if (threadIdx.x == 0) bn_publish_layer(blockIdx.x, n_layers, value, layer_values, total, counter, counter, 1);
}


// One workgroup per backward item: BH_BN_TILE consecutive elements of one layer (16-byte vectors when HW % 4 == 0).
__global__ __launch_bounds__(kBlock) void bn_bwd_kernel(BnPtrs ptrs, const bh_bn_layer* __restrict__ layers,
                                                        const bh_bn_item* __restrict__ items,
                                                        const float* __restrict__ coef, const float* __restrict__ gout,
                                                        float* __restrict__ grad_flat) {
  const bh_bn_item it = items[blockIdx.x];
  const bh_bn_layer L = layers[it.layer];
  const float* __restrict__ x = ptrs.p[it.layer];
  float* __restrict__ o = grad_flat + L.flat_off;
  const float g = gout ? gout[0] : 1.f;
  const float2* __restrict__ ab = reinterpret_cast<const float2*>(coef) + L.chan_off;
  const uint32_t begin = (uint32_t)it.a, count = (uint32_t)it.b;  // in float4s when vectorised, floats otherwise
  if ((L.HW & 3) == 0) {
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(o);
    const uint32_t unit = (uint32_t)(L.HW >> 2);
    float4 q[4];
    float2 k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t i = threadIdx.x + u * kBlock;
      if (i < count) {
        uint32_t plane, j, bidx, c;
        fast_divmod(begin + i, unit, L.div_unit_mul, L.div_unit_shr, plane, j);
        fast_divmod(plane, (uint32_t)L.C, L.div_c_mul, L.div_c_shr, bidx, c);
        q[u] = x4[begin + i];
        k[u] = ab[c];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t i = threadIdx.x + u * kBlock;
      if (i < count) {
        const float a = g * k[u].x, b = g * k[u].y;
        o4[begin + i] = make_float4(fmaf(b, q[u].x, a), fmaf(b, q[u].y, a), fmaf(b, q[u].z, a), fmaf(b, q[u].w, a));
      }
    }
  } else {
    const uint32_t unit = (uint32_t)L.HW;
    for (uint32_t i = threadIdx.x; i < count; i += kBlock) {
      uint32_t plane, j, bidx, c;
      fast_divmod(begin + i, unit, L.div_unit_mul, L.div_unit_shr, plane, j);
      fast_divmod(plane, (uint32_t)L.C, L.div_c_mul, L.div_c_shr, bidx, c);
      const float2 kk = ab[c];
      o[begin + i] = fmaf(g * kk.y, x[begin + i], g * kk.x);
    }
  }
}

// Backward of ONE layer fused with the accumulation into the activation gradient (the read-modify-write autograd would
// otherwise do with one ATen `add` per layer): out[i] = gin[i] + g * (A_c + B_c * x[i]); gin may be null (no other
// contribution reached this activation).  Same items / index arithmetic as bn_bwd_kernel, addresses local to the layer.
__global__ __launch_bounds__(kBlock) void bn_bwd_acc_kernel(const float* __restrict__ x, const float* __restrict__ gin,
                                                            float* __restrict__ o, const bh_bn_layer* __restrict__ layers,
                                                            const bh_bn_item* __restrict__ items,
                                                            const float* __restrict__ coef, const float* __restrict__ gout) {
  const bh_bn_item it = items[blockIdx.x];
  const bh_bn_layer L = layers[it.layer];
  const float g = gout ? gout[0] : 1.f;
  const float2* __restrict__ ab = reinterpret_cast<const float2*>(coef) + L.chan_off;
  const uint32_t begin = (uint32_t)it.a, count = (uint32_t)it.b;  // in float4s when vectorised, floats otherwise
  if ((L.HW & 3) == 0) {
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gin);
    float4* __restrict__ o4 = reinterpret_cast<float4*>(o);
    const uint32_t unit = (uint32_t)(L.HW >> 2);
    float4 q[4], acc[4];
    float2 k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t i = threadIdx.x + u * kBlock;
      if (i < count) {
        uint32_t plane, j, bidx, c;
        fast_divmod(begin + i, unit, L.div_unit_mul, L.div_unit_shr, plane, j);
        fast_divmod(plane, (uint32_t)L.C, L.div_c_mul, L.div_c_shr, bidx, c);
        q[u] = x4[begin + i];
        acc[u] = gin ? g4[begin + i] : make_float4(0.f, 0.f, 0.f, 0.f);
        k[u] = ab[c];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t i = threadIdx.x + u * kBlock;
      if (i < count) {
        const float a = g * k[u].x, b = g * k[u].y;
        // the statistic's term is rounded exactly as bn_bwd_kernel rounds it, then added like autograd's `add` would
        o4[begin + i] = make_float4(acc[u].x + fmaf(b, q[u].x, a), acc[u].y + fmaf(b, q[u].y, a), acc[u].z + fmaf(b, q[u].z, a),
                                    acc[u].w + fmaf(b, q[u].w, a));
      }
    }
  } else {
    const uint32_t unit = (uint32_t)L.HW;
    for (uint32_t i = threadIdx.x; i < count; i += kBlock) {
      uint32_t plane, j, bidx, c;
      fast_divmod(begin + i, unit, L.div_unit_mul, L.div_unit_shr, plane, j);
      fast_divmod(plane, (uint32_t)L.C, L.div_c_mul, L.div_c_shr, bidx, c);
      const float2 kk = ab[c];
      o[begin + i] = (gin ? gin[begin + i] : 0.f) + fmaf(g * kk.y, x[begin + i], g * kk.x);
    }
  }
}

// host: multiplier / shift of fast_divmod
void find_divisor(uint32_t d, uint32_t& mul, uint32_t& shr) {
  if (d <= 1u) {
    mul = 0u;
    shr = 0u;
    return;
  }
  uint32_t lg = 0;
  while ((1ull << lg) < d) ++lg;  // ceil(log2 d)
  const uint32_t p = 31u + lg;
  mul = (uint32_t)(((1ull << p) + d - 1ull) / d);
  shr = p - 32u;
}

// (the cap of the forward grid, the load depth and the finalize block are launch arguments: no mutable library state)


struct BnGeometry {
  int32_t S, narrow;
  int64_t fwd_items, bwd_items, flat;
};

bool bn_geometry(int32_t B, int32_t C, int32_t HW, BnGeometry& g) {
  if (B <= 0 || C <= 0 || HW <= 0) return false;
  const int64_t per_channel = (int64_t)B * HW, numel = per_channel * C;
  if (numel >= (int64_t)1 << 31) return false;  // 32-bit indices inside a layer
  bh::channel_geometry(per_channel, g.S, g.narrow);  // the rule kernel E shares (bh_common.h)
  const int64_t S = g.S;
  g.fwd_items = g.narrow ? (C + bh::kWavesPerBlock - 1) / bh::kWavesPerBlock : (int64_t)C * S;
  const int64_t units = (HW & 3) == 0 ? numel / 4 : numel;
  const int64_t per_item = (HW & 3) == 0 ? BH_BN_TILE / 4 : BH_BN_TILE;
  g.bwd_items = (units + per_item - 1) / per_item;
  g.flat = (numel + 3) & ~int64_t(3);
  return true;
}

}  // namespace

extern "C" {

int bh_prior_tv_norm(const float* x, int32_t B, int32_t H, int32_t W, float tv_scale, float inner_exp, float outer_exp,
                     float eps, int32_t double_opponents, float norm_scale, float norm_p, float* grad_out,
                     double* partials_dev, void* stream) {
  if (x == nullptr || grad_out == nullptr || partials_dev == nullptr || B <= 0 || H <= 0 || W <= 0) return BH_EINVAL;
  if (norm_scale != 0.f && norm_p == 0.f) return BH_EINVAL;
  const int64_t pixels = (int64_t)B * H * W;
  const int groups = double_opponents ? 6 : 3;
  // mean over [B, groups, H, W] (regularizers.py:147) / mean over [B, 3, H, W] (:197)
  const float tv_coef = (float)((double)tv_scale / ((double)pixels * groups));
  const float norm_val_coef = norm_scale != 0.f ? (float)((double)norm_scale / (double)norm_p / ((double)pixels * 3)) : 0.f;
  const float norm_grad_coef = norm_scale != 0.f ? (float)((double)norm_scale / ((double)pixels * 3)) : 0.f;
  const bool pq1 = inner_exp == 1.f && outer_exp == 1.f;
  hipStream_t st = bh::as_stream(stream);
  // 16-byte path for batches (at least one 256-thread workgroup of quads per CU: B >= 6 at 224 x 224): measured with rocprofv3 on
  // 8 x 3 x 224 x 224, 7.25 us against 8.1 us for the one-pixel-per-thread kernel alone (8.4 vs 8.2-8.5 us inside the see-through
  // loop); at B = 1 the launch is latency-bound and fewer, fatter threads LOSE (one wavefront per workgroup of quads: 5.1 us against
  // 4.1-4.3 us), so single images stay on the scalar kernel (profiles/r5_step_prior_probe_kernel_summary.csv).
  const int64_t quads = pixels >> 2;
  const bool vec4 = pq1 && (W & 3) == 0 && quads >= (int64_t)256 * kBlock &&
                    ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(grad_out)) & 15u) == 0;
  int grid;
  if (vec4) {
    const int64_t blocks = (quads + kBlock - 1) / kBlock;
    grid = (int)(blocks < BH_PRIOR_MAX_GRID ? blocks : BH_PRIOR_MAX_GRID);
    if (double_opponents)
      hipLaunchKernelGGL((tv_norm_vec4_kernel<true, kBlock>), dim3(grid), dim3(kBlock), 0, st, x, B, H, W, tv_coef, inner_exp, outer_exp, eps,
                         norm_val_coef, norm_grad_coef, norm_p, grad_out, partials_dev);
    else
      hipLaunchKernelGGL((tv_norm_vec4_kernel<false, kBlock>), dim3(grid), dim3(kBlock), 0, st, x, B, H, W, tv_coef, inner_exp, outer_exp, eps,
                         norm_val_coef, norm_grad_coef, norm_p, grad_out, partials_dev);
  } else {
    const int64_t blocks = (pixels + kBlock - 1) / kBlock;
    grid = (int)(blocks < BH_PRIOR_MAX_GRID ? blocks : BH_PRIOR_MAX_GRID);
#define BH_TV_LAUNCH(PQ1, OPP)                                                                                         \
  hipLaunchKernelGGL((tv_norm_kernel<PQ1, OPP>), dim3(grid), dim3(kBlock), 0, st, x, B, H, W, tv_coef, inner_exp,     \
                     outer_exp, eps, norm_val_coef, norm_grad_coef, norm_p, grad_out, partials_dev)
    if (pq1 && !double_opponents) BH_TV_LAUNCH(true, false);
    else if (pq1) BH_TV_LAUNCH(true, true);
    else if (!double_opponents) BH_TV_LAUNCH(false, false);
    else BH_TV_LAUNCH(false, true);
#undef BH_TV_LAUNCH
  }
  const int rc = bh::launch_status();
  return rc != 0 ? rc : grid;
}

int bh_bn_plan_size(int32_t n_layers, const int32_t* B, const int32_t* C, const int32_t* HW, int64_t* n_fwd_items,
                    int64_t* n_bwd_items, int64_t* flat_elems, int64_t* n_sum_pairs, int64_t* n_channels) {
  if (n_layers <= 0 || n_layers > BH_BN_MAX_LAYERS || B == nullptr || C == nullptr || HW == nullptr ||
      n_fwd_items == nullptr || n_bwd_items == nullptr || flat_elems == nullptr || n_sum_pairs == nullptr ||
      n_channels == nullptr)
    return BH_EINVAL;
  int64_t fwd = 0, bwd = 0, flat = 0, pairs = 0, chans = 0;
  for (int32_t l = 0; l < n_layers; ++l) {
    BnGeometry g;
    if (!bn_geometry(B[l], C[l], HW[l], g)) return BH_EINVAL;
    fwd += g.fwd_items;
    bwd += g.bwd_items;
    flat += g.flat;
    pairs += (int64_t)C[l] * g.S;
    chans += C[l];
  }
  if (fwd > INT32_MAX || bwd > INT32_MAX || chans > INT32_MAX) return BH_EINVAL;
  *n_fwd_items = fwd;
  *n_bwd_items = bwd;
  *flat_elems = flat;
  *n_sum_pairs = pairs;
  *n_channels = chans;
  return 0;
}

int bh_bn_plan_build(int32_t n_layers, const int32_t* B, const int32_t* C, const int32_t* HW, const float* weights,
                     bh_bn_layer* layers, bh_bn_item* fwd_items, int64_t n_fwd_items, bh_bn_item* bwd_items,
                     int64_t n_bwd_items) {
  int64_t fwd = 0, bwd = 0, flat = 0, pairs = 0, chans = 0;
  int rc = bh_bn_plan_size(n_layers, B, C, HW, &fwd, &bwd, &flat, &pairs, &chans);
  if (rc != 0) return rc;
  if (fwd != n_fwd_items || bwd != n_bwd_items || layers == nullptr || fwd_items == nullptr || bwd_items == nullptr)
    return BH_EINVAL;
  int64_t fi = 0, bi = 0;
  flat = pairs = chans = 0;
  for (int32_t l = 0; l < n_layers; ++l) {
    BnGeometry g;
    bn_geometry(B[l], C[l], HW[l], g);
    bh_bn_layer& L = layers[l];
    L.flat_off = flat;
    L.sums_off = pairs;
    L.chan_off = (int32_t)chans;
    L.B = B[l];
    L.C = C[l];
    L.HW = HW[l];
    L.S = g.S;
    L.narrow = g.narrow;
    L.weight = weights ? weights[l] : 1.f;
    L.fwd_items = (int32_t)g.fwd_items;
    const bool vec = (HW[l] & 3) == 0;
    find_divisor((uint32_t)(vec ? HW[l] / 4 : HW[l]), L.div_unit_mul, L.div_unit_shr);
    find_divisor((uint32_t)C[l], L.div_c_mul, L.div_c_shr);
    if (g.narrow) {
      for (int32_t c = 0; c < C[l]; c += bh::kWavesPerBlock) fwd_items[fi++] = bh_bn_item{l, c, 0, 0};
    } else {
      for (int32_t c = 0; c < C[l]; ++c)
        for (int32_t s = 0; s < g.S; ++s) fwd_items[fi++] = bh_bn_item{l, c, s, 0};
    }
    const int64_t numel = (int64_t)B[l] * C[l] * HW[l];
    const int64_t units = vec ? numel / 4 : numel, per_item = vec ? BH_BN_TILE / 4 : BH_BN_TILE;
    for (int64_t u = 0; u < units; u += per_item) {
      const int64_t cnt = (units - u) < per_item ? (units - u) : per_item;
      bwd_items[bi++] = bh_bn_item{l, (int32_t)u, (int32_t)cnt, 0};
    }
    flat += g.flat;
    pairs += (int64_t)C[l] * g.S;
    chans += C[l];
  }
  return (fi == n_fwd_items && bi == n_bwd_items) ? 0 : BH_EINVAL;
}

namespace {
bool fill_bn_ptrs(BnPtrs& out, const void* const* x_ptrs, int32_t n_layers, const int32_t* hw_host) {
  for (int32_t l = 0; l < n_layers; ++l) {
    const void* p = x_ptrs[l];
    if (p == nullptr) return false;
    if (hw_host != nullptr && (hw_host[l] & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15u) != 0) return false;
    out.p[l] = static_cast<const float*>(p);
  }
  for (int32_t l = n_layers; l < BH_BN_MAX_LAYERS; ++l) out.p[l] = nullptr;
  return true;
}
}  // namespace

int bh_bn_sums(int32_t n_layers, const void* const* x_ptrs, const int32_t* hw_host, const bh_bn_layer* layers_dev,
               const bh_bn_item* fwd_items_dev, int64_t n_fwd_items, double* sums_dev, int32_t grid_cap, int32_t load_depth,
               void* stream) {
  if (grid_cap < 0 || grid_cap > (1 << 20) || (load_depth != 0 && load_depth != 4 && load_depth != 8)) return BH_EINVAL;
  if (n_layers <= 0 || n_layers > BH_BN_MAX_LAYERS || x_ptrs == nullptr || hw_host == nullptr || layers_dev == nullptr ||
      fwd_items_dev == nullptr || n_fwd_items <= 0 || n_fwd_items > INT32_MAX || sums_dev == nullptr)
    return BH_EINVAL;
  BnPtrs ptrs;
  if (!fill_bn_ptrs(ptrs, x_ptrs, n_layers, hw_host)) return BH_EINVAL;
  const int64_t cap = grid_cap > 0 ? grid_cap : BH_BN_DEFAULT_GRID;
  const int64_t grid = n_fwd_items < cap ? n_fwd_items : cap;
  if (load_depth == 4)
    hipLaunchKernelGGL(bn_sums_kernel<4>, dim3((unsigned int)grid), dim3(kBlock), 0, bh::as_stream(stream), ptrs, layers_dev,
                       fwd_items_dev, (int)n_fwd_items, sums_dev);
  else
    hipLaunchKernelGGL(bn_sums_kernel<8>, dim3((unsigned int)grid), dim3(kBlock), 0, bh::as_stream(stream), ptrs, layers_dev,
                       fwd_items_dev, (int)n_fwd_items, sums_dev);
  return bh::launch_status();
}

int bh_bn_bwd_accumulate(const float* x, const float* gin, int32_t hw, const bh_bn_layer* layers_dev,
                         const bh_bn_item* layer_bwd_items_dev, int64_t n_items, const float* coef_dev, const float* gout_dev,
                         float* out, void* stream) {
  if (x == nullptr || out == nullptr || layers_dev == nullptr || layer_bwd_items_dev == nullptr || n_items <= 0 ||
      n_items > INT32_MAX || coef_dev == nullptr || hw <= 0)
    return BH_EINVAL;
  if ((hw & 3) == 0 && (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gin)) & 15u) != 0))
    return BH_EINVAL;
  hipLaunchKernelGGL(bn_bwd_acc_kernel, dim3((unsigned int)n_items), dim3(kBlock), 0, bh::as_stream(stream), x, gin, out,
                     layers_dev, layer_bwd_items_dev, coef_dev, gout_dev);
  return bh::launch_status();
}

int bh_bn_finalize(int32_t n_layers, const bh_bn_layer* layers_dev, const double* sums_dev, const float* running_mean,
                   const float* running_var, float* coef_dev, double* layer_values_dev, float* total_dev,
                   void* counter_dev, int32_t block_threads, void* stream) {
  if (block_threads != 0 && block_threads != 256 && block_threads != 512 && block_threads != 1024) return BH_EINVAL;
  if (n_layers <= 0 || n_layers > BH_BN_MAX_LAYERS || layers_dev == nullptr || sums_dev == nullptr ||
      running_mean == nullptr || running_var == nullptr || coef_dev == nullptr || layer_values_dev == nullptr ||
      total_dev == nullptr || counter_dev == nullptr)
    return BH_EINVAL;
  if ((reinterpret_cast<uintptr_t>(coef_dev) & 7u) != 0) return BH_EINVAL;  // read back as float2
#define BH_BN_FIN_LAUNCH(T)                                                                                             \
  hipLaunchKernelGGL(bn_finalize_kernel<T>, dim3(n_layers), dim3(T), 0, bh::as_stream(stream), n_layers, layers_dev,    \
                     sums_dev, running_mean, running_var, coef_dev, layer_values_dev, total_dev,                        \
                     static_cast<unsigned int*>(counter_dev))
  const int threads = block_threads > 0 ? block_threads : BH_BN_DEFAULT_FINALIZE_BLOCK;
  if (threads == 256) BH_BN_FIN_LAUNCH(256);
  else if (threads == 512) BH_BN_FIN_LAUNCH(512);
  else BH_BN_FIN_LAUNCH(1024);
#undef BH_BN_FIN_LAUNCH
  return bh::launch_status();
}

int bh_bn_bwd(int32_t n_layers, const void* const* x_ptrs, const int32_t* hw_host, const bh_bn_layer* layers_dev,
              const bh_bn_item* bwd_items_dev, int64_t n_bwd_items, const float* coef_dev, const float* gout_dev,
              float* grad_flat, void* stream) {
  if (n_layers <= 0 || n_layers > BH_BN_MAX_LAYERS || x_ptrs == nullptr || hw_host == nullptr || layers_dev == nullptr ||
      bwd_items_dev == nullptr || n_bwd_items <= 0 || n_bwd_items > INT32_MAX || coef_dev == nullptr ||
      grad_flat == nullptr || (reinterpret_cast<uintptr_t>(grad_flat) & 15u) != 0)
    return BH_EINVAL;
  BnPtrs ptrs;
  if (!fill_bn_ptrs(ptrs, x_ptrs, n_layers, hw_host)) return BH_EINVAL;
  hipLaunchKernelGGL(bn_bwd_kernel, dim3((unsigned int)n_bwd_items), dim3(kBlock), 0, bh::as_stream(stream), ptrs,
                     layers_dev, bwd_items_dev, coef_dev, gout_dev, grad_flat);
  return bh::launch_status();
}

}  // extern "C"
