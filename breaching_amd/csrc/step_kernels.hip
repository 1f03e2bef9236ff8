// Kernel B and friends: per-trial device state, loss commit, and the fused candidate step.
//
// The reference loop (breaching/attacks/optimization_based_attack.py:110-138) spends three host synchronisations
// per iteration (`objective_value < minimal_value_so_far` :119, `torch.isfinite` :131, `.item()` :135) and ~15 tiny
// launches for grad post-processing + torch.optim.Adam + projection.  Here the iteration counter, the loss history,
// the best-so-far bookkeeping and the non-finite flag live in a small device record, and one elementwise kernel does
//   assemble gradient -> Langevin noise -> norm clip -> sign -> Adam/AdamW -> box projection -> conditional best copy.
// The host never has to look at the device between callbacks, so the whole iteration can be captured in a hipGraph.

#include "bh_common.h"

namespace {

using bh::kBlock;

union Word {
  int32_t i;
  float f;
};

__global__ void state_reset_kernel(Word* st) {
  const int t = threadIdx.x;
  if (t >= BH_STATE_WORDS) return;
  Word w;
  w.i = 0;
  if (t == BH_STATE_IT || t == BH_STATE_FIRST_BAD) w.i = -1;
  if (t == BH_STATE_MIN) w.f = __builtin_inff();
  st[t] = w;
}

// Single workgroup.  Sums the regulariser partials in a fixed order, adds the optional scalar terms, commits.
__global__ __launch_bounds__(kBlock) void loss_commit_kernel(Word* __restrict__ st, float* __restrict__ history,
                                                             int history_len, const float* __restrict__ gm_loss,
                                                             const double* __restrict__ reg_partials, int n_reg,
                                                             const float* __restrict__ extra0,
                                                             const float* __restrict__ extra1) {
  __shared__ double lds[bh::kWavesPerBlock];
  double v[1] = {0.0};
  for (int i = threadIdx.x; i < n_reg; i += kBlock) v[0] += reg_partials[i];
  bh::block_sum<1>(v, lds);
  if (threadIdx.x != 0) return;
  // reference order: objective, then regularisers (optimization_based_attack.py:157-162), all fp32 adds
  float total = gm_loss ? gm_loss[0] : 0.f;
  if (n_reg > 0) total += (float)v[0];
  if (extra0) total += extra0[0];
  if (extra1) total += extra1[0];
  const int it = st[BH_STATE_IT].i + 1;
  st[BH_STATE_IT].i = it;
  st[BH_STATE_TOTAL].f = total;
  if (it >= 0 && it < history_len) history[it] = total;
  const bool dead = st[BH_STATE_DEAD].i != 0;
  const bool improved = !dead && (total < st[BH_STATE_MIN].f);  // :119 (false for NaN)
  st[BH_STATE_IMPROVED].i = improved ? 1 : 0;
  if (improved) st[BH_STATE_MIN].f = total;
  if (!dead && !isfinite(total)) {  // :131-133 -- the reference leaves the loop here
    st[BH_STATE_DEAD].i = 1;
    st[BH_STATE_FIRST_BAD].i = it;
  }
}

__device__ __forceinline__ float effective_grad(const float* __restrict__ g, const float* __restrict__ g_reg,
                                                const float* __restrict__ noise, float noise_coef, int64_t i) {
  float v = g[i];
  if (g_reg) v += g_reg[i];
  if (noise) v = fmaf(noise_coef, noise[i], v);  // optimization_based_attack.py:167-170
  return v;
}

__global__ __launch_bounds__(kBlock) void grad_sumsq_kernel(const Word* __restrict__ st, const float* __restrict__ g,
                                                            const float* __restrict__ g_reg,
                                                            const float* __restrict__ noise, int64_t n,
                                                            const double* __restrict__ sched, float langevin,
                                                            double* __restrict__ ws) {
  __shared__ double lds[bh::kWavesPerBlock];
  const int it = st[BH_STATE_IT].i;
  const float noise_coef = noise ? (float)((double)langevin * sched[(int64_t)it * BH_SCHED_STRIDE + 3]) : 0.f;
  float acc = 0.f;
  double dacc = 0.0;
  int cnt = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const float v = effective_grad(g, g_reg, noise, noise_coef, i);
    acc = fmaf(v, v, acc);
    if (++cnt == 32) {
      dacc += (double)acc;
      acc = 0.f;
      cnt = 0;
    }
  }
  double v[1] = {dacc + (double)acc};
  bh::block_sum<1>(v, lds);
  if (threadIdx.x == 0) ws[blockIdx.x] = v[0];
}

__global__ __launch_bounds__(kBlock) void grad_norm_finalize_kernel(Word* __restrict__ st, const double* __restrict__ ws,
                                                                    int rows) {
  __shared__ double lds[bh::kWavesPerBlock];
  double v[1] = {0.0};
  for (int i = threadIdx.x; i < rows; i += kBlock) v[0] += ws[i];
  bh::block_sum<1>(v, lds);
  if (threadIdx.x == 0) st[BH_STATE_GNORM].f = (float)sqrt(v[0]);
}

// The fused step.  fp32 arithmetic follows torch.optim's single-tensor Adam update term by term:
//   exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
//   denom = (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps); param.addcdiv_(exp_avg, denom, value=-lr/bc1)
struct StepScalars {
  float step_size, bc2_sqrt, decay, noise_coef, clip_mul, soft, w1, w2, beta2, eps;
  bool improved;
};

// `gnorm_in`: the gradient norm of THIS tensor when the caller has it already (the list kernel derives it from the partial rows in
// its preamble); NaN-free sentinel < 0 = read state[BH_STATE_GNORM] (the single-tensor launch behind bh_grad_norm).
__device__ __forceinline__ StepScalars step_scalars(const Word* __restrict__ st, const double* __restrict__ sched,
                                                    const bh_step_params& P, bool has_noise, float gnorm_in = -1.f) {
  StepScalars k;
  const int it = st[BH_STATE_IT].i;
  k.improved = st[BH_STATE_IMPROVED].i != 0;
  const double* row = sched + (int64_t)it * BH_SCHED_STRIDE;
  k.step_size = (float)row[0];
  k.bc2_sqrt = (float)row[1];
  k.decay = (float)row[2];
  k.noise_coef = has_noise ? (float)((double)P.langevin * row[3]) : 0.f;
  k.clip_mul = 1.f;
  if (P.grad_clip >= 0.f) {  // negative: clipping off (optim.grad_clip = None)
    const float gn = gnorm_in >= 0.f ? gnorm_in : st[BH_STATE_GNORM].f;
    if (gn > P.grad_clip) k.clip_mul = P.grad_clip / (gn + 1e-6f);  // :173-174
  }
  // soft sign factor (:176-180): python evaluates 1 - iteration / max_iterations in double
  k.soft = (float)(1.0 - (double)it / (double)P.max_iterations);
  // torch passes `1 - beta` (evaluated in double) and `beta2`, `eps` as Python scalars that become fp32 in the kernels
  k.w1 = (float)(1.0 - P.beta1);
  k.w2 = (float)(1.0 - P.beta2);
  k.beta2 = (float)P.beta2;
  k.eps = (float)P.eps;
  return k;
}

// One element: gr = assembled gradient (objective + prior + noise); lo / hi = the box of its channel.
__device__ __forceinline__ void step_elem(const StepScalars& k, const bh_step_params& P, float gr, float lo, float hi, float& xi,
                                          float& mi, float& vi) {
  gr *= k.clip_mul;
  if (P.sign_mode == BH_SIGN_HARD) {
    gr = bh::sgnf(gr);  // :181-182
  } else if (P.sign_mode == BH_SIGN_SOFT) {
    gr = tanhf(gr * k.soft) / k.soft;  // :180
  }
  if (P.decoupled_wd) xi *= k.decay;
  mi = fmaf(k.w1, gr - mi, mi);
  vi = fmaf(k.w2 * gr, gr, vi * k.beta2);
  const float denom = sqrtf(vi) / k.bc2_sqrt + k.eps;
  xi = xi - (k.step_size * mi) / denom;
  // :117-118  max(min(x, hi), lo); torch.min / torch.max propagate NaN, fminf / fmaxf would swallow it
  if (P.boxed) xi = (xi != xi) ? xi : fmaxf(fminf(xi, hi), lo);
}

__global__ __launch_bounds__(kBlock) void candidate_step_kernel(const Word* __restrict__ st, const double* __restrict__ sched,
                                                                bh_step_params P, float* __restrict__ x,
                                                                const float* __restrict__ g,
                                                                const float* __restrict__ g_reg,
                                                                const float* __restrict__ noise, float* __restrict__ m,
                                                                float* __restrict__ v, float* __restrict__ best) {
  const StepScalars k = step_scalars(st, sched, P, noise != nullptr);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * kBlock) {
    const float gr = effective_grad(g, g_reg, noise, k.noise_coef, i);
    float xi = x[i], mi = m[i], vi = v[i];
    const int c = P.boxed ? (int)((i / P.plane) % P.channels) : 0;
    step_elem(k, P, gr, P.lo[c], P.hi[c], xi, mi, vi);
    x[i] = xi;
    m[i] = mi;
    v[i] = vi;
    if (k.improved) best[i] = xi;  // :119-121 (clone taken after step + projection)
  }
}

// 16-byte variant: n and the channel plane are multiples of 4 and every buffer is 16-byte aligned, so a float4 never straddles
// a channel and the box of its four elements is one (lo, hi) pair, looked up once.  All loads of a float4 group are issued before
// the first use.  Element arithmetic = step_elem, i.e. bit-identical to the scalar kernel.  (At BASELINE configs[2], 8 x 3 x 224 x
// 224, the scalar kernel moved ~43-48 MB in 11.4-11.9 us with 4-byte accesses, profiles/r4_kernel_isa_census.txt.)
template <bool HAS_REG, bool HAS_NOISE>  // resolved per launch: `p ? p4[i] : zero` on a float4 is scalarised into four guarded dword loads
__global__ __launch_bounds__(kBlock) void candidate_step_vec4_kernel(const Word* __restrict__ st, const double* __restrict__ sched,
                                                                     bh_step_params P, float* __restrict__ x,
                                                                     const float* __restrict__ g,
                                                                     const float* __restrict__ g_reg,
                                                                     const float* __restrict__ noise, float* __restrict__ m,
                                                                     float* __restrict__ v, float* __restrict__ best) {
  const int64_t n4 = P.n >> 2, plane4 = P.plane >> 2;
  float4* __restrict__ x4 = reinterpret_cast<float4*>(x);
  float4* __restrict__ m4 = reinterpret_cast<float4*>(m);
  float4* __restrict__ v4 = reinterpret_cast<float4*>(v);
  float4* __restrict__ b4 = reinterpret_cast<float4*>(best);
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
  const float4* __restrict__ r4 = reinterpret_cast<const float4*>(g_reg);
  const float4* __restrict__ z4 = reinterpret_cast<const float4*>(noise);
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  // The launch covers the tensor in one sweep whenever n / 4 <= 2048 * 256 (every BASELINE candidate): the thread's six operand
  // loads are issued FIRST and the iteration scalars -- two dependent loads, state word -> schedule row -- are fetched while they
  // are in flight, instead of one round trip after the other in front of a 5-10 us kernel.
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool mine = i < n4;
  const int64_t i0 = mine ? i : 0;
  float4 gv = g4[i0];
  float4 rv = zero, zv = zero;
  if constexpr (HAS_REG) rv = r4[i0];
  if constexpr (HAS_NOISE) zv = z4[i0];
  float4 xv = x4[i0], mv = m4[i0], vv = v4[i0];
  const StepScalars k = step_scalars(st, sched, P, HAS_NOISE);
  if (!mine) return;
  for (;;) {
    const int c = P.boxed ? (int)((i / plane4) % P.channels) : 0;
    const float lo = P.lo[c], hi = P.hi[c];
    float ge[4] = {gv.x, gv.y, gv.z, gv.w};
    if constexpr (HAS_REG) ge[0] += rv.x, ge[1] += rv.y, ge[2] += rv.z, ge[3] += rv.w;
    if constexpr (HAS_NOISE) {  // optimization_based_attack.py:167-170
      ge[0] = fmaf(k.noise_coef, zv.x, ge[0]), ge[1] = fmaf(k.noise_coef, zv.y, ge[1]);
      ge[2] = fmaf(k.noise_coef, zv.z, ge[2]), ge[3] = fmaf(k.noise_coef, zv.w, ge[3]);
    }
    step_elem(k, P, ge[0], lo, hi, xv.x, mv.x, vv.x);
    step_elem(k, P, ge[1], lo, hi, xv.y, mv.y, vv.y);
    step_elem(k, P, ge[2], lo, hi, xv.z, mv.z, vv.z);
    step_elem(k, P, ge[3], lo, hi, xv.w, mv.w, vv.w);
    x4[i] = xv;
    m4[i] = mv;
    v4[i] = vv;
    if (k.improved) b4[i] = xv;
    i += (int64_t)gridDim.x * kBlock;
    if (i >= n4) break;
    gv = g4[i];
    if constexpr (HAS_REG) rv = r4[i];
    if constexpr (HAS_NOISE) zv = z4[i];
    xv = x4[i], mv = m4[i], vv = v4[i];
  }
}


// ---- the same two stages over a LIST of optimised tensors (joint data + label attack: 2 tensors) in ONE launch each -------------
// reference: optimization_with_label_attack.py:124-128 (only the data tensor is projected), :177-190 (noise, clip and sign applied to
// `[candidate, labels]` tensor by tensor, each clipped by ITS OWN norm).  Round 5 launched kernel B, the sum of squares and its
// finalize once per tensor: six latency-bound launches per iteration where two do.  The workgroups of a launch are dealt over the
// slots (`first_block`); inside a slot the arithmetic is the single-tensor kernels', statement for statement:
//   * sum of squares: slot s gets the SAME rows = min(ceil(n / 2048), BH_PRIOR_MAX_GRID) partial rows the single-tensor launch would use;
//   * step: every workgroup first adds its slot's rows in the order grad_norm_finalize_kernel adds them (thread t: rows t, t + 256,
//     ...; then the block sum) -- the rows are constants of this launch, so no workgroup waits for another, no ticket, no atomic --
//     and workgroup 0 of the slot publishes the norm in state[BH_STATE_GNORM + s].  Bit-identical to the per-tensor launches.
struct StepList {
  bh_step_slot slot[BH_STEP_MAX_SLOTS];
  int32_t first_block[BH_STEP_MAX_SLOTS + 1];  // workgroups [first_block[s], first_block[s + 1]) work on slot s
  int32_t row_begin[BH_STEP_MAX_SLOTS + 1];    // partial rows of slot s in the norm workspace
  int32_t n_slots;
};

__device__ __forceinline__ int slot_of_block(const StepList& L, int block) {
  int s = 0;
#pragma unroll
  for (int k = 1; k < BH_STEP_MAX_SLOTS; ++k)
    if (k < L.n_slots && block >= L.first_block[k]) s = k;
  return s;
}

__global__ __launch_bounds__(kBlock) void grad_sumsq_list_kernel(const Word* __restrict__ st, StepList L,
                                                                 const double* __restrict__ sched, double* __restrict__ ws) {
  __shared__ double lds[bh::kWavesPerBlock];
  const int s = slot_of_block(L, blockIdx.x);
  const bh_step_slot& S = L.slot[s];
  const int row = blockIdx.x - L.first_block[s], rows = L.first_block[s + 1] - L.first_block[s];
  const int it = st[BH_STATE_IT].i;
  const float noise_coef = S.noise ? (float)((double)S.params.langevin * sched[(int64_t)it * BH_SCHED_STRIDE + 3]) : 0.f;
  float acc = 0.f;
  double dacc = 0.0;
  int cnt = 0;
  for (int64_t i = (int64_t)row * kBlock + threadIdx.x; i < S.params.n; i += (int64_t)rows * kBlock) {
    const float v = effective_grad(S.g, S.g_reg, S.noise, noise_coef, i);
    acc = fmaf(v, v, acc);
    if (++cnt == 32) {
      dacc += (double)acc;
      acc = 0.f;
      cnt = 0;
    }
  }
  double v[1] = {dacc + (double)acc};
  bh::block_sum<1>(v, lds);
  if (threadIdx.x == 0) ws[L.row_begin[s] + row] = v[0];
}

__global__ __launch_bounds__(kBlock) void candidate_step_list_kernel(Word* __restrict__ st, const double* __restrict__ sched,
                                                                     StepList L, const double* __restrict__ ws) {
  __shared__ double lds[bh::kWavesPerBlock];
  __shared__ float gnorm_lds;
  const int s = slot_of_block(L, blockIdx.x);
  const bh_step_slot& S = L.slot[s];
  const bh_step_params& P = S.params;
  const int local = blockIdx.x - L.first_block[s], blocks = L.first_block[s + 1] - L.first_block[s];
  float gn = -1.f;
  if (P.grad_clip >= 0.f) {  // uniform per workgroup: every thread takes the barriers
    double v[1] = {0.0};
    for (int i = L.row_begin[s] + (int)threadIdx.x; i < L.row_begin[s + 1]; i += kBlock) v[0] += ws[i];  // grad_norm_finalize_kernel's order
    bh::block_sum<1>(v, lds);
    if (threadIdx.x == 0) gnorm_lds = (float)sqrt(v[0]);
    __syncthreads();
    gn = gnorm_lds;
    if (local == 0 && threadIdx.x == 0) st[BH_STATE_GNORM + s].f = gn;  // for observers; nothing in this launch reads it
  }
  const StepScalars k = step_scalars(st, sched, P, S.noise != nullptr, gn);
  const uintptr_t align = reinterpret_cast<uintptr_t>(S.x) | reinterpret_cast<uintptr_t>(S.g) | reinterpret_cast<uintptr_t>(S.m) |
                          reinterpret_cast<uintptr_t>(S.v) | reinterpret_cast<uintptr_t>(S.best);
  const bool vec4 = S.g_reg == nullptr && S.noise == nullptr && (P.n & 3) == 0 && (align & 15u) == 0 && (!P.boxed || (P.plane & 3) == 0);
  if (vec4) {  // candidate_step_vec4_kernel<false, false>'s body
    const int64_t n4 = P.n >> 2, plane4 = P.plane >> 2;
    float4* __restrict__ x4 = reinterpret_cast<float4*>(S.x);
    float4* __restrict__ m4 = reinterpret_cast<float4*>(S.m);
    float4* __restrict__ v4 = reinterpret_cast<float4*>(S.v);
    float4* __restrict__ b4 = reinterpret_cast<float4*>(S.best);
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(S.g);
    for (int64_t i = (int64_t)local * kBlock + threadIdx.x; i < n4; i += (int64_t)blocks * kBlock) {
      const float4 gv = g4[i];
      float4 xv = x4[i], mv = m4[i], vv = v4[i];
      const int c = P.boxed ? (int)((i / plane4) % P.channels) : 0;
      const float lo = P.lo[c], hi = P.hi[c];
      step_elem(k, P, gv.x, lo, hi, xv.x, mv.x, vv.x);
      step_elem(k, P, gv.y, lo, hi, xv.y, mv.y, vv.y);
      step_elem(k, P, gv.z, lo, hi, xv.z, mv.z, vv.z);
      step_elem(k, P, gv.w, lo, hi, xv.w, mv.w, vv.w);
      x4[i] = xv;
      m4[i] = mv;
      v4[i] = vv;
      if (k.improved) b4[i] = xv;
    }
  } else {  // candidate_step_kernel's body
    for (int64_t i = (int64_t)local * kBlock + threadIdx.x; i < P.n; i += (int64_t)blocks * kBlock) {
      const float gr = effective_grad(S.g, S.g_reg, S.noise, k.noise_coef, i);
      float xi = S.x[i], mi = S.m[i], vi = S.v[i];
      const int c = P.boxed ? (int)((i / P.plane) % P.channels) : 0;
      step_elem(k, P, gr, P.lo[c], P.hi[c], xi, mi, vi);
      S.x[i] = xi;
      S.m[i] = mi;
      S.v[i] = vi;
      if (k.improved) S.best[i] = xi;
    }
  }
}

int step_slot_ok(const bh_step_slot& S) {
  const bh_step_params& P = S.params;
  if (S.x == nullptr || S.g == nullptr || S.m == nullptr || S.v == nullptr || S.best == nullptr) return 0;
  if (P.n <= 0 || P.max_iterations <= 0) return 0;
  if (P.boxed && (P.channels <= 0 || P.channels > 4 || P.plane <= 0)) return 0;
  if (P.langevin > 0.f && S.noise == nullptr) return 0;
  if (P.sign_mode < BH_SIGN_NONE || P.sign_mode > BH_SIGN_SOFT) return 0;
  return 1;
}

int norm_rows(int64_t n) {  // bh_grad_norm's grid
  int64_t blocks = (n + (int64_t)kBlock * 8 - 1) / ((int64_t)kBlock * 8);
  if (blocks > BH_PRIOR_MAX_GRID) blocks = BH_PRIOR_MAX_GRID;
  return (int)(blocks < 1 ? 1 : blocks);
}

// Fills L from the caller's slots; returns 0 or BH_EINVAL.  `for_norm`: workgroups = partial rows; else kernel B's grid per slot.
int build_step_list(int32_t n_slots, const bh_step_slot* slots, bool for_norm, StepList& L) {
  if (slots == nullptr || n_slots <= 0 || n_slots > BH_STEP_MAX_SLOTS) return BH_EINVAL;
  L.n_slots = n_slots;
  L.first_block[0] = L.row_begin[0] = 0;
  for (int s = 0; s < BH_STEP_MAX_SLOTS; ++s) {
    if (s < n_slots) {
      if (!step_slot_ok(slots[s])) return BH_EINVAL;
      L.slot[s] = slots[s];
      if (L.slot[s].params.langevin <= 0.f) L.slot[s].noise = nullptr;
      const bh_step_params& P = slots[s].params;
      const int rows = P.grad_clip >= 0.f ? norm_rows(P.n) : 0;  // a slot without clipping has no rows and no norm workgroups
      int64_t blocks = rows;
      if (!for_norm) {
        const bool quads = (P.n & 3) == 0;  // upper bound of the workgroups the slot can use; the kernel re-derives the access width
        blocks = ((quads ? P.n >> 2 : P.n) + kBlock - 1) / kBlock;
        if (blocks > 2048) blocks = 2048;
      }
      L.first_block[s + 1] = L.first_block[s] + (int)blocks;
      L.row_begin[s + 1] = L.row_begin[s] + rows;
    } else {
      L.first_block[s + 1] = L.first_block[s];
      L.row_begin[s + 1] = L.row_begin[s];
    }
  }
  return 0;
}

}  // namespace

extern "C" {

int bh_state_reset(void* state_dev, void* stream) {
  if (state_dev == nullptr) return BH_EINVAL;
  hipLaunchKernelGGL(state_reset_kernel, dim3(1), dim3(64), 0, bh::as_stream(stream), static_cast<Word*>(state_dev));
  return bh::launch_status();
}

int bh_loss_commit(void* state_dev, float* history_dev, int32_t history_len, const float* gm_loss,
                   const double* reg_partials, int32_t n_reg, const float* extra0, const float* extra1, void* stream) {
  if (state_dev == nullptr || history_len < 0 || (history_len > 0 && history_dev == nullptr) || n_reg < 0 ||
      (n_reg > 0 && reg_partials == nullptr))
    return BH_EINVAL;
  hipLaunchKernelGGL(loss_commit_kernel, dim3(1), dim3(kBlock), 0, bh::as_stream(stream), static_cast<Word*>(state_dev),
                     history_dev, history_len, gm_loss, reg_partials, n_reg, extra0, extra1);
  return bh::launch_status();
}

int bh_grad_norm(void* state_dev, const float* g, const float* g_reg, const float* noise, int64_t n,
                 const double* sched_dev, float langevin, double* ws_dev, void* stream) {
  if (state_dev == nullptr || g == nullptr || n <= 0 || ws_dev == nullptr || (noise != nullptr && sched_dev == nullptr))
    return BH_EINVAL;
  int64_t blocks = (n + (int64_t)kBlock * 8 - 1) / ((int64_t)kBlock * 8);
  if (blocks > BH_PRIOR_MAX_GRID) blocks = BH_PRIOR_MAX_GRID;
  if (blocks < 1) blocks = 1;
  hipStream_t st = bh::as_stream(stream);
  hipLaunchKernelGGL(grad_sumsq_kernel, dim3((int)blocks), dim3(kBlock), 0, st, static_cast<const Word*>(state_dev), g,
                     g_reg, noise, n, sched_dev, langevin, ws_dev);
  int rc = bh::launch_status();
  if (rc != 0) return rc;
  hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(kBlock), 0, st, static_cast<Word*>(state_dev), ws_dev,
                     (int)blocks);
  return bh::launch_status();
}

int bh_candidate_step(const void* state_dev, const double* sched_dev, const bh_step_params* params, float* x,
                      const float* g, const float* g_reg, const float* noise, float* m, float* v, float* best,
                      void* stream) {
  if (state_dev == nullptr || sched_dev == nullptr || params == nullptr || x == nullptr || g == nullptr ||
      m == nullptr || v == nullptr || best == nullptr)
    return BH_EINVAL;
  const bh_step_params& P = *params;
  if (P.n <= 0 || P.max_iterations <= 0) return BH_EINVAL;
  if (P.boxed && (P.channels <= 0 || P.channels > 4 || P.plane <= 0)) return BH_EINVAL;
  if (P.langevin > 0.f && noise == nullptr) return BH_EINVAL;
  if (P.sign_mode < BH_SIGN_NONE || P.sign_mode > BH_SIGN_SOFT) return BH_EINVAL;
  if (P.langevin <= 0.f) noise = nullptr;
  const uintptr_t align = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(g_reg) |
                          reinterpret_cast<uintptr_t>(noise) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                          reinterpret_cast<uintptr_t>(best);
  const bool vec4 = (P.n & 3) == 0 && (align & 15u) == 0 && (!P.boxed || (P.plane & 3) == 0);
  const int64_t units = vec4 ? P.n >> 2 : P.n;
  int64_t blocks = (units + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
#define BH_STEP_VEC4(HAS_REG, HAS_NOISE)                                                                                \
  hipLaunchKernelGGL((candidate_step_vec4_kernel<HAS_REG, HAS_NOISE>), dim3((int)blocks), dim3(kBlock), 0,             \
                     bh::as_stream(stream), static_cast<const Word*>(state_dev), sched_dev, P, x, g, g_reg, noise, m, v, best)
  if (vec4 && g_reg && noise) BH_STEP_VEC4(true, true);
  else if (vec4 && g_reg) BH_STEP_VEC4(true, false);
  else if (vec4 && noise) BH_STEP_VEC4(false, true);
  else if (vec4) BH_STEP_VEC4(false, false);
#undef BH_STEP_VEC4
  else
    hipLaunchKernelGGL(candidate_step_kernel, dim3((int)blocks), dim3(kBlock), 0, bh::as_stream(stream),
                       static_cast<const Word*>(state_dev), sched_dev, P, x, g, g_reg, noise, m, v,
                       best);
  return bh::launch_status();
}

int32_t bh_step_list_norm_rows(int32_t n_slots, const bh_step_slot* slots) {
  StepList L;
  const int rc = build_step_list(n_slots, slots, true, L);
  return rc != 0 ? rc : L.row_begin[BH_STEP_MAX_SLOTS];
}

int bh_grad_norm_list(const void* state_dev, int32_t n_slots, const bh_step_slot* slots, const double* sched_dev, double* ws_dev,
                      void* stream) {
  StepList L;
  const int rc = build_step_list(n_slots, slots, true, L);
  if (rc != 0) return rc;
  if (state_dev == nullptr || sched_dev == nullptr) return BH_EINVAL;
  const int grid = L.first_block[BH_STEP_MAX_SLOTS];
  if (grid == 0) return 0;  // no slot clips: nothing to launch
  if (ws_dev == nullptr) return BH_EINVAL;
  hipLaunchKernelGGL(grad_sumsq_list_kernel, dim3(grid), dim3(kBlock), 0, bh::as_stream(stream), static_cast<const Word*>(state_dev), L,
                     sched_dev, ws_dev);
  return bh::launch_status();
}

int bh_candidate_step_list(void* state_dev, const double* sched_dev, int32_t n_slots, const bh_step_slot* slots, const double* ws_dev,
                           void* stream) {
  StepList L;
  const int rc = build_step_list(n_slots, slots, false, L);
  if (rc != 0) return rc;
  if (state_dev == nullptr || sched_dev == nullptr || (L.row_begin[BH_STEP_MAX_SLOTS] > 0 && ws_dev == nullptr)) return BH_EINVAL;
  hipLaunchKernelGGL(candidate_step_list_kernel, dim3(L.first_block[BH_STEP_MAX_SLOTS]), dim3(kBlock), 0, bh::as_stream(stream),
                     static_cast<Word*>(state_dev), sched_dev, L, ws_dev);
  return bh::launch_status();
}

int64_t bh_trial_key(float score, int32_t trial) {
  union {
    float f;
    uint32_t u;
  } bits;
  bits.f = score;
  const int64_t t = (int64_t)(uint32_t)trial;
  if (score != score || score == __builtin_inff()) return ((int64_t)0x7F800000 << 32) | t;
  if (score < 0.f) {  // below every non-negative key, relative order kept
    bits.f = -score;
    return -(((int64_t)bits.u << 32) | (int64_t)(0xFFFFFFFFu - (uint32_t)trial));
  }
  return ((int64_t)bits.u << 32) | t;
}

int bh_trial_key_unpack(int64_t key, float* score_out, int32_t* trial_out) {
  if (score_out == nullptr || trial_out == nullptr) return BH_EINVAL;
  union {
    float f;
    uint32_t u;
  } bits;
  if (key < 0) {
    const int64_t mag = -key;
    bits.u = (uint32_t)(mag >> 32);
    *score_out = -bits.f;
    *trial_out = (int32_t)(0xFFFFFFFFu - (uint32_t)(mag & 0xFFFFFFFF));
    return 0;
  }
  bits.u = (uint32_t)(key >> 32);
  *score_out = bits.f;
  *trial_out = (int32_t)(key & 0xFFFFFFFF);
  return 0;
}

int bh_event_create(void** event_out) {
  if (event_out == nullptr) return BH_EINVAL;
  hipEvent_t ev = nullptr;
  const int rc = bh::hip_status(hipEventCreate(&ev));
  *event_out = rc == 0 ? static_cast<void*>(ev) : nullptr;
  return rc;
}

int bh_event_destroy(void* event) {
  if (event == nullptr) return BH_EINVAL;
  return bh::hip_status(hipEventDestroy(static_cast<hipEvent_t>(event)));
}

int bh_event_record(void* event, void* stream) {
  if (event == nullptr) return BH_EINVAL;
  return bh::hip_status(hipEventRecord(static_cast<hipEvent_t>(event), bh::as_stream(stream)));
}

int bh_event_elapsed_ms(void* start, void* stop, float* ms_out) {
  if (start == nullptr || stop == nullptr || ms_out == nullptr) return BH_EINVAL;
  int rc = bh::hip_status(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
  if (rc != 0) return rc;
  return bh::hip_status(hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)));
}

int32_t bh_abi_version(void) { return BH_ABI_VERSION; }

const char* bh_build_arch(void) { return "gfx950"; }

}  // extern "C"
