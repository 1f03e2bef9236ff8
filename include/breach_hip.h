/*
 * breach_hip.h -- C ABI of libbreach_hip.so: the MI355X (gfx950) kernels behind the optimisation-based
 * gradient-inversion hot path of JonasGeiping/breaching.
 *
 * Every entry point
 *   - takes raw device pointers, plain sizes and an opaque hipStream_t (passed as void*),
 *   - enqueues work on that stream and returns immediately (no implicit synchronisation),
 *   - allocates nothing: every workspace is supplied by the caller,
 *   - returns 0 on success, BH_EINVAL (-1) for an invalid argument, or -(1000 + hipError_t) when the HIP
 *     runtime rejected the launch.  Nothing throws.
 *
 * The reference has no native code at all; each function below names the Python it
 * replaces as `reference: <file>:<lines>` relative to the reference checkout.
 *
 * All arithmetic is fp32 on fp32 storage (reference `impl.dtype: float`,
 * breaching/config/attack/_default_optimization_attack.yaml:40-43); reductions accumulate in fp64 and are
 * combined in a fixed order, so results are bitwise reproducible run to run for a fixed launch geometry.
 */
#ifndef BREACH_HIP_H
#define BREACH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BH_ABI_VERSION 3
#define BH_EINVAL (-1)

/* ------------------------------------------------------------------------------------------------------------------
 * Gradient-matching reduction over a per-parameter gradient list ("kernel A")
 * ---------------------------------------------------------------------------------------------------------------- */

/* Elements handled by one work item of the chunk table.  A tensor of n elements contributes ceil(n/BH_GM_CHUNK)
 * chunks; chunk starts are multiples of BH_GM_CHUNK inside their tensor, hence 16-byte aligned whenever the
 * tensor base is. */
#ifndef BH_GM_CHUNK
#define BH_GM_CHUNK 4096
#endif
/* Device pointers per launch that travel in the kernel-argument segment.  Longer lists are processed in several
 * launches by the library (transparent to the caller). */
#define BH_GM_MAX_PTRS 448
/* Doubles per row of the partial-sum workspace (one row per chunk). */
#define BH_GM_PARTIAL_STRIDE 4

/* One entry of the chunk table (24 bytes, device resident, built once per attack by bh_gm_build_table). */
typedef struct bh_gm_chunk {
  int64_t flat_off;   /* element offset of the chunk inside the packed flat buffers (multiple of 4) */
  int64_t tensor_off; /* element offset of the chunk inside its own tensor (multiple of BH_GM_CHUNK) */
  int32_t tensor;     /* index into the pointer list / per-tensor weight vector */
  int32_t len;        /* elements in this chunk, 1..BH_GM_CHUNK */
} bh_gm_chunk;

/* Objective selector.  reference: breaching/attacks/auxiliaries/objectives.py:496-506 (objective_lookup). */
enum bh_gm_kind {
  BH_GM_COSINE = 0,        /* CosineSimilarity._cosine_sim            objectives.py:183-196 */
  BH_GM_COSINE_MASKED = 1, /* MaskedCosineSimilarity (|d| > 1e-6)     objectives.py:233-244 */
  BH_GM_COSINE_FAST = 2,   /* FastCosineSimilarity (detached norms)   objectives.py:259-273 */
  BH_GM_ANGULAR = 3,       /* AngularSimilarity  acos(cos)/pi          objectives.py:210-214 */
  BH_GM_L2 = 4,            /* Euclidean._euclidean                     objectives.py:89-95   */
  BH_GM_L1 = 5,            /* L1Loss._l1loss                           objectives.py:158-166 */
  BH_GM_TAG = 6            /* EuclideanTag._weighted_euclidean_l1      objectives.py:133-141 */
};

/* Words of the statistics record written by bh_gm_finalize (fp32 each). */
enum bh_gm_stat {
  BH_GM_STAT_LOSS = 0, /* objective value, already multiplied by `scale` */
  BH_GM_STAT_C1 = 1,   /* backward coefficient 1 (see bh_gm_bwd) */
  BH_GM_STAT_C2 = 2,   /* backward coefficient 2 */
  BH_GM_STAT_S0 = 3,   /* raw sums: cosine family <r,d>, |r|^2, |d|^2 ; L2 family  sum (r-d)^2, sum |r-d|, - */
  BH_GM_STAT_S1 = 4,
  BH_GM_STAT_S2 = 5,
  BH_GM_STAT_SPAN_TICKS = 6, /* wall-clock ticks between the first block entering and the last block leaving bh_gm_fwd */
  BH_GM_STAT_WORDS = 8
};

/* Host-side helper: size the chunk table for a list of `n_tensors` tensors with `numel[i]` elements.
 * Writes the number of chunks and the number of elements of the packed flat layout (each tensor start rounded up to
 * 4 elements).  Pure host arithmetic, no HIP call. */
int bh_gm_table_size(int32_t n_tensors, const int64_t* numel, int64_t* n_chunks, int64_t* flat_elems);

/* Host-side helper: fill `chunks[n_chunks]` and `tensor_flat_off[n_tensors]` (host memory) for the same list.
 * The caller uploads `chunks` to the device once.  reference: replaces the Python `zip(gradient_rec,
 * gradient_data)` loop of objectives.py:190-193 -- pairs beyond the shorter list are dropped by the caller. */
int bh_gm_build_table(int32_t n_tensors, const int64_t* numel, bh_gm_chunk* chunks, int64_t n_chunks,
                      int64_t* tensor_flat_off);

/* Forward partial sums.  `rec_ptrs` is a HOST array of `n_tensors` DEVICE pointers (the tensors returned by
 * autograd this iteration, each contiguous fp32 and 16-byte aligned); `data_flat` is the packed observed gradient;
 * `chunks_dev` the device chunk table; `weights_dev` per-tensor fp32 weights (BH_GM_TAG only, else NULL).
 * `partials_dev` (32-byte aligned) must hold n_chunks rows of BH_GM_PARTIAL_STRIDE doubles; row c belongs to chunk c and is
 * overwritten, never accumulated.  `group_chunk_begin` (HOST, bh_gm_num_groups+1 entries, from bh_gm_group_bounds)
 * delimits the chunks of every launch group.
 * reference: objectives.py:89-95, 133-141, 158-166, 183-196, 233-244, 259-273 (the list reductions). */
int bh_gm_fwd(int32_t kind, int32_t n_tensors, const void* const* rec_ptrs, const float* data_flat,
              const bh_gm_chunk* chunks_dev, int64_t n_chunks, const int32_t* group_chunk_begin,
              const float* weights_dev, float tag_scale, double* partials_dev, void* stream, void* ev_start,
              void* ev_stop);
/* `ev_start` / `ev_stop` (here and in bh_gm_bwd): optional hipEvent_t handles from bh_event_create.  When given, the
 * launch goes through hipExtLaunchKernelGGL, so the events carry the dispatch's own begin / end timestamps (the
 * completion-signal times rocprofv3 reports) -- no host latency, no marker overhead.  Not usable during stream
 * capture.  NULL = plain launch. */

/* Number of launch groups for a list of n_tensors pointers: ceil(n_tensors / BH_GM_MAX_PTRS). */
int32_t bh_gm_num_groups(int32_t n_tensors);
/* Host helper: fill group_chunk_begin[bh_gm_num_groups+1] -- first chunk of every launch group (host chunk table). */
int bh_gm_group_bounds(int32_t n_tensors, const bh_gm_chunk* chunks_host, int64_t n_chunks,
                       int32_t* group_chunk_begin);

/* Fixed-order combine of the partial sums and objective epilogue: writes BH_GM_STAT_WORDS floats to `stats_dev`.
 * `fudge` is AngularSimilarity's clamp margin (objectives.py:208, 1e-7); ignored otherwise.
 * reference: objectives.py:95 (0.5*objective), :141, :166, :195, :211-214, :243, :271 and the `* self.scale`
 * at :86, :126, :155, :178. */
int bh_gm_finalize(int32_t kind, const double* partials_dev, int64_t n_rows, float scale, float tag_scale, float fudge,
                   float* stats_dev, double* span_accum_dev, void* stream);
/* Every forward workgroup stamps the constant-rate device wall clock (bh_wall_clock_khz) on entry and exit into the
 * spare word of its partial row; the finalize kernel reduces them to the launch's span (stats[BH_GM_STAT_SPAN_TICKS])
 * and, when `span_accum_dev` is non-NULL, adds it to span_accum_dev[0] and 1 to span_accum_dev[1] (doubles).  This is how
 * bench.py times kernel A *inside* hipGraph replays, where host-visible event pairs cannot be placed. */
int32_t bh_wall_clock_khz(void);

/* Backward: d objective / d rec_i for every tensor, written into the packed layout `grad_flat` (same offsets as
 * data_flat), multiplied by the upstream scalar *gout_dev (NULL means 1).
 *   cosine family: out = gout * (C1 * d + C2 * r)            [masked variant: zero where |d| <= 1e-6]
 *   L2 family:     out = gout * (C1 * (r - d) + C2 * w_t * sign(r - d))
 * reference: the autograd graph of the functions listed at bh_gm_fwd. */
int bh_gm_bwd(int32_t kind, int32_t n_tensors, const void* const* rec_ptrs, const float* data_flat,
              const bh_gm_chunk* chunks_dev, int64_t n_chunks, const int32_t* group_chunk_begin,
              const float* weights_dev, const float* stats_dev, const float* gout_dev, float* grad_flat, void* stream,
              void* ev_start, void* ev_stop);

/* Pack a list of device tensors into the flat layout (used once per attack for the observed gradient).
 * reference: base_attack.py:214-220 (_cast_shared_data keeps a list; we keep one packed copy). */
int bh_gm_pack(int32_t n_tensors, const void* const* src_ptrs, const bh_gm_chunk* chunks_dev, int64_t n_chunks,
               const int32_t* group_chunk_begin, float* flat_dst, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Image priors ("kernel C"): TotalVariation (+ optional L^p norm penalty) value and analytic gradient in one pass
 * ---------------------------------------------------------------------------------------------------------------- */

#define BH_PRIOR_MAX_GRID 1024
#define BH_PRIOR_PARTIAL_STRIDE 2 /* doubles per workgroup: [tv, norm] already scaled */

/* x: [B,3,H,W] contiguous fp32.  Computes
 *   tv   = tv_scale * mean_{b,g,i,j} ( (|dv|+eps)^p + (|dh|+eps)^p )^q    over g = 3 planes, or 6 with
 *          double_opponents (planes R-G, R-B, G-B appended), dv/dh forward differences with zero extension
 *   norm = norm_scale / norm_p * mean(x^norm_p)                               (skipped when norm_scale == 0)
 * and writes d(tv+norm)/dx into grad_out[B,3,H,W] (overwritten) plus per-workgroup partial values into
 * partials_dev[grid * 2].  Returns the grid size used (>0) or a negative error.
 * reference: regularizers.py:103-153 (TotalVariation), :184-200 (NormRegularization). */
int bh_prior_tv_norm(const float* x, int32_t B, int32_t H, int32_t W, float tv_scale, float inner_exp, float outer_exp,
                     float eps, int32_t double_opponents, float norm_scale, float norm_p, float* grad_out,
                     double* partials_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * DeepInversion batch-norm statistics prior ("kernel D")
 * ---------------------------------------------------------------------------------------------------------------- */

/* Slabs per channel used by bh_bnstat_sums for this shape (>= 1); sizes the workspace below. */
int32_t bh_bnstat_slabs(int32_t B, int32_t C, int64_t HW);

/* Per-channel sum and sum of squares of x[B,C,HW] over (b, hw), split in S = bh_bnstat_slabs(...) slabs per channel:
 * sums_dev[C * S * 2] doubles (overwritten).  Returns S (> 0) or a negative error.
 * reference: deepinversion.py:93-96 (mean / biased var of the BN input). */
int bh_bnstat_sums(const float* x, int32_t B, int32_t C, int64_t HW, double* sums_dev, void* stream);

/* From the sums: mean_c, var_c (biased) and r = |running_var - var|_2 + |running_mean - mean|_2 -> value_dev[0]
 * (fp32); also the backward coefficients coef_dev[2*C] (fp32): dr/dx[b,c,hw] = A_c + B_c * x[b,c,hw].
 * reference: deepinversion.py:96-101. */
int bh_bnstat_finalize(const double* sums_dev, int32_t B, int32_t C, int64_t HW, const float* running_mean,
                       const float* running_var, float* value_dev, float* coef_dev, double* scratch_dev /* [2*C] */,
                       void* stream);

/* grad_x = gout * (A_c + B_c * x); gout read from *gout_dev.  grad_x is overwritten. */
int bh_bnstat_bwd(const float* x, int32_t B, int32_t C, int64_t HW, const float* coef_dev, const float* gout_dev,
                  float* grad_x, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Trial state, loss commit and the fused candidate step ("kernel B")
 * ---------------------------------------------------------------------------------------------------------------- */

/* 32-bit words of the per-trial device state record (caller allocates BH_STATE_WORDS words, zero-initialised by
 * bh_state_reset). */
enum bh_state_word {
  BH_STATE_IT = 0,        /* int32: index of the iteration being processed, -1 before the first commit */
  BH_STATE_DEAD = 1,      /* int32: 1 once a non-finite objective was seen */
  BH_STATE_FIRST_BAD = 2, /* int32: iteration of the first non-finite objective, -1 if none */
  BH_STATE_IMPROVED = 3,  /* int32: 1 when this iteration's objective beat the best so far */
  BH_STATE_MIN = 4,       /* fp32: minimal objective so far (+inf initially) */
  BH_STATE_TOTAL = 5,     /* fp32: total objective of this iteration */
  BH_STATE_GNORM = 6,     /* fp32: L2 norm of the (noise-perturbed) candidate gradient, when clipping */
  BH_STATE_WORDS = 16
};

int bh_state_reset(void* state_dev, void* stream);

/* Advance the iteration counter and commit this iteration's objective:
 *   total = [*gm_loss] + sum(reg_partials[0..n_reg)) + [*extra0] + [*extra1]      (NULL terms skipped)
 *   history[it] = total ; improved = !dead && total < min ; min = improved ? total : min ;
 *   if !isfinite(total) and !dead: dead = 1, first_bad = it.
 * reference: optimization_based_attack.py:119-121 (best tracking), :131-135 (isfinite / stats append). */
int bh_loss_commit(void* state_dev, float* history_dev, int32_t history_len, const float* gm_loss,
                   const double* reg_partials, int32_t n_reg, const float* extra0, const float* extra1, void* stream);

/* Sum of squares of the effective gradient (g + g_reg + noise_coef[it] * noise) -> state[BH_STATE_GNORM] = sqrt(.)
 * Needs workspace ws_dev[BH_PRIOR_MAX_GRID] doubles.
 * reference: optimization_based_attack.py:171-172 (candidate.grad.norm()). */
int bh_grad_norm(void* state_dev, const float* g, const float* g_reg, const float* noise, int64_t n,
                 const double* sched_dev, float langevin, double* ws_dev, void* stream);

/* Per-iteration schedule row: 4 doubles {lr / bias_correction1, sqrt(bias_correction2), 1 - lr*weight_decay, lr}. */
#define BH_SCHED_STRIDE 4

enum bh_sign_mode { BH_SIGN_NONE = 0, BH_SIGN_HARD = 1, BH_SIGN_SOFT = 2 };

typedef struct bh_step_params {
  int64_t n;            /* elements of the candidate */
  int64_t plane;        /* H*W: elements per channel plane (box bounds are per channel) */
  int32_t channels;     /* C (<= 4 for per-channel box bounds; bounds of channel c = lo[c], hi[c]) */
  int32_t boxed;        /* clamp to [lo, hi] after the step                optimization_based_attack.py:117-118 */
  int32_t sign_mode;    /* bh_sign_mode                                    optimization_based_attack.py:175-184 */
  int32_t max_iterations; /* for the soft-sign factor 1 - it/max_iterations */
  float lo[4];
  float hi[4];
  double beta1, beta2, eps; /* doubles: torch derives 1-beta in double before rounding to fp32 (1 - 0.999f is 1.3e-5 off) */
  int32_t decoupled_wd; /* AdamW: x *= sched[2] first                      common.py:10-12 */
  float langevin;       /* langevin_noise (0 = off); noise must be non-NULL when > 0   :167-170 */
  float grad_clip;      /* <= 0 = off; uses state[BH_STATE_GNORM]                       :171-174 */
} bh_step_params;

/* One fused elementwise pass: assemble the gradient (g + g_reg + langevin*lr*noise), clip, sign, Adam/AdamW moment
 * and parameter update, box projection, and -- when state.improved -- copy the projected candidate into `best`.
 * reference: optimization_based_attack.py:165-184 (grad post-processing), torch.optim.Adam/AdamW single-tensor
 * update selected at auxiliaries/common.py:5-12, optimization_based_attack.py:117-121. */
int bh_candidate_step(const void* state_dev, const double* sched_dev, const bh_step_params* params, float* x,
                      const float* g, const float* g_reg, const float* noise, float* m, float* v, float* best,
                      void* stream);

/* Timing helpers (thin wrappers over hipEvent*, used by bench.py for the roofline leg). */
int bh_event_create(void** event_out);
int bh_event_destroy(void* event);
int bh_event_record(void* event, void* stream);
/* Blocks until `stop` has completed, then writes the elapsed milliseconds between the two events. */
int bh_event_elapsed_ms(void* start, void* stop, float* ms_out);

/* Library / build introspection. */
int32_t bh_abi_version(void);
const char* bh_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* BREACH_HIP_H */
