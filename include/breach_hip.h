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

/* 6: multi-tensor launch groups are BH_MT_MAX_PTRS = 112 tensors (two adjacent groups per launch for the forms with at most two
 *    pointer lists), bh_mt_* write outputs larger than the Infinity Cache with non-temporal stores; 5: tuning knobs became launch
 *    arguments (no mutable library state). */
#define BH_ABI_VERSION 7
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
/* Doubles per row of the partial-sum workspace (one row per persistent workgroup of the forward launch). */
#define BH_GM_PARTIAL_STRIDE 4
/* Workgroups (= rows) of one forward launch group: upper bound of the tuning knob, and its default (two resident
 * 256-thread workgroups per CU on 256 CUs -- measured fastest, see gm_kernels.hip). */
#define BH_GM_MAX_ROWS 2048
#define BH_GM_DEFAULT_ROWS 512

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
  BH_GM_TAG = 6,           /* EuclideanTag._weighted_euclidean_l1      objectives.py:133-141 */
  BH_GM_PEARL_L2 = 7       /* PearlmutterEuclidean residual: 0.5 sum (g-d)^2 plus |g|^2   objectives.py:451-460 */
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
  /* Pearlmutter finite differences (fd_eps > 0, cosine kinds and BH_GM_PEARL_L2), objectives.py:343-356: */
  BH_GM_STAT_PATCH_D = 8,   /* eps_n * coefficient of `data` in the first-order direction (read by bh_mt_patch) */
  BH_GM_STAT_PATCH_R = 9,   /* eps_n * coefficient of `grad` */
  BH_GM_STAT_FD_STEP = 10,  /* eps_n = fd_eps / |grad|_2 */
  BH_GM_STAT_FD_SCALE = 11, /* scale / eps_n: multiplies (dL_offset/dx - dL/dx) */
  BH_GM_STAT_WORDS = 12
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

/* Forward reduction.  `rec_ptrs` is a HOST array of `n_tensors` DEVICE pointers (the tensors returned by
 * autograd this iteration, each contiguous fp32 and 16-byte aligned); `data_flat` is the packed observed gradient;
 * `chunks_dev` the device chunk table; `weights_dev` per-tensor fp32 weights (BH_GM_TAG only, else NULL).
 * The launch is a persistent grid: bh_gm_fwd_rows() workgroups, each streaming every G-th chunk and writing ONE row
 * of BH_GM_PARTIAL_STRIDE doubles into `partials_dev` (32-byte aligned, bh_gm_fwd_rows rows; overwritten, never
 * accumulated).  `group_chunk_begin` (HOST, bh_gm_num_groups+1 entries, from bh_gm_group_bounds) delimits the chunks
 * of every launch group.  The caller follows up with bh_gm_finalize(kind, partials_dev, bh_gm_fwd_rows(...), ...).
 * reference: objectives.py:89-95, 133-141, 158-166, 183-196, 233-244, 259-273 (the list reductions). */
int bh_gm_fwd(int32_t kind, int32_t n_tensors, const void* const* rec_ptrs, const float* data_flat,
              const bh_gm_chunk* chunks_dev, int64_t n_chunks, const int32_t* group_chunk_begin,
              const float* weights_dev, float tag_scale, double* partials_dev, int32_t rows_cap, int32_t cache_policy,
              void* stream, void* ev_start, void* ev_stop);
/* `cache_policy` (here and in bh_gm_bwd): how the launch's streaming accesses treat the caches.  Both lists are read once per
 * launch; behind a producer whose dirty lines are still draining (autograd in the attack loop) plain loads of a list that does not
 * fit the 256 MiB Infinity Cache lose a third of their rate, non-temporal loads do not (BERT-base forward: 131.5 -> 94.6 us =
 * 0.65 -> 0.91 of peak); a list that fits wants plain loads (the backward re-reads what the forward pulled in). */
#define BH_GM_CACHE_AUTO 0        /* by the forward's bytes B = 2 x list bytes: forward STREAM when B > BH_GM_CACHE_AUTO_FWD_BYTES,
                                     backward STREAM_ALL when B > BH_GM_CACHE_AUTO_BYTES, else KEEP (measured in the attack loop) */
#define BH_GM_CACHE_KEEP 1        /* plain loads / stores */
#define BH_GM_CACHE_STREAM 2      /* non-temporal loads */
#define BH_GM_CACHE_STREAM_ALL 3  /* non-temporal loads and (backward) stores */
#define BH_GM_CACHE_AUTO_FWD_BYTES (128ll << 20)
#define BH_GM_CACHE_AUTO_BYTES (256ll << 20)
/* Rows (= workgroups over all launch groups) bh_gm_fwd writes for this list: chunks are dealt out evenly to at most
 * `rows_cap` workgroups per launch group (1..BH_GM_MAX_ROWS; 0 = BH_GM_DEFAULT_ROWS).  The cap is an argument of this
 * call and of bh_gm_fwd (pass the same value to both): the library holds no mutable tuning state, so two host threads
 * may use different caps.  Negative on invalid arguments. */
int32_t bh_gm_fwd_rows(int32_t n_tensors, const int32_t* group_chunk_begin, int32_t rows_cap);
/* `ev_start` / `ev_stop` (here and in bh_gm_bwd): optional hipEvent_t handles from bh_event_create.  When given, the
 * launch goes through hipExtLaunchKernelGGL, so the events carry the dispatch's own begin / end timestamps (the
 * completion-signal times rocprofv3 reports) -- no host latency, no marker overhead.  Not usable during stream
 * capture.  NULL = plain launch. */

/* Number of launch groups for a list of n_tensors pointers: ceil(n_tensors / BH_GM_MAX_PTRS). */
int32_t bh_gm_num_groups(int32_t n_tensors);
/* Host helper: fill group_chunk_begin[bh_gm_num_groups+1] -- first chunk of every launch group (host chunk table). */
int bh_gm_group_bounds(int32_t n_tensors, const bh_gm_chunk* chunks_host, int64_t n_chunks,
                       int32_t* group_chunk_begin);

/* Fixed-order combine of the partial rows and objective epilogue (one workgroup): writes BH_GM_STAT_WORDS floats to
 * `stats_dev`.
 * `fudge` is AngularSimilarity's clamp margin (objectives.py:208, 1e-7); ignored otherwise.
 * reference: objectives.py:95 (0.5*objective), :141, :166, :195, :211-214, :243, :271 and the `* self.scale`
 * at :86, :126, :155, :178. */
int bh_gm_finalize(int32_t kind, const double* partials_dev, int64_t n_rows, float scale, float tag_scale, float fudge,
                   float fd_eps, float* stats_dev, double* span_accum_dev, void* stream, void* ev_start, void* ev_stop);
/* `fd_eps` > 0 additionally fills the BH_GM_STAT_PATCH_* / FD_* words for the Pearlmutter objectives; 0 otherwise. */
/* Every forward workgroup stamps the constant-rate device wall clock (bh_wall_clock_khz) on entry and exit into the
 * spare word of its partial row; when `span_accum_dev` is non-NULL the finalize kernel reduces them to the launch's
 * span (stats[BH_GM_STAT_SPAN_TICKS]) and adds it to span_accum_dev[0] and 1 to span_accum_dev[1] (doubles).  This is
 * how bench.py times kernel A *inside* hipGraph replays, where host-visible event pairs cannot be placed; the product
 * path passes NULL (stats[BH_GM_STAT_SPAN_TICKS] = 0). */
int32_t bh_wall_clock_khz(void);

/* Backward: d objective / d rec_i for every tensor, written into the packed layout `grad_flat` (same offsets as
 * data_flat), multiplied by the upstream scalar *gout_dev (NULL means 1).
 *   cosine family: out = gout * (C1 * d + C2 * r)            [masked variant: zero where |d| <= 1e-6]
 *   L2 family:     out = gout * (C1 * (r - d) + C2 * w_t * sign(r - d))
 * reference: the autograd graph of the functions listed at bh_gm_fwd. */
int bh_gm_bwd(int32_t kind, int32_t n_tensors, const void* const* rec_ptrs, const float* data_flat,
              const bh_gm_chunk* chunks_dev, int64_t n_chunks, const int32_t* group_chunk_begin,
              const float* weights_dev, const float* stats_dev, const float* gout_dev, float* grad_flat, int32_t cache_policy,
              void* stream, void* ev_start, void* ev_stop);

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

/* The prior sums, over every BatchNorm2d layer l of the attacked model, a statistic of the layer's INPUT x_l[B,C,H*W]:
 *   r_l = | running_var - var_c(x_l) |_2 + | running_mean - mean_c(x_l) |_2     (biased variance over b, h, w)
 *   total = sum_l weight_l * r_l       (weight_l = scale * first_bn_multiplier for l = 0, scale otherwise)
 * reference: regularizers.py:203-230 (DeepInversion), deepinversion.py:93-101 (the per-layer hook; NVIDIA-NC licensed,
 * restated from the formula only).  All layers are processed by ONE launch per stage: the layer inputs travel as a
 * host array of device pointers (kernel arguments, at most BH_BN_MAX_LAYERS), their geometry in a device table. */
#define BH_BN_MAX_LAYERS 448
#define BH_BN_TILE 4096 /* elements per backward work item */
#define BH_BN_DEFAULT_GRID (1 << 20) /* cap of the forward grid; measured round 3 (profiles/r3_kernel_bench.json, 355.6 MB):
                                       * 512 / 1024 / 2048 / 4096 / uncapped workgroups = 111 / 74 / 69 / 64 / 62 us -- unlike
                                       * kernel A the item stream wants every workgroup it can get, so no cap by default */

typedef struct bh_bn_layer { /* 64 bytes, device resident, built once per attack by bh_bn_plan_build */
  int64_t flat_off;      /* element offset of the layer inside the packed gradient buffer (multiple of 4) */
  int64_t sums_off;      /* offset, in (sum, sum of squares) pairs, of its C x S partial sums */
  int32_t chan_off;      /* offset of its channels in the packed running statistics / coefficient arrays */
  int32_t B, C, HW;      /* x_l is [B, C, HW] contiguous fp32 (NCHW); 16-byte aligned when HW % 4 == 0 */
  int32_t S;             /* forward slabs per channel */
  int32_t narrow;        /* 1: B*HW is small, one wavefront per channel in the forward pass */
  float weight;          /* weight_l */
  uint32_t div_unit_mul, div_unit_shr; /* fast division by HW/4 (HW % 4 == 0) or HW */
  uint32_t div_c_mul, div_c_shr;       /* fast division by C */
  int32_t fwd_items;     /* forward items (stage 1) of this layer */
} bh_bn_layer;

typedef struct bh_bn_item { /* 16 bytes: forward (layer, channel [first of 4 when narrow], slab, -); */
  int32_t layer, a, b, c;   /*           backward (layer, first unit, units, -), unit = float4 or float */
} bh_bn_item;

/* Host arithmetic: table sizes for layers of shape [B[l], C[l], HW[l]]. */
int bh_bn_plan_size(int32_t n_layers, const int32_t* B, const int32_t* C, const int32_t* HW, int64_t* n_fwd_items,
                    int64_t* n_bwd_items, int64_t* flat_elems, int64_t* n_sum_pairs, int64_t* n_channels);
/* Host arithmetic: fill the three tables (host memory; the caller uploads them once).  `weights` may be NULL (= 1). */
int bh_bn_plan_build(int32_t n_layers, const int32_t* B, const int32_t* C, const int32_t* HW, const float* weights,
                     bh_bn_layer* layers, bh_bn_item* fwd_items, int64_t n_fwd_items, bh_bn_item* bwd_items,
                     int64_t n_bwd_items);

/* Stage 1: per-channel sum and sum of squares of every layer -- min(n_fwd_items, grid_cap) workgroups stream the forward
 * items -- into sums_dev[2 * n_sum_pairs] doubles (overwritten).  `x_ptrs` / `hw_host`: HOST arrays (device pointers, HW per
 * layer).  Tuning arguments of this launch (no library state): `grid_cap` 1..2^20 workgroups of the persistent grid
 * (0 = BH_BN_DEFAULT_GRID); `load_depth` 16-byte loads in flight per thread, 4 or 8 (0 = 8).
 * reference: deepinversion.py:93-96 (mean / biased var of the BN input). */
int bh_bn_sums(int32_t n_layers, const void* const* x_ptrs, const int32_t* hw_host, const bh_bn_layer* layers_dev,
               const bh_bn_item* fwd_items_dev, int64_t n_fwd_items, double* sums_dev, int32_t grid_cap, int32_t load_depth,
               void* stream);

#define BH_BN_DEFAULT_FINALIZE_BLOCK 1024

/* Stage 2 (one workgroup per layer): mean_c, var_c, r_l, the backward coefficients coef_dev[2 * n_channels] (fp32,
 * 8-byte aligned; d total / d x_l[b,c,hw] = A_c + B_c * x) and total_dev[0] = sum_l weight_l * r_l, added up in layer
 * order by the last workgroup to finish.  `running_mean` / `running_var`: packed per chan_off.  `layer_values_dev`:
 * n_layers doubles of workspace; `counter_dev`: a zeroed uint32 (re-zeroed by the kernel).  `block_threads`: threads of a
 * layer's workgroup, 256 / 512 / 1024 (0 = BH_BN_DEFAULT_FINALIZE_BLOCK) -- an argument of the launch, not library state.
 * reference: deepinversion.py:96-101, regularizers.py:222-227. */
int bh_bn_finalize(int32_t n_layers, const bh_bn_layer* layers_dev, const double* sums_dev, const float* running_mean,
                   const float* running_var, float* coef_dev, double* layer_values_dev, float* total_dev,
                   void* counter_dev, int32_t block_threads, void* stream);

/* Backward of all layers in one launch: grad_flat[flat_off_l + i] = gout * (A_c + B_c * x_l[i]); gout read from
 * *gout_dev (NULL = 1).  grad_flat (16-byte aligned, flat_elems floats) is overwritten. */
int bh_bn_bwd(int32_t n_layers, const void* const* x_ptrs, const int32_t* hw_host, const bh_bn_layer* layers_dev,
              const bh_bn_item* bwd_items_dev, int64_t n_bwd_items, const float* coef_dev, const float* gout_dev,
              float* grad_flat, void* stream);

/* Backward of ONE layer fused with the accumulation into the activation gradient: out[i] = gin[i] + gout * (A_c + B_c * x[i])
 * over the layer's n_items backward items (`layer_bwd_items_dev` points at the first of them inside the plan's table);
 * gin may be NULL (nothing else reached this activation), gout is read from *gout_dev (NULL = 1).  x / gin / out: the
 * layer's [B, C, HW] fp32 arrays (16-byte aligned when hw % 4 == 0); out may not alias x.  This is the read-modify-write
 * autograd otherwise does with one `add` per BatchNorm input (regularizers.py:222-227 summed into the main gradient). */
int bh_bn_bwd_accumulate(const float* x, const float* gin, int32_t hw, const bh_bn_layer* layers_dev,
                         const bh_bn_item* layer_bwd_items_dev, int64_t n_items, const float* coef_dev, const float* gout_dev,
                         float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Eval-mode BatchNorm of the attacker's private model copy, one launch per autograd order.
 * reference: base_attack.py:176-212 rebuilds the victim model and puts it in eval() whenever buffers are known; its
 * BatchNorm2d layers are then the per-channel affine map y = x * s_c + t_c (s_c = weight_c * inv_std_c,
 * t_c = bias_c - weight_c * mean_inv_c, inv_std = 1/sqrt(running_var + eps), mean_inv = running_mean * inv_std), which the
 * attack differentiates twice per iteration (objectives.py:40-46 with create_graph=True, optimization_based_attack.py:160).
 * x / y / gradients: [B, C, HW] contiguous fp32 (16-byte aligned when HW % 4 == 0); weight / bias may be NULL (affine=False);
 * per-channel sums in fp64, fixed order.  No allocation, no synchronisation. */
/* `stats` (may be NULL): 2 * C * S doubles receiving sum(x) and sum(x^2) per (channel, slab), S = bh_bn_eval_slabs -- the layout
 * bh_bn_finalize reads for a layer (point it at sums_dev + 2 * sums_off of that layer): the DeepInversion prior's statistics of
 * a BatchNorm input then come out of the pass that reads the input anyway, and bh_bn_sums is not needed. */
/* Epilogue (both optional): `residual` ([B,C,HW], NULL = none) is added to the affine map and `relu` (0 / 1) clamps the result at
 * zero -- y = relu(x * s_c + t_c + residual), the tail of a ResNet block in the BatchNorm's own launch. */
int bh_bn_eval_fwd(const float* x, const float* weight, const float* bias, const float* inv_std, const float* mean_inv, float* y,
                   double* stats, const float* residual, int32_t relu, int32_t B, int32_t C, int32_t HW, void* stream);
/* Slabs S a channel is cut into for this geometry -- the rule of bh_bn_plan_build (1 below 12 288 elements per channel: the
 * whole order is then ONE launch; otherwise about one per 8 192 elements, at most 64, and the backward orders take a small
 * second launch that adds the per-slab sums in slab order).  `workspace` below: 2 * C * S doubles (may be NULL when S == 1). */
int32_t bh_bn_eval_slabs(int32_t B, int32_t C, int32_t HW);
/* gx = gy * s_c (skipped when gx is NULL); gw_c = inv_std_c * sum(gy * x) - mean_inv_c * sum(gy); gb_c = sum(gy).
 * `tap_coef` (may be NULL; 8-byte aligned): the C (A_c, B_c) pairs bh_bn_finalize wrote for THIS layer (coef_dev + 2 * chan_off)
 * and `tap_gout` (device scalar, NULL = 1): the DeepInversion prior's backward of this BatchNorm input rides in the launch --
 * gx = gy * s_c + gout * (A_c + B_c * x) -- instead of a read-modify-write pass of its own (bh_bn_bwd_accumulate): x is read
 * here anyway, so the prior's backward costs no traffic on models whose BatchNorm runs through these kernels.
 * reference: deepinversion.py:93-103 (the statistic whose gradient this is; math only).
 * `y_mask` (may be NULL): the forward OUTPUT of a launch with relu = 1; the incoming gradient is then masked first, gz = gy * [y > 0],
 * and gz takes gy's place everywhere (gx, gw, gb).  `g_residual` (may be NULL): receives gz, the gradient of the residual input.
 * `gx_add` (may be NULL, [B,C,HW]): added to gx -- the OTHER gradient of the same BatchNorm input in the attack's outer pass (d_x of
 * bh_bn_eval_bwd_bwd), so that the framework's accumulation of the two needs no launch of its own. */
int bh_bn_eval_bwd(const float* gy, const float* x, const float* weight, const float* inv_std, const float* mean_inv, float* gx,
                   float* gw, float* gb, double* workspace, const float* tap_coef, const float* tap_gout, const float* y_mask,
                   float* g_residual, const float* gx_add, int32_t B, int32_t C, int32_t HW, void* stream);
/* Derivative of bh_bn_eval_bwd for incoming (ggx [B,C,HW], ggw [C], ggb [C]; each may be NULL = zero):
 * d_gy = ggx * s_c + ggw_c * (inv_std_c * x - mean_inv_c) + ggb_c;  d_x = ggw_c * inv_std_c * gy;  d_w_c = inv_std_c * sum(ggx * gy).
 * Outputs may be NULL (not computed).  With `y_mask` (the same forward output given to bh_bn_eval_bwd) and `gg_residual` (incoming
 * gradient of g_residual, may be NULL): d_gy = [y > 0] * (... + gg_residual), and gy is replaced by gz = gy * [y > 0] in d_x / d_w. */
int bh_bn_eval_bwd_bwd(const float* ggx, const float* ggw, const float* ggb, const float* gy, const float* x, const float* weight,
                       const float* inv_std, const float* mean_inv, float* d_gy, float* d_x, float* d_w, double* workspace,
                       const float* y_mask, const float* gg_residual, int32_t B, int32_t C, int32_t HW, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (elementwise affine) of the attacker's private model copy, one or two launches per
 * autograd order.  reference: the text attacks (config/attack/tag.yaml, optimization_with_label_attack.py:168-174 ->
 * objectives.py:40-46) differentiate every LayerNorm of the rebuilt victim model (base_attack.py:176-212) twice per
 * iteration.  x / y / gradients: [R, D] contiguous fp32 (R = all leading dimensions); gamma / beta [D] may be NULL.
 * Row sums in fp64, column sums in a fixed order; no allocation, no synchronisation. */
int bh_ln_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int32_t R, int32_t D,
              float eps, void* stream);
/* gx = rstd (g - M(g) - x_hat M(g x_hat)), g = gy gamma (skipped when gx is NULL); ggamma = sum_r gy x_hat; gbeta = sum_r gy
 * (one more launch; skipped when both are NULL).  mean / rstd: the [R] arrays bh_ln_fwd wrote. */
int bh_ln_bwd(const float* gy, const float* x, const float* gamma, const float* mean, const float* rstd, float* gx, float* ggamma,
              float* gbeta, int32_t R, int32_t D, void* stream);
/* Derivative of bh_ln_bwd for incoming u = d/d gx [R, D], s = d/d ggamma [D], t = d/d gbeta [D] (each may be NULL = zero):
 * d_gy = gamma rstd P u + s x_hat + t;  d_x = -rstd^2 (M(u w) x_hat + b P u + M(u x_hat) w) + rstd P (s gy);
 * d_gamma = sum_r gy rstd P u  (P v = v - M(v) - x_hat M(v x_hat), w = P g, b = M(g x_hat)).  Outputs may be NULL;
 * `row_scalars`: 2 * R floats of workspace, required for d_gamma. */
int bh_ln_bwd_bwd(const float* u, const float* s, const float* t, const float* gy, const float* x, const float* gamma,
                  const float* mean, const float* rstd, float* d_gy, float* d_x, float* d_gamma, float* row_scalars, int32_t R,
                  int32_t D, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-tensor elementwise kernels over per-parameter lists (FedAvg unroll, Pearlmutter offset) and batch kernels
 * ---------------------------------------------------------------------------------------------------------------- */

/* Tensors per base launch group.  Three pointer lists of this length travel in the kernel arguments of the one three-list form
 * ((a + alpha b) - c); the forms with at most two lists put two adjacent base groups (224 tensors) into one launch. */
#define BH_MT_MAX_PTRS 112
int32_t bh_mt_num_groups(int32_t n_tensors);
/* Like bh_gm_group_bounds for launch groups of BH_MT_MAX_PTRS tensors: group_chunk_begin[bh_mt_num_groups + 1]. */
int bh_mt_group_bounds(int32_t n_tensors, const bh_gm_chunk* chunks_host, int64_t n_chunks, int32_t* group_chunk_begin);

/* out = a + alpha * b            (c_ptrs == NULL)      one local SGD step of the FedAvg unroll, alpha = -lr
 * out = (a + alpha * b) - c      (c_ptrs != NULL)      the last step fused with `p_local - p_server`
 * for every tensor of the list, written into the packed layout of the chunk table (same as bh_gm_pack); mul, add and
 * sub round separately, exactly like torch's `param - lr * grad` and `p_local - p_server`.  Host arrays of device
 * pointers, each tensor contiguous fp32 and 16-byte aligned.  Cache policy (all bh_mt_* calls): operands are read with plain
 * loads while operands + output fit BH_GM_CACHE_AUTO_BYTES (the Infinity Cache), with non-temporal loads beyond; the
 * output is written with plain stores (the model's next forward pass reads it) unless it alone exceeds that size.  Same values.
 * reference: objectives.py:64-69 (`params = [param - lr * grad ...]`, `gradient = [p_local - p_server ...]`). */
int bh_mt_axpy(int32_t n_tensors, const void* const* a_ptrs, const void* const* b_ptrs, const void* const* c_ptrs,
               float alpha, const bh_gm_chunk* chunks_dev, int64_t n_chunks, const int32_t* group_chunk_begin,
               float* out_flat, void* stream);
/* out = alpha * a; a NULL entry of a_ptrs reads as zeros.  The backward of bh_mt_axpy with respect to b. */
int bh_mt_scale(int32_t n_tensors, const void* const* a_ptrs, float alpha, const bh_gm_chunk* chunks_dev, int64_t n_chunks,
                const int32_t* group_chunk_begin, float* out_flat, void* stream);
/* out = theta + mult * (coef[0] * data + coef[1] * grad): the model parameters offset along the first-order direction of the
 * gradient-matching objective, coefficients read from the device (the finalize kernel writes them at
 * stats[BH_GM_STAT_PATCH_D .. BH_GM_STAT_PATCH_R], already multiplied by the finite-difference step).
 * reference: objectives.py:347-352 (`torch._foreach_add_(model.parameters(), first_order_grad, alpha=eps_n)`), :468-486. */
int bh_mt_patch(int32_t n_tensors, const void* const* theta_ptrs, const void* const* grad_ptrs, const float* data_flat,
                const float* coef_dev, float mult, const bh_gm_chunk* chunks_dev, int64_t n_chunks,
                const int32_t* group_chunk_begin, float* out_flat, void* stream);
/* `mult` scales both coefficients: +1 forward differences, -1 backward, +-0.5 central (objectives.py:347, :375, :401-406). */

/* OrthogonalityRegularization on x[B, D] (D = C*H*W): value = sum over ordered pairs i != j of mean_k (x_ik * x_jk)^2
 * (the reference does not apply `scale`), analytic gradient into grad_out[B, D], per-workgroup partial values into
 * partials_dev[BH_PRIOR_MAX_GRID].  Returns the grid size (> 0) or a negative error.
 * reference: regularizers.py:156-181. */
int bh_prior_orthogonality(const float* x, int32_t B, int64_t D, float* grad_out, double* partials_dev, void* stream);

typedef struct bh_psnr_params {
  float mean[4]; /* de-normalisation: img = x * std[c] + mean[c]  (analysis.py:228-229) */
  float std[4];
  float factor;  /* peak value (1.0) */
  int32_t clip;  /* clamp both images to [0, 1] first */
} bh_psnr_params;
/* PSNR of rec[B, per_example] against ref (both normalised like the candidate): out_dev[0] = mean over examples,
 * out_dev[1] = max, out_dev[2 + b] = example b; +inf when an example matches exactly, NaN for a non-finite error.
 * mse_dev: B doubles of workspace.  reference: analysis/metrics.py:108-130 (psnr_compute, batched=False). */
int bh_metric_psnr(const float* rec, const float* ref, int32_t B, int64_t per_example, int64_t plane, int32_t channels,
                   const bh_psnr_params* params, double* mse_dev, float* out_dev, void* stream);

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
  BH_STATE_GNORM = 6,     /* fp32: L2 norm of the (noise-perturbed) candidate gradient, when clipping; words 6 .. 6 +
                           * BH_STEP_MAX_SLOTS - 1: one per slot of the list launches (bh_candidate_step_list) */
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
  float grad_clip;      /* < 0 = off (0 is a legal threshold); uses state[BH_STATE_GNORM] :171-174 */
} bh_step_params;

/* One fused elementwise pass: assemble the gradient (g + g_reg + langevin*lr*noise), clip, sign, Adam/AdamW moment
 * and parameter update, box projection, and -- when state.improved -- copy the projected candidate into `best`.
 * reference: optimization_based_attack.py:165-184 (grad post-processing), torch.optim.Adam/AdamW single-tensor
 * update selected at auxiliaries/common.py:5-12, optimization_based_attack.py:117-121. */
int bh_candidate_step(const void* state_dev, const double* sched_dev, const bh_step_params* params, float* x,
                      const float* g, const float* g_reg, const float* noise, float* m, float* v, float* best,
                      void* stream);

/* The same two stages over a LIST of optimised tensors, ONE launch each (ABI 7) -- the joint data + label attack optimises
 * `[candidate, labels]` and applies noise, clipping (each tensor by ITS OWN norm), sign and the Adam step tensor by tensor,
 * projecting only the data tensor: optimization_with_label_attack.py:124-128, :177-190.  A slot = one tensor with its own
 * bh_step_params (box, clip threshold, ...) and buffers; <= BH_STEP_MAX_SLOTS slots travel in the kernel-argument segment.
 *   bh_grad_norm_list:      per-slot partial sums of squares of the effective gradient into ws_dev -- slot s owns
 *                           min(ceil(n_s / 2048), BH_PRIOR_MAX_GRID) rows (bh_grad_norm's own grid), slots without clipping none;
 *                           ws_dev holds bh_step_list_norm_rows(...) doubles.  No launch when no slot clips.
 *   bh_candidate_step_list: every workgroup adds its slot's rows in bh_grad_norm's finalize order (constants of the launch: no
 *                           ticket, no atomic), then runs bh_candidate_step's arithmetic on its share of the slot; the norm of
 *                           slot s is also left in state[BH_STATE_GNORM + s].  Bit-identical to per-tensor launches of
 *                           bh_grad_norm + bh_candidate_step. */
#define BH_STEP_MAX_SLOTS 4
typedef struct bh_step_slot {
  bh_step_params params;
  float* x;
  const float* g;
  const float* g_reg; /* may be NULL */
  const float* noise; /* may be NULL unless params.langevin > 0 */
  float* m;
  float* v;
  float* best;
} bh_step_slot;
int32_t bh_step_list_norm_rows(int32_t n_slots, const bh_step_slot* slots);
int bh_grad_norm_list(const void* state_dev, int32_t n_slots, const bh_step_slot* slots, const double* sched_dev, double* ws_dev,
                      void* stream);
int bh_candidate_step_list(void* state_dev, const double* sched_dev, int32_t n_slots, const bh_step_slot* slots, const double* ws_dev,
                           void* stream);

/* Trial selection across ranks (host arithmetic; the collective itself is ONE all-reduce(MIN) on this key + one broadcast of the
 * winner, issued by the caller over RCCL -- `torch.distributed` in breaching_amd/trials.py, `ncclAllReduce(..., ncclInt64, ncclMin, ...)`
 * for a host without Python).  key = (IEEE-754 bits of the fp32 score << 32) | trial for scores >= 0 (their bit patterns order like
 * the floats); NaN and +inf map to the +inf pattern (the reference turns a non-finite score into +inf, :204); negative scores, which no
 * supported scoring produces, order below every non-negative one.  Ties go to the lower trial index, like torch.min in the
 * reference's sequential loop.  reference: optimization_based_attack.py:191-218. */
int64_t bh_trial_key(float score, int32_t trial);
/* Inverse for keys >= 0 and for the negative-score keys; returns 0, or BH_EINVAL for NULL outputs. */
int bh_trial_key_unpack(int64_t key, float* score_out, int32_t* trial_out);

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
