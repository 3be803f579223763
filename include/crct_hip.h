/* crct_hip.h -- C ABI of libcrct_hip.so: the MI355X (gfx950) CRCT co-attention training step.
 *
 * The reference (levymsn/CQA-CRCT) has no FFI: its boundary for this path is the Python surface
 * train.py uses.  This header is the native boundary underneath our mirror of that
 * surface (cqa-crct_amd/crct): plain pointers, sizes and a HIP stream -- no torch types.  Every
 * entry point cites the reference code whose arithmetic it replaces.  All device pointers are HBM
 * addresses owned by the caller; all kernels are enqueued on `stream` and never synchronise.
 *
 * Return value: 0 on success, non-zero on error (message via crct_last_error()).
 * Activations / activation gradients are bf16 (raw uint16 bits), statistics, parameters,
 * parameter gradients and optimizer state are fp32.
 */
#ifndef CRCT_HIP_H
#define CRCT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* crct_stream_t; /* hipStream_t */

enum { CRCT_ACT_NONE = 0, CRCT_ACT_GELU = 1, CRCT_ACT_RELU = 2, CRCT_ACT_LEAKY = 3, CRCT_ACT_TANH = 4 };

const char* crct_last_error(void);
int crct_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM with fused epilogue.  C[M][N] = epi( alpha * sum_k A'(m,k) B'(n,k) )
 *   ta = 0: A'(m,k) = A[m*lda + k]   ta = 1: A'(m,k) = A[k*lda + m]      (bf16)
 *   tb = 0: B'(n,k) = B[n*ldb + k]   tb = 1: B'(n,k) = B[k*ldb + n]      (bf16)
 * epilogue order: *alpha, +bias[n], store pre-activation, act, *act'(dact_src), dropout, +addend,
 * (+C if accumulate), store as bf16 or fp32.
 * Replaces nn.Linear forward / dgrad / wgrad everywhere in CRCT/backbone/vilbert.py
 * (:388-390 :425 :455 :468 :518-520 :662-675 :749-752 :959 :974 :1060) and regressor.py:8-34.
 * Requirements: lda, ldb, ldc, ld_aux, ld_add multiples of 8 elements, N % 4 == 0, K % 8 == 0,
 * and for a transposed operand its row extent (M for ta, N for tb) % 8 == 0; pointers 16-byte aligned.
 */
typedef struct CrctGemmArgs {
  const void* A; const void* B; void* C;
  const float* bias;        /* [N] or NULL */
  void* preact_out;         /* bf16 [M][ld_aux] or NULL: value before `act` */
  const void* dact_src;     /* bf16 [M][ld_aux] or NULL: multiply by act'(.) of kind `dact` */
  const void* addend;       /* bf16 [M][ld_add] or NULL */
  int64_t lda, ldb, ldc, ld_aux, ld_add;
  int32_t M, N, K;
  int32_t ta, tb;
  int32_t act, dact;
  int32_t c_is_f32, accumulate;
  int32_t tile;             /* -1 = auto; 0: 128x128, 1: 128x64, 2: 64x128, 3: 64x64 */
  float alpha;
  uint32_t drop_thr;        /* 0 = no dropout, else keep iff a 16-bit Philox slice >= thr >> 16 (drop probability thr / 2^32 to within 2^-16) */
  float drop_scale;         /* 1/(1-p) */
  uint32_t drop_site;
  uint64_t seed;
  float* rowsum_out;        /* fp32 [M] or NULL: rowsum_out[m] += sum_k A'[m][k] (A' = A as the GEMM reads it).  For a
                               weight gradient dW = dy^T x (ta = tb = 1) this is the BIAS gradient, computed by the
                               same kernel from the dy tiles it already holds (one extra MFMA against a fragment of
                               ones).  LDS-DMA kernel only (K % 64 == 0); other shapes are rejected. */
  /* fp8 forward (BASELINE configs[4]): fp8 != 0 -> A [M][lda] and B [N][ldb] are OCP e4m3 BYTES (ta = tb = 0, K % 128 == 0,
   * lda / ldb % 16 == 0), quantised per tensor as q = x * scale; the fp32 product is divided by (*scale_a) * (*scale_b)
   * (device scalars: delayed scaling never syncs the host).  q_out: optional e4m3 copy of the epilogue's result
   * [M][ld_q], quantised with *q_scale, while max |result| is max-ed into *q_amax (fp32 bits, >= 0) for the next step's scale. */
  int32_t fp8;              /* bit 0: fp8 GEMM; bit 1 (2): A is OCP e5m2 (a gradient) instead of e4m3; bit 2 (4): q_out is e5m2;
                               bit 3 (8): a data gradient (profiling label only: its operand layout is the forward's) */
  const float* scale_a; const float* scale_b;
  void* q_out; const float* q_scale; float* q_amax; int64_t ld_q;
  /* Which Linear of the model this launch belongs to (CRCT_SITE_*; 0 = untagged): only read by the live profile
   * (crct_prof_read_site) and the launch log -- never by a kernel. */
  int32_t site;
  /* K-partitioned launch (forward / data-gradient GEMMs of the LDS-DMA kernel whose output has too few tiles for 256 CUs:
   * M = 1600 rows x N = 768 is 78-156 workgroups).  split_k = S > 1: S workgroups per output tile each contract 1/S of the K
   * tiles, park their fp32 accumulators in splitk_ws (write-through stores) and draw a ticket from splitk_cnt[tile]; the one
   * that draws S - 1 adds the S slabs IN SLICE ORDER (so the result does not depend on which slice finished last), runs the
   * epilogue and puts the ticket back to 0.  splitk_ws: device fp32, >= tiles * S * BM * BN elements (crct_gemm_splitk_ws_elems);
   * splitk_cnt: device uint32 [tiles], zero before the first use.  Both belong to ONE stream: launches that share them must be
   * stream-ordered.  0 / 1 = off.  Not with ta (weight gradients), rowsum_out or fp8. */
  int32_t split_k;
  float* splitk_ws; uint32_t* splitk_cnt;
  /* The fp32 residual stream of the step engine (CrctStepCfg.residual_fp32): addend_f32 != 0 -> `addend` is fp32 [M][ld_add] (the previous
   * LayerNorm's fp32 output copy) instead of bf16; c_cached != 0 with c_is_f32 (and accumulate == 0) -> C is an activation (the
   * pre-LayerNorm sum, read by the next kernel) and is written with ordinary instead of streaming stores.  ld_add % 8 == 0 either way. */
  int32_t addend_f32, c_cached;
} CrctGemmArgs;

/* GEMM sites of the step.  The FFN group of BASELINE.md section 4 ("fraction of the FFN-GEMM roofline") = the four *_FFN_* sites. */
enum {
  CRCT_SITE_NONE = 0,
  CRCT_SITE_T_QKV = 1, CRCT_SITE_T_OUT = 2, CRCT_SITE_T_FFN_UP = 3, CRCT_SITE_T_FFN_DN = 4,      /* text stream: BertLayer + the text side of BertConnectionLayer's FFN */
  CRCT_SITE_V_QKV = 5, CRCT_SITE_V_OUT = 6, CRCT_SITE_V_FFN_UP = 7, CRCT_SITE_V_FFN_DN = 8,      /* visual stream */
  CRCT_SITE_C_QKV_T = 9, CRCT_SITE_C_QKV_V = 10, CRCT_SITE_C_OUT_T = 11, CRCT_SITE_C_OUT_V = 12, /* co-attention: query2/key2/value2, query1/key1/value1, biOutput.dense2, dense1 */
  CRCT_SITE_IMG_EMB = 13, CRCT_SITE_HEAD = 14, CRCT_SITE_COUNT = 15
};
enum { CRCT_KIND_FWD = 0, CRCT_KIND_DGRAD = 1, CRCT_KIND_WGRAD = 2 };
/* fp32 elements of split-K slab space an M x N output needs for `split_k` slices with the tile configuration the launcher
 * would use (an upper bound over the split-K configurations), and the number of ticket words. */
int64_t crct_gemm_splitk_ws_elems(int M, int N, int split_k);
int crct_gemm_splitk_tickets(int M, int N);

int crct_gemm_bf16(const CrctGemmArgs* args, crct_stream_t stream);
/* n <= 8 independent GEMMs in ONE grid (same ta/tb, no epilogue extras besides the output type and accumulate):
 * the weight gradients of one layer.  Problems that do not qualify are launched one by one. */
int crct_gemm_bf16_grouped(const CrctGemmArgs* args, int n, crct_stream_t stream);
/* Cap on the workgroups of a grouped launch (0 = one per tile): with a cap the workgroups are persistent and walk the tiles, so
 * a layer's weight gradients occupy at most that many CUs at a time beside the data-gradient chain.  Process-wide knob of the
 * developer tools (bench.py --wgrad-wgs); results are identical for every value. */
int crct_gemm_group_max_workgroups(int n);
// Target size of the persistent grid of a grouped bf16 weight-gradient launch issued through crct_gemm_bf16_grouped (default 0 = one
// workgroup per tile; the step engine has its own policy, crct_engine_set_wgrad_workgroups): the grid is
// ceil(tiles / ceil(tiles / target)) workgroups, rounded up to a multiple of 8, each walking whole rounds of the tile list.  Returns
// the previous target.  Placement only: results are bit-identical.
int crct_gemm_group_target_workgroups(int n);
/* Block -> tile placement of a grouped weight-gradient launch.  0 (default): every problem is cut into 8 rectangles, one per XCD
 * (each XCD touches every problem: 2.5 - 3.6 x the distinct operand bytes at the fabric).  on != 0: the tiles of all problems form
 * one list and each XCD takes one run of consecutive tiles, i.e. a compact part of one or two problems, so that its private L2
 * fetches ~1/8 of the group's operand panels.  Results are identical; measured neutral to slightly slower in the step
 * (EXPERIMENTS.md round 3), kept as the developer A/B switch behind that statement (bench.py --wgrad-concat). */
int crct_gemm_group_concat(int on);
/* Test hook: on != 0 makes crct_embed_text_bwd launch the position / type sums and the word-table scatter as two kernels one after
 * the other instead of one merged launch (identical results; tests/test_kernels_gpu.py, bench.py --embed-scatter-split). */
void crct_embed_scatter_split(int on);
/* Kernel configuration of the grouped weight-gradient launches of the step (all 128 x 128 tiles; see gemm.hip): 4 = plain loop,
 * 48 / 53 = loader waves (8 + 4 / 4 + 4 waves), 58 / 59 = loader waves + two fragment register sets.  Returns
 * the previous value; other values are ignored. */
int crct_gemm_group_wgrad_config(int cfg);

/* Kernel configuration of one shape class of the forward / data-gradient GEMMs: cls = 3 * rows_bucket + column_class with
 * rows_bucket 0: M <= 2000, 1: M <= 4000, 2: M > 4000 and column_class 0: N >= 2304 (wide), 1: N <= 1024 with K <= 1024, 2: N <= 1024
 * with K > 1024 (gemm.hip, pick_pipe_config).  cfg < 0 only reads.  Returns the previous configuration id (-1: bad class).  For
 * in-step sweeps (bench.py --class-policy); per-site overrides: crct_engine_set_site_policy. */
int crct_gemm_class_config(int cls, int cfg);

/* fp8 GEMMs (forward, data gradient, weight gradient): 1 (default) = the 128-deep K tile of an output tile as ONE
 * v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales -- gfx950's block-scaled instruction at twice the rate of the plain fp8
 * MFMA -- 0 = four v_mfma_f32_16x16x32_fp8_fp8 / _fp8_bf8 (round 3).  Same products, other order of the fp32 additions inside a K
 * tile.  on < 0 only reads.  Returns the previous setting. */
int crct_gemm_fp8_scaled_mfma(int on);

/* Tile the launcher would pick for an M x N output (0..3, see CrctGemmArgs.tile). */
int crct_gemm_pick_tile(int M, int N);

/* Test hook: on != 0 routes every GEMM through the register-staged kernel (any K % 8 == 0) instead of
 * the LDS-DMA pipelined kernel (K % 64 == 0), so the parity tests can cover both code paths. */
int crct_gemm_force_generic(int on);

/* Live measurement for bench.py: when enabled, every GEMM launch is bracketed by HIP events on its
 * launch stream.  variant = config*3 + {0 forward, 1 dgrad (tb), 2 wgrad (ta,tb)}, config 0..15 = LDS-DMA
 * kernel configurations (tile / waves / stages, see gemm.hip), 16..19 = register-staged kernel tiles 0..3.  crct_prof_read
 * synchronises on the recorded events and returns launches, summed algorithmic FLOPs (2MNK) and
 * summed elapsed milliseconds of that variant since the last reset. */
int crct_prof_enable(int on);      /* 0 off; 1: the GEMM kernels; 2: also every other kernel the library launches (crct_prof_stamp_*) */
int crct_prof_reset(void);
/* With crct_prof_enable(2): begin / end stamps of EVERY kernel launch of the library since the last crct_prof_reset, in launch order
 * over all streams.  crct_prof_stamp_read(i): the stream (hipStream_t) launch i ran on and its begin / end in milliseconds after the
 * first stamped launch began (synchronises).  bench.py derives config.critical_path from them: launches per step, kernels and busy
 * time per hardware queue, time with k kernels in flight. */
int crct_prof_stamp_count(void);
int crct_prof_stamp_read(int i, void** stream, double* t0_ms, double* t1_ms);
int crct_prof_read(int variant, long* count, double* flops, double* ms);
/* The same stamps keyed by model site (CRCT_SITE_*) and kind (CRCT_KIND_*).  A grouped launch (the weight gradients of a layer
 * in one grid) is ONE kernel: its duration is apportioned to its member problems by their share of the launch's FLOPs
 * (`apportioned` != 0 tells the caller that this (site, kind) contains such shares). */
int crct_prof_read_site(int site, int kind, long* count, double* flops, double* ms, int* apportioned);
/* Launch log (developer tooling: tools/pmc_sites.py matches it against a rocprofv3 counter collection, which serialises
 * dispatches in host enqueue order): while enabled, every GEMM kernel launch appends one record. */
typedef struct CrctLaunchRec {
  int32_t site, kind, M, N, K, cfg, split_k, grid, n_problems;   /* grouped launch: site = -1, M/N/K of problem 0, n_problems > 1 */
  double flops;
} CrctLaunchRec;
int crct_launch_log_enable(int on);      /* on != 0 also clears the log */
int crct_launch_log_count(void);
int crct_launch_log_read(int i, CrctLaunchRec* out);

/* ---------------------------------------------------------------------------------------------
 * Row LayerNorm (TF style, eps inside sqrt) -- BertLayerNorm, vilbert.py:281-294 -- over rows that
 * already hold dense(x)+dropout+residual (the GEMM epilogue adds them).
 *   y = gamma * (x - mean) * rstd + beta ; optional inverted dropout AFTER the norm (embeddings,
 *   vilbert.py:356-357, :1494-1495).  x, y bf16 [M][H]; mean, rstd fp32 [M].
 */
int crct_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                       float* mean, float* rstd, int M, int H, float eps,
                       uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                       crct_stream_t stream);

/* The same, also writing an OCP e4m3 copy q_out [M][H] of y (quantised as q = y * *q_scale, saturating at +-448) and
 * max-ing max |y| into *q_amax (fp32 bits): the operand of the fp8 forward GEMMs (BASELINE configs[4]). */
int crct_layernorm_fwd_q(const void* x, const float* gamma, const float* beta, void* y,
                         float* mean, float* rstd, int M, int H, float eps,
                         uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                         void* q_out, const float* q_scale, float* q_amax, crct_stream_t stream);

/* crct_layernorm_fwd / _q with the arguments in a struct (q_out == NULL: no e4m3 copy). */
typedef struct CrctLnFwdArgs {
  const void* x; const float* gamma; const float* beta; void* y; float* mean; float* rstd;
  int32_t M, H; float eps; uint32_t drop_thr; float drop_scale; uint32_t drop_site; uint64_t seed;
  void* q_out; const float* q_scale; float* q_amax;
  /* The fp32 residual stream (CrctStepCfg.residual_fp32): x_f32 != 0 -> x is fp32 [M][H] (the pre-LayerNorm sum as the GEMM epilogue
   * left it, CrctGemmArgs.c_cached); y_f32 != NULL -> y is ALSO written as fp32 [M][H] (the next block's residual addend,
   * CrctGemmArgs.addend_f32).  The bf16 y stays what GEMMs read. */
  int32_t x_f32; float* y_f32;
} CrctLnFwdArgs;
int crct_layernorm_fwd_args(const CrctLnFwdArgs* a, crct_stream_t stream);

/* Every amax "value" below and in CrctGemmArgs / CrctStepCfg / CrctFp8Shadow is CRCT_FP8_AMAX_LANES consecutive fp32 words
 * (the kernels spread their atomic maxima over them; crct_fp8_update_scales takes the maximum of the words): an amax array
 * for n tensors has n * CRCT_FP8_AMAX_LANES words and entry i starts at word i * CRCT_FP8_AMAX_LANES. */
#define CRCT_FP8_AMAX_LANES 64
/* fp8 (OCP e4m3, per-tensor delayed scaling) helpers.  All scales / amax values are device fp32.
 *  crct_fp8_quantize_bf16   q[i] = e4m3(x[i] * *scale), *amax = max(*amax, max |x|)            (n % 8 == 0)
 *  crct_fp8_update_scales   scale[i] = fmax / amax[i] where amax[i] > 0 (fmax = 448 for e4m3 tensors, 57344 for the e5m2
 *                           gradient copies of the fp8 backward; <= 0 means 448); reset != 0 also clears amax[i]; skip_if (device fp32,
 *                           may be NULL) != 0 leaves everything untouched (GradScaler's found_inf: a skipped optimizer step
 *                           does not requantise the weight shadow, so its scales must stay too).  The amax values are
 *                           RUNNING maxima: call it once per step with reset = 0 and every few hundred steps with reset = 1
 *                           (a history window); clearing every step makes every wave of every producer issue an atomic
 *  crct_fp8_quantize_weights  exact per-tensor scaling of fp32 weights into the flat e4m3 shadow (same element offsets as
 *                           the fp32 buffer): tensors (seg_off, seg_len) with scale slot seg_slot[s] >= 0, chunk table
 *                           (blk_seg, blk_off) from crct_adamw_plan; writes scale[slot] = 448 / max |w|.  Used at start-up
 *                           and after a load_state_dict; the optimizer keeps the shadow current afterwards (CrctFp8Shadow). */
int crct_fp8_quantize_bf16(const void* x, void* q, const float* scale, float* amax, int64_t n, crct_stream_t stream);
int crct_fp8_update_scales(float* scale, float* amax, int n, int reset, const float* skip_if, float fmax, crct_stream_t stream);
int crct_fp8_quantize_weights(const float* p, void* q, const int64_t* seg_off, const int64_t* seg_len, const int32_t* seg_slot,
                              const int32_t* blk_seg, const int64_t* blk_off, int64_t n_blk, float* scale, float* amax, int n_slots,
                              crct_stream_t stream);
/* Transposed copy of the e4m3 weight shadow for the fp8 data-gradient GEMMs (dx = dy W contracts over W's rows): weight i lies
 * at byte offset w_off[i] of q as [w_out[i]][w_in[i]] and is written to the same offset of qt as [w_in[i]][w_out[i]]
 * (w_out, w_in multiples of 16).  tile_begin[i] = first 64 x 64 tile of weight i in the launch grid, n_tiles = their total.
 * All arrays device memory.  Run once per optimizer step behind the update that rewrote q.  max_workgroups > 0 caps the launch
 * (a persistent grid): beside a forward pass that is starting on another stream the full grid would take its CUs. */
int crct_fp8_transpose_weights(const void* q, void* qt, const int64_t* w_off, const int32_t* w_out, const int32_t* w_in,
                               const int64_t* tile_begin, int n_w, int64_t n_tiles, int max_workgroups, crct_stream_t stream);

/* LayerNorm backward.  dy bf16 [M][H] (gradient w.r.t. y, or w.r.t. post-norm-dropout output when
 * post_* is set), x = saved pre-norm rows.  Writes
 *   dx      bf16 [M][H]   gradient w.r.t. the pre-norm sum (also the residual branch's gradient)
 *   dx_lin  bf16 [M][H]   (optional) dx with the PRE-norm dropout mask of the producing Linear
 *                         re-applied (site/seed of that Linear's epilogue) = gradient of dense(x)
 *   dgamma, dbeta, dbias_lin  fp32 [H]: column sums (dbias_lin = colsum(dx_lin or dx)), written or
 *                         accumulated (`accumulate`).  `partials` is fp32 scratch sized [3][4 * nblk][H] (the kernel fills one row per workgroup, or one per
 *                         wave when [3][4][H] fp32 do not fit 64 KB of LDS),
 *                         nblk = crct_layernorm_bwd_blocks(M).
 */
int crct_layernorm_bwd_blocks(int M);
int crct_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                       const float* gamma, void* dx, void* dx_lin,
                       float* dgamma, float* dbeta, float* dbias_lin, float* partials,
                       int M, int H, int accumulate,
                       uint32_t post_thr, float post_scale, uint32_t post_site,
                       uint32_t lin_thr, float lin_scale, uint32_t lin_site, uint64_t seed,
                       crct_stream_t stream);

/* The same in two launches (rows pass / column pass), so that the column pass can be enqueued on another
 * stream off the critical path; `partials` must stay untouched until the finalize has run. */
int crct_layernorm_bwd_rows(const void* dy, const void* x, const float* mean, const float* rstd,
                            const float* gamma, void* dx, void* dx_lin, float* partials, int M, int H,
                            uint32_t post_thr, float post_scale, uint32_t post_site,
                            uint32_t lin_thr, float lin_scale, uint32_t lin_site, uint64_t seed,
                            crct_stream_t stream);
/* crct_layernorm_bwd_rows with the arguments in a struct (+ the optional fp8 copy). */
typedef struct CrctLnBwdArgs {
  const void* dy; const void* x; const float* mean; const float* rstd; const float* gamma; void* dx; void* dx_lin; float* partials;
  int32_t M, H; uint32_t post_thr; float post_scale; uint32_t post_site; uint32_t lin_thr; float lin_scale; uint32_t lin_site;
  uint64_t seed;
  /* fp8 backward (BASELINE configs[4]): optional OCP e5m2 copy [M][H] of the gradient that leaves towards the producing Linear
   * (dx_lin when given, else dx), quantised as q = g * *q_scale (saturating at +-57344), max |g| max-ed into *q_amax
   * (CRCT_FP8_AMAX_LANES words): the A operand of that Linear's fp8 data-gradient GEMM. */
  void* q_out; const float* q_scale; float* q_amax;
  int32_t x_f32;            /* x (the saved pre-norm rows) is fp32 [M][H]: the fp32 residual stream */
} CrctLnBwdArgs;
int crct_layernorm_bwd_rows_args(const CrctLnBwdArgs* a, crct_stream_t stream);      /* crct_layernorm_bwd_rows from the struct (incl. q_out) */
int crct_layernorm_bwd_finalize(const float* partials, float* dgamma, float* dbeta, float* dbias_lin,
                                int M, int H, int accumulate, crct_stream_t stream);

/* Column sum of a bf16 [M][ld] matrix into fp32 out[N] (bias gradients).  partials: [nblk][N]. */
int crct_colsum_blocks(int M);
int crct_colsum_bf16(const void* x, int64_t ld, float* out, float* partials, int M, int N,
                     int accumulate, crct_stream_t stream);

/* Key masks (uint8, 1 = attend) from the batch as the data loader ships it -- see CrctBatch.  km_t / km_v may be NULL. */
int crct_build_keymasks(const int64_t* sep_indices, const int64_t* hist_len, int sep_stride, const int64_t* image_mask,
                        uint8_t* km_t, uint8_t* km_v, int B, int T, int V, crct_stream_t stream);

/* Row softmax fp32 [M][F] -> bf16 [M][F]  (F.softmax(image_feat), vilbert.py:1476). */
int crct_softmax_rows_f32_bf16(const float* x, void* y, int M, int F, crct_stream_t stream);
/* The same from bf16 features (a data loader / input pipeline that ships bf16 halves the step's host -> device bytes). */
int crct_softmax_rows_bf16_bf16(const void* x, void* y, int M, int F, crct_stream_t stream);

/* Stand-in for a gradient all-reduce on a box with ONE GPU (bench.py --ghost-ranks N; CRCT/train.py:138-143 is what it stands in for):
 * `channels` workgroups on `stream` (RCCL runs one per channel) stream `bytes` at ptr (16-byte aligned, left unchanged) through HBM
 * `passes` times and pace themselves so that the kernel lasts `microseconds` -- the CUs, the HBM bandwidth and the hardware queue a real
 * ring all-reduce would hold while the bytes cross xGMI.  Nothing is reduced: a measuring device, not a collective. */
int crct_ghost_collective(void* ptr, int64_t bytes, int channels, int passes, double microseconds, crct_stream_t stream);
/* fp32 -> bf16 copy (weight shadow refresh). */
int crct_cast_f32_bf16(const float* x, void* y, int64_t n, crct_stream_t stream);
/* bf16 -> fp32 copy (a gradient bucket exchanged as bf16 put back into the fp32 gradient buffer for callers that read .grad). */
int crct_cast_bf16_f32(const void* x, float* y, int64_t n, crct_stream_t stream);
/* y[off[r] + i] = bf16(x[off[r] + i]) for the runs (off[r], len[r]) of element offsets (x fp32, y bf16, same element offsets; 16-byte
 * aligned bases), walked by the chunk table (blk_seg, blk_off) of crct_adamw_plan(len, ...).  The data-parallel exchange packs with
 * it what the weight-gradient GEMMs have not already written into the bf16 buffer (CrctStepCfg.grads_bf16). */
int crct_cast_runs_f32_bf16(const float* x, void* y, const int64_t* off, const int64_t* len, const int32_t* blk_seg, const int64_t* blk_off,
                            int64_t n_blk, crct_stream_t stream);
/* The reverse over the same table: y[off[r] + i] = float(x[off[r] + i]) (x bf16, y fp32).  After the exchange the all-reduced values
 * of those runs go back into the fp32 gradient buffer, so that every .grad view backward ACCUMULATES into holds the reduced gradient
 * (DistributedDataParallel's contract, CRCT/train.py:138-143) even when the Linear weight gradients stay in the bf16 buffer. */
int crct_cast_runs_bf16_f32(const void* x, float* y, const int64_t* off, const int64_t* len, const int32_t* blk_seg, const int64_t* blk_off,
                            int64_t n_blk, crct_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Scaled-dot-product attention, one workgroup per (batch, head), operands in LDS:
 * P = softmax(q k^T / sqrt(d) + (1-keymask)*-10000), dropout(P), ctx = P v.
 * Replaces vilbert.py:392-412 (text self), :522-543 (visual self), :684-701 / :704-723 (co-attn).
 * q [B][Tq][ldq], k/v [B][Tk][ldk] bf16 with head h at column h*d; keymask fp32/int-free: uint8 [B][Tk]
 * (1 = attend); ctx bf16 [B][Tq][ldo].
 * Lengths: Tq, Tk <= CRCT_ATTN_MAX_LEN (512: the text position table of config/vilbert.json, i.e. every length the reference model can
 * embed; CRCT/options.py:27's default max_seq_len is 256, config/plotqa.json:5-6 trains at 124 text tokens x 44 visual elements) for
 * head sizes 32 / 48 / 64 -- the three of config/vilbert.json; any other head size (d <= 64,
 * d % 8 == 0) only up to 112 x 112.  Anything else is refused with an error, never truncated.
 */
#define CRCT_ATTN_MAX_LEN 512
int crct_attention_fwd(const void* q, const void* k, const void* v, const uint8_t* keymask, void* ctx,
                       int B, int heads, int Tq, int Tk, int d,
                       int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                       uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                       crct_stream_t stream);
/* Three implementations share these entry points: Tq, Tk <= 112 with d in {32, 48, 64} run register-resident MFMA kernels
 * (attention_mfma.hip); longer sequences with those head sizes a key-tile loop with online softmax on MFMA (attention_long.hip);
 * other head sizes up to 112 x 112 (and everything up to 112 x 112 after crct_attention_force_valu(1)) the fp32 VALU kernels.
 * Same dropout stream in all three. */
void crct_attention_force_valu(int on);
/* Test hook: on != 0 sends every shape the long-sequence kernels can take (d in {32, 48, 64}) through them, short ones included. */
void crct_attention_force_long(int on);
/* Test hook of the MFMA kernels: n = 1 / 2 / 4 waves per (batch, head) where the tile counts allow it, 0 = automatic.  Every
 * count computes the same bits (tests/test_kernels_gpu.py). */
void crct_attention_force_split(int n);
/* Backward: recomputes P from q,k (no probabilities are stored).  dq/dk/dv have the layout of q/k/v.
 * accumulate_kv != 0 adds into dk/dv instead of overwriting (unused by the step; kept for tests). */
int crct_attention_bwd(const void* q, const void* k, const void* v, const uint8_t* keymask,
                       const void* dctx, void* dq, void* dk, void* dv,
                       int B, int heads, int Tq, int Tk, int d,
                       int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                       int64_t lddq, int64_t lddk, int64_t lddv,
                       uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                       crct_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Text embeddings: BertEmbeddingLocation.forward, vilbert.py:320-358.
 *  sum = word[ids] + pos[posid]*(qa) + type[seg']*(seg != 0) + (W_loc loc + b_loc)*(|loc|_1 != 0)
 *  y = dropout(LN(sum)).  Writes the pre-norm sum (bf16), y (bf16), mean/rstd.
 */
int crct_embed_text_fwd(const int64_t* ids, const int64_t* segs, const float* loc,
                        const float* word, const float* pos, const float* type,
                        const float* w_loc, const float* b_loc, const float* gamma, const float* beta,
                        void* sum_out, void* y, float* mean, float* rstd,
                        int B, int T, int H, int n_pos, float eps,
                        uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                        crct_stream_t stream);
/* Backward.  With rows_scratch (fp32 [B*T][H]) + idx_scratch (int32 [2][B*T]) every table gradient is summed in a fixed order
 * (bitwise reproducible): the word table by one wave per token row (the first row of each id adds all rows of that id), the
 * position / type tables by one workgroup per table row; with rows_scratch == NULL all three fall back to fp32 atomics.  Reduces the
 * loc-Linear and LayerNorm parameter gradients.  partials: fp32 [9][4 * nblk][H], nblk = crct_layernorm_bwd_blocks(B*T)
 * (7 row sets + the sums of token types 0 and 1).
 * n_types = rows of the type table.  All parameter-gradient outputs are ACCUMULATED (caller zeroes per step). */
int crct_embed_text_bwd(const void* dy, const void* sum_saved, const float* mean, const float* rstd,
                        const int64_t* ids, const int64_t* segs, const float* loc, const float* gamma,
                        float* d_word, float* d_pos, float* d_type, float* d_wloc, float* d_bloc,
                        float* d_gamma, float* d_beta, float* partials,
                        int B, int T, int H, int n_pos,
                        uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                        float* rows_scratch, int32_t* idx_scratch, int n_types,
                        crct_stream_t stream);
/* The same with an index for the word-table sums: word_index = int32 [2][n_vocab] in device memory, ALL ZERO on entry and all zero again on
 * exit (the kernels clean up what they set).  The row kernel leaves the first and last token row of every id there; the scatter kernel then
 * lets every wave that is not its id's first row return at once and scans only [first, last] for the others -- same owner, same summation
 * order, same bits as crct_embed_text_bwd, without its scan of all B*T ids per row (B*T = 9 920: 320 -> 35 us).  NULL / n_vocab 0: the scan. */
void crct_embed_word_index(int on);      /* test / timing hook: 0 = crct_embed_text_bwd_indexed ignores its index (default 1) */
int crct_embed_text_bwd_indexed(const void* dy, const void* sum_saved, const float* mean, const float* rstd,
                                const int64_t* ids, const int64_t* segs, const float* loc, const float* gamma,
                                float* d_word, float* d_pos, float* d_type, float* d_wloc, float* d_bloc,
                                float* d_gamma, float* d_beta, float* partials,
                                int B, int T, int H, int n_pos,
                                uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                                float* rows_scratch, int32_t* idx_scratch, int n_types, int32_t* word_index, int n_vocab,
                                crct_stream_t stream);

/* Image embeddings: BertImageEmbeddings.forward (dataset 'plotqa'), vilbert.py:1474-1496.
 *  sum = img_lin (bf16 [M][H], = new_image_embeddings(softmax(feat)) from the GEMM) + W_loc loc + b_loc
 *        + color_emb[target];  y = dropout(LN(sum)). */
int crct_embed_image_fwd(const void* img_lin, const float* loc, const int64_t* target,
                         const float* w_loc, const float* b_loc, const float* color,
                         const float* gamma, const float* beta,
                         void* sum_out, void* y, float* mean, float* rstd,
                         int M, int H, float eps,
                         uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                         crct_stream_t stream);
/* Backward: d_sum (bf16 [M][H], feeds the wgrad GEMM of new_image_embeddings), colour-table sums (gather-sum pass
 * through rows_scratch fp32 [M][H] + idx_scratch int32 [M] over the n_color table rows, or fp32 atomics when
 * rows_scratch == NULL), loc-Linear / LayerNorm / image-Linear-bias gradients (accumulated).
 * partials: fp32 [7][4 * nblk][H]. */
int crct_embed_image_bwd(const void* dy, const void* sum_saved, const float* mean, const float* rstd,
                         const float* loc, const int64_t* target, const float* gamma,
                         void* d_sum, float* d_color, float* d_wloc, float* d_bloc, float* d_bimg,
                         float* d_gamma, float* d_beta, float* partials,
                         int M, int H,
                         uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed,
                         float* rows_scratch, int32_t* idx_scratch, int n_color,
                         crct_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Heads + losses: BertPreTrainingHeads.forward (vilbert.py:1048-1062), the tail of
 * PlotQA_Regressor_v20 (regressor.py:31-41: Linear(256,1)+Tanh), the regression bookkeeping and
 * CrossEntropyLoss(ignore_index=-1) of vilbert.py:1583-1657, and the loss combination of
 * encoder_decorator.py:144-153 -- one launch, no host sync.
 *  pooled_t, pooled_v  bf16 [B][Hb] (relu'd pooler outputs), fus_h bf16 [B][256] (LeakyReLU(fusion.4(..)))
 *  outputs: logits fp32 [B][2]; reg fp32 [5][B] = pred*scale, reg_loss, reg_l1, raw tanh output, dist5;
 *           stats fp32 [17] = loss, nsp_loss, mean_B reg_loss, n_needs, n_right5, n_rightT, n_valid_labels, 0, then [8..16] =
 *           the 9 floats train.py:181-189 all-reduces every iteration: loss, lm_loss (0), nsp_loss, mean reg_loss and mean
 *           reg_5_dist over the rows that need regression (0 if none), legend_loss (0), num_regs, reg_5_right, reg_t_right
 *  gradient seeds (upstream gradients: grad_scale * g_nsp_dev / g_reg_dev when given, else nsp_coeff*g and reg_coeff*g/B as
 *  in encoder_decorator.py:144-153 with g = grad_scale * (*g_loss_dev, or 1); grad_scale is the data-parallel 1 / world):
 *  d_pooled_t / d_pooled_v
 *  bf16 [B][Hb] = gradient w.r.t. the poolers' PRE-activations (dropout and ReLU undone),
 *  d_fus_h bf16 [B][256] = gradient w.r.t. fusion.4's pre-activation; parameter gradients of
 *  bi_seq_relationship / fusion.6 are ACCUMULATED into d_w_cls[2][Hb], d_b_cls[2], d_w_f6[256], d_b_f6[1].
 */
typedef struct CrctHeadArgs {
  const void* pooled_t; const void* pooled_v; const void* fus_h;
  const float* w_cls; const float* b_cls; const float* w_f6; const float* b_f6;
  const float* R;                 /* [B][4] gt, needs, tol, scale */
  const int64_t* labels;          /* [B] or NULL (evaluation) */
  float* logits; float* reg; float* stats;
  float* scratch;                 /* fp32 [B][8] per-row records (dlogits, dz, loss terms) */
  void* d_pooled_t; void* d_pooled_v; void* d_fus_h;   /* NULL in evaluation */
  float* d_w_cls; float* d_b_cls; float* d_w_f6; float* d_b_f6;
  const float* g_nsp_dev;         /* optional device scalar: dLoss/d nsp_loss */
  const float* g_reg_dev;         /* optional device [B]:    dLoss/d reg_loss[b] */
  const float* g_loss_dev;        /* optional device scalar: dLoss/d stats[0] (the combined loss); multiplies the default seeds */
  int32_t B, Hb;
  int32_t fusion_sum;             /* 0 = 'mul' (default, vilbert.py:163), 1 = 'sum' */
  int32_t use_l1;                 /* params['L1'] : L1Loss vs SmoothL1Loss(beta=0.5), vilbert.py:1525-1528 */
  int32_t kind_l1;                /* reg_loss_kind == 'L1' (evaluation) -> no zeroing of |target|>1 */
  float tol_margin, nsp_coeff, reg_coeff, grad_scale;
  uint32_t drop_thr; float drop_scale; uint32_t drop_site; uint64_t seed;   /* cls.dropout (0.1) */
} CrctHeadArgs;
int crct_head_loss(const CrctHeadArgs* args, crct_stream_t stream);

/* Evaluation scoring, answer selection per question (replaces the per-question Python loop with .item() syncs of
 * evaluation.py:281-292): question q owns num_ans[q] consecutive candidate rows of the N scored rows;
 * prob0 = softmax(logits [N][2])[:, 0] (:249), answers[q] = argmax of prob0 over its rows (first maximum), or
 * forced_answers[q] when given ('_REGS' question files, :283-284); sel_out / sel_err / sel_terr [Q] gather the regressed
 * value, its relative error and its tick error (regression[0] / [4] / [2], :255-257) from the chosen row.
 * prob0 [N] is optional.  Integer results are exact; an out-of-range forced answer selects +inf errors. */
int crct_eval_select(const float* logits, const float* reg_out, const float* reg_err, const float* reg_terr,
                     const int64_t* num_ans, const int64_t* forced_answers, int Q, int64_t N, float* prob0,
                     int64_t* answers, float* sel_out, float* sel_err, float* sel_terr, crct_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused multi-tensor AdamW (torch.optim.AdamW semantics as constructed by utils.py:228-249) over
 * the flat fp32 parameter / gradient / moment buffers, refreshing the bf16 weight shadow.
 * seg_* describe the tensors that receive gradients: element offset, length, lr, weight decay.
 * blk_seg / blk_off: per-workgroup (segment id, element offset inside it), built by the caller
 * with crct_adamw_plan().  inv_scale_dev: optional device scalar dividing the gradients (GradScaler).
 * max_workgroups: 0 = one workgroup per 4096-element block; > 0 = grid-stride launch of at most that many workgroups
 * (throttle for an update that overlaps other work on another stream).
 * zero_grads != 0: every gradient element is overwritten with 0 right after it has been read (optimizer.zero_grad()
 * folded into the update: the separate 953 MB memset disappears).
 */
/* Loss scaling as torch.amp.GradScaler drives it (train.py:157,208-212: scaler.scale(loss).backward(); scaler.step(optimizer)):
 * grad_scale = the scaler's scale (the gradients are divided by it inside the update), found_inf != 0 skips the whole step,
 * step = device counter of the steps really taken (replaces the host `step` in the bias corrections; crct_adamw_advance
 * increments it unless found_inf is set).  All device pointers, each may be NULL. */
typedef struct CrctAmpState { const float* grad_scale; const float* found_inf; const int32_t* step; } CrctAmpState;
int crct_adamw_advance(int32_t* step_dev, const float* found_inf_dev, crct_stream_t stream);
/* e4m3 shadow of the weights the fp8 forward GEMMs read: q = flat byte buffer with the element offsets of the fp32 buffer,
 * seg_slot[s] = scale slot of AdamW segment s (-1: tensor has no fp8 shadow), scale = device fp32 [n_slots], amax = device fp32
 * [n_slots * CRCT_FP8_AMAX_LANES].  The update
 * quantises the NEW weights with scale[slot] and max-es max |w| into amax[slot]; the caller runs crct_fp8_update_scales on
 * (scale, amax) before the NEXT update (delayed scaling).  All pointers device memory; q == NULL switches it off.
 * qt / seg_in / seg_t_base / seg_t_ld (all or none; may be NULL): the TRANSPOSED shadow of the fp8 data-gradient GEMMs
 * (crct_fp8_transpose_weights' layout) kept current by the update itself.  seg_in[s] > 0: segment s is a band of whole rows
 * [seg_len / seg_in][seg_in] (both multiples of 64) of a weight [seg_t_ld[s]][seg_in] -- the weight itself, or one of the three
 * parameter tensors of a fused QKV weight -- and its bytes are also written to qt, transposed: element (r, c) of the band goes to
 * byte seg_t_base[s] + c * seg_t_ld[s] + r (seg_t_base = the weight's byte offset + the band's first row).  0 = no transposed
 * copy of that segment.  The update walks such a band in 64 x 64 tiles, which costs one extra byte written per element and
 * no extra launch (a separate transposing launch beside the next forward pass cost the step 0.12-0.33 ms: EXPERIMENTS.md). */
typedef struct CrctFp8Shadow {
  void* q; const int32_t* seg_slot; const float* scale; float* amax;
  void* qt; const int32_t* seg_in; const int64_t* seg_t_base; const int32_t* seg_t_ld;
} CrctFp8Shadow;
int64_t crct_adamw_plan(const int64_t* seg_len, int n_seg, int32_t* blk_seg, int64_t* blk_off, int64_t cap);
int crct_adamw_step(float* p, float* g, float* m, float* v, void* p_bf16,
                    const int64_t* seg_off, const int64_t* seg_len, const float* seg_lr, const float* seg_wd,
                    const int32_t* blk_seg, const int64_t* blk_off, int64_t n_blk,
                    float beta1, float beta2, float eps, int step, const float* inv_scale_dev, const CrctAmpState* amp,
                    const CrctFp8Shadow* fp8_shadow, int max_workgroups, int zero_grads, const void* g_bf16, crct_stream_t stream);
/* g_bf16 (may be NULL): bf16 gradient buffer with the element offsets of g -- the data-parallel exchange's payload
 * (crct/ddp.py: each bucket is packed to bf16, all-reduced, and consumed here as it lies: 2 B instead of 4 B per parameter on
 * the wire and in this kernel's reads).  When set, every gradient element is read from it; g is only written (zero_grads). */

/* fp8 copies of the attention results (BASELINE configs[4]): ctx also as OCP e4m3 (the input of the attention-output
 * projection's fp8 forward GEMM and weight gradient), dq / dk / dv also as OCP e5m2 (the input of the QKV projections' fp8
 * data and weight gradients).  Every copy has the shape and leading dimension of its bf16 twin (one byte per element), is
 * quantised from the bf16-rounded value with *scale (saturating) and max-es max |.| into *amax (CRCT_FP8_AMAX_LANES words);
 * dk and dv share one scale (they are columns of one fused gradient buffer).  NULL pointers = no copy.  Only the MFMA kernels
 * write copies: crct_attention_quant_ok(Tq, Tk, d) != 0 says whether a shape runs on them; the _q calls fail otherwise. */
typedef struct CrctAttnQuant {
  void* ctx_q; const float* ctx_scale; float* ctx_amax;
  void* dq_q; void* dk_q; void* dv_q;
  const float* dq_scale; float* dq_amax; const float* dkv_scale; float* dkv_amax;
  /* Row statistics kept from the forward (independent of fp8; the long-sequence kernels use them, the others ignore them):
   * crct_attention_fwd_q writes row_lse[B][heads][Tq] fp32 = log2 of the softmax denominator in the kernel's exp2 domain; a
   * crct_attention_bwd_q that is given the SAME array and the forward's output (`ctx` bf16 [B][Tq][ld_ctx], head h at column h d) drops
   * its statistics sweep: P = exp2(s - lse), delta_i = dctx_i . ctx_i.  Both NULL: everything is recomputed from q, k, v. */
  float* row_lse; const void* ctx; int64_t ld_ctx;
} CrctAttnQuant;
int crct_attention_quant_ok(int Tq, int Tk, int d);
int crct_attention_fwd_q(const void* q, const void* k, const void* v, const uint8_t* keymask, void* ctx,
                         int B, int heads, int Tq, int Tk, int d, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                         uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed, const CrctAttnQuant* quant,
                         crct_stream_t stream);
int crct_attention_bwd_q(const void* q, const void* k, const void* v, const uint8_t* keymask,
                         const void* dctx, void* dq, void* dk, void* dv,
                         int B, int heads, int Tq, int Tk, int d,
                         int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t lddq, int64_t lddk, int64_t lddv,
                         uint32_t drop_thr, float drop_scale, uint32_t drop_site, uint64_t seed, const CrctAttnQuant* quant,
                         crct_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Step engine: the whole forward + loss + backward of one batch as one native call
 * (encoder_decorator.forward + BertForMultiModalPreTraining.forward + autograd of both;
 * encoder_decorator.py:73-158, vilbert.py:1540-1661, layer order vilbert.py:852-939).
 */
typedef struct CrctModelDims {
  int32_t vocab, n_pos, n_types, H, L, heads, I;
  int32_t Fv, Hv, Lv, v_heads, Iv, Hb, b_heads, n_color;
  int32_t n_conn;
  int32_t v_biatt[32], t_biatt[32];
  int32_t fusion_sum, with_coattention;
  float p_hidden, p_attn, p_v_hidden, p_v_attn, p_cls;
} CrctModelDims;

typedef struct crct_engine crct_engine_t;

/* names: '\n'-joined parameter keys (reference state_dict names without 'bert_pretrained.'),
 * offsets / sizes: element offset and element count of each in the flat buffers; pass size 0 for a tensor
 * that never receives a gradient (it is then excluded from the backward segments' gradient ranges). */
crct_engine_t* crct_engine_create(const CrctModelDims* dims, const char* names, const int64_t* offsets,
                                  const int64_t* sizes, int n_params, int max_B, int max_T, int max_V);
void crct_engine_destroy(crct_engine_t*);
size_t crct_engine_workspace_bytes(const crct_engine_t*);
int crct_engine_num_segments(const crct_engine_t*);
/* Backward segment s (0 = heads ... last = embeddings) has finished writing gradient elements
 * [lo, hi) of the flat gradient buffer once crct_engine_backward(.., s, ..) returns. */
int crct_engine_segment_range(const crct_engine_t*, int seg, int64_t* lo, int64_t* hi);

typedef struct CrctBatch {
  const int64_t* tokens; const int64_t* segments; const float* loc; const uint8_t* text_keymask;
  const void* image_feat;    /* fp32 [B][V][Fv], or bf16 when image_feat_bf16 is set */
  const float* image_loc; const int64_t* image_target; const uint8_t* image_keymask;
  const float* R; const int64_t* labels;   /* labels NULL => evaluation */
  int32_t B, T, V;
  /* When text_keymask / image_keymask is NULL the engine builds it in its workspace (one launch, crct_build_keymasks) from
   * what the data loader ships: sep_indices int64 [B][sep_stride] + hist_len int64 [B] (key t attended iff
   * t < sep_indices[b][hist_len[b]] + 1, encoder_decorator.py:118-120) / image_mask int64 [B][V] (attended iff != 0). */
  const int64_t* sep_indices; const int64_t* hist_len; const int64_t* image_mask;
  int32_t sep_stride;
  int32_t image_feat_bf16;                 /* != 0: image_feat points at bf16 [B][V][Fv] instead of fp32 */
} CrctBatch;

typedef struct CrctStepCfg {
  int32_t training;          /* dropout on/off */
  int32_t use_l1, kind_l1;
  float tol_margin, nsp_coeff, reg_coeff, grad_scale;
  uint64_t seed;             /* dropout seed: the value, or (1<<63 | device address of a u64 holding it, re-read at run time) */
  const float* g_nsp_dev;    /* optional upstream gradients from autograd (device) */
  const float* g_reg_dev;
  const float* g_loss_dev;   /* optional device scalar: upstream gradient of the combined loss stats[0]; used when g_nsp_dev / g_reg_dev are NULL */
  const void* const* seg_ready_events;   /* optional HOST array [crct_engine_num_segments] of hipEvent_t (or NULL entries):
                                forward makes the stream(s) of the layers of segment s wait for event s before they run --
                                lets a per-segment optimizer update of step n overlap the forward of step n+1 */
  const void* const* seg_done_events;    /* optional HOST array [4 * crct_engine_num_segments] of hipEvent_t, used by
                                crct_engine_backward(seg < 0): after segment s has been enqueued, events 4s .. 4s+3 are recorded
                                on the four internal streams (text, text weight-gradient, visual, visual weight-gradient); once
                                all four have fired, the gradient range of segments 0 .. s is final -- a data-parallel caller
                                starts that range's all-reduce behind them while the rest of backward keeps running */
  /* fp8 forward (BASELINE configs[4]).  fp8 != 0 with all four pointers set: every Linear of the encoder whose two dimensions are
   * multiples of 128 (QKV, attention output / biOutput, FFN) runs its forward GEMM on e4m3 operands (fp32 accumulate) -- the inputs
   * are e4m3 copies written by the producing LayerNorm / GELU epilogue / attention kernel; embeddings, poolers and heads stay bf16.
   *   params_fp8      flat e4m3 weight shadow, element offsets of params_f32 (kept current by crct_adamw_step / CrctFp8Shadow)
   *   fp8_w_scale     device fp32 [crct_engine_fp8_weights()]: scale of weight slot i (q = w * scale)
   *   fp8_act_scale   device fp32 [crct_engine_fp8_sites()]: scales the producers quantise the activations with
   *   fp8_act_amax    device fp32 [same]: max |activation| seen by this pass, for the caller's crct_fp8_update_scales */
  int32_t fp8;               /* 0 off, 1 on, 2 calibration: copies and maxima are written, the GEMMs read the bf16 operands (a dry forward pass
                                before the first fp8 one: its maxima are those of the bf16 forward) */
  const void* params_fp8; const float* fp8_w_scale; const float* fp8_act_scale; float* fp8_act_amax;
  /* fp8 backward (configs[4], backward only; needs fp8 != 0).  fp8_bwd = 1: the data-gradient GEMMs of the same Linears (dx = dy W)
   * read an OCP e5m2 copy of dy -- written by the producing LayerNorm-backward kernel / GELU' epilogue / attention-backward kernel
   * with the per-site scales fp8_grad_scale[crct_engine_fp8_grad_sites()], maxima into fp8_grad_amax -- and the TRANSPOSED e4m3
   * weight shadow params_fp8_t (crct_fp8_transpose_weights of params_fp8, kept current by crct_adamw_step / CrctFp8Shadow.qt;
   * weight scales = fp8_w_scale).  fp8_bwd = 2: calibration -- the maxima are collected, the GEMMs run in bf16 (the first
   * backward pass).
   * fp8_wgrad != 0 (with fp8_bwd = 1): their WEIGHT gradients dW = dy^T x also read fp8 operands -- the same e5m2 copy of dy and
   * the e4m3 copy of x the forward GEMM read, both token-major, through the transposing LDS load (gemm.hip); the bias gradients
   * are column sums of the bf16 dy.  Shapes the MFMA attention kernels do not cover (crct_attention_quant_ok) keep the
   * attention-output forward and the QKV gradients of that layer in bf16. */
  int32_t fp8_bwd; int32_t fp8_wgrad;
  const void* params_fp8_t; const float* fp8_grad_scale; float* fp8_grad_amax;
  /* Host callback of crct_engine_backward(seg < 0): invoked on the calling thread, INSIDE the call, right after the four
   * seg_done_events of segment s have been recorded -- i.e. while the host is still enqueuing the rest of backward.  A
   * data-parallel caller launches the gradient all-reduce of the bucket that segment s completes from here (behind those
   * events, on its own stream), the way torch DDP's autograd hooks do (train.py:138-143): the first collective is in RCCL's
   * queue ~0.3 ms into backward instead of after the host has enqueued all of it.  Must not call back into the engine. */
  void (*seg_enqueued)(int seg, void* user);
  void* seg_enqueued_user;
  const int32_t* seg_done_mask;   /* optional HOST array [crct_engine_num_segments]: only segments with a non-zero entry get their
                                events recorded and the callback (a bucketed exchange needs them at bucket ends only); NULL = all */
  int32_t wgrad_overwrite;   /* backward only.  != 0: the caller guarantees that nothing has been accumulated into the weight
                                gradients listed by crct_engine_wgrad_owned since they were last consumed; those gradients are
                                then WRITTEN instead of added to (bit-identical to adding into zeros) and need not be zeroed --
                                no 0.96 GB zero fill and no read-modify-write of the weight gradients per step.  All other
                                gradients (biases, LayerNorm, embeddings, heads) are still accumulated and must be zero. */
  void* grads_bf16;          /* backward only, with wgrad_overwrite; may be NULL.  A bf16 buffer with the element offsets of grads_f32
                                (the data-parallel exchange's payload, crct/ddp.py): the OWNED weight gradients are then written there,
                                rounded to bf16 by the GEMM epilogue, and NOT into grads_f32 -- the exchange need not pack them (1.4 GB
                                of traffic per step less) and the weight-gradient GEMMs write half the bytes.  Every other gradient
                                still goes to grads_f32. */
  int32_t residual_fp32;     /* forward AND backward of one step alike.  != 0: the residual stream of the encoder is carried in fp32, as
                                the reference's autocast path carries it (vilbert.py:424-428 etc. add in fp32 there: the custom LayerNorm
                                runs in fp32): the pre-LayerNorm sums (Linear output + dropout + residual) are written and read as fp32, and
                                every LayerNorm also leaves an fp32 copy of its output for the next block's residual add; GEMM operands
                                stay bf16.  0: both are stored as bf16 (rounds 1 - 5).  Measured on the CPU oracle: storing them as bf16 is
                                what costs the bf16 path its gradient fidelity beyond the bf16-autocast yardstick (median cosine deficit
                                x 5 - 10 on ordinary draws); gradients of the stream stay bf16 -- rounding THEM changes nothing. */
} CrctStepCfg;

int crct_engine_forward(crct_engine_t*, const float* params_f32, const void* params_bf16,
                        const CrctBatch* batch, const CrctStepCfg* cfg, void* workspace,
                        float* logits, float* reg, float* stats, crct_stream_t stream);
/* Backward of the batch whose forward was the last crct_engine_forward on this workspace.
 * seg < 0 runs every segment, else segments must be issued in order 0 .. num_segments-1.
 * Gradients are ACCUMULATED into grads_f32 (zero it per optimizer step).  Segment 0 re-evaluates the
 * loss kernel with gradient outputs on, so logits / reg / stats are (re)written identically. */
int crct_engine_backward(crct_engine_t*, const float* params_f32, const void* params_bf16,
                         const CrctBatch* batch, const CrctStepCfg* cfg, void* workspace,
                         float* grads_f32, float* logits, float* reg, float* stats, int seg,
                         crct_stream_t stream);
/* Internal concurrency (default: both on): the visual stream's layers run on a second HIP stream and all
 * weight-gradient GEMMs / bias column sums on two more, forked from and joined to `stream` inside every call.
 * use_wgrad_streams = 2: ONE side stream for the weight gradients of both data streams (a data-parallel run gives the hardware
 * queue this frees to the gradient exchange: MI355X schedules HIP streams onto 4 hardware queues, and streams that share one
 * are serialised).  Results do not depend on the setting (tests compare them bit for bit). */
int crct_engine_set_streams(crct_engine_t*, int use_visual_stream, int use_wgrad_streams);
/* The auxiliary HIP stream of the step: the engine places its internal streams and this one on hardware queues that do not
 * collide with `main_stream`'s or with each other (probed once per engine, csrc/streams.hip: streams that share one of the
 * GPU's 4 hardware queues are serialised, and which ones share is otherwise an accident of creation order).  The host-side
 * glue runs the overlapped optimizer update (during the next forward) and the data-parallel exchange (during backward) on
 * it -- nothing else should.  queue_classes (may be NULL): hardware-queue classes the probe saw, 4 when every stream has its own. */
crct_stream_t crct_engine_aux_stream(crct_engine_t*, crct_stream_t main_stream, int* queue_classes);
/* The engine's internal streams, for labelling the stamps of crct_prof_stamp_read: out[0] = visual data stream, out[1] / out[2] = text /
 * visual weight-gradient streams, out[3] = auxiliary stream (NULL where a stream does not exist yet).  The text data stream is the caller's. */
int crct_engine_streams(crct_engine_t*, crct_stream_t out[4]);
// Where a layer's queued weight-gradient GEMMs leave for the side stream (default 1).  0: one grouped launch at the end of the layer.  Bit 0: the
// FFN block's two (with its LayerNorm column pass) right behind the FFN-up data gradient, the rest at the end of the layer.  Bit 1:
// the attention-output projection's right behind its data gradient.
// Scheduling only: the gradients are bit-identical in every mode.
int crct_engine_set_wgrad_flush(crct_engine_t* e, int mode);
// Persistent-grid policy of the engine's grouped bf16 weight-gradient launches: target_wgs workgroups (0 = one per tile) for the
// groups of a data stream with at most max_rows token rows, and only while every data stream has a side stream of its own (a
// shared side stream -- the mode a gradient exchange uses -- would become the critical path).  Defaults 96 / 3000.
int crct_engine_set_wgrad_workgroups(crct_engine_t* e, int target_wgs, int max_rows);
/* Per-site launch policy of the forward / data-gradient GEMMs (A/B switch of the developer tools and the tests; the defaults
 * are the measured choices).  phase 0 = the text-only part of the schedule (layers t0 .. before the first
 * co-attention layer: nothing else on the chip's data path), phase 1 = beside the visual stream; phase < 0 sets both.
 * cfg = kernel configuration id (-1 = the shape-class default), split_k = K slices (0 / 1 = off).  kind = CRCT_KIND_WGRAD: cfg
 * only (4 / 9: the layer's grouped weight-gradient launch takes the configuration of its first problem).  Every choice computes
 * the same function; split_k changes the summation order over K (deterministically). */
int crct_engine_set_site_policy(crct_engine_t*, int site, int kind, int phase, int cfg, int split_k);
/* The weight gradients (flat offsets / element counts into grads_f32, sorted by offset) that exactly one weight-gradient
 * GEMM per backward pass produces and nothing else adds to: every Linear weight of the encoder layers, the image embedding,
 * the poolers and the regressor pipes.  The set is fixed by the schedule at crct_engine_create (it does not depend on the
 * batch size); only these gradients are overwritten under CrctStepCfg.wgrad_overwrite.  Returns the count, fills up to `cap`. */
int crct_engine_wgrad_owned(crct_engine_t*, int64_t* offsets, int64_t* numels, int cap);
/* fp32 zero fill of `n_runs` ranges base[off[i] .. off[i] + len[i]) (device arrays off / len; blk_* from crct_adamw_plan
 * over len): the gradients that stay accumulate-only under wgrad_overwrite.  Non-temporal stores. */
int crct_zero_runs(float* base, const int64_t* off, const int64_t* len, const int32_t* blk_seg, const int64_t* blk_off,
                   int64_t n_blk, crct_stream_t stream);
/* fp8 forward: number of activation scale sites, and the (flat offset, element count) of every weight that has an e4m3 shadow
 * (index = its slot in CrctStepCfg.fp8_w_scale).  Both are fixed at crct_engine_create. */
int crct_engine_fp8_sites(const crct_engine_t*);
int crct_engine_fp8_grad_sites(const crct_engine_t*);      /* gradient scale sites of the fp8 backward (CrctStepCfg.fp8_grad_scale) */
int crct_engine_fp8_weights(const crct_engine_t*, int64_t* offsets, int64_t* numels, int cap);
/* Ordering events for same-device stream dependencies of the host-side glue (optimizer overlap, data-parallel buckets, the
 * CrctStepCfg event arrays): created without timing and without the system-scope fence a default HIP / torch event carries in
 * every record (a same-device hand-off is ~3 us shorter, the following kernels keep their L2 contents).  NOT for data the
 * host or a peer device reads. */
void* crct_event_create(void);
void crct_event_destroy(void* ev);
int crct_event_record(void* ev, crct_stream_t stream);
int crct_stream_wait_event(crct_stream_t stream, void* ev);
int crct_event_synchronize(void* ev);
int crct_event_query(void* ev);      /* 1 = complete, 0 = not yet, < 0 = error */

/* Debug taps: copy a named bf16 activation ("emb.t", "t3.t", "c0.v", "seq_t" ...) of the last
 * forward (batch B, T, V) into `out` (device, bf16); returns the element count or -1. */
int64_t crct_engine_tap(crct_engine_t*, const void* workspace, const char* name, int B, int T, int V,
                        void* out, int64_t cap, crct_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
