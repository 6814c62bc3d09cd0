/*
 * passl_hip.h — C ABI of libpassl_hip.so: the MI355X (gfx950) kernels behind the
 * MoCo-v2 ResNet-50 data-parallel training step.
 *
 * PASSL (the reference) is 100 % Python and has NO FFI / operator boundary of its own:
 * every op on the hot path is a PaddlePaddle call made from Python
 * (SURVEY.md §8b).  This header therefore defines the boundary a maintainer would bind
 * (ctypes — see INTEGRATION.md); each entry point cites the reference call site whose
 * Paddle op(s) it replaces.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *   - every function enqueues work on `stream` (hipStream_t passed as void*) and returns
 *     immediately: no allocation, no host synchronisation, graph-capture safe
 *   - returns 0 on success, negative passl_status otherwise (argument errors are detected
 *     before anything is launched)
 *   - activations are NHWC ("channels-last"), channel count a multiple of 8;
 *     `dtype` selects the arithmetic/storage type of activations and packed weights:
 *     PASSL_F32 (exact-fp32 MFMA, parity runs) or PASSL_BF16 (bf16 storage, fp32 accumulate)
 *   - master parameters, gradients, optimizer state, BN statistics, q/k/queue are fp32
 */
#ifndef PASSL_HIP_H_
#define PASSL_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* passl_stream_t;

enum passl_status {
  PASSL_OK = 0,
  PASSL_EINVAL = -1,       /* bad argument (null pointer, misaligned, unsupported shape) */
  PASSL_ELAUNCH = -2,      /* hipLaunch failed */
  PASSL_EUNSUPPORTED = -3  /* dtype/shape combination not built */
};

enum passl_dtype { PASSL_F32 = 0, PASSL_BF16 = 1 };

/* The value passl_hip_abi_version() of a library built from THIS header returns: a caller compiled against another
 * layout of the descriptors below must not call it (tools/kbench and passl_amd/hip/lib.py check at start-up). */
#define PASSL_HIP_ABI_VERSION 15
int passl_hip_abi_version(void);
/* Kernel-selection knobs (defaults are tuned for MI355X; tests use them to force a path):
 *   "igemm_ring" 0/1            use the LDS-DMA ring conv kernel when it applies (1)
 *   "igemm_ring_min_nk" n       ... only for reductions of at least n 64-element K-tiles (8)
 *   "igemm_ring_min_tiles" n    ... and launches with at least n output tiles (1)
 *   "igemm_ring_bm" 128|256     row-tile of the ring kernel: 128 (4 waves, 2 LDS stages, two
 *                               workgroups per CU; default) or 256 (8 waves, 3 stages, one)
 *   "igemm_8p" 0|1|2            the 256 x 256-tile 8-phase kernel: never / when its cost model prefers it to
 *                               the ring kernel (default) / whenever the launch is inside its envelope
 *   "igemm_8p_min_nk" n         ... (mode 1) only for reductions of at least n 64-element K-tiles (8)
 *   "igemm_8p_tk" / "_te" / "_ring_tk" / "_ring_te" / "_margin"   the cost model's constants (0.01 us per K-tile and
 *                               per tile of either kernel, margin in %: conv_igemm_8p.hip)
 *   "conv3x3_wave" 0/1          the wave-per-patch kernel (conv3x3_wave.hip) for 3x3 / stride 1 / pad 1 launches with 64 input
 *                               and 64 output channels whose image sides are multiples of 8 (ResNet-50 stage 1, forward and
 *                               data gradient; affine / ReLU, fused statistics or BatchNorm-backward epilogue): the nine weight
 *                               taps stay in LDS, every wave walks its own patches, no workgroup barrier (1);
 *                               "conv3x3_wave_rows" 4|8: 4 x 8 patches on eight waves (default) / 8 x 8 patches on four;
 *                               "conv3x3_wave_modes" bit mask of the launches that take it (1 plain / affine / ReLU, 2 fused
 *                               statistics, 4 BatchNorm-backward sums; default 7 — in-step experiments)
 *   "igemm_persist" 0/1         persistent form of the register-staged kernel for dense 1x1 launches: bit-identical, measured
 *                               slower (profiles/r06_negative_results.txt): off;  "igemm_persist_grid" n: its grid (tests)
 *   "wgrad_halo" 0|1|2          spatially tiled 3x3 / stride 1 weight-gradient kernel (conv_wgrad_halo.inc): off / images
 *                               whose sides are multiples of 8 / every such layer (default: the 8 x 8 patches overhang,
 *                               out-of-image pixels are fetched as zeros);  "wgrad_halo_stages" 2|3.
 *                               Its grid is one workgroup per 64 x 64 block of dw and slice: pass ~512 / blocks slices.
 *   "stem_pool_form" 0|1        passl_hip_bn_relu_maxpool_bwd_reduce: one 2048-item block per workgroup / workgroups that
 *                               walk over their share of the items with the next item's loads in flight (default; tensors
 *                               below 2^30 elements);  "stem_pool_wgs" n: workgroups of the latter (1024).  Both change
 *                               what passl_hip_bn_relu_maxpool_blocks returns: set them before sizing a slab.
 * Returns PASSL_EINVAL for an unknown name. */
int passl_hip_set_option(const char* name, int value);
/* Which kernel the most recent passl_hip_conv_igemm call of this process launched: 0 = igemm_kernel
 * (register-staged), 1 = igemm_ring_kernel, 2 = stem_kernel, 3 = igemm_8p_kernel, 4 = conv3x3_wave_kernel; -1
 * before the first call.
 * Diagnostics for tests and benchmarks (not thread-safe). */
int passl_hip_last_igemm_kernel(void);
/* Human readable text for a passl_status. */
const char* passl_hip_strerror(int status);

/* ---------------------------------------------------------------- flat-buffer ops */

/* k[i] = k[i]*m + q[i]*(1-m) over a flat fp32 buffer; if k_lp != NULL also writes the bf16
 * copy of the new k (the key encoder's compute weights).
 * Replaces the 269 per-tensor paddle.assign launches of
 * MoCo._momentum_update_key_encoder, passl_v110/modeling/architectures/moco.py:82-90. */
int passl_hip_ema_update(float* k, const float* q, void* k_lp, int64_t n, float m,
                         passl_stream_t stream);

/* Inference-form BatchNorm affine of ALL layers of an encoder in one launch (the key encoder's fused
 * BN: scale folded into the conv epilogues): for i < n
 *   scale[i] = flat[gamma_idx[i]] * rsqrt(flat[var_idx[i]] + eps)
 *   shift[i] = flat[beta_idx[i]] - flat[mean_idx[i]] * scale[i]
 * The index lists address the encoder's flat parameter/statistics buffer.  Replaces the per-layer
 * running-statistics BatchNorm of the frozen key encoder (passl_v110/modules/freeze.py:18-23 +
 * paddle.nn.BatchNorm2D in eval form). */
int passl_hip_bn_fold(const float* flat, const int64_t* gamma_idx, const int64_t* beta_idx,
                      const int64_t* mean_idx, const int64_t* var_idx, int64_t n, float eps,
                      float* scale, float* shift, passl_stream_t stream);

/* Momentum-SGD with L2 decay folded into the gradient, over flat fp32 buffers:
 *   g' = g*grad_scale + wd*p;  v = mu*v + g';  p = p - lr*v
 * Replaces paddle.optimizer.Momentum.step() called from
 * passl_v110/hooks/optimizer_hook.py:43-47 (rule restated in-tree at
 * passl/optimizer/momentum.py:150-158). */
int passl_hip_momentum_sgd(float* p, const float* g, float* v, int64_t n, float lr, float mu,
                           float wd, float grad_scale, passl_stream_t stream);
/* The same update with the learning rate read from DEVICE memory (hyper[0]) when the kernel runs: the value of
 * a launch is not frozen at enqueue time, so a captured HIP graph of the training step can be replayed while the
 * schedule (reference passl_v110/hooks/lr_scheduler_hook.py:26-28, stepped every iteration) moves on — the host
 * writes the new value into pinned memory, a captured H2D copy node carries it to `hyper`. */
int passl_hip_momentum_sgd_dev(float* p, const float* g, float* v, int64_t n, const float* hyper, float mu,
                               float wd, float grad_scale, passl_stream_t stream);

/* LARS momentum over a flat fp32 buffer holding many parameter tensors ("segments").
 * The buffer is cut into blocks of at most 4096 elements, each inside ONE segment
 * (blk_off: element offset, 4-aligned; blk_len; blk_seg: segment index); per segment s:
 *   local_lr = lr*lars_coeff*|p_s| / (|g_s| + wd_s*|p_s| + epsilon)   if wd_s > 0, |p_s| > 0, |g_s| > 0
 *            = lr                                                     otherwise
 *   v = mu*v + local_lr*(g*grad_scale + wd_s*p);   p = p - v
 * norms: caller-owned fp32 workspace of 2*(n_seg + n_blocks) floats ([n_seg][2] squared norms followed by
 * the per-block partial sums; fully written).  blk_seg must be ascending (the blocks of one parameter are
 * consecutive).  Three launches (per-block partial sums, fixed-order per-parameter reduction — no atomics,
 * so data-parallel replicas compute bit-identical local learning rates — and the update).
 * Replaces paddle.fluid.optimizer.LarsMomentumOptimizer.minimize called from
 * passl_v110/hooks/optimizer_hook.py:44-45 (registered at passl_v110/solver/optimizer.py:25). */
int passl_hip_lars_momentum(float* p, const float* g, float* v, const int64_t* blk_off,
                            const int32_t* blk_len, const int32_t* blk_seg, int n_blocks,
                            const float* seg_wd, int n_seg, float* norms, float lr, float mu,
                            float lars_coeff, float epsilon, float grad_scale,
                            passl_stream_t stream);
/* ... with lr = hyper[0] read on the device (see passl_hip_momentum_sgd_dev). */
int passl_hip_lars_momentum_dev(float* p, const float* g, float* v, const int64_t* blk_off,
                                const int32_t* blk_len, const int32_t* blk_seg, int n_blocks,
                                const float* seg_wd, int n_seg, float* norms, const float* hyper, float mu,
                                float lars_coeff, float epsilon, float grad_scale, passl_stream_t stream);

/* MomentumLARC over the same segmented flat buffer and workspace (blk_*, seg_wd, norms as above), lr = hyper[0]:
 *   if |p_s| != 0 and |g_s| != 0:  a = trust_coefficient*|p_s| / (|g_s| + |p_s|*wd_s + epsilon)
 *                                  (clip != 0: a = min(a / lr, 1));   g' = a*(g*grad_scale + wd_s*p)
 *   else                        :  g' = g*grad_scale                 (no weight decay)
 *   v = mu*v + g';   p = p - lr*v
 * Replaces passl/optimizer/momentum_larc.py:56-111 (MomentumLARC.step: a Python loop of per-tensor norm / scale /
 * copy ops), the optimizer of tasks/ssl/simsiam/configs/simsiam_resnet50_lp_in1k_1n8c_dp_fp32.yaml. */
int passl_hip_larc_momentum_dev(float* p, const float* g, float* v, const int64_t* blk_off,
                                const int32_t* blk_len, const int32_t* blk_seg, int n_blocks,
                                const float* seg_wd, int n_seg, float* norms, const float* hyper, float mu,
                                float trust_coefficient, float epsilon, int clip, float grad_scale,
                                passl_stream_t stream);

/* dst_bf16[i] = bf16(src[i]) (round-to-nearest-even). */
int passl_hip_cast_f32_to_bf16(const float* src, void* dst, int64_t n, passl_stream_t stream);

/* Weight (re)packing for the implicit-GEMM kernels.  One launch executes `n_jobs` jobs read
 * from device memory; each job writes   dst[c][tr][ts][k] (or [k][tr][ts][c] when
 * !transpose) = src[k][r_base + tr*r_step][s_base + ts*s_step][c]   from the fp32 master
 * weights (physically [K][R][S][C]) into the compute-dtype buffer.  `jobs` is an array of
 * passl_pack_job, `block_job`/`block_start` map each 256-thread block to (job, first element). */
typedef struct passl_pack_job {
  int64_t src_off;   /* element offset into src (fp32) */
  int64_t dst_off;   /* element offset into dst (compute dtype) */
  int32_t K, R, S, C;          /* source dims */
  int32_t TR, TS;              /* taps kept */
  int32_t r_base, r_step, s_base, s_step;
  int32_t transpose;           /* 1: dst is [C][TR][TS][K]; 0: dst is [K][TR][TS][C];
                                * fast paths chosen by the host for whole-tensor jobs: 2 = R=S=1 transpose
                                * dst[c][k] = src[k][c] (block_start = index of a 32x32 tile, k fastest),
                                * 3 = contiguous cast dst[e] = src[e] (block_start = first element) */
  int32_t c_pad;               /* dst innermost-dim padding: dst C (or K) dim is c_pad wide (>= C), zero filled; 0 = no pad */
} passl_pack_job;
int passl_hip_pack_weights(const float* src, void* dst, int dtype, const passl_pack_job* jobs,
                           const int32_t* block_job, const int32_t* block_start, int n_blocks,
                           passl_stream_t stream);

/* x fp32 NCHW [N,C,H,W] -> y NHWC [N, H+2*pad, Wp, Cp] in `dtype`, zero border / zero extra
 * channels (Wp >= W+2*pad, Cp >= C).  The stem conv reads this padded image.
 * Replaces the NCHW input handling of paddle.nn.Conv2D at
 * passl_v110/modeling/backbones/resnetimagenet.py:190-195. */
int passl_hip_nchw_to_nhwc_pad(const float* x, void* y, int N, int C, int H, int W, int pad,
                               int Wp, int Cp, int dtype, passl_stream_t stream);

/* ---------------------------------------------------------------- implicit-GEMM convolution */

/* One descriptor drives forward convs, data-gradient convs (as convs over dy with repacked
 * weights, optionally writing a strided sub-lattice of dx) and Linear layers (1x1, OP=OQ=1).
 *
 *   y[n,op,oq,col] = epi( sum_{r,s,c} A[n, op*sh + r - ph, oq*sw + s - pw, c] * B[col][r][s][c] )
 *   epi(v) = relu?( v*scale[col] + shift[col] + residual[n,op,oq,col] )
 *
 * A is addressed with explicit element strides (a_sn,a_sh,a_sw; channel stride 1); taps
 * outside [0,IH)x[0,IW) contribute zero.  B is [NCOLS][R*S*C] row-major in `dtype`.
 * If C is not a multiple of the K-tile (64 bf16 / 32 fp32) the kernel decomposes k per
 * 16-byte chunk (stem).  Output rows are addressed with (y_sn,y_sh,y_sw), channel stride 1.
 * Replaces paddle.nn.Conv2D / nn.Linear forward+backward-data as used by
 * resnetimagenet.py:114-131,190-198,216-224 and necks/base_neck.py:80-97. */
typedef struct passl_conv_desc {
  const void* a;
  const void* b;
  void* y;
  const float* scale;      /* [NCOLS] or NULL (=1) */
  const float* shift;      /* [NCOLS] or NULL (=0) */
  const void* residual;    /* addressed like y, dtype = out dtype; or NULL */
  float* stats;            /* NULL, or the slab of fused BatchNorm statistics of the STORED output values
                              (bf16 output without residual only): stats_tiles = ceil(M/128) row tiles,
                                stats[t][col][0..1] = sum (v - s), sum (v - s)^2 over the rows of tile t,
                                s = shifts[t][col] = the tile's first-row value, stored behind the sums at
                                stats + stats_tiles*NCOLS*2 (shifted sums: no cancellation).
                              stats_tiles*NCOLS*3 floats, fully written by plain stores (no atomics, no
                              zeroing): feed it to passl_hip_bn_finalize with nblocks = stats_tiles,
                              rows_per_block = 128. */
  /* BatchNorm-backward statistics fused into a data-gradient launch (all NULL/0 when unused).  The
   * launch's output is the gradient w.r.t. the OUTPUT of a BatchNorm(+ReLU) layer whose input was
   * bnb_y (addressed like y).  The epilogue masks the gradient with that layer's ReLU mask
   * (bnb_relu: 0 none, 2 recomputed as bnb_y*bnb_scale + bnb_shift > 0, 3 bit mask bnb_mask as written
   * by passl_hip_bn_apply), STORES the masked gradient g, and writes per 128-row tile t
   *   bnb_partial[bnb_tile_off + t][col][0..1] = sum g, sum g * (bnb_y - bnb_mean[col]) * bnb_invstd[col]
   * (plain stores; passl_hip_bn_bwd_finalize consumes the slab, passl_hip_bn_bwd_reduce never runs).
   * bf16 only; relu must be 0. */
  const void* bnb_y;
  const uint8_t* bnb_mask;
  const float* bnb_mean;
  const float* bnb_invstd;
  const float* bnb_scale;
  const float* bnb_shift;
  float* bnb_partial;
  int32_t N, OP, OQ;       /* M = N*OP*OQ */
  int32_t NCOLS;
  int32_t R, S, C;
  int32_t IH, IW;
  int32_t sh, sw, ph, pw;
  int64_t a_sn, a_sh, a_sw;
  int64_t y_sn, y_sh, y_sw;
  int32_t relu;
  int32_t dtype;           /* passl_dtype of A, B */
  int32_t out_f32;         /* 1: y (and residual) are fp32 regardless of dtype */
  int32_t stats_tiles;     /* must equal ceil(N*OP*OQ / 128) when stats != NULL */
  int32_t bnb_relu;
  int32_t bnb_tile_off;    /* first slab row of this launch (residue-class launches share one slab) */
  /* A SECOND BatchNorm fed by the same gradient (all NULL = none; needs bnb_partial).  The output of a bottleneck block
   * with a downsample branch is relu(bn3(y3) + bn_ds(y_ds)): the masked gradient g this launch stores is the output
   * gradient of BOTH BatchNorm layers.  With bnb2_* set the epilogue also writes, per 128-row tile t,
   *   bnb2_partial[bnb_tile_off + t][col][0..1] = sum g, sum g * (bnb2_y - bnb2_mean[col]) * bnb2_invstd[col]
   * (bnb2_y addressed like y), so the second layer's backward needs no reduce pass over (g, bnb2_y) either.
   * Built for dense 1x1 / stride-1 launches with C % 64 == 0 and NCOLS >= 128 (the conv1 data gradients that follow a
   * downsample block: resnetimagenet.py:139-153); PASSL_EUNSUPPORTED otherwise — launch without and run
   * passl_hip_bn_bwd_reduce for the second layer. */
  const void* bnb2_y;
  const float* bnb2_mean;
  const float* bnb2_invstd;
  float* bnb2_partial;
} passl_conv_desc;
int passl_hip_conv_igemm(const passl_conv_desc* d, passl_stream_t stream);

/* Weight gradient:  dw[col][r][s][c] += sum_{n,op,oq} dy[n,op,oq,col] * A[n, op*sh+r-ph, oq*sw+s-pw, c]
 * dw is fp32 [NCOLS][R*S*C] and is ACCUMULATED into (caller zeroes it).
 * dy is [M][NCOLS] dense (row stride dy_ld elements).  `splits` = number of M-slices (>=1): every
 * slice is one partial tile set.  With a workspace `ws` of at least splits*NCOLS*R*S*C floats
 * (`ws_floats`) each slice writes its tiles into its own slab and a second launch adds the slabs to dw
 * in slice order — bit-reproducible.  ws == NULL: slices > 1 accumulate with fp32 atomics (order-
 * dependent rounding).  The workspace may be shared by all layers (stream-ordered use).
 * Replaces Conv2D/Linear backward-filter (autograd of the call sites above). */
typedef struct passl_wgrad_desc {
  const void* a;
  const void* dy;
  float* dw;
  int32_t N, OP, OQ;
  int32_t NCOLS;
  int32_t R, S, C;
  int32_t IH, IW;
  int32_t sh, sw, ph, pw;
  int64_t a_sn, a_sh, a_sw;
  int64_t dy_ld;
  float* ws;               /* NULL or the split workspace (16-byte aligned) */
  int64_t ws_floats;
  int32_t dtype;
  int32_t splits;
} passl_wgrad_desc;

/* out[i] (+)= sum_{z < slabs} ws[z*n + i], slabs added in ascending order (fixed-order replacement of
 * atomic accumulation); n % 4 == 0, 16-byte aligned pointers. */
int passl_hip_slab_reduce(const float* ws, float* out, int64_t n, int slabs, int accumulate,
                          passl_stream_t stream);
int passl_hip_conv_wgrad(const passl_wgrad_desc* d, passl_stream_t stream);

/* ---------------------------------------------------------------- BatchNorm (+ReLU, +residual) */

/* Training-mode BN over NHWC rows x[M][C].  Three launches:
 *  stats:     slab b = rows [b*rpb, (b+1)*rpb), rpb = ceil(M/nblocks):
 *             partial[b][c][0..1] = sum (x - s), sum (x - s)^2 with s = shifts[b][c] = the slab's first row,
 *             shifts stored behind the sums at partial + nblocks*C*2  (nblocks*C*3 floats in total;
 *             shifted sums: the variance never comes from E[x^2] - mean^2 of large numbers).
 *             The conv epilogue writes the same layout with rpb = 128 (passl_conv_desc.stats).
 *  finalize:  fixed-order fp64 combine of the slabs (re-centred on slab 0's shift; ONE launch for any slab
 *             height since ABI 14) -> mean/invstd
 *             (biased var, eps), scale=gamma*invstd, shift=beta-mean*scale,
 *             running = momentum*running + (1-momentum)*batch   [Paddle: momentum 0.9, biased var]
 *  apply:     z = relu?( x*scale + shift + residual )
 * No atomics anywhere: results are bit-reproducible run to run.
 * Replaces paddle.nn.BatchNorm2D + ReLU (+ `out += identity`) at
 * resnetimagenet.py:133-153,196-197,236-238. */
/* Size (floats) of a `partial` buffer for nblocks slabs of C channels: the slab itself — [nblocks][C][2]
 * sums (+ [nblocks][C] shifts when shifted != 0: forward statistics) — followed by the fp64 scratch the
 * finalize kernels use to combine tall slabs segment by segment.  Every `partial` handed to
 * passl_hip_bn_finalize / passl_hip_bn_bwd_finalize (and the conv descriptor's `stats` / `bnb_partial`)
 * must be this large. */
int64_t passl_hip_bn_partial_floats(int nblocks, int C, int shifted);
int passl_hip_bn_stats(const void* x, float* partial, int64_t M, int C, int nblocks, int dtype,
                       passl_stream_t stream);
int passl_hip_bn_finalize(const float* partial, int nblocks, int64_t M, int C, int rows_per_block,
                          const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* mean, float* invstd, float* scale,
                          float* shift, passl_stream_t stream);
/* relu_mask (optional, needs relu): one bit per element of z (bit e of byte i <-> element 8*i+e,
 * set where z > 0), M*C/8 bytes — the ReLU mask the backward kernels can read instead of z. */
int passl_hip_bn_apply(const void* x, const float* scale, const float* shift, const void* residual,
                       void* z, uint8_t* relu_mask, int64_t M, int C, int relu, int dtype,
                       passl_stream_t stream);
/* Backward.  g = dz * relu-mask;   reduce: partial[b][c] = (sum g, sum g*xhat);
 * finalize: dgamma += .., dbeta += .. (accumulated) and the per-channel coefficients (A,B,Cc)
 *           of dx = A*g + B*x + Cc;
 * apply: writes dx and, if dres != NULL, dres = g (gradient of the residual branch).
 * `relu` selects where the ReLU mask comes from: 0 no ReLU; 1 `z` is the forward output
 * (mask z > 0); 2 recomputed as x*scale + shift > 0 from the forward's scale/shift (BN+ReLU
 * without residual: z is never read); 3 `z` is the bit mask written by passl_hip_bn_apply. */
int passl_hip_bn_bwd_reduce(const void* dz, const void* z, const void* x, const float* mean,
                            const float* invstd, const float* scale, const float* shift,
                            float* partial, int64_t M, int C, int nblocks, int relu, int dtype,
                            passl_stream_t stream);
int passl_hip_bn_bwd_finalize(const float* partial, int nblocks, int64_t M, int C,
                              const float* gamma, const float* mean, const float* invstd,
                              float* dgamma, float* dbeta, float* coef /* [3][C] */,
                              passl_stream_t stream);
/* Cross-rank (Sync) BatchNorm — reference passl/models/simsiam.py:160-162 (nn.SyncBatchNorm.convert_sync_batchnorm
 * under data parallelism): statistics over the batches of ALL ranks.  Forward: bn_moments folds this rank's slab to
 * fp64 {mean[C], M2[C], n[C]} (mom: 3*C doubles); the caller all-gathers them (rank order) and bn_finalize_moments
 * combines them with Chan's update and produces what bn_finalize produces (running statistics from the GLOBAL batch).
 * Backward: bn_bwd_sums folds the slab to fp64 {sum g, sum g*xhat} (2*C doubles), all-gathered; bn_bwd_finalize_sums
 * accumulates THIS rank's sums into dgamma / dbeta and writes the apply coefficients from the totals and the global
 * row count.  The collectives are the caller's (torch.distributed): the library never communicates. */
int passl_hip_bn_moments(const float* partial, int nblocks, int64_t M, int C, int rows_per_block, double* mom,
                         passl_stream_t stream);
int passl_hip_bn_finalize_moments(const double* mom_all, int world, int C, const float* gamma, const float* beta,
                                  float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                  float* invstd, float* scale, float* shift, passl_stream_t stream);
int passl_hip_bn_bwd_sums(const float* partial, int nblocks, int64_t M, int C, double* sums, passl_stream_t stream);
int passl_hip_bn_bwd_finalize_sums(const double* sums_all, int world, int rank, int64_t M_total, int C,
                                   const float* gamma, const float* mean, const float* invstd, float* dgamma,
                                   float* dbeta, float* coef, passl_stream_t stream);
int passl_hip_bn_bwd_apply(const void* dz, const void* z, const void* x, const float* coef,
                           const float* scale, const float* shift, void* dx, void* dres, int64_t M,
                           int C, int relu, int dtype, passl_stream_t stream);
/* ---------------------------------------------------------------- pooling */

/* 3x3 stride-2 pad-1 max pool, NHWC; idx[n,p,q,c] (uint8) = winning tap r*3+s (first max in
 * row-major window order).  Replaces nn.MaxPool2D at resnetimagenet.py:198,239. */
int passl_hip_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int N, int H, int W, int C,
                               int dtype, passl_stream_t stream);
int passl_hip_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W,
                               int C, int dtype, passl_stream_t stream);
/* Training-mode BatchNorm + ReLU + the 3x3 / stride 2 / pad 1 max pool of the ResNet stem as ONE pass per direction
 * (csrc/stem_pool.hip): the BatchNorm output z and the pool's input gradient dz are never written.  x = the
 * BatchNorm input [N,H,W,C] (the stem convolution's output), scale / shift / mean / invstd / coef as produced by
 * passl_hip_bn_finalize / passl_hip_bn_bwd_finalize, idx as passl_hip_maxpool3x3s2_fwd writes it.
 *   fwd:        y[n,p,q,c] = max over the window of T(relu(x*scale + shift)), idx = the winning tap
 *   bwd_reduce: partial[b][c][0..1] = sum g, sum g*(x - mean)*invstd with g = T(max-pool gradient of dy) masked by
 *               x*scale + shift > 0; nblocks = passl_hip_bn_relu_maxpool_blocks(N,H,W,C) slab rows (buffer sized by
 *               passl_hip_bn_partial_floats(nblocks, C, 0)), consumed by passl_hip_bn_bwd_finalize with M = N*H*W.
 *               idx must come from the forward pass (a tap outside the image is never the arg-max).
 *   bwd_apply:  dx = A g + B x + C
 * Element for element the arithmetic of passl_hip_bn_apply -> maxpool3x3s2_fwd resp. maxpool3x3s2_bwd -> bn_bwd_reduce
 * (relu = 2) -> bn_bwd_apply: y, idx and (for equal coefficients) dx are bit-identical to that chain; the slab is
 * summed in another fixed order.  C/8 must divide 256; PASSL_EUNSUPPORTED otherwise (use the separate passes).
 * Replaces BatchNorm2D + ReLU + MaxPool2D at resnetimagenet.py:196-198 and their autograd. */
int passl_hip_bn_relu_maxpool_blocks(int N, int H, int W, int C);
int passl_hip_bn_relu_maxpool_fwd(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx, int N,
                                  int H, int W, int C, int dtype, passl_stream_t stream);
int passl_hip_bn_relu_maxpool_bwd_reduce(const void* dy, const uint8_t* idx, const void* x, const float* mean,
                                         const float* invstd, const float* scale, const float* shift, float* partial,
                                         int nblocks, int N, int H, int W, int C, int dtype, passl_stream_t stream);
int passl_hip_bn_relu_maxpool_bwd_apply(const void* dy, const uint8_t* idx, const void* x, const float* coef,
                                        const float* scale, const float* shift, void* dx, int N, int H, int W, int C,
                                        int dtype, passl_stream_t stream);
/* Global average pool [N,HW,C] -> [N,C] and its backward.  Replaces AdaptiveAvgPool2D((1,1)) at
 * necks/base_neck.py:79,94. */
int passl_hip_avgpool_fwd(const void* x, void* y, int N, int HW, int C, int dtype,
                          passl_stream_t stream);
int passl_hip_avgpool_bwd(const void* dy, void* dx, int N, int HW, int C, int dtype,
                          passl_stream_t stream);
/* dx = dy * (y > 0), n elements (multiple of 8) — backward of the projector's ReLU
 * (necks/base_neck.py:83). */
int passl_hip_relu_bwd(const void* dy, const void* y, void* dx, int64_t n, int dtype,
                       passl_stream_t stream);
/* out[c] = sum_m x[m][c] (fp32 out) — Linear bias gradient.  Rows are cut into at most 1024 slabs;
 * with a workspace `ws` (>= 1024*C floats, `ws_floats` = its size) the slab partials are added in
 * slab order (bit-reproducible); ws == NULL falls back to fp32 atomics when there is more than one slab. */
int passl_hip_colsum(const void* x, float* out, int64_t M, int C, int dtype, float* ws,
                     int64_t ws_floats, passl_stream_t stream);
/* out[c] += sum_m x[m][c] — accumulates straight into the (zero-initialised) gradient buffer. */
int passl_hip_colsum_acc(const void* x, float* out, int64_t M, int C, int dtype, float* ws,
                         int64_t ws_floats, passl_stream_t stream);

/* ---------------------------------------------------------------- contrastive head */

/* y = x / max(||x||_2, eps) per row (fp32), saves the clamped norm.
 * Replaces F.normalize(axis=1) at moco.py:159,170. */
int passl_hip_l2norm_fwd(const float* x, float* y, float* norm, int N, int D, float eps,
                         passl_stream_t stream);
/* dx = (dy - y*(y.dy)) / norm ; written as fp32 (dtype F32) or bf16. */
int passl_hip_l2norm_bwd(const float* dy, const float* y, const float* norm, void* dx, int N,
                         int D, int dtype, passl_stream_t stream);
/* SimSiam's criterion (reference passl/models/simsiam.py:69,93: nn.CosineSimilarity(axis=1) of the predictor output
 * against the stop-gradient projector output, negated and averaged): loss[0] = -(1/N) sum_i a_i.b_i / max(|a_i||b_i|,
 * eps); a, b fp32 [N,D]; stats [N,4] is saved for the backward.  The mean is one fixed-order sum. */
int passl_hip_cosine_loss_fwd(const float* a, const float* b, int N, int D, float eps, float* stats, float* loss,
                              passl_stream_t stream);
/* da [N,D] = gloss[0] * d loss / d a  (b is a constant); gloss: device scalar. */
int passl_hip_cosine_loss_bwd(const float* a, const float* b, const float* stats, const float* gloss, int N, int D,
                              float* da, passl_stream_t stream);

/* Fused InfoNCE forward (moco.py:178-180 + heads/contrastive_head.py:47-78):
 *   l_pos[i] = q[i].k[i];  l_neg[i][j] = q[i].queue[:,j];  logits = [l_pos | l_neg]/T
 *   loss = mean_i( logsumexp(logits[i]) - logits[i][0] ),  acc1/acc5 = % rows whose column 0
 *   has rank < 1 / < 5 among strictly greater logits.
 * q,k: [N][D] fp32, queue: [D][K] fp32 (dim-major, as the reference stores it).
 * Outputs: out[0..2] = loss, acc1, acc5 (fixed-order mean, fully written); row_lse[N]; optional
 * logits [N][K+1] (may be NULL).
 * workspace: >= passl_hip_infonce_workspace_bytes(N,K) bytes. D must be 128, K % 128 == 0. */
int64_t passl_hip_infonce_workspace_bytes(int N, int K);
int passl_hip_infonce_fwd(const float* q, const float* k, const float* queue, int N, int D, int K,
                          float T, float* out, float* row_lse, float* logits, void* workspace,
                          passl_stream_t stream);
/* dq[i] = gscale/(N*T) * ( (p_i0 - 1)*k[i] + sum_j p_ij * queue[:,j] ),  p = softmax(logits).
 * dq is fully written (no zeroing): every 128-column queue slice writes its own [N][D] slab of
 * `workspace` (>= passl_hip_infonce_bwd_workspace_bytes(N,K) bytes) and the slabs are added in slice
 * order — no atomics, bit-reproducible.  `gscale` device pointer to the upstream scalar gradient
 * (or NULL = 1). */
int64_t passl_hip_infonce_bwd_workspace_bytes(int N, int K);
int passl_hip_infonce_bwd(const float* q, const float* k, const float* queue,
                          const float* row_lse, const float* gscale, int N, int D, int K, float T,
                          float* dq, void* workspace, passl_stream_t stream);
/* queue[:, ptr:ptr+B] = keys^T  (keys [B][D]).  Replaces the slice-assign of
 * MoCo._dequeue_and_enqueue, moco.py:101-102 (the pointer arithmetic stays on the host). */
int passl_hip_enqueue(float* queue, const float* keys, int D, int K, int ptr, int B,
                      passl_stream_t stream);
/* The same with the pointer in DEVICE memory (MoCo's `queue_ptr` buffer itself, int64[1], moco.py:80): the launch
 * reads *ptr, writes the keys and then advances *ptr = (*ptr + B) % K — nothing about the step's position in
 * the queue is frozen at enqueue time (HIP-graph replay).  K % B == 0 (moco.py:99). */
int passl_hip_enqueue_dev(float* queue, const float* keys, int D, int K, int64_t* ptr, int B,
                          passl_stream_t stream);

/* Fused NT-Xent + CO2 head of SimCLR (passl_v110/modeling/heads/simclr_contrastive_head.py:42-102).
 * a, b: [B][D] fp32 rows of this rank (hidden1, hidden2); a_all, b_all: [BL][D] the column sets
 * (a_all = a, b_all = b, BL = B, row_offset = 0 reproduces the reference, which never gathers;
 * with all-gathered embeddings row i's positive column is row_offset + i).  D must be 128, BL >= 2.
 *   loss = mean_i( LSE([ab_i|aa_i]) - ab_i,pos + LSE([ba_i|bb_i]) - ba_i,pos )
 *          + co2_weight * sum_i( KL(Pb_i||Pa_i) + KL(Pa_i||Pb_i) ) / B          (self/positive masks
 *   as in the reference), acc1 = fraction of rows whose positive is the arg-max of ab_i.
 * out[0..1] = loss, acc1; rowstats [B][8] = (lse_ce_a, lse_ce_b, lse_Pa, lse_Pb, co2_i, pos, rank, 0)
 * is what the backward needs.  The B x BL logits are never written. */
int passl_hip_ntxent_fwd(const float* a, const float* b, const float* a_all, const float* b_all,
                         int B, int BL, int row_offset, int D, float T, float co2_weight,
                         float* out, float* rowstats, passl_stream_t stream);
/* Gradients (accumulated with atomics into zeroed buffers): da, db [B][D] = through the row role,
 * da_all, db_all [BL][D] = through the column role (the caller adds / reduce-scatters them).
 * The kl_div target carries no gradient (Paddle's kldiv_loss_grad). gscale: device scalar or NULL. */
int passl_hip_ntxent_bwd(const float* a, const float* b, const float* a_all, const float* b_all,
                         const float* rowstats, const float* gscale, int B, int BL, int row_offset,
                         int D, float T, float co2_weight, float* da, float* db, float* da_all,
                         float* db_all, passl_stream_t stream);

/* ---------------------------------------------------------------- ViT / MAE
 * Reference: class MAE, passl_v110/modeling/backbones/mae.py:318-564 (= passl/models/mae.py:37-290),
 * Mlp/Attention/Block :61-189.  Activations [rows][C] in `dtype`, C % 8 == 0. */

/* y = (x - mean)/sqrt(var + eps) * gamma + beta per row (biased var); saves mean, rstd [M]. */
int passl_hip_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y,
                            float* mean, float* rstd, int64_t M, int C, float eps, int dtype,
                            passl_stream_t stream);
/* dx (+ dres when non-NULL: the gradient arriving through the residual branch that forked off x, so
 * that `x + f(LN(x))` needs no separate add); dgamma/dbeta (fp32 [C]) are ACCUMULATED into: every block writes
 * its column sums to a slab of `ws` (>= passl_hip_layernorm_bwd_ws_floats(M, C) floats) and a second launch adds
 * the slabs in block order — no floating-point atomics, the result is bit-reproducible.  C <= 2048. */
int64_t passl_hip_layernorm_bwd_ws_floats(int64_t M, int C);
int passl_hip_layernorm_bwd(const void* dy, const void* x, const float* gamma, const float* mean,
                            const float* rstd, const void* dres, void* dx, float* dgamma,
                            float* dbeta, int64_t M, int C, int dtype, float* ws, int64_t ws_floats,
                            passl_stream_t stream);
/* layernorm_bwd with dgamma == dbeta == NULL leaves its per-block partial sums in ws (which must then be the caller's
 * own buffer, not a shared scratch); this folds them — dgamma / dbeta += the fixed-order sums — on any stream. */
int passl_hip_layernorm_param_reduce(const float* ws, int64_t M, int C, float* dgamma, float* dbeta,
                                     passl_stream_t stream);
/* exact (erf) GELU and its backward dx = dy * gelu'(x); n % 8 == 0. */
int passl_hip_gelu_fwd(const void* x, void* y, int64_t n, int dtype, passl_stream_t stream);
int passl_hip_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype,
                       passl_stream_t stream);
/* y = tanh(x);  dx = dy * (1 - tanh(x)^2)   (n a multiple of 8, 16-byte aligned).  Reference: the
 * `representation_size` head of the v2 VisionTransformer, passl/models/vision_transformer.py:318-321,340-343. */
int passl_hip_tanh_fwd(const void* x, void* y, int64_t n, int dtype, passl_stream_t stream);
int passl_hip_tanh_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype, passl_stream_t stream);
/* Fused softmax(q k^T * scale) v per (image, head) on the fused projection qkv [B,T,3,H,DH];
 * out [B,T,H,DH]; lse [B,H,T] (row log-sum-exp, saved for the backward).  DH in {32, 64},
 * T <= 208 (PASSL_EUNSUPPORTED otherwise).  Replaces Attention.forward, mae.py:141-155.
 * causal != 0: key j visible to query i iff j <= i (the CLIP text tower's additive triu(-inf, 1)
 * mask, passl_v110/modeling/backbones/clip.py:284-286 + vision_transformer.py:107-113). */
int passl_hip_attention_fwd(const void* qkv, void* out, float* lse, int B, int T, int H, int DH,
                            float scale, int causal, int dtype, passl_stream_t stream);
/* dqkv [B,T,3,H,DH] (fully written) from dout [B,T,H,DH]. */
int passl_hip_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse,
                            void* dqkv, int B, int T, int H, int DH, float scale, int causal,
                            int dtype, passl_stream_t stream);
/* random_masking (mae.py:461-488) without the sort: ids_restore[b,l] = rank of noise[b,l] in its row
 * (ties by index), ids_keep[b,rank] = l for rank < len_keep, mask[b,l] = rank >= len_keep. */
int passl_hip_mae_mask(const float* noise, int B, int L, int len_keep, int32_t* ids_keep,
                       int32_t* ids_restore, float* mask, passl_stream_t stream);
/* encoder input (mae.py:494-503): out[b,0] = cls + pos[0]; out[b,1+k] = x[b,ids_keep[b,k]] +
 * pos[1+ids_keep[b,k]];  x [B,L,D], pos [L+1,D], out [B,K+1,D]. */
int passl_hip_mae_gather(const void* x, const float* cls, const float* pos, const int32_t* ids_keep,
                         void* out, int B, int L, int K, int D, int dtype, passl_stream_t stream);
/* its backward: dx [B,L,D] fully written (zeros at masked patches), dcls += sum_b dout[b,0] (images added in
 * a fixed order). */
int passl_hip_mae_gather_bwd(const void* dout, const int32_t* ids_restore, void* dx, float* dcls,
                             int B, int L, int K, int D, int dtype, passl_stream_t stream);
/* decoder input (mae.py:516-527): out[b,0] = x[b,0] + pos[0]; out[b,1+l] = (r = ids_restore[b,l]) < K
 * ? x[b,1+r] : mask_token, + pos[1+l];  x [B,K+1,D], out [B,L+1,D]. */
int passl_hip_mae_unshuffle(const void* x, const float* mask_token, const float* pos,
                            const int32_t* ids_restore, void* out, int B, int L, int K, int D,
                            int dtype, passl_stream_t stream);
/* its backward: dx [B,K+1,D] fully written, dmask_token += sum over masked positions (token-block slabs in
 * `ws`, >= 1024 * D floats, added in order: no atomics). */
int passl_hip_mae_unshuffle_bwd(const void* dout, const int32_t* ids_keep, const int32_t* ids_restore,
                                void* dx, float* dmask_token, int B, int L, int K, int D, int dtype,
                                float* ws, int64_t ws_floats, passl_stream_t stream);
/* imgs fp32 NCHW -> out [B*L, p*p*C] in `dtype`, column order (ph, pw, c): the patch-embed conv
 * (mae.py:87-121) as a GEMM, and MAE.patchify's order (mae.py:433-445). */
int passl_hip_patchify(const float* img, void* out, int B, int C, int H, int W, int p, int dtype,
                       passl_stream_t stream);
/* forward_loss (mae.py:541-557): pred fp32 [B, L+1, P] (row 0 of every image = cls, ignored);
 * loss[0] = sum_l mask * mean_P (pred - target)^2 / denom, target = patches of img, normalised per
 * patch ((x - mean)/sqrt(var_unbiased + 1e-6)) when norm_pix.  denom = sum(mask).  ws: B * (L + 1) floats (the
 * per-patch terms, summed in one fixed order). */
int passl_hip_mae_loss_fwd(const float* img, const float* pred, const float* mask, float* loss, int B,
                           int C, int H, int W, int p, int norm_pix, float denom, float* ws,
                           int64_t ws_floats, passl_stream_t stream);
int passl_hip_mae_loss_bwd(const float* img, const float* pred, const float* mask,
                           const float* gscale, float* dpred, int B, int C, int H, int W, int p,
                           int norm_pix, float denom, passl_stream_t stream);
/* AdamW over a flat fp32 buffer (paddle adamw op): p *= 1 - lr*wd; m = b1 m + (1-b1) g;
 * v = b2 v + (1-b2) g^2; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps*sqrt(1-b2^t)), with
 * g scaled by grad_scale.  Replaces paddle.optimizer.AdamW.step (solver/optimizer.py:22). */
int passl_hip_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float epsilon, float weight_decay, float beta1_pow, float beta2_pow,
                    float grad_scale, passl_stream_t stream);
/* ... with the step-dependent scalars read on the device: hyper = {lr, beta1^t, beta2^t} (see
 * passl_hip_momentum_sgd_dev).  Both forms derive sqrt(1-b2^t) etc. inside the kernel: identical bits. */
int passl_hip_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1,
                        float beta2, float epsilon, float weight_decay, float grad_scale, passl_stream_t stream);

/* ---------------------------------------------------------------- measurement hooks */

/* When enabled, every passl_hip_conv_igemm / passl_hip_conv_wgrad launch is bracketed by HIP
 * events on its own stream; passl_hip_prof_collect synchronises those events and returns the
 * accumulated kernel time (ms) and launch count per kernel class (0 = ring igemm, 1 = wgrad, 2 = register-staged
 * igemm / stem, 3 = 8-phase igemm). */
int passl_hip_prof_enable(int on);
int passl_hip_prof_collect(int kernel_class, double* total_ms, int64_t* launches);
/* Algorithmic work of the launches timed since the last call for this class: FLOPs (2 x MACs of the real
 * taps) and HBM bytes (every operand element once).  Classes: 0 igemm_ring_kernel, 1 weight gradients,
 * 2 igemm_kernel (register-staged). */
int passl_hip_prof_collect_work(int kernel_class, double* flops, double* bytes);
/* Average reading (us) of n back-to-back event pairs with nothing in between on `stream`: what the event bracket of
 * prof_enable adds to a kernel's own duration (synchronises the stream). */
int passl_hip_prof_event_overhead(int n, passl_stream_t stream, double* avg_us);

/* ---------------------------------------------------------------- CLIP
 * Reference: class CLIP, passl_v110/modeling/backbones/clip.py:183-336; QuickGELU
 * base_transformer.py:25-28; CLIPHead, passl_v110/modeling/heads/clip_head.py:24-36. */

/* QuickGELU y = x * sigmoid(1.702 x) and dx = dy * d/dx; n % 8 == 0. */
int passl_hip_quick_gelu_fwd(const void* x, void* y, int64_t n, int dtype, passl_stream_t stream);
int passl_hip_quick_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int dtype,
                             passl_stream_t stream);
/* encode_text input (clip.py:295-298): out[b,t] = table[text[b,t]] + pos[t]; text int64 [B,T],
 * table fp32 [vocab,C], pos fp32 [T,C], out [B*T,C] in `dtype`.  Ids outside [0,vocab) read as 0. */
int passl_hip_embed_fwd(const int64_t* text, const float* table, const float* pos, void* out, int B,
                        int T, int C, int vocab, int dtype, passl_stream_t stream);
/* its backward: dtable[text[b,t]] += dout[b,t] (scatter-add), dpos[t] += sum_b dout[b,t]; both fp32,
 * ACCUMULATED into.  C <= 2048.  Bit-reproducible: the scatter-add accumulates 64-bit FIXED-POINT values (one
 * power-of-two scale per launch, chosen from max|dout|) with integer atomics — exact and order-independent —
 * in `acc` (>= passl_hip_embed_bwd_acc_bytes(vocab, C) bytes, a PERSISTENT buffer of the caller that is zero
 * before the first call and is left zero by every call); the position sums go through slabs in `ws`
 * (>= passl_hip_embed_bwd_ws_floats(B, T, C) floats, per-call scratch). */
int64_t passl_hip_embed_bwd_acc_bytes(int vocab, int C);
int64_t passl_hip_embed_bwd_ws_floats(int B, int T, int C);
int passl_hip_embed_bwd(const int64_t* text, const void* dout, float* dtable, float* dpos, int B,
                        int T, int C, int vocab, int dtype, void* acc, int64_t acc_bytes, float* ws,
                        int64_t ws_floats, passl_stream_t stream);
/* out[r] = x[idx[r]] for r < n (class-token rows x[:, 0], EOT rows x[i][argmax text[i]]). */
int passl_hip_gather_rows(const void* x, const int32_t* idx, void* out, int n, int C, int dtype,
                          passl_stream_t stream);
/* backward of gather_rows for distinct idx: dx [rows_total, C] = 0 except dx[idx[r]] = dout[r]. */
int passl_hip_scatter_rows(const void* dout, const int32_t* idx, void* dx, int n, int64_t rows_total,
                           int C, int dtype, passl_stream_t stream);
/* idx[b] = b*T + argmax_t text[b,t] (first maximum) — clip.py:303-306. */
int passl_hip_eot_index(const int64_t* text, int B, int T, int32_t* idx, passl_stream_t stream);
/* CLIP.forward's logits (clip.py:317-336) from fp32 features img, txt [B,D] (D % 16 == 0):
 * logits [B,B] = exp(logit_scale) * (img/|img|) (txt/|txt|)^T = image_logits; text_logits is its
 * transpose.  logit_scale (device scalar, the parameter) is then clipped in place to
 * [clip_lo, clip_hi].  ws: caller-owned fp32 workspace of passl_hip_clip_logits_ws_floats(B, D) floats
 * that must survive until the backward call. */
int64_t passl_hip_clip_logits_ws_floats(int B, int D);
int passl_hip_clip_logits_fwd(const float* img, const float* txt, float* logit_scale, int B, int D,
                              float clip_lo, float clip_hi, float* ws, float* logits,
                              passl_stream_t stream);
/* dimg, dtxt [B,D] fully written; dlogit_scale[0] += sum(dlogits .* logits) (<= 256 block partials in
 * `scratch` (>= 256 floats), added in block order). */
int passl_hip_clip_logits_bwd(const float* dlogits, const float* logits, const float* ws, int B, int D,
                              float* dimg, float* dtxt, float* dlogit_scale, float* scratch,
                              passl_stream_t stream);
/* Building blocks of the CROSS-RANK form of the same logits (BASELINE configs[4]: the negatives of
 * every rank; pattern of passl/models/mocov3.py:187-198 — gather the other modality's features, labels
 * arange(B) + B*rank): image_logits = exp(s) * I^_local . T^_all^T  [B][W*B] and the text counterpart.
 *   clip_scale:  alpha = exp(*logit_scale), then *logit_scale = clip(*logit_scale) (clip.py:309-311)
 *   gemm_f32_nt: C[M][N] = *alpha * A[M][K] . B[N][K]^T        (exact-fp32 MFMA; K % 16 == 0)
 *   gemm_f32_gx: C[M][N] = *alpha * op(G) . X[K][N], op(G) = G [M][K] (trans 0) or G^T, G [K][M] (trans 1)
 *   dot_acc:     *out += sum_e a[e]*b[e]  (fixed order: <= 256 block partials in ws (>= 256 floats), then one sum)
 * Row normalisation without epsilon = passl_hip_l2norm_fwd/bwd with eps 0; the row cross-entropy with
 * offset labels = passl_hip_softmax_ce_fwd/bwd. */
int passl_hip_clip_scale(float* logit_scale, float* alpha, float clip_lo, float clip_hi,
                         passl_stream_t stream);
int passl_hip_gemm_f32_nt(const float* A, const float* B, float* C, int M, int N, int K, const float* alpha,
                          passl_stream_t stream);
int passl_hip_gemm_f32_gx(const float* G, const float* X, float* C, int M, int N, int K, int trans,
                          const float* alpha, passl_stream_t stream);
int passl_hip_dot_acc(const float* a, const float* b, int64_t n, float* out, float* ws,
                      passl_stream_t stream);
/* CLIPHead (clip_head.py:24-36) with labels arange(B): out = {CE over the rows of logits (img_loss),
 * CE over its columns (= rows of text_logits; text_loss), their sum (loss)}; lse [2B] = row and
 * column log-sum-exp, saved for the backward.  ws: 2 * B floats (per-row / per-column terms, summed in a fixed
 * order). */
int passl_hip_clip_ce_fwd(const float* logits, int B, float* lse, float* out, float* ws, int64_t ws_floats,
                          passl_stream_t stream);
/* dlogits[i][j] = gloss/B * (softmax_row_i[j] + softmax_col_j[i] - 2 [i == j]); gloss: device scalar. */
int passl_hip_clip_ce_bwd(const float* logits, const float* lse, const float* gloss, int B,
                          float* dlogits, passl_stream_t stream);

/* ---------------------------------------------------------------- linear probe
 * Reference: ClasHead.loss + accuracy, passl_v110/modeling/heads/clas_head.py:47-72. */

/* scores fp32 [N,C], labels int64 [N] -> lse [N] (row log-sum-exp, saved for the backward),
 * out = {mean cross-entropy, acc1 (%), acc5 (%)}.  Top-k membership by rank counting (ties resolve to
 * the lower index).  A label outside [0,C) makes the loss NaN.  ws: 3 * N floats (per-row terms, summed in a
 * fixed order). */
int passl_hip_softmax_ce_fwd(const float* scores, const int64_t* labels, int N, int C, float* lse,
                             float* out, float* ws, int64_t ws_floats, passl_stream_t stream);
/* dscores[i][j] = gloss/N * (softmax(scores_i)[j] - [j == labels[i]]); gloss: device scalar. */
int passl_hip_softmax_ce_bwd(const float* scores, const float* lse, const int64_t* labels,
                             const float* gloss, int N, int C, float* dscores, passl_stream_t stream);

/* ---------------------------------------------------------------- native step plans
 * Reference: the iteration of passl_v110/engine/trainer.py:287-337 (model forward, then OptimizerHook:
 * clear_grad -> backward -> step, hooks/optimizer_hook.py:25-50) is ~1 400 launches here, each of which the
 * reference-side host code (Python) would issue one by one.  A plan is that launch list recorded ONCE while a
 * step executes normally, and replayed from one call per segment:
 *   create -> record_begin -> [the step runs: every kernel this library launches, on any thread, is appended
 *   with its stream and a copy of its arguments; the host reports its cross-stream edges with event_record /
 *   stream_wait and closes a segment with cut wherever something that is not a library launch must happen in
 *   between, e.g. a collective] -> record_end -> replay(segment) ... -> destroy.
 * Contract of a replay: every pointer a recorded launch carried must still be valid and mean the same thing
 * (the host keeps the recorded step's allocations in a private pool and step-varying scalars in device memory);
 * streams are the recorded ones.  One plan records at a time per process; a plan records once.
 * event_record returns the id (>= 0) of a plan-owned event = "everything enqueued on `stream` up to here";
 * while recording nothing is enqueued for it (the executing step orders itself with the caller's own events).
 * plan_info(what): 0 segments, 1 kernel launches, 2 event records, 3 stream waits, 4 memsets, 5 argument bytes,
 * 6 replays so far, 7 distinct streams. */
typedef struct passl_plan passl_plan_t;
int passl_hip_plan_create(passl_plan_t** out);
int passl_hip_plan_destroy(passl_plan_t* plan);
int passl_hip_plan_record_begin(passl_plan_t* plan);
int passl_hip_plan_cut(passl_plan_t* plan);                 /* -> index of the segment that starts here */
int passl_hip_plan_record_end(passl_plan_t* plan);
int passl_hip_plan_event_record(passl_plan_t* plan, passl_stream_t stream);
int passl_hip_plan_stream_wait(passl_plan_t* plan, passl_stream_t stream, int event_id);
int passl_hip_plan_replay(passl_plan_t* plan, int segment);
int64_t passl_hip_plan_info(passl_plan_t* plan, int what);

/* The step's remaining framework launches as library kernels, so that a recorded step holds library launches
 * only (reference call sites: `clear_grad()` optimizer_hook.py:31; `self.queue.clone().detach()` moco.py:180;
 * the implicit dtype casts Paddle inserts around fp32 heads; the stem filter's padded gradient):
 *   fill_zero: p[0:bytes] = 0                      copy_bytes: dst[0:bytes] = src[0:bytes] (no overlap)
 *   cast_bf16_to_f32: dst[i] = float(src[i])       (both 16-byte aligned)
 *   unpad_add: dst[row][r][s][c] += src[row][r][s][c], s < dst_S, c < dst_C; src rows are [R][src_S][src_C] */
int passl_hip_fill_zero(void* p, int64_t bytes, passl_stream_t stream);
int passl_hip_copy_bytes(void* dst, const void* src, int64_t bytes, passl_stream_t stream);
int passl_hip_cast_bf16_to_f32(const void* src, float* dst, int64_t n, passl_stream_t stream);
int passl_hip_unpad_add(const float* src, float* dst, int64_t rows, int R, int dst_S, int dst_C, int src_S,
                        int src_C, passl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PASSL_HIP_H_ */
