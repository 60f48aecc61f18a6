/*
 * ttk.h - C-ABI of libttk_hip.so: the MI355X (gfx950) kernels behind the pose-estimator training
 * step of opentrack/neuralnet-tracker-traincode.
 *
 * The reference has NO native boundary for this path (it is 100 % Python calling torch ATen ops,
 * SURVEY.md §0.1, §8b): every entry point below replaces a group of ATen calls issued by the
 * reference file:line that is cited next to it (paths relative to the reference's
 * trackertraincode/).  The Python host (neuralnet-tracker-traincode_amd/trackertraincode/) binds
 * these symbols with ctypes; INTEGRATION.md shows the binding a maintainer of the reference would
 * add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (PyTorch's
 *    caching allocator in the shipped host code).  The library never allocates, frees or
 *    synchronises, keeps no mutable global state except the thread-local error string, and
 *    launches only on the stream it is given (backward runs on autograd's worker thread).
 *  - return value: 0 = ok; negative = argument/shape error detected on the host before any launch;
 *    positive = hipError_t of the failed launch.  ttk_last_error_string() describes the last error
 *    of the calling thread.
 *  - Activation layout.  The network input x[B][1][H][W] is the reference's NCHW tensor.  Every activation-sized tensor of the
 *    MobileNet path (raw conv outputs y, materialised block inputs, their gradients) is stored as CHANNEL BLOCKS of 32:
 *        [C / 32][M][32],  M = B*H*W pixels in (n, h, w) order;  element (m, c) at ((c >> 5) * M + m) * 32 + (c & 31)
 *    (C = 32: plain channels-last).  A depthwise workgroup's 32-channel slab and a GEMM's k32 step are then contiguous runs of
 *    pixels x 128 bytes; over channels-last rows the same kernels touched 128-byte pieces of 4C-byte rows and streamed 10-20 %
 *    slower (profiles/r03_stream_sweep.txt).  These tensors only travel from kernel to kernel; ttk_bn_act hands out a plain
 *    channels-last copy (MobileNet's intermediate feature maps), ttk_avgpool_fwd the [B][C] features.
 *    The ResNet18 entry points (ttk_conv_*, ttk_stem7_*, ttk_maxpool_*, ttk_bn_add_act, ttk_bn_bwd_apply) keep channels-last rows
 *    y[n][h][w][c]; ttk_avgpool_* serve both (TTK_LAYOUT_ROWS).
 *  - training-mode BatchNorm is split in three: the producing conv writes its RAW output y and
 *    per-workgroup partial sums  part[row][0][c] = sum(y), part[row][1][c] = sum(y*y);
 *    ttk_bn_fwd_finalize folds them (fp64) into the layer's constant block bn[TTK_BN_ROWS][C]
 *    (+ running statistics); the CONSUMER applies
 *        a = max(scale*(y - mean) + beta (+ skip), 0)
 *    while loading.  Backward mirrors it: the producer of a gradient writes g = dL/d(bn output) and
 *    partials sum(g), sum(g*(y - mean)); ttk_bn_bwd_finalize adds ga, gb, gmean to the block such that
 *        dL/dy = ga*(g - gmean) + gb*(y - mean),
 *    which the consumer applies while loading.  Both forms subtract first: the one-fma-per-tensor
 *    variants cancel catastrophically in fp32 for channels that are nearly constant over the batch.
 */
#ifndef TTK_H_
#define TTK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ttk_stream_t; /* hipStream_t */

#define TTK_ABI_VERSION 28

/* rows of a layer's BatchNorm constant block  float bn[TTK_BN_ROWS][C] */
enum {
  TTK_BN_SCALE = 0, /* gamma * rstd                      (forward)  */
  TTK_BN_BETA = 1,  /* copy of the bias parameter         (forward)  */
  TTK_BN_MEAN = 2,  /* batch mean (eval: running_mean)    (both)     */
  TTK_BN_RSTD = 3,  /* 1/sqrt(var + eps)                             */
  TTK_BN_GA = 4,    /* gamma * rstd                       (backward) */
  TTK_BN_GB = 5,    /* -ga * rstd^2 * mean(g*(y-mean))    (backward) */
  TTK_BN_GMEAN = 6, /* mean(g)                            (backward) */
  TTK_BN_AUX = 7,   /* magnitude bounds of the tensors this layer forms (TTK_AUX_*), for the fp16-split GEMMs */
  TTK_BN_ROWS = 8
};
/* Row TTK_BN_AUX.  The pointwise GEMMs on the fp16 matrix pipe (csrc/pwconv_f16.hip) scale each operand tensor by a
 * power of two taken from an upper bound of its magnitude; the bounds live here.  They are NON-NEGATIVE floats that
 * several workgroups raise with an integer atomicMax on the bit pattern, so the row must be ZERO before the forward
 * pass of a step (the shipped host allocates all blocks of a step from one zeroed arena).  0 = unknown: scale 1.
 *   ACT_BOUND >= max |max(scale*(y-mean)+beta, 0)|        written by ttk_bn_fwd_finalize (Cauchy-Schwarz on the batch
 *                                                         variance: |y-mean| <= sqrt(count*var)); eval: stays 0
 *   GMAX      =  max |g|  of the gradient w.r.t. this layer's output, raised by the kernel that produces g
 *                                                         (ttk_avgpool_bwd, ttk_dwconv3x3_bwd_data)
 *   DY_BOUND  >= max |ga*(g-gmean) + gb*(y-mean)|         written by ttk_bn_bwd_finalize from GMAX and the variance */
enum { TTK_AUX_ACT_BOUND = 0, TTK_AUX_DY_BOUND = 1, TTK_AUX_GMAX = 2 };
/* Storage of the activation-sized tensors of the MobileNet path.  Entry points with an `act_bf16` argument take them as
 * `void*`; the argument is a set of TTK_STORE_* bits:
 *   0                                      everything float32 - the reference's precision: what the depthwise (ttk_dwconv3x3_*),
 *                                          pointwise (ttk_pwconv1x1_*) and pooling (ttk_avgpool_*) entry points accept.  The storage-only
 *                                          bf16 variants of rounds 2-5 (`--precision bf16 | bf16-all`) are RETIRED: those entry points
 *                                          return -1 for either bit and name the bf16-compute path (ttk_bc_*, below)
 *   TTK_STORE_ACT_BF16|TTK_STORE_GRAD_BF16 activations AND gradients bfloat16: accepted by the stem pair only (ttk_stem_fwd,
 *                                          ttk_stem_bwd_weight), which also serves the bf16-compute path (C = 32: same bytes in either
 *                                          block layout).  Rounded to nearest even on store; BatchNorm statistics from the values as stored. */
#define TTK_STORE_ACT_BF16 1
#define TTK_STORE_GRAD_BF16 2
/* The two entry points both backbones share (ttk_avgpool_fwd / ttk_avgpool_bwd) take this bit in the same argument: their
 * activation tensors are channels-last rows [pixels][C] (the ResNet18 path) instead of channel blocks (see "Activation layout"). */
#define TTK_LAYOUT_ROWS 4
/* ... and this one: channel blocks of 64 (the bf16-compute path below; C >= 64) instead of 32. */
#define TTK_LAYOUT_CB64 8
#define TTK_MAX_PARTIAL_ROWS_ELEMENTWISE 1024
#define TTK_GEMM_BLOCK_M 128

int ttk_abi_version(void);
const char* ttk_last_error_string(void);
/* Entry points report launch failures through hipGetLastError(), which is per thread and sticky: call this first to drop an
 * error that some EARLIER HIP call of the process left pending (returns it).  The shipped host does so before every call. */
int ttk_clear_error(void);

/* Number of rows of the `part` buffer ([rows][2][C] floats) that a producer writes. */
int ttk_partial_rows_elementwise(int64_t work_items); /* stem / depthwise / pool kernels        */
int ttk_partial_rows_dwconv(int B, int H, int W, int C, int stride, int backward); /* depthwise fwd (0) / data-grad (1) */
int ttk_partial_rows_gemm(int64_t M);                 /* MFMA kernels with 128-row partial sums: ceil(M/128) (convolutions, fused backward inputs) */
/* rows of partial sums ttk_pwconv1x1_fwd (K = Cin, Nout = Cout, dgrad = 0) / ttk_pwconv1x1_bwd_data (K = Cout, Nout = Cin, dgrad = 1) write
 * for M rows: one per row block of the kernel that runs the shape - ceil(M/128), or the row-block tiling of csrc/pwconv_r.hip */
int ttk_partial_rows_pwconv(int64_t M, int K, int Nout, int dgrad);
/* rows of the MFMA tile that runs the shape: 128, 192 or 256 for the row-block kernels of csrc/pwconv_r.hip (chosen per (M, K, Nout) by their
 * cost model so that the tiles fill whole rounds of the CUs), 0 for every other kernel.  ABI 19; a query for tests and tools. */
int ttk_pwconv_tile_rows(int64_t M, int K, int Nout, int dgrad);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d statistics - replaces F.batch_norm(training=True, momentum, eps) as called through
 * nn.BatchNorm2d at backbones/mobilenet_v1.py:30,66,68,79,84,125,162.
 * running_var receives the UNBIASED batch variance, normalisation uses the biased one;
 * num_batches_tracked (int64 scalar on the device, may be NULL) is incremented.
 * ------------------------------------------------------------------------------------------- */
/* `part` is scratch: when part_rows > 1280 the finalize kernels first fold it IN PLACE to 1024 rows.
 * Writes rows SCALE, BETA, MEAN, RSTD of bn and raises bn[TTK_BN_AUX][TTK_AUX_ACT_BOUND].
 *
 * The statistics PIVOT.  Every forward producer below (ttk_stem_fwd, ttk_dwconv3x3_fwd, ttk_pwconv1x1_fwd, ttk_conv_fwd,
 * ttk_stem7_fwd) takes `pivot` (float[C] on the device, or NULL = zeros) and leaves in `part` the sums of (y - pivot) and
 * (y - pivot)^2; the finalisation given the SAME pivot forms mean = pivot + S1/n and var = S2/n - (S1/n)^2.  The partial rows are
 * fp32, so without a pivot a channel whose mean is k standard deviations from zero loses ~k^2 * 2^-24 of its variance to
 * cancellation (13 sigma: 1e-5 relative); with the layer's running_mean as the pivot - what the host passes; the finalisation
 * reads it before it writes the update, so pivot may alias running_mean - k is the distance between the batch mean and the
 * running mean instead. */
int ttk_bn_fwd_finalize(float* part, const float* pivot, int part_rows, int C, int64_t count,
                        const float* gamma, const float* beta,
                        float* running_mean, float* running_var, int64_t* num_batches_tracked,
                        float momentum, float eps, float* bn, ttk_stream_t stream);
/* eval mode: SCALE, BETA, MEAN, RSTD from the running statistics. */
int ttk_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, int C, float* bn, ttk_stream_t stream);
/* part[row][0][c] = sum(g), part[row][1][c] = sum(g*(y-mean)).  Writes rows GA, GB, GMEAN of bn, raises
 * bn[TTK_BN_AUX][TTK_AUX_DY_BOUND] (when the producer of g left its maximum in TTK_AUX_GMAX) and writes
 * the parameter gradients dgamma/dbeta (nullable; accumulate != 0: += instead of =). */
int ttk_bn_bwd_finalize(float* part, int part_rows, int C, int64_t count, const float* gamma,
                        float* bn, float* dgamma, float* dbeta, int accumulate, ttk_stream_t stream);

/* Backward constants of a FROZEN BatchNorm (eval-mode statistics while the convolutions train: NetworkWithPointHead
 * .prepare_finetune() + train(), models.py:378-394, modelcomponents.py:208-215): rows GA = SCALE, GB = 0, GMEAN = 0, and
 * TTK_AUX_DY_BOUND from TTK_AUX_GMAX.  bn holds ttk_bn_eval_prepare's rows; no partial sums are read, gamma / beta get no
 * gradient (the reference freezes them). */
int ttk_bn_bwd_frozen(float* bn, int C, ttk_stream_t stream);
/* Forward half: after ttk_bn_eval_prepare, raise bn[TTK_BN_AUX][TTK_AUX_ACT_BOUND] to a bound of relu(scale*(y-mean)+beta) over
 * this batch from the producer's partial sums (as ttk_bn_fwd_finalize does for batch statistics), so that the fp16 GEMMs of the
 * backward pass scale their operands; nothing else of bn is written. */
int ttk_bn_frozen_bound(float* part, const float* pivot, int part_rows, int C, int64_t count, float* bn, ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Stem: nn.Conv2d(1, 32, 5, stride 2, pad 2, bias=False)  - mobilenet_v1.py:122-124,161.
 * x[B][H][W] -> y[B][Ho][Wo][32], Ho = (H+1)/2.  w is the reference's weight (32,1,5,5) as is.
 * part may be NULL (eval).
 * ------------------------------------------------------------------------------------------- */
int ttk_stem_fwd(const float* x, const float* w, void* y, float* part, const float* pivot, int B, int H, int W,
                 int act_bf16, ttk_stream_t stream);
/* dW[32][25] (+)= sum dy * x, dy formed on load from (g, y, bn = the stem's BatchNorm block). */
/* partial (nullable): scratch of ttk_stem_wgrad_partial_bytes() - the workgroups store their partial sums there and a
 * second kernel folds them in a fixed order (bitwise reproducible) instead of fp32 atomics. */
size_t ttk_stem_wgrad_partial_bytes(void);
int ttk_stem_bwd_weight(const void* g, const void* y, const float* bn, const float* x, float* dw,
                        int accumulate, float* partial, int B, int H, int W, int act_bf16,
                        ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Depthwise 3x3, pad 1, stride 1|2, groups=C, bias=False - DepthWiseBlock.conv_dw,
 * mobilenet_v1.py:57-65,78.  The block input is formed on load from the previous layer's raw output:
 *   a_in = max(scale*(yprev - mean) + beta (+ skip_prev), 0)   with bn_prev = the producer's block
 *                                                 (bn + relu (+ residual), mobilenet_v1.py:79-90 / :162-163)
 * a_out (nullable): materialise a_in, needed when THIS block has a residual connection
 * (mobilenet_v1.py:70,86-88).  w is the reference's weight (C,1,3,3) as is.
 * ------------------------------------------------------------------------------------------- */
int ttk_dwconv3x3_fwd(const void* yprev, const float* bn_prev, const void* skip_prev, void* a_out,
                      const float* w, void* y, float* part, const float* pivot, int B, int H, int W, int C,
                      int stride, int act_bf16, ttk_stream_t stream);
/* Gradient w.r.t. the block input, masked by relu and handed to the producer's BatchNorm:
 *   G      = convT3x3(dy_dw) (+ skip_grad)        dy_dw formed on load from (g_dw, y_dw, bn_dw)
 *   g_prev = G * [a_in > 0]                        -> written, with partials sum(g_prev), sum(g_prev*(yprev-mean))
 * a_in (nullable): materialised block input; if NULL it is recomputed from yprev/bn_prev/skip_prev.
 * dw (nullable): FUSED weight gradient dW[C][9] (+)= sum dy_dw * a_in(taps) - every (dy, a_in) pair it
 * needs is already in registers here.
 * Raises bn_prev[TTK_BN_AUX][TTK_AUX_GMAX] to max |g_prev| (the bound the previous block's GEMMs scale by).
 * dw_partial (nullable): scratch of ttk_partial_rows_dwconv(.., 1) * 9 * C floats - the fused weight gradient is then
 * folded from per-workgroup rows in a fixed order (bitwise reproducible) instead of fp32 atomics.  * dw_accumulate == 2 (with dw_partial): the rows stay UNFOLDED in dw_partial and the caller folds them with ttk_bc_bn_bwd_finalize_fold, in the
 * launch that finalises the producer's BatchNorm backward (the product's default since round 5: one launch instead of float atomics). */
int ttk_dwconv3x3_bwd_data(const void* g_dw, const void* y_dw, const float* bn_dw, const float* w,
                           const void* skip_grad, const void* yprev, float* bn_prev,
                           const void* skip_prev, const void* a_in, void* g_prev, float* part,
                           float* dw, int dw_accumulate, float* dw_partial, int B, int H, int W, int C,
                           int stride, int act_bf16, ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Pointwise 1x1 conv = GEMM on the matrix cores - DepthWiseBlock.conv_sep, mobilenet_v1.py:67,82.
 *   y[M][Cout] = a_dw[M][Cin] . w[Cout][Cin]^T,   a_dw = max(bn_dw(ydw), 0) on load,  M = B*Ho*Wo
 * Compute-bound shapes run as split-operand products on the 16-bit matrix pipe with fp32-chain accuracy: two fp16
 * pieces per operand and three products (csrc/pwconv_f16.hip; the default), or TTK_GEMM=bf16x3: three bf16 pieces and
 * six products (csrc/pwconv_split.hip).  The HBM-bound early layers run on v_mfma_f32_32x32x2_f32 (TTK_GEMM=f32mfma:
 * every layer).  The fp16 form needs the operand bounds of row TTK_BN_AUX: bn_dw[AUX][ACT_BOUND] (forward, weight
 * gradient), bn_pw[AUX][DY_BOUND] (both gradients).
 * wsplit (forward and data gradient): scratch of ttk_pwconv_prepared_bytes(Cin, Cout) for the split weight operand,
 * or a block that ttk_pwconv_prepare_weights filled (then w / wt == NULL); NULL selects the fp32 MFMA kernels.
 * ------------------------------------------------------------------------------------------- */
int ttk_pwconv1x1_fwd(const void* ydw, const float* bn_dw, const float* w, void* y, float* part, const float* pivot,
                      int64_t M, int Cin, int Cout, void* wsplit, int act_bf16, ttk_stream_t stream);
/* g_dw[M][Cin] = (dy[M][Cout] . w[Cout][Cin]) * [bn_dw(ydw) > 0],  dy formed on load from (g, y, bn_pw);
 * wt = w transposed ([Cin][Cout], ttk_transpose).  partials: sum(g_dw), sum(g_dw*(ydw-mean)). */
int ttk_pwconv1x1_bwd_data(const void* g, const void* y, const float* bn_pw, const float* wt,
                           const void* ydw, const float* bn_dw, void* g_dw, float* part, int64_t M,
                           int Cin, int Cout, void* wsplit, int act_bf16, ttk_stream_t stream);
/* dw[Cout][Cin] += dy^T . a_dw.  dw must be zeroed (or hold the running gradient) before the call.
 * partial == NULL: the M dimension is split over workgroups that add atomically (fp32 atomics: the result depends on
 * the order the hardware commits them).  partial = scratch of ttk_pwconv_wgrad_partial_bytes(M, Cin, Cout) (0 = this
 * shape / GEMM mode has no such form): every slice of M stores its tile and a second kernel adds the slices to dw in a
 * fixed order - bitwise reproducible.  For Cin, Cout multiples of 256 the slice form is also the FASTER one (256 x 256 tiles with transposed
 * LDS reads, csrc/pwconv_r.hip): ttk_pwconv_wgrad_scratch_bytes says how much scratch the default mode wants to be handed as `partial`
 * for a shape (0: none, the atomic form runs). */
size_t ttk_pwconv_wgrad_partial_bytes(int64_t M, int Cin, int Cout);
size_t ttk_pwconv_wgrad_scratch_bytes(int64_t M, int Cin, int Cout);
int ttk_pwconv1x1_bwd_weight(const void* g, const void* y, const float* bn_pw, const void* ydw,
                             const float* bn_dw, float* dw, float* partial, int64_t M, int Cin, int Cout,
                             int act_bf16, ttk_stream_t stream);
/* Weight operands of n (<= 16) pointwise layers (w[i]: [Cout][Cin] fp32 device pointers; w, cin, cout and
 * prepared are HOST arrays): two launches for all layers (workgroup |w| maxima; planes).  prepared[i]: device scratch of
 * ttk_pwconv_prepared_bytes(cin[i], cout[i]); pass it as `wsplit` with w == NULL (forward) / wt == NULL (data gradient)
 * to skip the per-call split and transpose launches. */
/* Weight AND data gradient of the first pointwise layers (Cin -> Cout = 32 -> 64, 64 -> 128, 128 -> 128; fp32 storage) in one
 * kernel: g, y and ydw are read once instead of twice (these layers are HBM-bound).  32 -> 64 runs on the fp32 matrix pipe
 * from the raw weights w[Cout][Cin] (wsplit unused); 128 -> 128 on the fp16 pipe from the block that
 * ttk_pwconv_prepare_weights filled (wsplit; w unused), with the operand bounds of row TTK_BN_AUX; 64 -> 128 on the fp16
 * pipe too in the default TTK_GEMM mode (w AND wsplit: the raw weights, cut in the kernel with the block's |w| scale),
 * on fp32 MFMA otherwise.  dw accumulates
 * (fp32 atomics per workgroup, or - partial = scratch of ttk_pwconv1x1_bwd_fused_partial_bytes - workgroup rows folded in a
 * fixed order); g_dw and part as ttk_pwconv1x1_bwd_data, with ttk_pwconv1x1_bwd_fused_rows(M, Cin, Cout) partial rows
 * (0 = this shape has no fused form). */
int ttk_pwconv1x1_bwd_fused_rows(int64_t M, int Cin, int Cout);
size_t ttk_pwconv1x1_bwd_fused_partial_bytes(int64_t M, int Cin, int Cout);
int ttk_pwconv1x1_bwd_fused(const float* g, const float* y, const float* bn_pw, const float* w, const void* wsplit,
                            const float* ydw, const float* bn_dw, float* g_dw, float* dw, float* partial, float* part,
                            int64_t M, int Cin, int Cout, ttk_stream_t stream);
size_t ttk_pwconv_prepared_bytes(int Cin, int Cout);
int ttk_pwconv_prepare_weights(int n, const float* const* w, const int* cin, const int* cout,
                               void* const* prepared, ttk_stream_t stream);
int ttk_transpose(const float* in, float* out, int rows, int cols, ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * AdaptiveAvgPool2d(1) + view over the last block's output - mobilenet_v1.py:143,180-181.
 *   feat[B][C] = mean_hw max(bn(y) (+ skip), 0)
 * ------------------------------------------------------------------------------------------- */
int ttk_avgpool_fwd(const void* y, const float* bn, const void* skip, float* feat, int B, int HW,
                    int C, int act_bf16, ttk_stream_t stream);
/* g[B][HW][C] = gfeat[B][C]/HW * [bn(y)+skip > 0]; partials sum(g), sum(g*(y-mean)); raises
 * bn[TTK_BN_AUX][TTK_AUX_GMAX] to max |g|. */
int ttk_avgpool_bwd(const float* gfeat, const void* y, float* bn, const void* skip, void* g,
                    float* part, int B, int HW, int C, int act_bf16, ttk_stream_t stream);
/* a[rows][C] = max(bn(y) (+ skip), 0): materialises a post-activation tensor (the `intermediates` list
 * MobileNet.forward returns, mobilenet_v1.py:165-186).  y, skip: channel blocks; a: plain channels-last rows. */
int ttk_bn_act(const float* y, const float* bn, const float* skip, float* a, int64_t rows, int C,
               ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The bf16-COMPUTE path of the MobileNet backbone (`--precision bf16-compute`, BASELINE config 5's bf16 leg; csrc/bc_*.hip).  The
 * reference has no such mode (it is fp32 everywhere: scripts/train_poseestimator.py:442-454 sets no precision); the entry points
 * replace the same reference lines as their fp32 counterparts above (depthwise conv mobilenet_v1.py:57-66,78-80; pointwise conv
 * :67,82).  Differences from the fp32 entry points:
 *  - every activation-sized tensor AND its gradient is bfloat16 in CHANNEL BLOCKS OF 64:  [C / 64][M][64], element (m, c) at
 *    ((c >> 6) * M + m) * 64 + (c & 63)  (C = 32: plain [M][32]) - a pixel of a block is one 128-byte line.  ttk_stem_fwd /
 *    ttk_stem_bwd_weight (C = 32: the same bytes in either layout) serve this path with act_bf16 = TTK_STORE_ACT_BF16 |
 *    TTK_STORE_GRAD_BF16; the ttk_bn_* finalisations are shared unchanged; the pool has its own pair (ttk_bc_avgpool_*);
 *  - the pointwise products run as ONE bf16 MFMA product with fp32 accumulation: operands rounded to bf16 after the BatchNorm (+ ReLU)
 *    / BatchNorm-backward map on load, weights rounded once per step by ttk_bc_prepare_weights (`wprep`: two images per layer,
 *    ttk_bc_prepared_bytes); no operand bounds (row TTK_BN_AUX is not read);
 *  - the BatchNorm maps are applied in their one-fma forms (scale*y + shift; ga*g + gb*y + c0); BatchNorm statistics come from the
 *    values as stored; weight gradients (fp32) are reduced in a fixed order (bitwise reproducible in every mode);
 *  - part rows: ttk_bc_partial_rows_pw(M, K, Nout) for ttk_bc_pw_fwd (K = Cin, Nout = Cout) / ttk_bc_pw_bwd_data (K = Cout,
 *    Nout = Cin); ttk_bc_partial_rows_dw for the depthwise pair;
 *  - ttk_bc_pw_bwd_weight needs `scratch` of ttk_bc_pw_wgrad_scratch_bytes(M, Cin, Cout) bytes and ADDS to dw.
 * ------------------------------------------------------------------------------------------- */
size_t ttk_bc_prepared_bytes(int Cin, int Cout);
int ttk_bc_prepare_weights(int n, const float* const* w, const int* cin, const int* cout, void* const* prepared, ttk_stream_t stream);
int ttk_bc_partial_rows_pw(int64_t M, int K, int Nout);
int ttk_bc_partial_rows_dw(int B, int H, int W, int C, int stride, int backward);
int ttk_bc_pw_fwd(const void* ydw, const float* bn_dw, const void* wprep, void* y, float* part, const float* pivot, int64_t M, int Cin,
                  int Cout, ttk_stream_t stream);
int ttk_bc_pw_bwd_data(const void* g, const void* y, const float* bn_pw, const void* wprep, const void* ydw, const float* bn_dw,
                       void* g_dw, float* part, int64_t M, int Cin, int Cout, ttk_stream_t stream);
size_t ttk_bc_pw_wgrad_scratch_bytes(int64_t M, int Cin, int Cout);
int ttk_bc_pw_bwd_weight(const void* g, const void* y, const float* bn_pw, const void* ydw, const float* bn_dw, float* dw,
                         float* scratch, int64_t M, int Cin, int Cout, ttk_stream_t stream);
/* dw == NULL (ttk_bc_pw_bwd_weight, ttk_bc_pw_bwd_fused): the slice tiles stay in `scratch` as [slices][Cout * Cin] (slices =
 * ttk_bc_pw_wgrad_slices / ttk_bc_pw_bwd_fused_rows) and the caller adds them to dw with ttk_bc_bn_bwd_finalize_fold, in the launch that
 * finalises the BatchNorm backward of the layer's input (same fixed order, same results). */
int ttk_bc_pw_wgrad_slices(int64_t M, int Cin, int Cout);
/* ttk_bc_pw_bwd_weight AND ttk_bc_pw_bwd_data of the early layers (32 -> 64, 64 -> 128, 128 -> 128: the largest pixel counts, HBM-bound)
 * in one kernel that reads g, y and ydw once: same results as the pair (the weight gradient is reduced over a different slicing of the
 * pixels).  ttk_bc_pw_bwd_fused_rows = the rows of `part` it writes, 0 = the shape has no fused form (use the pair);
 * `scratch` of ttk_bc_pw_bwd_fused_scratch_bytes bytes; ADDS to dw. */
int ttk_bc_pw_bwd_fused_rows(int64_t M, int Cin, int Cout);
size_t ttk_bc_pw_bwd_fused_scratch_bytes(int64_t M, int Cin, int Cout);
int ttk_bc_pw_bwd_fused(const void* g, const void* y, const float* bn_pw, const void* wprep, const void* ydw, const float* bn_dw,
                        void* g_dw, float* dw, float* scratch, float* part, int64_t M, int Cin, int Cout, ttk_stream_t stream);
/* AdaptiveAvgPool2d(1) over the last block's output and its backward (as ttk_avgpool_fwd / _bwd) on the 64-channel-block bf16 tensors;
 * the backward writes ttk_bc_partial_rows_pool(B, HW, C) partial rows. */
int ttk_bc_partial_rows_pool(int B, int HW, int C);
int ttk_bc_avgpool_fwd(const void* y, const float* bn, const void* skip, float* feat, int B, int HW, int C, ttk_stream_t stream);
int ttk_bc_avgpool_bwd(const float* gfeat, const void* y, const float* bn, const void* skip, void* g, float* part, int B, int HW, int C,
                       ttk_stream_t stream);
int ttk_bc_dw_fwd(const void* yprev, const float* bn_prev, const void* skip_prev, void* a_out, const float* w, void* y, float* part,
                  const float* pivot, int B, int H, int W, int C, int stride, ttk_stream_t stream);
int ttk_bc_dw_bwd_data(const void* g_dw, const void* y_dw, const float* bn_dw, const float* w, const void* skip_grad,
                       const void* yprev, const float* bn_prev, const void* skip_prev, const void* a_in, void* g_prev, float* part,
                       float* dw, int dw_accumulate, float* dw_partial, int B, int H, int W, int C, int stride, ttk_stream_t stream);
/* dw_accumulate == 2 (with dw_partial): the kernel leaves its ttk_bc_partial_rows_dw(..., 1) rows of [C][9] weight-gradient sums in dw_partial
 * UNFOLDED; the caller folds them in the same launch as the BatchNorm-backward finalisation that follows the kernel anyway:
 * ttk_bn_bwd_finalize(part, ...) + "fold_out (+)= sum over fold_rows rows of fold_partial[row][fold_n]" as ONE launch (two dependent
 * few-microsecond launches less per depthwise layer; same arithmetic, same results as the two separate calls). */
int ttk_bc_bn_bwd_finalize_fold(float* part, int part_rows, int C, int64_t count, const float* gamma, float* bn, float* dgamma, float* dbeta,
                                int accumulate, const float* fold_partial, int fold_rows, int64_t fold_n, float* fold_out, int fold_accumulate,
                                ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Dense convolutions of the ResNet18 backbone variant (backbones/resnet.py:52-104; arithmetic =
 * torchvision.models.resnet.BasicBlock/conv3x3/conv1x1, an un-vendored dependency of the reference) as implicit
 * GEMMs on the matrix cores.  Channels-last activations a[B][H][W][C] that are already post-BatchNorm/ReLU
 * ("materialised"); k in {1,3}, stride in {1,2}, pad = k/2; Cin % 32 == 0, Cout % 64 == 0.
 *   ttk_conv_weight_repack  w[Cout][Cin][KH][KW] -> w_fwd[.][KH*KW*Cin/32][Cout][32], w_bwd[.][KH*KW*Cout/32][Cin][32] (either may be
 *                           NULL; each a buffer of 3 * 2 bytes per weight): the 16-bit piece planes the GEMM producers
 *                           move without arithmetic - two fp16 planes of w * 2^s followed by a float header holding
 *                           max |w| (default), or TTK_GEMM=bf16x3: the three planes (h, m, l) of the exact bf16 split
 *   ttk_conv_fwd            y[B][Ho][Wo][Cout] raw conv output + part[ttk_partial_rows_gemm(B*Ho*Wo)][2][Cout].
 *                           a_bound: device float >= max |a_in| (the fp16 form scales by it) - the TTK_AUX_ACT_BOUND slot
 *                           of the BatchNorm block that formed a_in (ttk_bn_fwd_finalize bounds relu(bn(y)) and hence
 *                           its max-pool; ttk_bn_add_act raises the slot to max a)
 *   ttk_conv_bwd_data       g_in[B][H][W][Cin] = conv^T(dy), dy = ga*(g-gmean)+gb*(y-mean) formed on load from the conv
 *                           output's gradient g, raw output y and BatchNorm block bn.  With mask_y/mask_bn (the conv
 *                           input was relu(mask_bn(mask_y))): masked, and part[ttk_partial_rows_gemm(B*H*W)][2][Cin]
 *                           receives (sum g_in, sum g_in*(mask_y-mean)) and mask_bn[TTK_BN_AUX][TTK_AUX_GMAX] is raised
 *                           to max |g_in|; without: raw gradient, part untouched.  bn[TTK_BN_AUX][TTK_AUX_DY_BOUND]
 *                           must bound |dy| (ttk_bn_bwd_finalize, from the TTK_AUX_GMAX the producer of g raised).
 *   ttk_conv_bwd_weight     dw[Cout][Cin][KH][KW] += sum_pixels dy (x) a_in; a_bound as above.  partial = scratch of
 *                           ttk_conv_wgrad_partial_bytes(...) bytes (0 = the call takes none): the slices of the pixel range
 *                           store their tiles side by side and a second kernel folds them in a fixed order into dw
 *                           (bitwise reproducible, and several times faster than the alternative, partial == NULL: one
 *                           fp32 atomicAdd per slice and element at torch's 36-byte tap stride)
 *                           Both gradients accept y == NULL: g then holds dy as ttk_bn_bwd_apply wrote it - two fp16
 *                           planes [rows][Cout] (h, then l) of dy * 2^s (fp16 kernels only): half the operand bytes
 *                           through the L1, which is what bounds these kernels, and producers that only move data.
 * ------------------------------------------------------------------------------------------- */
int ttk_conv_weight_repack(const float* w, void* w_fwd, void* w_bwd, int Cout, int Cin, int KH, int KW,
                           ttk_stream_t stream);
/* the same for n <= 24 weight tensors (square kernels of size ksize[i] in {1,3}) in two launches: a training step's 19 */
int ttk_conv_prepare_weights(int n, const float* const* w, void* const* w_fwd, void* const* w_bwd, const int* cout,
                             const int* cin, const int* ksize, ttk_stream_t stream);
int ttk_conv_fwd(const float* a_in, const float* a_bound, const void* w_fwd, float* y, float* part, const float* pivot, int B, int H,
                 int W, int Cin, int Cout, int KH, int KW, int stride, int pad, ttk_stream_t stream);
int ttk_conv_bwd_data(const float* g, const float* y, const float* bn, const void* w_bwd, const float* mask_y,
                      float* mask_bn, float* g_in, float* part, int B, int H, int W, int Cin, int Cout,
                      int KH, int KW, int stride, int pad, ttk_stream_t stream);
size_t ttk_conv_wgrad_partial_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int ttk_conv_bwd_weight(const float* g, const float* y, const float* bn, const float* a_in, const float* a_bound, float* dw,
                        float* partial, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * The non-GEMM kernels of the ResNet18 variant (backbones/resnet.py:52-104: torchvision ResNet18, 1-channel 7x7 stem).
 *   ttk_stem7_fwd          y[B][65][65][64] = conv7x7/s2/p3(x[B][1][H][W]) + part[ttk_partial_rows_elementwise(B*Ho*Wo*16)][2][64]
 *   ttk_stem7_bwd_weight   dw[64][1][7][7] (overwritten) from dy = ga*(g-gmean)+gb*(y-mean); partial (nullable) = scratch of
 *                          ttk_stem7_wgrad_partial_bytes: workgroup partials folded in a fixed order instead of atomics
 *   ttk_maxpool3x3s2_fwd   a[B][Ho][Wo][C] = maxpool3x3/s2/p1(relu(bn(y))), idx = window position of the first maximum;
 *                          raises bn[TTK_BN_AUX][TTK_AUX_ACT_BOUND] to max a
 *   ttk_maxpool3x3s2_bwd   g[B][H][W][C] = gradient w.r.t. bn(y) (ReLU mask applied) from ga (+ gb) w.r.t. the pooled
 *                          activation; part[ttk_partial_rows_elementwise(B*H*W*C/4)][2][C] = (sum g, sum g*(y-mean))
 *   ttk_bn_add_act         a = relu(bn(y) + r); r = res, or res_bn(res) when res is the raw downsample-conv output
 *                          (BasicBlock: out = relu(bn2(conv2) + identity)), or nothing.  bn[TTK_BN_AUX][TTK_AUX_ACT_BOUND]
 *                          becomes the a_bound of the convolutions that read a: measure == 0 (training) - the bound of
 *                          relu(bn(y)) that ttk_bn_fwd_finalize left there, plus *res_bound (bound of r: the shortcut
 *                          activation's a_bound, or res_bn's TTK_AUX_ACT_BOUND slot; NULL without r); measure != 0
 *                          (eval: no batch statistics) - raised to the measured max a.
 *   ttk_bn_bwd_apply       dy = ga*(g-gmean) + gb*(y-mean), materialised once for a convolution's two gradients in the form
 *                          the fp16-split GEMMs consume: dy[2][rows][C] fp16 = h = fp16(dy*2^s), l = fp16(dy*2^s - h) with
 *                          2^s * bn[TTK_BN_AUX][TTK_AUX_DY_BOUND] in [2^14, 2^15) (4 bytes per element, like fp32)
 *   ttk_residual_bwd       gs = (ga (+ gb)) * [a > 0]; part = sums for bn(y); partd (with yd, bnd) = sums for the
 *                          downsample BatchNorm; rows of both = ttk_partial_rows_elementwise(rows*C/4).  Raises
 *                          TTK_AUX_GMAX of bn (and bnd) to max |gs|.
 * ------------------------------------------------------------------------------------------- */
int ttk_stem7_fwd(const float* x, const float* w, float* y, float* part, const float* pivot, int B, int H, int W, ttk_stream_t stream);
size_t ttk_stem7_wgrad_partial_bytes(int B, int H, int W);
int ttk_stem7_bwd_weight(const float* g, const float* y, const float* bn, const float* x, float* dw, float* partial, int B,
                         int H, int W, ttk_stream_t stream);
int ttk_maxpool3x3s2_fwd(const float* y, float* bn, float* a, unsigned char* idx, int B, int H, int W, int C,
                         ttk_stream_t stream);
int ttk_maxpool3x3s2_bwd(const float* ga, const float* gb, const unsigned char* idx, const float* y, const float* bn,
                         float* g, float* part, int B, int H, int W, int C, ttk_stream_t stream);
int ttk_bn_add_act(const float* y, float* bn, const float* res, const float* res_bn, float* a, const float* res_bound,
                   int measure, int64_t rows, int C, ttk_stream_t stream);
int ttk_bn_bwd_apply(const float* g, const float* y, const float* bn, void* dy, int64_t rows, int C, ttk_stream_t stream);
/* BlurPool2D of the ResNet18 variant's use_blurpool (backbones/resnet.py:31-49,63-66; neuralnets/modelcomponents.py:187-205) on
 * channels-last rows: t[B][Ho][Wo][C] = depthwise 3x3 of a[B][H][W][C] with the binomial kernel [1 2 1]^T [1 2 1] / 16, zero padding 1,
 * stride 1 | 2 (Ho = (H-1)/stride + 1); C % 4 == 0.  ttk_blur3x3_bwd: g[B][H][W][C] = the transposed map of (ga + gb) (gb nullable) -
 * the gradient w.r.t. a.  max |t| <= max |a|: the a_bound of a is a valid bound of t. */
int ttk_blur3x3_fwd(const float* a, float* t, int B, int H, int W, int C, int stride, ttk_stream_t stream);
int ttk_blur3x3_bwd(const float* ga, const float* gb, float* g, int B, int H, int W, int C, int stride, ttk_stream_t stream);
int ttk_residual_bwd(const float* ga, const float* gb, const float* a, const float* y, float* bn, const float* yd,
                     float* bnd, float* gs, float* part, float* partd, int64_t rows, int C, ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Multi-task heads - everything NetworkWithPointHead.forward does after the backbone
 * (neuralnets/models.py:345-376): BoundingBox :177-197, PositionSizeOutput :200-215,
 * DirectQuaternionWithNormalization :127-150 (+ rotrepr.py:36-48), FeaturesAsTriangularScale
 * (negloglikelihood.py:187-242), two LocalToGlobalCoordinateOffset modules
 * (modelcomponents.py:136-184) and Landmarks3dOutput (models.py:96-124, modelcomponents.py:38-82).
 * The linear layers are passed stacked: wcat[NZ][F], bcat[NZ] with the row order
 *   box 4 | xy 2 | size 1 | quat 4 | [coord-scale neck 7 | pose-scale neck 7] | [shape 50]
 * NZ = ttk_heads_num_rows(enable_uncertainty, enable_point_head, enable_6drot).  ids: int32 dataset ids (NULL = row 0,
 * the reference's set_id=None path).  Outputs per sample: roi[4] coord[3] rot[4] (ijkw) qu[4]
 * (= unnormalized_quat) Lc[9] Lr[9] (= coord_scales, pose_scales_tril) pts[68][3] shp[50]; z[B][NZ] is
 * saved for backward.
 * enable_6drot (RotRepr6dWithNormalization, models.py:153-174; torch6drotation.py:27-49; Mat33Repr, rotrepr.py:63-98):
 * the rotation rows are 6 instead of 4, rot is [B][9] (row-major 3x3: Gram-Schmidt of the two 3-vectors, identity
 * where max|R R^T - I| > 1e-3, times the dataset offset rotation), qu is [B][6] (= unnormalized_6drepr).
 * ------------------------------------------------------------------------------------------- */
int ttk_heads_num_rows(int enable_uncertainty, int enable_point_head, int enable_6drot);
int ttk_heads_fwd(const float* feat, const float* wcat, const float* bcat, const int* ids,
                  const float* P, const float* Pk, const float* keypts, const float* keyeig, int B, int F,
                  int NZ, int enable_uncertainty, int enable_point_head, int use_offset, int enable_6drot,
                  float* z, float* roi, float* coord, float* rot, float* qu, float* Lc, float* Lr, float* pts,
                  float* shp, ttk_stream_t stream);
/* dz[B][NZ], dprow[B][8]: scratch.  Outputs dfeat[B][F], dwcat[NZ][F], dbcat[NZ], dP[8][4], dPk[8][4]
 * (overwritten). */
int ttk_heads_bwd(const float* feat, const float* wcat, const float* z, const int* ids, const float* P,
                  const float* Pk, const float* keypts, const float* keyeig, int B, int F, int NZ,
                  int enable_uncertainty, int enable_point_head, int use_offset, int enable_6drot,
                  const float* g_roi, const float* g_coord, const float* g_rot, const float* g_qu,
                  const float* g_Lc, const float* g_Lr, const float* g_pts, const float* g_shp, float* dz,
                  float* dprow, float* dfeat, float* dwcat, float* dbcat, float* dP, float* dPk,
                  ttk_stream_t stream);
/* DiagonalScaleParameter (negloglikelihood.py:50-65): out[n] from hidden[n+1]. */
int ttk_diag_scale_fwd(const float* hidden, float* out, int n, ttk_stream_t stream);
int ttk_diag_scale_bwd(const float* hidden, const float* gout, float* ghidden, int n, ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Losses: per-sample values v[n] and gradients w.r.t. the predictions (gv[n] = upstream gradient).
 *   rot       1-(q.t)^2                     losses.py:42-50 / torchquaternion.py:225-228
 *   quatreg   (1-|qu|)^2                    losses.py:116-125
 *   mse_rows  mean_d (p-t)^2                losses.py:67-97,163-173 (xy, size, box, shape params)
 *   points    mean_p w_p sum_{d<dim}(p-t)^2 losses.py:128-160 (w: chin/eye weights, keypoints68.py)
 *   nllrot    tangent-space MVN + uniform   negloglikelihood.py:245-274
 *   nllcoord  correlated MVN + uniform      negloglikelihood.py:113-126
 *   normal    -mean Normal.log_prob         negloglikelihood.py:129-177 (points != 0: [n][68][3] weighted)
 *   gmm       shape plausibility, float64   losses.py:100-113, modelcomponents.py:278-290
 *             ck[k] = log w_k + sum_d log sinv_kd - 25 log 2pi; post[n][K] scratch (responsibilities)
 * ------------------------------------------------------------------------------------------- */
int ttk_loss_rot_fwd(const float* q, const float* t, int n, float* v, ttk_stream_t stream);
int ttk_loss_rot_bwd(const float* q, const float* t, const float* gv, int n, float* gq, ttk_stream_t stream);
/* 6D-rotation variants (--enable-6drot): Rot6dReprLoss 0.75 - 0.25 tr(R T^T), T = tomatrix(target quaternion)
 * (losses.py:53-58, torch6drotation.py:68-72); Rot6dNormalizationSoftConstraint mean((M M^T - I_2)^2) on the raw 6D
 * features (losses.py:61-64, torch6drotation.py:20-24); Mat33Repr.as_quat = torchquaternion.from_matrix (:94-168:
 * best-conditioned of four candidates, positivereal) - used by QuatPoseNLLLoss and the eval-mode `pose` output. */
int ttk_loss_rot6d_fwd(const float* R, const float* t, int n, float* v, ttk_stream_t stream);
int ttk_loss_rot6d_bwd(const float* t, const float* gv, int n, float* gR, ttk_stream_t stream);
int ttk_loss_ortho6d_fwd(const float* z6, int n, float* v, ttk_stream_t stream);
int ttk_loss_ortho6d_bwd(const float* z6, const float* gv, int n, float* gz6, ttk_stream_t stream);
int ttk_mat_to_quat_fwd(const float* m, int n, float* q, ttk_stream_t stream);
int ttk_mat_to_quat_bwd(const float* m, const float* gq, int n, float* gm, ttk_stream_t stream);
int ttk_loss_quatreg_fwd(const float* q, int n, float* v, ttk_stream_t stream);
int ttk_loss_quatreg_bwd(const float* q, const float* gv, int n, float* gq, ttk_stream_t stream);
int ttk_loss_mse_rows_fwd(const float* p, const float* t, int n, int D, float* v, ttk_stream_t stream);
int ttk_loss_mse_rows_bwd(const float* p, const float* t, const float* gv, int n, int D, float* gp, ttk_stream_t stream);
/* The same over the column window [c0, c0+Dc) of rows Dt floats apart (PoseXYLoss / PoseSizeLoss on coord[..., :2] and
 * coord[..., 2], reference neuralnets/losses.py:66-85, without materialising the slices); gp is the gradient of the whole
 * [n][Dt] tensor, zero outside the window. */
int ttk_loss_mse_cols_fwd(const float* p, const float* t, int n, int Dt, int c0, int Dc, float* v, ttk_stream_t stream);
int ttk_loss_mse_cols_bwd(const float* p, const float* t, const float* gv, int n, int Dt, int c0, int Dc, float* gp, ttk_stream_t stream);
/* Loss bookkeeping of one step (reference train.py:372-439, default_compute_loss) in single launches; src, dst, count,
 * val, sample_w, w, gval are HOST arrays of n <= 32 entries.
 *   multi_copy      : dst[k][0..count[k]) = src[k] ? src[k][..] : 0      (concatenate sub-batch values / slice gradients)
 *   weighted_sum_fwd: out = scale * sum_k w[k] * sum_i sample_w[k][i] * val[k][i]  (sample_w or sample_w[k] NULL = 1),
 *                     one workgroup, fixed summation order, double accumulators
 *   weighted_sum_bwd: gval[k][i] = gout[0] * scale * w[k] * sample_w[k][i] */
int ttk_multi_copy(int n, const float* const* src, float* const* dst, const int64_t* count, ttk_stream_t stream);
int ttk_weighted_sum_fwd(int n, const float* const* val, const float* const* sample_w, const float* w, const int* count,
                         float scale, float* out, ttk_stream_t stream);
int ttk_weighted_sum_bwd(int n, const float* gout, const float* const* sample_w, const float* w, const int* count,
                         float scale, float* const* gval, ttk_stream_t stream);
int ttk_loss_points_fwd(const float* p, const float* t, int n, int dim, float chin, float eye, float* v, ttk_stream_t stream);
int ttk_loss_points_bwd(const float* p, const float* t, const float* gv, int n, int dim, float chin, float eye, float* gp, ttk_stream_t stream);
int ttk_loss_nllrot_fwd(const float* q, const float* t, const float* L, int n, float* v, ttk_stream_t stream);
int ttk_loss_nllrot_bwd(const float* q, const float* t, const float* L, const float* gv, int n, float* gq, float* gL, ttk_stream_t stream);
int ttk_loss_nllcoord_fwd(const float* c, const float* t, const float* L, int n, float* v, ttk_stream_t stream);
int ttk_loss_nllcoord_bwd(const float* c, const float* t, const float* L, const float* gv, int n, float* gc, float* gL, ttk_stream_t stream);
int ttk_loss_normal_fwd(const float* mu, const float* sigma, const float* x, int n, int per, int points, int dim, float chin, float eye, float* v, ttk_stream_t stream);
int ttk_loss_normal_bwd(const float* mu, const float* sigma, const float* x, const float* gv, int n, int per, int points, int dim, float chin, float eye, float* gmu, float* gsigma, ttk_stream_t stream);
/* The non-default kinds of the reference's loss switches (single launches, not part of ttk_loss_batch):
 *   laplace       the `normal` entry points with Laplace(mu, b) in place of Normal(mu, sigma) - DISTRIBUTION_CLASS_MAP["laplace"],
 *                 negloglikelihood.py:68-69 (CoordPoseNLLLoss, BoxNLLLoss, Points3dNLLLoss, ShapeParamsNLLLoss)
 *   elem          v[s] = sum_d colw[d] * f(p[s][d] - t[s][d]) over rows of D floats, f by kind: TTK_ELEM_L2 e^2, TTK_ELEM_L1 |e|,
 *                 TTK_ELEM_SMOOTH_L1 (|e| < beta ? e^2 / (2 beta) : |e| - beta / 2) - LOSS_OBJECT_MAP, losses.py:16-21 (the reference's
 *                 smooth_l1 has beta = 0.01).  colw[D] (device) carries each loss class's reduction: 1/Dc inside a column window
 *                 (PoseXYLoss, PoseSizeLoss, BoxLoss), point weight / 68 on the first `dim` coordinates of a landmark (Points3dLoss).
 *   rot_geodesic  smooth_l1(|rotation_delta(q, t)|, beta = 1 degree) / pi - smooth_geodesic_distance, losses.py:24-32 */
enum { TTK_ELEM_L2 = 0, TTK_ELEM_L1 = 1, TTK_ELEM_SMOOTH_L1 = 2 };
int ttk_loss_laplace_fwd(const float* mu, const float* b, const float* x, int n, int per, int points, int dim, float chin, float eye, float* v, ttk_stream_t stream);
int ttk_loss_laplace_bwd(const float* mu, const float* b, const float* x, const float* gv, int n, int per, int points, int dim, float chin, float eye, float* gmu, float* gb, ttk_stream_t stream);
int ttk_loss_elem_fwd(const float* p, const float* t, const float* colw, int n, int D, int kind, float beta, float* v, ttk_stream_t stream);
int ttk_loss_elem_bwd(const float* p, const float* t, const float* colw, const float* gv, int n, int D, int kind, float beta, float* gp, ttk_stream_t stream);
int ttk_loss_rot_geodesic_fwd(const float* q, const float* t, int n, float* v, ttk_stream_t stream);
int ttk_loss_rot_geodesic_bwd(const float* q, const float* t, const float* gv, int n, float* gq, ttk_stream_t stream);
int ttk_loss_gmm_fwd(const float* x, const double* ck, const double* mu, const double* sinv, int K, double fudge, int n, float* v, double* post, ttk_stream_t stream);
int ttk_loss_gmm_bwd(const float* x, const double* mu, const double* sinv, const double* post, int K, double fudge, const float* gv, int n, float* gx, ttk_stream_t stream);

/* Up to TTK_LOSS_BATCH_MAX of the loss ops above in ONE launch.  The ~30 loss kernels of a training step move a few KB each and
 * do not depend on one another, but on one stream each costs ~4.5 us of launch-to-completion latency.  An op carries the
 * arguments of its entry point in signature order - pointers in p[], ints in i[], floats in f[], the double in d - and
 * `items`, the threads it needs (n for the per-sample ops rot / rot6d / ortho6d / quatreg / nllrot / nllcoord; 64 n for the
 * wave-per-sample reductions mse_rows_fwd, mse_cols_fwd, points_fwd, normal_fwd, gmm_fwd; one per output element for the other
 * backward ops).  The ops of one call must be independent (none reads what another writes).  `ops` is a HOST array. */
#define TTK_LOSS_BATCH_MAX 32
enum {
  TTK_OP_ROT_FWD, TTK_OP_ROT_BWD, TTK_OP_ROT6D_FWD, TTK_OP_ROT6D_BWD, TTK_OP_ORTHO6D_FWD, TTK_OP_ORTHO6D_BWD,
  TTK_OP_QUATREG_FWD, TTK_OP_QUATREG_BWD, TTK_OP_MSE_ROWS_FWD, TTK_OP_MSE_ROWS_BWD, TTK_OP_MSE_COLS_FWD, TTK_OP_MSE_COLS_BWD,
  TTK_OP_POINTS_FWD, TTK_OP_POINTS_BWD, TTK_OP_NLLROT_FWD, TTK_OP_NLLROT_BWD, TTK_OP_NLLCOORD_FWD, TTK_OP_NLLCOORD_BWD,
  TTK_OP_NORMAL_FWD, TTK_OP_NORMAL_BWD, TTK_OP_GMM_FWD, TTK_OP_GMM_BWD, TTK_OP_COUNT
};
typedef struct ttk_loss_op {
  int kind;  /* TTK_OP_* */
  int items;
  const void* p[6];
  int i[4];
  float f[2];
  double d;
} ttk_loss_op;
int ttk_loss_batch(int nops, const ttk_loss_op* ops, ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * On-GPU intensity augmentation - the kornia chain of trackertraincode/pipelines.py:508-532
 * (container: datatransformation/batch/intensity.py:30-41) in one pass: one workgroup per image, the image stays in
 * LDS from its single read to its single write.  x, y: [B][H][W] grey levels in [0,1] (y may alias x);
 * params[B][TTK_INTENSITY_PARAMS]: which operations fire for each sample and their sampled magnitudes (0 / negative =
 * off), applied in the reference's order equalize -> posterize -> gamma -> contrast -> brightness -> 5x5 Gaussian
 * blur (sigma 1.5, reflect border) -> + noise_std * noise -> clip to [0,1] -> + out_shift (whitening, -0.5).
 * noise: standard normal draws [B][H][W] or NULL.  2*H*W*4 + 2 KB of LDS per image (129x129: 132 KB).
 * kornia is not available to this build: parity unpinned, checker oracle/intensity.py.
 * ------------------------------------------------------------------------------------------- */
#define TTK_INTENSITY_EQUALIZE 0        /* > 0: histogram equalisation */
#define TTK_INTENSITY_POSTERIZE_BITS 1  /* 1..7: bits kept; 0 or 8: off */
#define TTK_INTENSITY_GAMMA 2           /* > 0: v^gamma */
#define TTK_INTENSITY_CONTRAST 3        /* > 0: v * factor */
#define TTK_INTENSITY_BRIGHTNESS 4      /* > 0: v + (factor - 1) */
#define TTK_INTENSITY_BLUR 5            /* > 0: blur */
#define TTK_INTENSITY_NOISE_STD 6       /* > 0: additive Gaussian noise */
#define TTK_INTENSITY_PARAMS 8
int ttk_intensity_augment(const float* x, float* y, const float* params, const float* noise, int B, int H, int W,
                          float out_shift, ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser step: torch.nn.utils.clip_grad_norm_(params, max_norm) over ALL tensors followed by
 * torch.optim.Adam (scripts/train_poseestimator.py:147-167 create_optimizer, :442-445
 * gradient_clip_val=1.0) in two launches, no host sync.  Device tables: ptrs[ntensors][4] =
 * {param, grad (0 = no gradient), exp_avg, exp_avg_sq} addresses; numel[ntensors]; group[ntensors]
 * (index into the host arrays lr4/wd4 of TTK_ADAM_MAX_GROUPS floats each); chunk_tensor/chunk_offset[nchunks] cut the tensors into
 * chunks of chunk_size elements.  steps[ntensors] (DEVICE floats): torch.optim.Adam's per-parameter `step`, kept on
 * the device - the call adds 1 for every tensor that has a gradient and derives the bias corrections
 * 1 - beta^step from it (fp64), so nothing about the step count is a launch argument and the call can sit inside a
 * captured hipGraph.  grad_scale: the gradients in memory are read as grad_scale * g everywhere (norm, clip, update) -
 * 1/world for data-parallel replicas whose all-reduce left SUMS in place.  partial[nchunks]: scratch; out_norm
 * (nullable): total gradient norm (of the scaled gradients).  hyper_dev (nullable, DEVICE
 * float[TTK_ADAM_HYPER_FLOATS]): when given, the per-group learning rates and weight decays are read from it instead
 * of lr4/wd4 (a captured graph then follows a scheduler without re-capture).
 * ------------------------------------------------------------------------------------------- */
#define TTK_ADAM_MAX_GROUPS 128 /* parameter groups (NetworkWithPointHead.prepare_finetune: one per backbone sub-module, 66) */
#define TTK_ADAM_HYPER_LR 0     /* [TTK_ADAM_MAX_GROUPS] */
#define TTK_ADAM_HYPER_WD 128   /* [TTK_ADAM_MAX_GROUPS] */
#define TTK_ADAM_HYPER_FLOATS 256
int ttk_clip_adam(const int64_t* ptrs, const int32_t* numel, const int32_t* group,
                  const int32_t* chunk_tensor, const int32_t* chunk_offset, int nchunks, int chunk_size,
                  const float* lr4, const float* wd4, float beta1, float beta2, float eps, float max_norm,
                  float grad_scale, float* steps, float* partial, float* out_norm, const float* hyper_dev,
                  ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * On-GPU affine-warp augmentation (the reference does this per sample on the CPU with OpenCV inside
 * DataLoader workers: datatransformation/batch/geometric.py:193-231; semantic oracle for the image =
 * tensors/image_geometric_torch.py:60-98, for the labels = tensors/affinetrafo.py:37-148).
 *   ttk_view_roi       GeneralFocusRoi._compute_view_roi (:108-157) + torch.round().to(int32) (:205);
 *                      view_roi[B][4] int32 is BIT-EXACT (explicitly rounded fp32 ops, round-half-even)
 *   ttk_roi_transform  tr[B][2][3] = center_rotation(angle) @ range_remap(view_roi -> [0,N]^2) (:159-177)
 *   ttk_affine_warp    out[B][1][N][N] = bilinear(src[B][1][Hs][Ws], tr^-1(pixel centre)) * mul + add,
 *                      zero padding, align_corners=False; src uint8 (src_is_u8) or float32
 *   ttk_affine_labels  in place: coord[B][3], pose[B][4] (ijkw), roi[B][4] (nullable each), pts_in ->
 *                      pts_out [B][68][3] (68-point flip map when det < 0); with N > 0 followed by the
 *                      pixel -> [-1,1] normalisation of normalize_batch (batch/normalization.py:20-56)
 * ------------------------------------------------------------------------------------------- */
int ttk_view_roi(const float* face_roi, const float* scales, const float* translations,
                 float beyond_border_shift, int B, int* view_roi, ttk_stream_t stream);
int ttk_roi_transform(const int* view_roi, const float* angles, int B, int N, float* tr, ttk_stream_t stream);
int ttk_affine_warp(const void* src, int src_is_u8, int B, int Hs, int Ws, const float* tr, float* out,
                    int N, float mul, float add, ttk_stream_t stream);
int ttk_affine_labels(const float* tr, int B, int N, float* coord, float* pose, float* roi,
                      const float* pts_in, float* pts_out, ttk_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Streaming probe (measurement infrastructure: bench.py `copy_probe`, tools/stream_sweep.py; no reference counterpart).
 * Streams `nread` (1, 2 or 5) source tensors of `rows` x `row_bytes`, `stream_bytes` apart, in the access shape of the
 * step's HBM-bound kernels - every workgroup reads `seg_bytes`-wide column slices of consecutive rows, 16 bytes per lane,
 * `unroll` (1, 2, 4, 8) x nread loads per lane in flight (seg_bytes == row_bytes: a linear sweep; 128 / 256: the
 * 32- / 64-channel slab of a channels-last tensor) - adds them and writes the sum to `nwrite` destination tensors of the
 * same shape (0: read only; *sink receives a dummy value that keeps the loads alive).  nontemporal: streaming loads.
 * `blocks` persistent workgroups of 256 threads.  The caller times it with events on `stream`.
 * ------------------------------------------------------------------------------------------- */
int ttk_stream_probe(const float* src, float* dst, float* sink, int64_t rows, int row_bytes, int seg_bytes, int nread, int nwrite,
                     int64_t stream_bytes, int unroll, int nontemporal, int blocks, ttk_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TTK_H_ */
