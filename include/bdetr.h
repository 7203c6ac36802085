/* bdetr.h - C ABI of the MI355X-native DETR training-step hot path.
 *
 * One shared library, libbdetr.so (built from boosted_detr_amd/csrc with hipcc for
 * gfx950).  The reference (mvenouziou/Boosted_DETR) exposes no FFI: its seam is
 * the Keras object surface and all arithmetic is delegated to TensorFlow /
 * tensorflow_addons / scipy.  Each entry point below therefore cites the
 * reference CALL SITE (ModelComponents/<file>:<line>) whose third-party op it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless named host_*; fp32 unless stated
 *  - the caller owns every buffer including workspaces; the library allocates nothing
 *  - `stream` is a hipStream_t passed as void*; all work is stream-ordered, no
 *    implicit synchronisation, safe to capture into a hipGraph
 *  - return value: 0 = ok, <0 = argument/shape error, >0 = hipError_t; the message
 *    is available from bdetr_last_error() (thread-local); nothing throws or aborts
 *  - activations are NHWC, conv weights are OHWI ([Cout][KH][KW][Cin]), dense
 *    weights are [out][in]; the Python host converts from/to the Keras layouts
 *    (HWIO, [in][out]) at set_weights/get_weights time
 */
#ifndef BDETR_H
#define BDETR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define BDETR_ABI_VERSION 8

int         bdetr_abi_version(void);
const char* bdetr_last_error(void);
/* number of CUs of the current device (used by the host to size split-K) */
int         bdetr_device_cus(void);

/* A non-blocking stream of the lowest priority class of the device (never destroyed: process lifetime), for
 * work that must not delay the caller's critical path (the host's weight-gradient side stream), and the
 * device's priority range (numerically larger = lower priority). */
int         bdetr_low_priority_stream_create(void** stream_out);
int         bdetr_stream_priority_range(int* least, int* greatest);
/* (ABI 8) Candidates for a PLACED side stream: HIP binds a new stream to one of a few hardware queues in creation order, and the queue it
 * shares - or does not share - a dispatch pipe with decides how two streams overlap (measured: the same training step at 25.1 / 25.3 / 25.2 /
 * 44.3 ms on the four queues a low-priority stream can land on; engine.side_stream).  Creates `ncand` (1..8) lowest-priority non-blocking
 * streams (-> streams_out[ncand], process lifetime) and measures for each how long `ticks` one-workgroup kernels take back to back on
 * `main_stream` while the candidate runs `loads` launches of a streaming kernel over a 64-MB scratch buffer (-> scores_ms[ncand]; smaller
 * = the two streams get in each other's way less; unloaded_ms, may be null: the same ticks with no load beside them).  Allocates and
 * frees its scratch; synchronises the streams - not inside a capture.  bdetr_stream_destroy: for the candidates not kept. */
int         bdetr_side_stream_candidates(void* main_stream, int ncand, int loads, int ticks, void** streams_out, float* scores_ms, float* unloaded_ms);
int         bdetr_stream_destroy(void* stream);

/* Arithmetic of the conv/GEMM family (inputs and outputs are always fp32):
 *  BDETR_GEMM_FP32    every product on v_mfma_f32_32x32x2_f32 (exact fp32 products);
 *  BDETR_GEMM_BF16X3  every operand is split on the fly into hi + lo bf16 halves and a product is
 *                     hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation
 *                     (~2^-18 relative error per product; TensorFlow's TF32 default is 2^-11);
 *  BDETR_GEMM_MIXED   forward products - conv2d_fwd and GEMMs with grad == 0 - are exact fp32 (class
 *                     ids, match indices and ReLU masks are decided by them), gradient products -
 *                     conv2d_bwd_* and GEMMs with grad != 0 - are split-bf16;
 *  BDETR_GEMM_SPLIT   gradient products split-bf16; forward products split-fp16: hi + lo fp16 halves
 *                     (11 + 11 significant bits), three products on v_mfma_f32_32x32x16_f16 - fp32-grade
 *                     products (~2^-22), but forward operands must stay below 65504 in magnitude (a
 *                     larger value raises the range guard).  Since ABI 5 the pre-split (P16) operands keep
 *                     the lo half UNSCALED in ONE accumulator (the f16 MFMA keeps subnormal operands;
 *                     conv weights are packed as 2^8 w and the epilogue multiplies by 2^-8); only
 *                     igemm.hip's IN-KERNEL split (stem, neck, transformer Dense: operands without a
 *                     producer that could pre-scale them) still scales lo by 2^11 into a second accumulator.
 *  BDETR_GEMM_BF16X6  fp32-grade products ON the 16-bit MFMA with fp32's exponent range (round 4): every operand is split on the
 *                     fly into THREE bf16 terms (8 + 8 + 8 significant bits) and a product keeps the six terms down to 2^-16 of
 *                     it - six v_mfma_f32_32x32x16_bf16, ~2^-22.5 relative error per product.  Meant as the policy of a backward
 *                     pass behind a BDETR_GEMM_SPLIT forward (Python: Model.train_grad_precision = "bf16x6"); the attention core
 *                     runs exact fp32 under it.
 * The env variable BDETR_GEMM_PRECISION=fp32|bf16x3|mixed|split|bf16x6 picks the initial value.
 * This policy is the library's ONE piece of mutable state (a mode word like a rounding mode, not data): it is thread-local,
 * so a host thread's launches are unaffected by another thread's policy, and every launch reads it once on the host at
 * enqueue time - kernels already enqueued (or captured into a hipGraph) keep the arithmetic they were enqueued with. */
enum { BDETR_GEMM_FP32 = 0, BDETR_GEMM_BF16X3 = 1, BDETR_GEMM_MIXED = 2, BDETR_GEMM_SPLIT = 3, BDETR_GEMM_BF16X6 = 4 };
int         bdetr_set_gemm_precision(int mode);
int         bdetr_get_gemm_precision(void);

/* Live profiling of the MFMA (igemm) kernel family for bench.py's roofline leg: when enabled,
 * every conv/GEMM launch is bracketed by hipEvents on its own stream.  bdetr_prof_read (after a
 * stream synchronise) returns the summed kernel time, the launch count and the algorithmic FLOPs
 * (2*I*J*R per GEMM) of everything launched since bdetr_prof_enable(1). */
int bdetr_prof_enable(int on);
int bdetr_prof_read(double* total_ms, int64_t* launches, double* flops);
/* the same sums restricted to the launches that ran one arithmetic: 0 exact fp32, 1 split-bf16, 2 split-fp16 */
int bdetr_prof_read_arith(int arith, double* total_ms, int64_t* launches, double* flops);
/* debug aid: one CSV row per recorded launch (shape, tile, loader kinds, ms, GFLOP) */
int bdetr_prof_dump(const char* path);

/* ---- activation codes for the GEMM / conv epilogue ---- */
enum { BDETR_ACT_NONE = 0, BDETR_ACT_RELU = 1, BDETR_ACT_TANH = 2 };

/* ------------------------------------------------------------------------
 * K1  image preparation - backbone.py:49-56 (clip, Resizing, convert_image_dtype
 *     uint8 round trip, resnet50.preprocess_input caffe mode).
 *     in  : [B,h,w,3] fp32 in [0,1] (any h,w; bilinear half-pixel resize to HxW)
 *     out : [B,H,W,4] fp32, channels BGR minus mean, 4th channel = 0 (so the 7x7
 *           stem conv reads 16-byte pixels)
 * ---------------------------------------------------------------------- */
int bdetr_image_prep(const float* in, int B, int h, int w, float* out, int H, int W, void* stream);

/* Targets that are already tokenised and resident in HBM - tokenizers.py:40-82 minus the StringLookup
 * (strings never reach the device): category ids [rows] are range-checked against C, the attribute id
 * slots [rows][slots] become the multi-hot matrix [rows][A] (tf.one_hot + reduce_max over the slot axis,
 * tokenizers.py:72-82; <PAD> slots set bit 0).  An id outside its vocabulary maps to 1 (<OOV>), as
 * StringLookup maps an unknown string.  rows = B*M. */
int bdetr_tokens_prepare(const int* cat_ids, const int* att_ids, int64_t rows, int slots, int C, int A,
                         int* cat_out, float* att_hot, void* stream);

/* Input-step augmentations (pipeline.py:274-341): per image, bilinear down-size to
 * (new_h,new_w), place at (off_h,off_w) on a zero canvas of the original size, then tf.image
 * adjust_contrast / adjust_brightness / adjust_saturation with the given factors.  in/out: [B,H,W,3]
 * (out != in); iparams int32 [B][4] = new_h,new_w,off_h,off_w; fparams [B][3] = contrast, brightness
 * delta, saturation; ws: bdetr_augment_ws_floats(B) floats.  The random draws are the host's. */
int bdetr_augment_ws_floats(int B);
int bdetr_augment(const float* in, float* out, const int32_t* iparams, const float* fparams,
                  int B, int H, int W, float* ws, void* stream);
/* tf.image.random_jpeg_quality / adjust_jpeg_quality (pipeline.py:319-325): per image, float [0,1] -> uint8 (x * 255.5 truncated,
 * saturating) -> baseline JPEG of quality[b] with 4:2:0 chroma -> decode -> / 255.  Only the lossy stages of the codec run (colour
 * conversion, chroma down-sampling, 8x8 integer DCT, quantisation with the quality-scaled Annex K tables, inverse DCT, fancy
 * up-sampling); entropy coding is lossless and skipped.  The codec is a third-party dependency of the reference (libjpeg-turbo inside
 * TensorFlow): bit-exact against a real libjpeg-turbo through oracle/jpeg_oracle.py (tests/test_jpeg_quality.py).  in == out allowed.
 * quality: int32 [B] on the device; ws: bdetr_jpeg_quality_ws_bytes(B, H, W) bytes. */
int64_t bdetr_jpeg_quality_ws_bytes(int B, int H, int W);
int bdetr_jpeg_quality(const float* in, float* out, const int32_t* quality, int B, int H, int W, void* ws, void* stream);
/* bdetr_augment with the JPEG round trip between brightness and saturation, where the reference has it (pipeline.py:364-383);
 * quality == NULL: exactly bdetr_augment */
int bdetr_augment_jpeg(const float* in, float* out, const int32_t* iparams, const float* fparams, const int32_t* quality,
                       int B, int H, int W, float* ws, void* jpeg_ws, void* stream);

/* ------------------------------------------------------------------------
 * K2/K5/K6  MFMA implicit-GEMM family (v_mfma_f32_32x32x2_f32, LDS-staged tiles).
 * ---------------------------------------------------------------------- */
typedef struct {
    int N, H, W, C;          /* input  NHWC  */
    int K, R, S;             /* output channels, kernel height, kernel width */
    int stride, pad;         /* symmetric zero padding */
    int OH, OW;              /* output spatial size */
} bdetr_conv_desc;

/* conv forward - Keras Conv2D inside keras.applications.ResNet50 (backbone.py:37-38,57)
 * and BackboneNeck.conv2d_downscaler (backbone.py:76-78).
 *   y[N,OH,OW,K] = act(conv(x, w) + bias)
 *   stat_sum/stat_sq (optional, may be NULL): per-row-chunk partial column sums of
 *   y and y*y, shape [bdetr_conv2d_fwd_stat_chunks(d)][K] each - consumed by bdetr_bn_stats_finalize */
int bdetr_conv2d_fwd(const float* x, const float* w, const float* bias, float* y,
                     const bdetr_conv_desc* d, int act, float* stat_sum, float* stat_sq, void* stream);
/* number of partial-statistics rows bdetr_conv2d_fwd writes for this geometry */
int bdetr_conv2d_fwd_stat_chunks(const bdetr_conv_desc* d);
/* dx[N,H,W,C] (+)= conv_transpose(dy, w).  accumulate!=0 adds into dx (residual merge).
 * For stride>1 only 1x1/pad 0 is supported (Keras ResNet-50 v1 strides its 1x1s) and
 * the untouched pixels of dx are written as zero unless accumulate. */
int bdetr_conv2d_bwd_data(const float* dy, const float* w, float* dx,
                          const bdetr_conv_desc* d, int accumulate, void* stream);
/* dw[K,R,S,C] = sum over pixels; split-K partials are combined with fp32 atomics, so the
 * caller must zero dw first when splitk>1 (bdetr_zero).  splitk<=0 = choose automatically. */
int bdetr_conv2d_bwd_weight(const float* x, const float* dy, float* dw,
                            const bdetr_conv_desc* d, int splitk, void* stream);
int bdetr_conv2d_bwd_weight_splitk(const bdetr_conv_desc* d);
/* Deterministic split-K (ABI 6): the same launch, but every r-slice STORES its partial dw into its own slab of the caller's
 * workspace `ws` (16-byte aligned, ws_elems >= bdetr_splitk_workspace_elems(K, R*S*C, splitk) floats) and a second launch adds
 * the slabs to dw in the fixed order z = 0, 1, ...: two runs give bit-identical gradients.  ws == NULL: the atomic form above.
 * Replaces the same Keras autodiff call sites (backbone.py:37-38,57); the reference itself (TF on CPU) is deterministic. */
int bdetr_conv2d_bwd_weight_ws(const float* x, const float* dy, float* dw,
                               const bdetr_conv_desc* d, int splitk, float* ws, int64_t ws_elems, void* stream);
int64_t bdetr_splitk_workspace_elems(int64_t I, int64_t J, int splitk);

/* ------------------------------------------------------------------------
 * Pre-split ("P16") operand path of the same convolutions - csrc/sgemm.hip.  Replaces the same reference
 * call sites as bdetr_conv2d_* (Keras Conv2D + autodiff inside keras.applications ResNet-50,
 * backbone.py:37-38,57); used by the training step under BDETR_GEMM_SPLIT for every layer whose channel
 * counts allow it (bdetr_p16_supported).
 *
 * P16 layout of a row-major [rows][C] matrix, C % 8 == 0, 4 bytes per element like fp32: every group of 8
 * consecutive elements occupies 32 bytes = [8 x hi (16 bit)][8 x lo (16 bit)].
 *   f16 pair  (forward operands):  hi = f16(x),  lo = f16(x - hi), unscaled;  |x| < 65504 or the value is lost.  For |x| < 2^-3 the
 *             lo half is an f16 subnormal (absolute error <= 2^-25: fp32-grade against an operand tensor of RMS ~1; the MFMA keeps
 *             subnormal operands).  The forward copy of a conv WEIGHT holds 2^8 w (weights are ~1e-2: their lo halves stay normal);
 *             bdetr_p16_conv2d_fwd multiplies by 2^-8 in its epilogue.  All three split products share ONE accumulator.
 *   bf16 pair (gradient operands): hi = bf16(x), lo = bf16(x - hi);          fp32 range
 * The producer of a tensor writes it (bdetr_bn_apply_p16 / bdetr_bn_bwd_p16 / bdetr_p16_pack*), the GEMM moves
 * it HBM -> LDS with buffer_load ... lds and evaluates a product as three 16-bit MFMA products, fp32
 * accumulate - the arithmetic of BDETR_GEMM_SPLIT without the in-kernel split.
 *
 * overflow_flag (device int*, may be null): set to 1 when a value cannot be represented by the f16 pair
 * (|x| >= 65504 or not finite); never cleared by the library. */
/* BatchNormalization apply / backward of bdetr_bn_apply / bdetr_bn_bwd (same reference call sites, same fp32
 * arithmetic) with P16 outputs for the consumer convolutions: out_f16 = forward operand, out_bf16 = weight-
 * gradient operand, out32 = the fp32 tensor; any of the three may be null.  residual_p16 != 0: `residual` is the
 * f16 pair copy of the shortcut tensor (block outputs need not exist in fp32).  bdetr_bn_bwd_p16 writes the input
 * gradient as a bf16 pair (dx_bf16) and, when dx32 != null, in fp32 too.  ReLU mask source of the backward pass
 * (`out`, needed only when a residual was added - otherwise the mask is recomputed from x): out_p16 = 0 the fp32
 * forward output, 1 its bf16 pair copy (the hi half has fp32's range: hi > 0 <=> value > 0), 2 the bit mask that
 * bdetr_bn_apply_p16 writes to relu_mask (uint64 words, (rows*C/4 + 63)/64*4 of them: one bit per element instead of
 * re-reading a 4-byte-per-element tensor in both backward passes). */
/* residual_p16 = 2: `residual` is the RAW output of the projection shortcut's convolution and residual_bn its BatchNorm
 * (batch statistics already reduced): out = relu?(bn(x) + residual_bn(residual)) in one pass - the shortcut's normalised
 * tensor (Keras conv*_block1_0_bn) is never written or read back. */
typedef struct { const float* mean; const float* rstd; const float* gamma; const float* beta; } bdetr_bn_affine;
int bdetr_bn_apply_p16(const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const void* residual, int residual_p16, const bdetr_bn_affine* residual_bn, int relu, float* out32,
                       void* out_f16, void* out_bf16, uint64_t* relu_mask, int* overflow_flag, int64_t rows, int C,
                       void* stream);
int bdetr_bn_bwd_p16(const float* dout, const void* out, int out_p16, const float* x, const float* mean,
                     const float* rstd, const float* gamma, const float* beta, int relu, int frozen,
                     float* dx32, void* dx_bf16, float* dgamma, float* dbeta, float* dresidual,
                     float* ws, const float* pre_g, const float* pre_gx, int pre_n, int64_t rows, int C, void* stream);
/* pre_g / pre_gx [pre_n][C] (may be null): partial sums of g and g * xhat already produced by the backward-data
 * epilogue that wrote dout (bdetr_p16_conv2d_bwd_data_bnstats); the reduction pass over dout and x is skipped. */
/* The same for a dout [N,H,W,C] that is zero outside the pixels (2i, 2j): the gradient of a stage's last unit, which reaches it only
 * through the backward-data of the next stage's stride-2 1x1 convolutions (keras ResNet50 conv{3,4,5}_block1_{0,1}_conv).  The
 * reduction pass visits those N * ceil(H/2) * ceil(W/2) rows only - a quarter of the bytes; the apply pass is the dense one. */
int bdetr_bn_bwd_p16_even_pixels(const float* dout, const void* out, int out_p16, const float* x, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, int relu, int frozen,
                                 float* dx32, void* dx_bf16, float* dgamma, float* dbeta, float* dresidual,
                                 float* ws, int N, int H, int W, int C,
                                 int dout_compact /* ABI 7: dout is the compact [N, H/2, W/2, C] tensor of those pixels alone (H, W even;
                                 dresidual must be null): see bdetr_p16_conv2d_bwd_data_masked_accum_compact */, void* stream);
int bdetr_p16_supported(const bdetr_conv_desc* d);
int bdetr_p16_pack(const float* x, int64_t n, void* f16_out, void* bf16_out, int* overflow_flag, void* stream);
int bdetr_p16_unpack(const void* p, int is_f16, int64_t n, float* out, void* stream);
/* w [K][R][S][C] fp32 -> w_f16 P16-f16 [K][R*S*C] (forward B operand) and wt_bf16 P16-bf16 [C][R*S][K] with
 * flipped taps, wt[c][r][s][k] = w[k][R-1-r][S-1-s][c] (backward-data B operand); either may be null */
int bdetr_p16_pack_conv_weights(const float* w, int K, int R, int S, int C, void* w_f16, void* wt_bf16,
                                int* overflow_flag, void* stream);
/* the same for many weight tensors in one launch: `table` (device) holds ntensors rows of 7 int64
 * {w, w_f16, wt_bf16, K, R, S, C}; null output pointers skip that copy */
int bdetr_p16_pack_conv_weights_multi(const int64_t* table, int ntensors, int* overflow_flag, void* stream);
/* y fp32 [N,OH,OW,K] = conv(x_f16 P16-f16 [N,H,W,C], w_f16) + bias, act, BatchNorm partial column sums as
 * bdetr_conv2d_fwd (stat_* sized with bdetr_p16_conv2d_fwd_stat_chunks rows) */
int bdetr_p16_conv2d_fwd_stat_chunks(const bdetr_conv_desc* d);
int bdetr_p16_conv2d_fwd(const void* x_f16, const void* w_f16, const float* bias, float* y,
                         const bdetr_conv_desc* d, int act, float* stat_sum, float* stat_sq, void* stream);
/* dx fp32 [N,H,W,C] (+)= conv_transpose(dy_bf16 P16-bf16 [N,OH,OW,K], wt_bf16) */
int bdetr_p16_conv2d_bwd_data(const void* dy_bf16, const void* wt_bf16, float* dx,
                              const bdetr_conv_desc* d, int accumulate, void* stream);
/* The same product with the BatchNorm-backward REDUCTION of the layer that produced this conv's input fused into the
 * epilogue: dx is the gradient of out = relu?(gamma * (y - mean) * rstd + beta) (no residual), and while the dx tile is
 * stored its per-column partial sums of g = dx * [out > 0] and g * xhat are written to part_g / part_gx
 * [bdetr_p16_conv2d_bwd_data_stat_chunks(d)][C] - the pass of bdetr_bn_bwd(_p16) that would re-read dx and y.  Hand
 * them to bdetr_bn_bwd_p16 (pre_g / pre_gx / pre_n). */
typedef struct {
    const float* y; const float* mean; const float* rstd; const float* gamma; const float* beta; int relu;
    float* part_g; float* part_gx;
    const uint64_t* relu_mask;      /* may be null; else out = relu(bn(y) + shortcut): the ReLU decision is bdetr_bn_apply_p16's bit mask (1x1 convs only) */
    /* A second BatchNorm over the SAME gradient (may be null; bdetr_p16_conv2d_bwd_data_masked_accum with relu_mask only): the projection
     * shortcut of a stage's first unit, out = relu(bn3(y3) + bn0(y0)) (keras.applications.resnet block1 with conv_shortcut, reference
     * backbone.py:37-38).  Its sum of g is part_g; part_gx2 receives its sum of g * xhat0, xhat0 = (y2 - mean2) * rstd2. */
    const float* y2; const float* mean2; const float* rstd2; float* part_gx2;
} bdetr_bn_bwd_fuse;
int bdetr_p16_conv2d_bwd_data_stat_chunks(const bdetr_conv_desc* d);
/* x <- x * relu_mask in place (n elements, n % 4 == 0): materialises a gradient that was handed on with its unit's ReLU mask
 * still to be applied, for consumers other than bdetr_p16_conv2d_bwd_data_masked_accum / bdetr_bn_bwd_p16(out_p16 = 2). */
int bdetr_relu_mask_apply(float* x, const uint64_t* relu_mask, int64_t n, void* stream);
/* 1x1 stride-1 backward-data that merges the skip branch of a residual unit: on entry dx holds the gradient of the unit's
 * OUTPUT (after the ReLU), on exit dx = conv_transpose(dy) + dx * relu_mask, relu_mask = the 1-bit-per-element mask
 * bdetr_bn_apply_p16 wrote for that unit.  The masked gradient of the skip branch (Keras: the Add + Activation of
 * keras.applications.resnet block1, reference backbone.py:37-38) is never materialised. */
int bdetr_p16_conv2d_bwd_data_masked_accum(const void* dy_bf16, const void* wt_bf16, float* dx, const uint64_t* relu_mask,
                                           const bdetr_conv_desc* d, const bdetr_bn_bwd_fuse* bn /* may be null: the BatchNorm-backward
                                           sums of the unit whose output gradient this launch completes */, void* stream);
/* (ABI 7) The same merge when the unit's output gradient exists only at the even pixels: old_even is the COMPACT fp32
 * [N, H/2, W/2, C] tensor that the stride-2 1x1 backward-data products of the next stage wrote densely (Keras: the stride-2 conv1 and
 * projection shortcut of the next stage's first block, reference backbone.py:37-38), zero everywhere else - no zero-filled dense
 * gradient is ever written.  dx (fp32 [N,H,W,C], uninitialised on entry) = conv_transpose(dy) + expand(old_even) * relu_mask. */
int bdetr_p16_conv2d_bwd_data_masked_accum_compact(const void* dy_bf16, const void* wt_bf16, float* dx, const float* old_even,
                                                   const uint64_t* relu_mask, const bdetr_conv_desc* d, const bdetr_bn_bwd_fuse* bn, void* stream);
int bdetr_p16_conv2d_bwd_data_bnstats(const void* dy_bf16, const void* wt_bf16, float* dx,
                                      const bdetr_conv_desc* d, const bdetr_bn_bwd_fuse* bn, void* stream);
/* dw fp32 [K][R][S][C] += sum over pixels of dy x patches(x); with splitk > 1 the slices add with float
 * atomics, so dw must hold zeros (or the running sum) on entry */
int bdetr_p16_conv2d_bwd_weight_splitk(const bdetr_conv_desc* d);
int bdetr_p16_conv2d_bwd_weight(const void* x_bf16, const void* dy_bf16, float* dw,
                                const bdetr_conv_desc* d, int splitk, void* stream);
/* The same with x given as the P16-f16 tensor the FORWARD of this convolution read: the kernel converts each fragment to a bf16 pair
 * in registers ((hi + lo) is exact in fp32, then the bf16 split), so a producer (bdetr_bn_apply_p16, bdetr_p16_pack) need not
 * write a bf16 copy of an activation at all: 4 bytes per element of HBM traffic and of saved-activation memory less. */
int bdetr_p16_conv2d_bwd_weight_xf16(const void* x_f16, const void* dy_bf16, float* dw,
                                     const bdetr_conv_desc* d, int splitk, void* stream);
/* deterministic split-K form of the two above (see bdetr_conv2d_bwd_weight_ws); x_is_f16 selects the xf16 operand flavour */
int bdetr_p16_conv2d_bwd_weight_ws(const void* x, int x_is_f16, const void* dy_bf16, float* dw,
                                   const bdetr_conv_desc* d, int splitk, float* ws, int64_t ws_elems, void* stream);

/* strided-batched GEMM - tf Dense / tf.linalg.matmul call sites
 * (transformers.py:41-48,62-65,86,97,101,174-177; prediction_heads.py:40-43,106-110,175-179)
 *   C[b][i][j] = act(alpha * sum_r A(b,i,r) * B(b,j,r) + bias[j])      (accumulate: C += ...)
 *   a_rcontig: A(b,i,r) at a + i*lda + r, else at a + r*lda + i           (same for B with j)
 *   batch index b = b0*nb1 + b1, operand offset = b0*s?0 + b1*s?1
 *   splitk>1 splits the r range over gridDim.z and combines with atomics (C must be zeroed,
 *   nb0*nb1 must be 1, bias/act must be off). */
typedef struct {
    int I, J, R;
    int nb0, nb1;
    const float* a; int64_t lda, sa0, sa1; int a_rcontig;
    const float* b; int64_t ldb, sb0, sb1; int b_rcontig;
    float*       c; int64_t ldc, sc0, sc1;
    const float* bias; float alpha; int act; int accumulate; int splitk;
    int grad;                /* != 0: a gradient (backward) product - see BDETR_GEMM_MIXED */
} bdetr_gemm_desc;
int bdetr_gemm(const bdetr_gemm_desc* g, void* stream);
/* bdetr_gemm with a deterministic split-K reduction (g->splitk > 1, dense C: ldc == J): slabs in `ws`
 * (>= bdetr_splitk_workspace_elems(I, J, splitk) floats), fixed-order fold into C (C += sum of the slices).  ws == NULL: bdetr_gemm. */
int bdetr_gemm_ws(const bdetr_gemm_desc* g, float* ws, int64_t ws_elems, void* stream);
/* n (1..4) independent GEMMs that share J, R, leading dimensions, operand flavours and epilogue flags
 * (their pointers, bias and row count I may differ) in ONE launch - the Q/K/V projections of an
 * attention block (transformers.py:68-70) and their input gradients.  No batching inside.  splitk > 1 (the same value in every
 * problem, problems of one shape, no bias / activation): every problem's r range is cut into the same slices, which ADD into C
 * with float atomics (C zeroed by the caller) - the weight gradients of one layer's Dense kernels in one launch. */
int bdetr_gemm_grouped(const bdetr_gemm_desc* g, int n, void* stream);

/* column sums: out[j] = sum_i x[i][j]  (bias gradients; Keras autodiff of Dense/Conv bias).
 * Deterministic two-level reduction; ws: cols * bdetr_colsum_chunks(rows) floats. */
int bdetr_colsum_chunks(int64_t rows);
int bdetr_colsum(const float* x, int64_t rows, int cols, float* out, float* ws, void* stream);
/* the same sum ADDED into out with float atomics in one launch (no workspace): out must hold zeros or the running
 * sum - the host's flat gradient buffer is zero-filled once per step */
int bdetr_colsum_accumulate(const float* x, int64_t rows, int cols, float* out, void* stream);
/* n <= 4 such column sums of one width in ONE launch (the bias gradients of an attention block's Q / K / V projections,
 * transformers.py:68-70 + autodiff).  xs / rows / outs are HOST arrays of n entries; outs[m] must hold zeros or the running sum. */
int bdetr_colsum_accumulate_group(const float* const* xs, const int64_t* rows, int cols, float* const* outs, int n, void* stream);

/* ------------------------------------------------------------------------
 * K3  BatchNormalization, training mode (keras BN inside ResNet-50, backbone.py:79-80,
 *     prediction_heads.py:42,108,177; semantics SURVEY S5).  x is [rows][C].
 * ---------------------------------------------------------------------- */
/* partial column sums of x and x*x over row chunks: part_* are [bdetr_bn_bwd_chunks(rows)][C].
 * (The conv/GEMM epilogue produces the same partials for free - see bdetr_conv2d_fwd.) */
int bdetr_colstats(const float* x, int64_t rows, int C, float* part_sum, float* part_sq, void* stream);
/* reduce nparts partial rows (from bdetr_colstats or a conv epilogue) in fp64, fixed order, into
 * mean[C], rstd[C] (biased variance); if moving_mean!=NULL update the moving statistics
 * (momentum; bessel!=0 uses the unbiased variance for the moving estimate).  x is unused. */
int bdetr_bn_stats(const float* x, int64_t rows, int C, const float* part_sum, const float* part_sq,
                   int nparts, float eps, float momentum, int bessel,
                   float* mean, float* rstd, float* moving_mean, float* moving_var, float* fold_ws,
                   int* guard_flag, void* stream);
/* guard_flag (device int*, may be null): the step's range guard.  The P16-f16 producers set it when a forward
 * activation leaves the f16 pair's range; bn_stats sets it when the batch statistics are not finite and skips
 * the moving-statistics update while it is set; bdetr_flag_nonfinite sets it for a non-finite loss; the
 * optimizer applies nothing while it is set.  The host reads and clears it and redoes the step on the
 * exact-fp32 forward (the reference's fp32 arithmetic has no such range limit). */
int bdetr_flag_nonfinite(const float* x, int64_t n, int* flag, void* stream);
/* The guard word logged per step without a host synchronisation.  ordinal: one device int, advanced by each call's kernel;
 * host_ring: 1 + ring_len ints of pinned, device-mapped HOST memory.  The kernel stores *flag at host_ring[1 + ordinal % ring_len]
 * and then the ordinal at host_ring[0], so a host that finds host_ring[0] >= k may read step k's entry (Model._guard_poll).
 * Capturable: it belongs to the step (the last launch of its optimizer segment), nothing is enqueued between steps. */
int bdetr_flag_snapshot(const int* flag, int* ordinal, int* host_ring, int ring_len, void* stream);
/* Diagnostic of the hipGraph replay path (no reference counterpart; tools/graph_segment_checksums.py): appends
 * {wrapping sum of x's bit patterns, count of non-finite elements, tag} to log[3 * (*cursor)++] (device memory, `cap` entries;
 * scratch: two zeroed device words).  Order-independent, stream-ordered, capturable: every replay of a segment appends an entry. */
int bdetr_debug_checksum(const float* x, int64_t n, uint64_t* scratch, uint64_t* log, int* cursor, int cap, uint64_t tag, void* stream);
/* Diagnostic of the hipGraph replay path (ABI 7; no reference counterpart): counts[t] = nodes of hipGraphNodeType t in `graph` (a
 * hipGraph_t; types >= ncounts are counted in the last slot).  The captured training step must hold kernel nodes only - a hipMemset node
 * replayed wrongly on this runtime (round 4) - and engine.SegmentedCapture enforces that with this call under BDETR_GRAPH_CENSUS=1. */
int bdetr_graph_node_census(void* graph, int64_t* counts, int ncounts);
/* fold_ws (optional): 2*C*bdetr_bn_stats_fold_rows() floats; lets bn_stats pre-reduce thousands of
 * epilogue partial rows with a wide grid before the fp64 finalise */
int bdetr_bn_stats_fold_rows(void);
/* inference-mode statistics: mean = moving_mean, rstd = 1/sqrt(moving_var+eps) */
int bdetr_bn_stats_frozen(const float* moving_mean, const float* moving_var, int C, float eps,
                          float* mean, float* rstd, void* stream);
/* out = [relu]( gamma*(x-mean)*rstd + beta [+ residual] ) */
int bdetr_bn_apply(const float* x, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, const float* residual, int relu, float* out,
                   int64_t rows, int C, void* stream);
/* backward of bn_apply.  dout: gradient w.r.t. out; out: forward output (for the ReLU mask;
 * may be NULL when relu==0, or when relu==1 and there was no residual: the mask is then recomputed
 * from x with the forward's exact affine map, saving one full read of `out` per pass).  Produces dx, dgamma[C], dbeta[C] and, when dresidual!=NULL,
 * the masked gradient that flows to the residual branch (dresidual may alias dout).
 * frozen!=0: statistics were constants (inference-mode BN).  ws: 2*C*nchunks floats where
 * nchunks = bdetr_bn_bwd_chunks(rows). */
int bdetr_bn_bwd_chunks(int64_t rows);
int bdetr_bn_bwd(const float* dout, const float* out, const float* x, const float* mean,
                 const float* rstd, const float* gamma, const float* beta, int relu, int frozen,
                 float* dx, float* dgamma, float* dbeta, float* dresidual,
                 float* ws, int64_t rows, int C, void* stream);

/* ------------------------------------------------------------------------
 * K4  3x3/2 max-pool with 1-pixel zero pad (ResNet-50 pool1_pad + pool1_pool)
 * ---------------------------------------------------------------------- */
int bdetr_maxpool3x3s2_fwd(const float* x, float* y, int N, int H, int W, int C, int OH, int OW, void* stream);
int bdetr_maxpool3x3s2_bwd(const float* x, const float* y, const float* dy, float* dx,
                           int N, int H, int W, int C, int OH, int OW, void* stream);

/* The ResNet stem's tail in one pass each way (training, batch statistics): BatchNormalization -> ReLU -> ZeroPadding2D(1) ->
 * MaxPool 3x3/2 over the RAW conv1 output y [N,H,W,C] (keras ResNet50 conv1_bn .. pool1_pool, reached from backbone.py:79-80).
 * fwd: pooled tensor [N,PH,PW,C] (PH = (H-1)/2+1) as fp32 (out32) and / or as the f16 pair of the P16 layout (out_f16), plus
 *      tap [N,PH,PW,C] bytes: which of the window's 9 taps (row-major) held the maximum - the first one on a tie.
 * bwd: dpool [N,PH,PW,C] -> dy [N,H,W,C] (gradient of the raw conv output), dgamma, dbeta; the normalised full-resolution tensor
 *      and its gradient are never materialised.  ws: 2*C*bdetr_stem_pool_bwd_chunks(N*H*W) floats.  C % 8 == 0, N*H*W < 2^31. */
int bdetr_stem_pool_bwd_chunks(int64_t rows);
int bdetr_stem_pool_fwd(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        int N, int H, int W, int C, float* out32, void* out_f16, uint8_t* tap, int* overflow_flag, void* stream);
int bdetr_stem_pool_bwd(const float* dpool, const uint8_t* tap, const float* y, const float* mean, const float* rstd, const float* gamma,
                        const float* beta, int N, int H, int W, int C, float* dy, int dy_p16 /* ABI 7: write dy as the P16 bf16 pair
                        (the operand of bdetr_p16_stem_bwd_weight) instead of fp32 */, float* dgamma, float* dbeta, float* ws, void* stream);
/* The stem's weight gradient on pre-split operands (ABI 7).  keras ResNet50 conv1_conv - 7x7 / stride 2 on the ZeroPadding2D(3) image,
 * reference backbone.py:37-38 - reads a 4-channel image, which the P16 layout (groups of 8 channels) cannot hold; space-to-depth makes it
 * a size-preserving stride-1 4x4 convolution over x2[n][i][j][(a,b,c)] = x[n][2i+a][2j+b][c] (16 channels, low-side padding 2), tap (r,s) =
 * (2r'+a-1, 2s'+b-1).  bdetr_p16_s2d_pack_bf16: x fp32 [N,H,W,4] (H, W even) -> x2 P16-bf16 [N,H/2,W/2,16].  bdetr_p16_stem_bwd_weight:
 * dw2 fp32 [K][4][4][16] += sum over pixels of dy x patches(x2) (dw2 must hold zeros: split-K float atomics; K % 64 == 0; dy P16-bf16
 * [N,H/2,W/2,K]).  bdetr_p16_s2d_unpack_dw: dw [K][7][7][4] <- dw2 (the r = -1 / s = -1 taps of the 4x4 form do not exist and are dropped). */
int bdetr_p16_s2d_pack_bf16(const float* x, int N, int H, int W, void* out_bf16, void* stream);
int bdetr_p16_stem_bwd_weight(const void* x2_bf16, const void* dy_bf16, float* dw2, int N, int H2, int W2, int K, void* stream);
int bdetr_p16_s2d_unpack_dw(const float* dw2, float* dw, int K, void* stream);

/* ------------------------------------------------------------------------
 * K7  attention softmax (transformers.py:88-94): p = softmax(scale*s) row-wise, in place ok.
 *     bwd: ds = scale * p * (dp - sum_k dp*p)
 * ---------------------------------------------------------------------- */
int bdetr_softmax_rows_fwd(const float* s, float* p, int64_t rows, int cols, float scale, void* stream);
int bdetr_softmax_rows_bwd(const float* p, const float* dp, float* ds, int64_t rows, int cols, float scale, void* stream);

/* Fused attention core for head dimension bdetr_attention_head_dim() (= 32): the [B,h,q,k] scores
 * never reach HBM (online softmax in registers, K/V streamed through LDS, fp32 MFMA).
 *   q [B,nq,h*32], k/v [B,nk,h*32]  ->  o [B,h,nq,32] (the layout transformers.py:100 reshapes
 *   without a permute), lse [B,h,nq] = log-sum-exp of the scaled scores (kept for the backward).
 *   bwd: dq [B,nq,h*32], dk/dv [B,nk,h*32]; dvec_ws: B*h*nq floats of workspace. */
int bdetr_attention_head_dim(void);
int bdetr_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse,
                        int B, int h, int nq, int nk, float scale, void* stream);
int bdetr_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o,
                        const float* lse, float* dq, float* dk, float* dv, float* dvec_ws,
                        int B, int h, int nq, int nk, float scale, void* stream);

/* ------------------------------------------------------------------------
 * K8  residual + dropout + LayerNormalization (transformers.py:135-137,178-180)
 *     h = x + dropout(y) ; out = gamma*(h-mean)*rstd + beta   (eps 1e-3, biased var)
 *     dropout: inverted, keep-prob 1-rate, counter-based RNG keyed by (seed, element index);
 *     rate==0 disables.  seed_base (device uint64*, may be null): a per-step value kept in HBM and folded
 *     into `seed`, so that a captured hipGraph draws fresh masks on every replay (the host updates the word
 *     between replays); forward and backward of one layer must be given the same pair.
 * ---------------------------------------------------------------------- */
int bdetr_add_dropout_layernorm_fwd(const float* x, const float* y, const float* gamma, const float* beta,
                                    float* out, float* mean, float* rstd, int64_t rows, int D,
                                    float eps, float rate, uint64_t seed, const uint64_t* seed_base, void* stream);
/* given dout: dx (gradient to x, = dh), dy (gradient to y, = dh * dropout mask), dgamma, dbeta.
 * `out` is the forward output (h is recovered as (out-beta)/gamma is NOT used; xhat is
 * recomputed from x,y,mean,rstd).  ws: 2*D*bdetr_ln_bwd_chunks(rows) floats. */
int bdetr_ln_bwd_chunks(int64_t rows);
int bdetr_add_dropout_layernorm_bwd(const float* dout, const float* x, const float* y, const float* gamma,
                                    const float* mean, const float* rstd, float* dx, float* dy,
                                    float* dgamma, float* dbeta, float* ws, int64_t rows, int D,
                                    float rate, uint64_t seed, const uint64_t* seed_base, int accumulate_dx, void* stream);

/* ----------------------------------------------------------------------
 * Row chain (csrc/rowchain.hip, ABI 6): the row-local part of a transformer layer between two attention cores as ONE launch
 * per direction, model width 256 (bdetr_rowchain_width).  Replaces, for an AttentionBlock followed by a FeedForwardBlock
 * (transformers.py:101 OutputProjection, 135-137 Add + Dropout + LayerNormalization, 174-177 DenseRelu / DenseLinear, 178-180
 * Add + Dropout + LayerNormalization) the five forward and ~twenty backward launches of the unfused path:
 *     a = ctx Wo^T + bo;  x1 = LN1(resid + drop1(a));  h = relu(x1 W1^T + b1);  f = h W2^T + b2;  x2 = LN2(x1 + drop2(f))
 * nstages = 3: the whole chain; nstages = 1: stage 1 only (an AttentionBlock followed by another AttentionBlock:
 * DecoderBlock's self-attention, transformers.py:378-383).  All activations fp32 [M][256] row-major; weights PRE-PACKED by
 * bdetr_rowchain_pack_weights (per-optimizer-step, one launch for all matrices): forward copy = f16 pairs of 2^8 W in MFMA
 * fragment order, backward copy = bf16 pairs of W^T (the 'split' policy's arithmetic: BDETR_GEMM_SPLIT above; other policies
 * use the unfused path).  Dropout: the hash, seeds and element indices of bdetr_add_dropout_layernorm_* - same masks.
 * Forward saves what the backward needs: pre1 / pre2 (the LayerNorm inputs resid + drop(a), x1 + drop(f)), mean / rstd, h.
 * ---------------------------------------------------------------------- */
typedef struct {
    int64_t M; int nstages; float eps, rate;
    const float* ctx; const float* resid;
    const void* w[3];            /* packed forward copies of Wo, W1, W2 */
    const float* bias[3];
    const float* g1; const float* b1; const float* g2; const float* b2;      /* LayerNorm gamma / beta */
    float* pre1; float* x1; float* mean1; float* rstd1;                      /* x1 = the result when nstages == 1 */
    float* h; float* pre2; float* x2; float* mean2; float* rstd2;
    uint64_t seed1, seed2; const uint64_t* seed_base;
} bdetr_rowchain_fwd_desc;
int bdetr_rowchain_fwd(const bdetr_rowchain_fwd_desc* d, void* stream);
/* Backward of the same chain from dout = gradient of x2 (nstages 3) or x1 (nstages 1): writes dctx and dresid, the three Dense
 * layers' output gradients G0 (of a), G1 (of the ReLU's input), G2 (of f) - the operands of their weight gradients, which stay
 * GEMMs of their own (dW = G^T X: bdetr_gemm) - and one row of partial sums per 32-token workgroup for the seven vector
 * gradients {dgamma2, dbeta2, dbias2, dbias1, dgamma1, dbeta1, dbias0} (for nstages == 1 only the last three are written):
 * partials holds bdetr_rowchain_partial_rows(M) x 7 x 256 floats; bdetr_rowchain_reduce folds them in a fixed order into up to
 * seven destinations (dst7[v] may be null; accumulate7[v] != 0: add to what is there). */
typedef struct {
    int64_t M; int nstages; float rate;
    const float* dout;
    const float* pre2; const float* mean2; const float* rstd2; const float* g2;
    const float* h;
    const float* pre1; const float* mean1; const float* rstd1; const float* g1;
    const void* wt[3];           /* packed backward copies of Wo, W1, W2 */
    float* G2; float* G1; float* G0;
    float* dresid; float* dctx;
    float* partials;
    uint64_t seed1, seed2; const uint64_t* seed_base;
} bdetr_rowchain_bwd_desc;
int bdetr_rowchain_bwd(const bdetr_rowchain_bwd_desc* d, void* stream);
int bdetr_rowchain_reduce(const float* partials, int nparts, float* const* dst7, const int* accumulate7, void* stream);
int bdetr_rowchain_width(void);
int64_t bdetr_rowchain_pack_elems(void);                 /* floats (4-byte units) of one packed copy of a 256 x 256 matrix */
int bdetr_rowchain_partial_rows(int64_t M);
/* table: device int64 [n][3] rows {W fp32 [256 out][256 in], forward copy | 0, backward copy | 0}; overflow_flag: the range guard
 * (a weight whose 2^8 multiple leaves the f16 range raises it) */
int bdetr_rowchain_pack_weights(const int64_t* table, int n, int* overflow_flag, void* stream);

/* ------------------------------------------------------------------------
 * Panoptic head (forward only - the reference never wires it into a loss):
 * panoptic_neck.py:20-21 Resizing (bilinear, half-pixel centres), 118-121 / 164-167 channel LayerNormalization
 * (eps 1e-3) + ReLU(negative_slope), 33-47 Concatenate / transpose; transformers.py:519 LayerNormalization.
 * Channel counts of the head are arbitrary (66, 44, 29 ...): tensors are stored with the channel dimension
 * zero-padded to a multiple of 4 (ld), and the convolutions run on bdetr_conv2d_fwd with zero-padded weights.
 *   resize        in [B,h,w,C] -> out [B,H,W,C], C % 4 == 0
 *   layernorm_act out[r][c] = leaky_slope(gamma[c] * (x[r][c] - mean_r) * rstd_r + beta[c]) for c < C, 0 for C <= c < ldo
 *   copy_cols     dst[r][dst_col0 + c] = src[r][c], c < C            (channel concatenation / padding)
 *   nhwc_to_nchw  out[b][c][p] = in[b][p][c], c < C of ld_in columns
 * ---------------------------------------------------------------------- */
int bdetr_resize_bilinear_nhwc(const float* in, int B, int h, int w, int C, float* out, int H, int W, void* stream);
int bdetr_layernorm_act_fwd(const float* x, int64_t rows, int C, int ldx, const float* gamma, const float* beta,
                            float eps, float slope, float* out, int ldo, void* stream);
int bdetr_copy_cols(const float* src, int64_t rows, int C, int ld_src, float* dst, int ld_dst, int dst_col0, void* stream);
int bdetr_nhwc_to_nchw(const float* in, int B, int P, int C, int ld_in, float* out, void* stream);

/* ------------------------------------------------------------------------
 * K9  head activations (prediction_heads.py:44,60-62,111,127-129,180,197-199)
 * ---------------------------------------------------------------------- */
int bdetr_softmax_lastdim_fwd(const float* logits, float* p, int64_t rows, int cols, void* stream);
int bdetr_softmax_lastdim_bwd(const float* p, const float* dp, float* dlogits, int64_t rows, int cols, void* stream);
int bdetr_sigmoid_fwd(const float* x, float* y, int64_t n, void* stream);
int bdetr_sigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);
/* y = 3*sigmoid(x/100) - 1 */
int bdetr_boxsigmoid_fwd(const float* x, float* y, int64_t n, void* stream);
int bdetr_boxsigmoid_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);

/* elementwise helpers */
int bdetr_zero(float* p, int64_t n, void* stream);
int bdetr_add(const float* a, const float* b, float* out, int64_t n, void* stream);          /* out = a + b */
int bdetr_add_bcast_rows(const float* a, const float* row, float* out, int64_t rows, int64_t rowlen, void* stream); /* out[r] = a[r] + row (batch broadcast) */
int bdetr_sum_over_batch(const float* x, float* out, int64_t batch, int64_t n, int accumulate, void* stream);       /* out = sum_b x[b] */
int bdetr_tanh_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);
int bdetr_relu_bwd(const float* y, const float* dy, float* dx, int64_t n, void* stream);
int bdetr_axpy(float alpha, const float* x, float* y, int64_t n, void* stream);               /* y += alpha*x */

/* ------------------------------------------------------------------------
 * K10-K12  set criterion (losses_and_metrics.py:111-161, 176-192, 195-251)
 *   One workgroup per image.  Inputs:
 *     cat_pred [B,N,C] probabilities, att_pred [B,N,A], box_pred [B,N,4]
 *     cat_ids  int32 [B,M] (0=<PAD>), att_hot [B,M,A] multi-hot fp32 (may be NULL if A==0 or
 *     attribute_weight==0), bbox [B,M,4] COCO format, num_objects int32 [B]
 *   bdetr_cost_matrix : cost[B,M,N] = 1000*cat + box_w*box + att_w*att  (fp32, summed in the
 *                       reference's order, line 130); optional component outputs.  Rows
 *                       m >= num_objects[b] (never read by the matcher, masked out of the loss)
 *                       are written as 0 when num_objects != NULL
 *   bdetr_lsa         : exact rectangular shortest-augmenting-path (Crouse / scipy semantics,
 *                       fp64 in LDS, scipy's scan order and tie rule), rows i<num_objects[b];
 *                       match[b][m] = assigned prediction index or -1 ;  int32 [B,M]
 *   bdetr_set_loss    : masked reductions + existence BCE + IoU metric and the gradient of
 *                       sum_b total_b w.r.t. the three prediction tensors.
 *                       losses[6][B] = total, category, attribute, box, exist, iou
 *                       loss_scale multiplies the gradients (1/num_replicas under DP, S14).
 * ---------------------------------------------------------------------- */
typedef struct {
    int B, M, N, C, A;
    float category_weight, attribute_weight, box_weight, exist_weight;
} bdetr_loss_desc;
int bdetr_cost_matrix(const bdetr_loss_desc* d, const float* cat_pred, const float* att_pred,
                      const float* box_pred, const int32_t* cat_ids, const float* att_hot,
                      const float* bbox, const int32_t* num_objects, float* cost, float* cost_cat,
                      float* cost_att, float* cost_box, void* stream);
int bdetr_lsa(const float* cost, const int32_t* num_objects, int B, int M, int N,
              int32_t* match, void* stream);
int bdetr_set_loss(const bdetr_loss_desc* d, const float* cat_pred, const float* att_pred,
                   const float* box_pred, const int32_t* cat_ids, const float* att_hot,
                   const float* bbox, const int32_t* num_objects, const int32_t* match,
                   float* losses, float* d_cat, float* d_att, float* d_box, float loss_scale,
                   void* stream);
/* mask[B,M,N] = 1 where match[b][m]==n (MatchingAssignment output, for inspection/tests) */
int bdetr_match_to_mask(const int32_t* match, float* mask, int B, int M, int N, void* stream);

/* ------------------------------------------------------------------------
 * K13  optimizer: per-tensor clip-by-norm + Keras SGD(nesterov) (notebook cell 26, S15).
 *      Multi-tensor: `ptrs` is a device array of ntensors x {w,g,v} pointers, `sizes` the
 *      element counts; norms: ntensors floats of workspace.
 *      Work is split in slabs of bdetr_sgd_slab_elems() elements; the host builds the slab
 *      table once: slab_tensor[nslabs] (owning tensor of each slab) and slab_first[ntensors+1]
 *      (first slab of each tensor), both int64 on the device.  partial: nslabs floats.
 *      lr is read from device memory (so a captured graph sees schedule updates).
 *      skip_flag (optional, the range guard's device int): nothing is applied while it is up, and the call itself raises it
 *      when a tensor's gradient norm is not finite (a NaN / Inf born in the backward pass) - no tensor is then updated. */
int bdetr_sgd_slab_elems(void);
int bdetr_sgd_nesterov_clipnorm(const uint64_t* ptrs, const int64_t* sizes, int ntensors,
                                const int64_t* slab_tensor, const int64_t* slab_first, int nslabs,
                                float* partial, float* norms, const float* lr, float momentum,
                                float clipnorm, float grad_scale, int* skip_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BDETR_H */
