/* utv2.h - C ABI of the MI355X (gfx950) kernels behind the Unbiased-Teacher-v2 training step.
 *
 * The reference (facebookresearch/unbiased-teacher-v2) is pure Python and has no FFI of its own;
 * every entry point below replaces the native op the reference reaches through PyTorch /
 * Detectron2 / fvcore / torchvision at the cited call site (paths relative to the reference
 * root).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller unless the name ends in _host;
 *     tensors are contiguous fp32 / int32 / int64 / uint8 as declared;
 *   - activations are NHWC ([N][H][W][C]); conv weights are [Cout][KH][KW][Cin];
 *   - the last argument is the hipStream_t to launch on; nothing synchronises the stream,
 *     nothing allocates; scratch memory is passed in (size from the *_workspace_* twin);
 *   - return 0 on success, -(hipError_t) on a launch failure, -1000 on a bad argument.
 *   - re-entrant; no mutable global state (the A/B environment switches and two one-time kernel attributes are read / set once per
 *     process, thread-safely, and never change a result).
 */
#ifndef UTV2_H
#define UTV2_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* utv2_stream_t; /* == hipStream_t */

/* ---- convolution (implicit GEMM on v_mfma_f32_32x32x2_f32) ------------------------------------
 * Replaces ATen conv2d fwd / dgrad / wgrad reached from D2 ResNet/FPN, fcos/fcos.py:252-304
 * (towers + prediction convs), backbone/fpn.py:21-22 (P6/P7) and the RCNN heads.
 * y = relu?(conv(x,w) * scale[co] + bias[co] + residual) (+ y when accumulate).
 * Kred = length of one weight row (KH*KW*C, or the 16-padded 208 for the C==4 stem image).
 * in_dil > 1: transposed gather (dgrad of a strided conv) - see ubteacher/hip.py:conv2d_dgrad. */
int utv2_conv2d_nhwc_fwd(const float* x, const float* w, float* y, const float* scale, const float* bias,
                         const float* residual, int N, int H, int W, int C, int K, int KH, int KW, int stride, int pad,
                         int in_dil, int OH, int OW, int relu, int accumulate, int Kred, utv2_stream_t stream);
/* D2 BasicStem conv1 on the zero-padded NHWC4 image (C == 4, weight rows padded to Kred % 16 == 0); y_dtype selects
 * the output element type (UTV2_F32 / UTV2_BF16, defined below) */
int utv2_conv2d_stem_fwd(const float* x, const float* w, void* y, int y_dtype, const float* scale, const float* bias, int N,
                         int H, int W, int K, int KH, int KW, int stride, int pad, int OH, int OW, int relu, int Kred,
                         utv2_stream_t stream);
int utv2_conv2d_wgrad_splits(int N, int OH, int OW, int K, int Kred);
int64_t utv2_conv2d_wgrad_workspace_floats(int N, int OH, int OW, int K, int Kred);
/* dw[K][KH*KW*C] (+)= sum_m dy[m][k] * im2col(x)[m][:]; deterministic split reduction through ws. */
int utv2_conv2d_nhwc_wgrad(const float* x, const float* dy, float* dw, float* ws, int N, int H, int W, int C, int K,
                           int KH, int KW, int stride, int pad, int OH, int OW, int accumulate, utv2_stream_t stream);
/* db[C] (+)= column sums of g[M][C] (conv bias gradient).  ws >= 1024*C floats. */
int utv2_colsum(const float* g, float* db, float* ws, int M, int C, int accumulate, utv2_stream_t stream);
/* wt[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci]  (weight image consumed by dgrad) */
int utv2_weight_flip_transpose(const float* w, float* wt, int K, int KH, int KW, int C, utv2_stream_t stream);

/* Multi-level "same" conv on a level-first [sum_l N*H_l*W_l][C] buffer: ONE launch for every FPN level of a
 * shared head (fcos/fcos.py:338-376 loops the towers over the levels; D2 StandardRPNHead likewise).
 * Doubles as its own dgrad (flipped/transposed weights, pad = k-1-pad).  H_host/W_host: host int[nlev]. */
int utv2_conv2d_ml_fwd(const float* x, const float* w, float* y, const float* scale, const float* bias,
                       const float* residual, int nlev, const int* H_host, const int* W_host, int N, int C, int K, int KH,
                       int KW, int pad, int relu, int accumulate, utv2_stream_t stream);
int utv2_conv2d_ml_wgrad(const float* x, const float* dy, float* dw, float* ws, int nlev, const int* H_host,
                         const int* W_host, int N, int C, int K, int KH, int KW, int pad, int accumulate,
                         utv2_stream_t stream);

/* ---- mixed precision (the reference's SOLVER.AMP.ENABLED configs: autocast at engine/trainer.py:194-198,318-349):
 * bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 accumulate, weights from a bf16 mirror of the fp32 master arena.
 * Activations and activation gradients are bf16 in HBM (autocast stores conv outputs in 16 bit too); every tensor
 * argument carries its element type (`*_dtype`: UTV2_F32 or UTV2_BF16) so fp32 tensors - the loss-side head outputs,
 * their gradients, RoIAlign output - enter and leave without a cast pass.  `residual` has y's type.
 * Same contracts as the fp32 entry points otherwise. */
#define UTV2_F32 0
#define UTV2_BF16 1   /* the library's 16-bit float type: bfloat16 in libutv2_hip.so, IEEE fp16 in libutv2_hip_f16.so (the same sources built
                       * with -DUTV2_H16=_Float16; csrc/common.h h16_t) - "bf16" in the names below reads "the 16-bit type" there */
int utv2_conv2d_bf16_supported(int C, int KH, int KW);

/* Measurement aid (bench.py roofline.sustained_clock_ghz; synchronises the device, never on the training path): the shader clock in GHz
 * that workgroup 0 of the LAST persistent-grid multi-level launch (FCOS tower / RPN head) of the 256 x 256 forward / dgrad conv tile ran at (s_memtime ticks per 10 ns
 * s_memrealtime tick over that workgroup's lifetime) and the lifetime in microseconds; 0 / 0 before the first such launch.  The MFMA peak
 * is quoted at 2.4 GHz; under full-chip MFMA load the board's power limit holds the clock at 1.45-1.9 GHz.  No reference counterpart. */
int utv2_conv_clock_probe(double* ghz, double* lifetime_us);
/* mask (optional, y's type and shape): y = mask > 0 ? conv*scale+bias : 0, before the residual add - the ReLU backward of the
 * layer that produced the input, fused into the dgrad launch that computes its gradient; post_mask (optional, same type and shape):
 * y = post_mask > 0 ? y : 0 AFTER the residual add - the ReLU whose OUTPUT this gradient flows into (a bottleneck's input), so the
 * block that produced that output needs no separate mask pass */
int utv2_conv2d_nhwc_fwd_bf16(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                              const float* bias, const void* residual, const void* mask, const void* post_mask, int N, int H, int W, int C, int K,
                              int KH, int KW, int stride, int pad, int in_dil, int OH, int OW, int relu, int accumulate,
                              utv2_stream_t stream);
/* the same with the per-output-pixel geometry table of utv2_conv2d_wgrad_bf16 (rowinfo, optional; in_dil == 1): the tile prologues of
 * the bf16-input kernels load their rows' geometry instead of decoding it (2 us of a 70 us tile on the 256-tile kernel) */
int utv2_conv2d_nhwc_fwd_bf16_ri(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                                 const float* bias, const void* residual, const void* mask, const void* post_mask, int N, int H, int W, int C,
                                 int K, int KH, int KW, int stride, int pad, int in_dil, int OH, int OW, int relu, int accumulate,
                                 const int* rowinfo, utv2_stream_t stream);
/* the same with ReLU masks as BIT planes (16-bit y, K % 8 == 0): uint8 [N*OH*OW][K / 8], bit q of byte c = channel 8c + q.
 * relu_bits (optional, WRITTEN): bit = the stored output is > 0 - all the backward of the ReLU needs of it; mask_bits / post_mask_bits
 * (optional, read) take the place of mask / post_mask (one form per mask).  The dgrads of a bottleneck (engine/trainer.py's backward
 * through D2 BottleneckBlock) then read K / 8 bytes per pixel for a sign instead of re-reading 2 K bytes of forward activation. */
int utv2_conv2d_nhwc_fwd_bf16_bits(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                                   const float* bias, const void* residual, const void* mask, const void* post_mask, int N, int H, int W,
                                   int C, int K, int KH, int KW, int stride, int pad, int in_dil, int OH, int OW, int relu, int accumulate,
                                   const int* rowinfo, void* relu_bits, const void* mask_bits, const void* post_mask_bits,
                                   utv2_stream_t stream);
int utv2_conv2d_ml_fwd_bf16(const void* x, int x_dtype, const void* w16, void* y, int y_dtype, const float* scale,
                            const float* bias, const void* residual, int nlev, const int* H_host, const int* W_host, int N,
                            int C, int K, int KH, int KW, int pad, int relu, int accumulate, utv2_stream_t stream);
/* The same over column slices and / or GROUPED (the paired FCOS towers - cls | bbox, two independent 256 -> 256 chains of
 * fcos/fcos.py:252-304 - run as ONE launch per depth): x has row pitch x_pitch elements and group g reads its channels
 * [g*C, (g+1)*C) (C = input channels PER GROUP), w16 = bf16 [K][KH*KW*C], y / residual have row pitch y_pitch >= K.
 * Anything but (groups 1, x_pitch C, y_pitch K) needs bf16 x, C % 32 == 0, K % 4 == 0, pitches % 8 == 0, (K / groups) % 128 == 0.
 * gn_part (optional; bf16 y, K % 8 == 0): fp32 [ceil(P / 32)][K / 8][2] - per 32-row block and 8-channel group the sum and the sum of
 * squares of y as stored: the statistics pass of the GroupNorm that consumes y (fcos/fcos.py:263-264), see ..._seg_fwd_p32.
 * rowinfo (optional): device int32[P][2], the geometry table of utv2_conv2d_wgrad_bf16 for this conv (same k and pad). */
int utv2_conv2d_ml_fwd_bf16_g(const void* x, int x_dtype, int x_pitch, const void* w16, void* y, int y_dtype, int y_pitch,
                              const float* scale, const float* bias, const void* residual, int nlev, const int* H_host,
                              const int* W_host, int N, int C, int K, int KH, int KW, int pad, int relu, int accumulate, int groups,
                              float* gn_part, const int* rowinfo, utv2_stream_t stream);
/* The dgrad that produces the gradient of a GroupNorm + ReLU OUTPUT (the FCOS towers' conv -> GN -> ReLU chain, fcos/fcos.py:252-304,
 * in backward): a 16-bit level-first conv as ..._g with dense y (pitch K) whose epilogue applies the ReLU mask - mask_bits: [P * K / 8]
 * bytes, the plane utv2_groupnorm_relu_seg_fwd_p32b wrote - and leaves GroupNorm backward's first reduction, taken while the rows are in
 * registers: gnb_part fp32 [ceil(P / 64)][K][2] = {sum y, sum y * gnb_x} per 64-row block and channel over the rows as stored
 * (gnb_x: [P][K], the GroupNorm's input, 16-bit).  utv2_groupnorm_seg_bwd_p64 finishes the backward from them.
 * C % 64 == 0, KH * KW * C >= 1024, (K / groups) % 128 == 0. */
int utv2_conv2d_ml_fwd_bf16_gnb(const void* x, int x_pitch, const void* w16, void* y, int nlev, const int* H_host, const int* W_host, int N,
                                int C, int K, int KH, int KW, int pad, int groups, const int* rowinfo, const void* mask_bits,
                                const void* gnb_x, float* gnb_part, utv2_stream_t stream);
/* bf16 wgrad (+ fused bias gradient); rowinfo = device int32[M][2] per OUTPUT pixel:
 * {input pixel index of tap (0,0), (W << 16) | mask of the taps that fall inside the image}; KH*KW <= 16 */
int utv2_conv2d_wgrad_bf16_splits(int M, int K, int Kred);
int64_t utv2_conv2d_wgrad_bf16_workspace_floats(int M, int K, int Kred);
/* rowscale (optional, [K]): per-output-channel multiplier of the result (the folded FrozenBN scale: the kernels then
 * consume the gradient of the BN OUTPUT directly); the dgrad weight image takes the same multiplier at flip time */
int utv2_conv2d_wgrad_bf16(const void* x, int x_dtype, const void* dy, int dy_dtype, float* dw, float* db, float* ws,
                           const int* rowinfo, const float* rowscale, int M, int C, int K, int KH, int KW, int accumulate,
                           utv2_stream_t stream);
/* x with pixel pitch x_pitch (a channel slice of a wider matrix), dy with row pitch dy_pitch >= K (the first K columns of a
 * zero-padded matrix) and / or grouped: dw rows [g*K/groups, (g+1)*K/groups) correlate dy with input channels [g*C, (g+1)*C)
 * (C per group; dw = [K][KH*KW*C]); (K / groups) % 128 == 0, x_pitch >= groups * C */
int utv2_conv2d_wgrad_bf16_g(const void* x, int x_dtype, int x_pitch, const void* dy, int dy_dtype, int dy_pitch, float* dw, float* db,
                             float* ws, const int* rowinfo, const float* rowscale, int M, int C, int K, int KH, int KW, int accumulate,
                             int groups, utv2_stream_t stream);
/* The same launch with its split-K tail (slab reduction, bias reduction) RECORDED in a caller-owned pending table - host memory of
 * utv2_wgrad_fold_table_bytes() bytes, zero-initialised - instead of launched: utv2_wgrad_fold_flush runs the recorded tails of up to 8
 * launches as ONE kernel, with the arithmetic of the separate kernels (bit-identical gradients).  dw / db are valid after the flush on the
 * same stream; every recorded launch needs its own ws until then; two recorded launches must not share dw or db (EARG: flush first).
 * utv2_wgrad_fold_pending: recorded tails (0 = nothing to flush).  The per-layer tails of a backward are 60-70 dispatches of 5-25 us. */
int utv2_conv2d_wgrad_bf16_d(const void* x, int x_dtype, int x_pitch, const void* dy, int dy_dtype, int dy_pitch, float* dw, float* db,
                             float* ws, const int* rowinfo, const float* rowscale, int M, int C, int K, int KH, int KW, int accumulate,
                             int groups, void* pending, utv2_stream_t stream);
int64_t utv2_wgrad_fold_table_bytes(void);
int utv2_wgrad_fold_pending(const void* pending);
int utv2_wgrad_fold_flush(void* pending, utv2_stream_t stream);
/* D2 BasicStem conv1 on bf16 MFMA: xpad16 = bf16 [N][H+6][W+8][4], the normalised NHWC4 image inside a zero border
 * (written by utv2_preprocess_image_bf16pad); w16s = bf16 [K][7][32] (7 taps x 4 channels + 4 zeros per kernel row) */
int utv2_conv2d_stem_fwd_bf16(const void* xpad16, const void* w16s, void* y, int y_dtype, const float* scale,
                              const float* bias, int N, int H, int W, int K, int OH, int OW, int relu, utv2_stream_t stream);
/* the geometry table above, built on the device: out = int32 [N*OH*OW][2] for an [N, OH, OW] output over an [N, H, W] input whose first
 * pixel has index `start` (level-concatenated inputs: one call per level with the running pixel offset); replaces a host-side build + copy
 * per new canvas (Detectron2 ImageList.from_tensors pads every batch to its own canvas: reference data/dataset_mapper.py, INPUT.MIN_SIZE_TRAIN) */
int utv2_rowinfo_nhwc(int* out, int N, int H, int W, int OH, int OW, int stride, int pad, int KH, int KW, int64_t start,
                      utv2_stream_t stream);
int utv2_f32_to_bf16(const float* src, void* dst16, int64_t n, utv2_stream_t stream);
int utv2_weight_flip_transpose_bf16(const float* w, void* wt16, const float* scale, int K, int KH, int KW, int C,
                                    utv2_stream_t stream);
/* dst16 bf16 [rows][cpad] = src [rows][c] (UTV2_F32 / UTV2_BF16) rounded to nearest even, columns [c, cpad) zero; c, cpad % 8 == 0.
 * The output gradient of a layer whose channel count is no multiple of 32 (the 80-channel FCOS prediction convs, fcos.py:283-307),
 * padded for the LDS-DMA dgrad kernel. */
int utv2_pad_cols_bf16(const void* src, int src_dtype, void* dst16, int64_t rows, int c, int cpad, utv2_stream_t stream);
/* every dgrad weight image of a model in one launch; table = device array of nlayers 48-byte records
 * {int64 w_off (elements into arena), int64 dst_off (into bank), int64 scale_off (into scales, -1 = none), int32 K, KH, KW, C,
 *  int32 Kpad (>= K: pitch of the image's output-channel axis, channels [K, Kpad) zero), int32 0} */
int utv2_weight_flip_transpose_bf16_batched(const float* arena, const float* scales, void* bank, const void* table, int nlayers,
                                            utv2_stream_t stream);

/* ---- teacher EMA: engine/trainer.py:468-486 (FCOS), :950-968 (RCNN) --------------------------
 * teacher = student*(1-keep) + teacher*keep, evaluated with the reference's three roundings. */
int utv2_ema_axpby(float* teacher, const float* student, int64_t n, double keep_rate, utv2_stream_t stream);
/* ---- SGD+momentum+weight-decay on a flat arena: torch.optim.SGD via D2 build_optimizer
 * (engine/trainer.py:50,625; optimizer.step at :425-429,:912). */
int utv2_sgd_momentum(float* param, float* grad, float* mom_buf, int64_t n, float lr, float momentum, float weight_decay,
                      float grad_scale, int zero_grad, utv2_stream_t stream);
/* Dynamic loss scaling of the fp16 AMP mode with torch.cuda.amp.GradScaler's semantics (reference engine/trainer.py:207,424-426),
 * on the device: state = fp32 {loss scale, found_inf flag, growth tracker}.  utv2_amp_found_inf sets the flag when the (scaled)
 * gradient arena holds a non-finite value; utv2_sgd_momentum_amp applies the SGD step to grad / scale (* grad_scale), or nothing at
 * all when the flag is set; utv2_amp_update_scale: flag ? scale *= backoff : (every growth_interval clean steps scale *= growth),
 * then clears the flag. */
int utv2_amp_found_inf(const float* grad, int64_t n, float* state, utv2_stream_t stream);
int utv2_sgd_momentum_amp(float* param, const float* grad, float* mom_buf, int64_t n, float lr, float momentum, float weight_decay,
                          float grad_scale, const float* state, utv2_stream_t stream);
int utv2_amp_update_scale(float* state, float growth_factor, float backoff_factor, int growth_interval, utv2_stream_t stream);
/* the three updates above that ALSO write the 16-bit mirror of the arena range they update (mirror16[i] = the library's 16-bit rounding of
 * the new value; the copy the mixed-precision convs read): the weights are not re-read by a conversion pass before the next forward */
int utv2_ema_axpby_m16(float* teacher, const float* student, void* mirror16, int64_t n, double keep_rate, utv2_stream_t stream);
int utv2_sgd_momentum_m16(float* param, float* grad, float* mom_buf, void* mirror16, int64_t n, float lr, float momentum, float weight_decay,
                          float grad_scale, int zero_grad, utv2_stream_t stream);
int utv2_sgd_momentum_amp_m16(float* param, const float* grad, float* mom_buf, void* mirror16, int64_t n, float lr, float momentum,
                              float weight_decay, float grad_scale, const float* state, utv2_stream_t stream);

/* ---- elementwise pieces of ResNet / FPN ([D2-recall]) ------------------ */
int utv2_relu_bwd_scale(const void* dy, const void* y, const float* scale, void* out, int64_t M, int C, int dtype,
                        utv2_stream_t stream);
int utv2_add(const float* a, const float* b, float* out, int64_t n, utv2_stream_t stream);
int utv2_maxpool3x3s2_nhwc(const void* x, int x_dtype, void* y, int y_dtype, int N, int H, int W, int C, int OH, int OW,
                           utv2_stream_t stream);
/* its backward (a trainable stem, MODEL.BACKBONE.FREEZE_AT < 1; ATen max_pool2d_with_indices_backward): dx = the sum of dpool over the
 * windows whose FIRST maximum (kh, kw scan order) the pixel is; relu != 0: times (x > 0), the ReLU in front of the pool.  x = the
 * pool's input (x_dtype); dpool, dx: g_dtype (f32 / f32, 16-bit / 16-bit, 16-bit x with f32 gradients) */
int utv2_maxpool3x3s2_bwd_nhwc(const void* x, int x_dtype, const void* dpool, void* dx, int g_dtype, int N, int H, int W, int C, int OH,
                               int OW, int relu, utv2_stream_t stream);
int utv2_upsample2x_add_nhwc(const void* lateral, const void* top, void* out, int N, int H, int W, int C, int dtype,
                             utv2_stream_t stream);
int utv2_downsample2x_sum_nhwc(const void* g, void* dtop, int N, int TH, int TW, int C, int accumulate, int dtype,
                               utv2_stream_t stream);
/* dgrad of a stride-2 1x1 conv (D2 STRIDE_IN_1X1 bottlenecks): dst[n,2i,2j,:] = src[n,i,j,:], zeros elsewhere;
 * src is [N][(H+1)/2][(W+1)/2][C] - the compact gradient comes from a plain GEMM instead of a 4x larger masked one */
int utv2_zero_interleave2x_nhwc(const void* src, const void* mask, void* dst, int N, int H, int W, int C, int dtype, utv2_stream_t stream);
/* the same on 16-bit tensors (C % 8 == 0) with the fusions of the stride-2 bottlenecks' backward: dst = add + (even pixel ? masked src : 0);
 * the mask as a 16-bit tensor or as a bit plane (utv2_conv2d_nhwc_fwd_bf16_bits); `add` = the gradient another consumer of the same
 * activation produced (already masked by its producer) - no elementwise add pass.  mask, mask_bits, add optional. */
int utv2_zero_interleave2x_add_nhwc(const void* src, const void* mask, const void* mask_bits, const void* add, void* dst, int N, int H, int W,
                                    int C, utv2_stream_t stream);
/* modeling/one_stage_detector.py:88-90 / meta_arch/rcnn.py:18 (preprocess_image + ImageList pad) */
int utv2_preprocess_image(const void* src, int is_u8, float* dst, int H, int W, int Hp, int Wp,
                          const float* mean3_host, const float* std3_host, utv2_stream_t stream);
/* same, as bf16 at pixel offset (3,3) of one [Hp+6][Wp+8][4] slot of the zero-bordered bf16 stem input */
int utv2_preprocess_image_bf16pad(const void* src, int is_u8, void* dst16, int H, int W, int Hp, int Wp,
                                  const float* mean3_host, const float* std3_host, utv2_stream_t stream);
/* the whole batch in one launch: src_host = host array of N <= 32 device pointers ([3][H_i][W_i] images, all uint8 or all fp32),
 * dst16 = the N slots bf16 [N][Hp+6][Wp+8][4] */
int utv2_preprocess_images_bf16pad(const void* const* src_host, int is_u8, void* dst16, const int* H_host, const int* W_host, int N,
                                   int Hp, int Wp, const float* mean3_host, const float* std3_host, utv2_stream_t stream);
/* D2 FrozenBatchNorm2d.forward: scale = w*rsqrt(var+eps), shift = b - mean*scale, all layers at once */
int utv2_frozenbn_fold(const float* w, const float* b, const float* mean, const float* var, float* scale, float* shift,
                       int n, float eps, utv2_stream_t stream);
/* GroupNorm(32)+ReLU of the FCOS towers: fcos/fcos.py:263-264,283.  The _seg forms normalise every
 * (image, FPN level) segment of a level-first [rows][C] buffer in one launch (seg_rows_host: host int[nseg]). */
int64_t utv2_groupnorm_seg_workspace_floats(int nseg, const int* seg_rows_host, int C);
int utv2_groupnorm_relu_seg_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                float* ws, int nseg, const int* seg_rows_host, int C, int G, float eps, int relu, int dtype,
                                utv2_stream_t stream);
/* the same for bf16 x with 8 channels per group when the conv that produced x left the statistics partials in part32
 * (utv2_conv2d_ml_fwd_bf16_g gn_part: fp32 [ceil(rows / 32)][G][2]): one tensor pass less */
int utv2_groupnorm_relu_seg_fwd_p32(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                    const float* part32, int nseg, const int* seg_rows_host, int C, int G, float eps, int relu,
                                    utv2_stream_t stream);
/* ... that also writes the ReLU mask as a bit plane: relu_bits (optional; C % 32 == 0) [rows * C / 8] bytes, bit (row * C + c) = y > 0 in
 * fp32, before the 16-bit rounding (the reference's ReLU follows an fp32 GroupNorm under autocast): the mask ..._seg_bwd recomputes from x */
int utv2_groupnorm_relu_seg_fwd_p32b(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                                     const float* part32, int nseg, const int* seg_rows_host, int C, int G, float eps, int relu,
                                     void* relu_bits, utv2_stream_t stream);
/* beta (optional): the ReLU mask is recomputed from x with the forward expression instead of read from y (y may be null) */
int utv2_groupnorm_relu_seg_bwd(const void* dy, const void* y, const void* x, const float* mean, const float* rstd,
                                const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, float* ws,
                                int nseg, const int* seg_rows_host, int C, int G, int relu, int dtype, utv2_stream_t stream);
/* the same with colsum_part (optional, fp32 [utv2_groupnorm_seg_chunks(...)][C]) = per-chunk column sums of dx as stored: summed over
 * the chunks (utv2_colsum) they are the bias gradient of the convolution in front of the GroupNorm (fcos.py:263: conv -> GN -> ReLU), whose
 * dY is this dx - no separate pass over dY */
int64_t utv2_groupnorm_seg_chunks(int nseg, const int* seg_rows_host);
int utv2_groupnorm_relu_seg_bwd_colsum(const void* dy, const void* y, const void* x, const float* mean, const float* rstd,
                                       const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, float* ws,
                                       int nseg, const int* seg_rows_host, int C, int G, int relu, int dtype, float* colsum_part,
                                       utv2_stream_t stream);
/* GroupNorm (+ ReLU) backward from the partial sums of utv2_conv2d_ml_fwd_bf16_gnb: g = the incoming gradient with the ReLU mask already
 * applied (16-bit [rows][C]), part64 = that conv's gnb_part.  dx / dgamma / dbeta / colsum_part / ws as ..._seg_bwd_colsum. */
int utv2_groupnorm_seg_bwd_p64(const void* g, const void* x, const float* mean, const float* rstd, const float* gamma, void* dx,
                               float* dgamma, float* dbeta, float* ws, int nseg, const int* seg_rows_host, int C, int G,
                               const float* part64, float* colsum_part, utv2_stream_t stream);
/* db[c] (+)= sum_b partial[b][c], partial fp32 [nb][K] (the colsum_part above; fixed summation order) */
int utv2_colsum_partials(const float* partial, float* db, int nb, int K, int accumulate, utv2_stream_t stream);
int64_t utv2_groupnorm_workspace_floats(int N, int HW, int C);
int utv2_groupnorm_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                            float* ws, int N, int HW, int C, int G, float eps, int relu, utv2_stream_t stream);
int utv2_groupnorm_relu_bwd(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                            const float* gamma, float* dx, float* dgamma, float* dbeta, float* ws, int N, int HW, int C,
                            int G, int relu, utv2_stream_t stream);

/* ---- FCOS targets / losses / decode: modeling/fcos/fcos_outputs.py --------------------------- */
/* :649-698,:772-906; center_radius > 0 = CENTER_SAMPLE with POS_RADIUS (get_sample_region :700-770), 0 = plain in-box test.
 * drop_empty: bit 0 = images without gt drop all their locations (label -1; keep_locations of :804-815,310-311), bit 1 = ignore_near
 * (:841-848): a location inside some box but inside no box's centre-sampling region is dropped.
 * H,W,strides,soi are HOST arrays.  img_active (optional, device uint8[N]):
 * images flagged 0 get label -1 everywhere (ignored by the loss kernels) - the two student passes of one iteration
 * (trainer.py:396-411) run as one batch and each loss branch sees only its own images. */
int utv2_fcos_targets(int num_levels, const int* H_host, const int* W_host, const int* strides_host,
                      const float* soi_host, int N, int MAXG, const float* gt_boxes, const int* gt_classes,
                      const unsigned char* gt_valid, const float* gt_std, int num_classes, int drop_empty,
                      float center_radius, const unsigned char* img_active, int* labels, float* reg_targets, float* bvars,
                      int* gt_inds, utv2_stream_t stream);
/* the same for a batch of N images of which only [gt_img0, gt_img0 + gt_imgs) carry this loss branch's ground truth (the gt arrays
 * hold gt_imgs images); every other image gets label -1 - no padded copies of the gt arrays, no activity mask */
int utv2_fcos_targets_range(int num_levels, const int* H_host, const int* W_host, const int* strides_host, const float* soi_host, int N,
                            int MAXG, const float* gt_boxes, const int* gt_classes, const unsigned char* gt_valid, const float* gt_std,
                            int gt_img0, int gt_imgs, int num_classes, int drop_empty, float center_radius,
                            const unsigned char* img_active, int* labels, float* reg_targets, float* bvars, int* gt_inds,
                            utv2_stream_t stream);
/* fvcore sigmoid_focal_loss_jit at :329-338,:619-628 with on-the-fly one-hot.  ws >= 1024 floats */
int utv2_sigmoid_focal_fwd(const float* logits, const int* labels, int64_t P, int C, float alpha, float gamma,
                           float* loss_sum, float* ws, utv2_stream_t stream);
int utv2_sigmoid_focal_bwd(const float* logits, const int* labels, int64_t P, int C, float alpha, float gamma,
                           const float* coef, float* dlogits, utv2_stream_t stream);
/* :340-416 / :514-590 fused over positive locations: Integral(:44-77), centerness target + BCE,
 * GIoU (layers/iou_loss.py:26-76), NLL (layers/kl_loss.py:69-105), TS-better L1 (:552-569).
 * sums[8] = {n_pos, sum ctr_t, sum bce, sum giou*ctr_t, sum nll*iou, n_sel, sum_sel|d-t|, 0}
 * ws >= 4096 floats. */
int utv2_fcos_loc_terms_fwd(const int* labels, const float* box, int box_stride, const float* reg_targets,
                            const float* bvars, int64_t P, int num_classes, int reg_max, float ts_better, float ts_cert,
                            int flags, float* sums, float* ws, utv2_stream_t stream);
int utv2_fcos_loc_terms_bwd(const int* labels, const float* box, int box_stride, const float* reg_targets,
                            const float* bvars, int64_t P, int num_classes, int reg_max, float ts_better, float ts_cert,
                            int flags, const float* coef, float* dbox, utv2_stream_t stream);
/* The backward launches of SEVERAL loss branches of one fused student pass into ONE gradient tensor per head output (the supervised,
 * pseudo-classification and pseudo-regression target sets of engine/trainer.py:396-411 all differentiate the same logits / box
 * buffers): coef = the d total / d sum vector of utv2_fcos_loss_combine for this branch (focal: 1 float; loc terms: the 8 floats of
 * d total / d sums[8], read at [2], [3], [4], [6]), gscale = optional device scalar multiplied in (the node's upstream gradient).
 * accumulate 0: writes every row (zeros where the branch has no gradient); 1: rows without a gradient (label < 0; for the location
 * terms also background) are left untouched, the others ADDED - no per-branch gradient tensors, no elementwise add passes. */
int utv2_sigmoid_focal_bwd_acc(const float* logits, const int* labels, int64_t P, int C, float alpha, float gamma, const float* coef,
                               const float* gscale, float* dlogits, int accumulate, utv2_stream_t stream);
int utv2_fcos_loc_terms_bwd_acc(const int* labels, const float* box, int box_stride, const float* reg_targets, const float* bvars,
                                int64_t P, int num_classes, int reg_max, float ts_better, float ts_cert, int flags, const float* coef8,
                                const float* gscale, float* dbox, int accumulate, utv2_stream_t stream);
/* scalar tail of the FCOS losses of a fused student pass (normalisation fcos_outputs.py:317-321,361-362,381-416; pseudo branch
 * :504-585; loss weighting engine/trainer.py:396-417): the raw sums of utv2_sigmoid_focal_fwd / utv2_fcos_loc_terms_fwd of the
 * supervised, pseudo-cls and pseudo-reg target sets -> rec[8] = {cls, loc, ctr, cls_pseudo, ctr_pseudo, loc_pseudo,
 * teacher_better_student, weighted total} and coef[26] = d total / d {focal_sup[1], sums_sup[8], focal_cls[1], sums_cls[8], sums_reg[8]}.
 * norm: optional [6] all-reduced (n_pos, sum ctrness) per branch (NULL: the local sums); world = data-parallel world size;
 * flags 1 KL_LOSS, 2 KL type "klloss", 4 UNIFY_CTRCLS, 8 tsbetter; loss k (order cls, loc, ctr, cls_p, ctr_p, loc_p) enters the total
 * as value * wmul_host[k] / wdiv_host[k]. */
int utv2_fcos_loss_combine(const float* focal_sup, const float* sums_sup, const float* focal_cls, const float* sums_cls,
                           const float* sums_reg, const float* norm, float world, int flags, float kl_weight, const float* wmul_host,
                           const float* wdiv_host, float* rec, float* coef, utv2_stream_t stream);
/* :1146-1195 ranking score -> sortable int64 key (method 0 cls, 1 cls_n_ctr, 2 ctr, 3 cls_n_loc); image n's HW*C keys
 * start at keys + n*key_row_stride (>= HW*C: rows of a wider matrix shared by all FPN levels) */
int utv2_fcos_rank_keys(const float* logits, const float* box, int box_stride, int reg_max, int N, int HW, int C,
                        float thr, int method, long long* keys, int64_t key_row_stride, utv2_stream_t stream);
/* :1238-1241 `topk(pre_nms_top_n)` for every (image, level) row at once: exact MSD radix select over ragged rows
 * (row r = keys[row_off[r] .. row_off[r+1]), row_off device int64[rows+1]); out[rows][k] descending, -1 padded; k <= 8192 */
int64_t utv2_topk_rows_workspace_bytes(int rows, int k);
int utv2_topk_rows_i64(const long long* keys, const long long* row_off, int rows, int64_t max_width, int k, long long* out,
                       void* ws, utv2_stream_t stream);
/* :1093-1104,:1258-1296 decode of the selected candidates of one level into padded slots */
int utv2_fcos_decode(const long long* topkeys, int K, const float* logits, const float* box, int box_stride, int reg_max,
                     int N, int HW, int Wl, int C, int stride, int level, int method, int MAXC, int slot0, float* oboxes,
                     float* oscores, int* ocls, float* oloc, float* octr, float* oconf, float* ostd, int* olevel,
                     unsigned char* ovalid, utv2_stream_t stream);
/* Scale layer fcos/fcos.py:22-28,356-357 on the first ncols columns of strided rows */
int utv2_scale_cols(float* y, int64_t rows, int row_stride, int ncols, const float* s, utv2_stream_t stream);
int utv2_scale_cols_bwd(float* g, const float* ypost, int64_t rows, int row_stride, int ncols, const float* s, float* dsum,
                        float* ws, utv2_stream_t stream);
/* the Scale layers of ALL levels of a level-first matrix at once (one launch forward, two backward instead of four per level):
 * row0_host = host int64[nlev + 1] (first row of each level, then the end), s_host / sgrad_host = host arrays of nlev DEVICE pointers
 * (the per-level scalar and its gradient, accumulated: sgrad_l += sum(g_in * ypost) / s_l).  nlev <= 8; ws >= nlev * 1024 floats */
int utv2_scale_cols_ml(float* y, int nlev, const int64_t* row0_host, int row_stride, int ncols, const float* const* s_host,
                       utv2_stream_t stream);
int utv2_scale_cols_bwd_ml(float* g, const float* ypost, int nlev, const int64_t* row0_host, int row_stride, int ncols,
                           const float* const* s_host, float* const* sgrad_host, float* ws, utv2_stream_t stream);
/* the same backward, out of place, straight into the zero-padded 16-bit matrix the prediction conv's dgrad / weight gradient read:
 * out16 = 16-bit [rows][cpad] (cpad >= row_stride, % 4 == 0): [:, :ncols] = g * s_l, [:, ncols:row_stride] = g, [:, row_stride:] = 0; g is
 * not modified (no clone of the incoming gradient, no separate utv2_pad_cols_bf16 pass).  ws >= nlev * 1024 floats */
int utv2_scale_cols_bwd_ml_pad16(const float* g, const float* ypost, int nlev, const int64_t* row0_host, int row_stride, int ncols,
                                 const float* const* s_host, float* const* sgrad_host, float* ws, void* out16, int cpad, utv2_stream_t stream);

/* ---- NMS / IoU: layers/ml_nms.py:27, D2 batched_nms / pairwise_iou ---------------------------- */
int utv2_nms_mpad(int M);
int64_t utv2_nms_workspace_bytes(int N, int M);
int utv2_nms_batched(const float* boxes, const float* scores, const int* cls, const unsigned char* valid, int N, int M,
                     float iou_thr, int class_aware, int post_topk, int max_out, int* keep, int* keep_count, void* ws,
                     utv2_stream_t stream);
int utv2_box_iou(const float* a, const float* b, int A, int B, float* out, utv2_stream_t stream);

/* ---- Faster-RCNN path ----------------------------------------------------------------------- */
/* D2 pairwise_iou + Matcher (proposal_generator/rpn.py:112-148, roi_heads/roi_heads.py:213-231):
 * per candidate box the max IoU over the image's valid gts and its (first) argmax; optionally the
 * per-gt max over boxes (uint32 float bits, caller zero-fills) for the low-quality pass. */
int utv2_match_boxes(const float* boxes, int64_t box_img_stride, int N, int P, const float* gt_boxes,
                     const unsigned char* gt_valid, int G, float* max_iou, int* arg, unsigned* gt_max_bits,
                     utv2_stream_t stream);
int utv2_match_lowq(const float* boxes, int64_t box_img_stride, int N, int P, const float* gt_boxes,
                    const unsigned char* gt_valid, int G, const unsigned* gt_max_bits, unsigned char* lowq,
                    utv2_stream_t stream);
/* torchvision roi_align(aligned=True, sampling_ratio=0) through D2 ROIPooler level assignment
 * (roi_heads/roi_heads.py:28-45,118).  feats_host / dfeats_host: HOST arrays of device pointers. */
int utv2_roi_align_fwd(int num_levels, int min_level, const void* const* feats_host, const int* H_host,
                       const int* W_host, const float* scales_host, const float* rois, const int* roi_batch,
                       const unsigned char* roi_valid, int R, int C, int PH, int PW, void* out, int dtype,
                       utv2_stream_t stream);
/* features / out / dy are `dtype` (UTV2_F32 / UTV2_BF16); the gradient buffers dfeats stay fp32 (atomics) */
int utv2_roi_align_bwd(int num_levels, int min_level, float* const* dfeats_host, const int* H_host, const int* W_host,
                       const float* scales_host, const float* rois, const int* roi_batch, const unsigned char* roi_valid,
                       int R, int C, int PH, int PW, const void* dy, int dtype, utv2_stream_t stream);
/* the same gradient as a deterministic gather over 8 x 8 pixel tiles (no atomics; the backward of the ROIPooler call at
 * roi_heads/roi_heads.py:118): the ROIs of image n are rois[n*rois_per_image .. (n+1)*rois_per_image), C <= 256, PH, PW <= 7;
 * every element of every dfeats[l] ([N][H_l][W_l][C], element type out_dtype) is written - no zero-fill, no fp32 staging */
int utv2_roi_align_bwd_tiled(int num_levels, int min_level, void* const* dfeats_host, const int* H_host, const int* W_host,
                             const float* scales_host, const float* rois, const unsigned char* roi_valid, int N, int rois_per_image,
                             int C, int PH, int PW, const void* dy, int dy_dtype, int out_dtype, utv2_stream_t stream);
/* RPN proposal decoding, the part of detectron2's RPN.predict_proposals / find_top_rpn_proposals in front of the NMS that
 * proposal_generator/rpn.py:60-76 inherits.  head: level-first RPN head output, rows (level, image, pixel) x ch floats = A objectness
 * logits then 4A anchor-major deltas; hw_host[l] = pixels of level l.
 * rank_keys: keys[t] for every logit in memory order (rows x A), 63-bit, descending key order == (logit desc, anchor index asc) within
 * the logit's (level, image) row - the input of utv2_topk_rows_i64.
 * decode: top = [num_levels * N][maxk] selected keys (-1 = empty), k_host[l] candidates kept per image on level l; writes, per image,
 * the K = sum k_l candidates in level order: boxes [N][K][4] (apply_deltas with weights_host[4], clipped to image_hw[n] = (h, w)),
 * scores (0 where non-finite), lvls, keep (finite and both sides > min_size). */
int utv2_rpn_rank_keys(const float* head, int num_levels, const int* hw_host, int N, int A, int ch, int64_t* keys, utv2_stream_t stream);
int utv2_rpn_decode(const int64_t* top, int maxk, const float* head, const float* anchors, const float* image_hw, int num_levels,
                    const int* hw_host, const int* k_host, int N, int A, int ch, const float* weights_host, float scale_clamp,
                    float min_size, float* boxes, float* scores, int* lvls, unsigned char* keep, utv2_stream_t stream);
/* The predictor's inference (roi_heads/fast_rcnn.py:1094-1125,1162-1225; D2 fast_rcnn_inference) around the top-k and the NMS:
 * keys:   probs [N*P][K+1] (softmax), deltas [N*P][4], prop [N][P][4], valid [N][P], whwh [N][4] = (w, h, w, h) of the image ->
 *         boxes [N][P][4] = Box2BoxXYXYTransform.apply_deltas (weights wx, wy; clamp +-scale_clamp) clipped to the image, and
 *         keys [N][P*K]: per foreground class (order-preserving bits of its probability) << 32 | (2^32 - 1 - (p*K + c)), the key of
 *         -1 when prob <= thr / the slot is invalid / box or probabilities are not finite.  Descending key order = (prob desc, index asc).
 * gather: top [N][k] (the k largest keys of each image, descending) -> sc, rows (proposal index, int64), cls, cb [N][k][4], valid (sc > thr)
 * pack:   kidx [N][D] / cnt [N] of utv2_nms_batched on (cb, sc, cls, valid) -> the padded detections + stdl [N*P][4] of their rows */
/* the survivors of utv2_nms_batched (kidx [N][D], -1 padded; cnt [N]) gathered into padded outputs: oboxes [N][D][4], oscores [N][D],
 * ovalid [N][D] = slot < cnt (slots beyond the count repeat candidate 0) - the tail of D2 find_top_rpn_proposals in one launch */
int utv2_nms_pack(const int* kidx, const int* cnt, const float* boxes, const float* scores, int N, int M, int D, float* oboxes, float* oscores,
                  unsigned char* ovalid, utv2_stream_t stream);
int utv2_roi_infer_keys(const float* probs, const float* deltas, const float* prop, const unsigned char* valid, const float* whwh, int N, int P,
                        int K, float wx, float wy, float scale_clamp, float thr, float* boxes, int64_t* keys, utv2_stream_t stream);
int utv2_roi_infer_gather(const int64_t* top, const float* boxes, int N, int P, int K, int k, float thr, float* sc, int64_t* rows, int* cls,
                          float* cb, unsigned char* valid, utv2_stream_t stream);
int utv2_roi_infer_pack(const int* kidx, const int* cnt, const float* cb, const float* sc, const int* cls, const int64_t* rows, const float* stdl,
                        int N, int P, int k, int D, float* oboxes, float* oscores, int* ocls, float* ostd, int64_t* orows, unsigned char* ovalid,
                        utv2_stream_t stream);
/* PseudoLabRPN.losses on the sampled anchors (proposal_generator/rpn.py:153-225): sums[0] = sum of BCE-with-logits over the sampled
 * positives (pos_idx [N][npos], int64 anchor indices, pos_valid) and negatives (neg_idx [N][nneg]) - every term times the score of the
 * anchor's matched pseudo box when gt_scores is given, zero when the image has no gt (has_gt [N]) -, sums[1] = sum over the valid
 * positives of |delta - Box2BoxTransform.get_deltas(anchor, matched gt, weights_host[4])|; gobj [N][npos+nneg] / gdl [N][npos][4] =
 * the derivatives of the two sums.  head = 0: obj [N][R], deltas [N][R][4]; head = 1: obj = deltas = the level-first RPN head output
 * (utv2_rpn_decode's layout).  matched [N][R] int32 (utv2_match_boxes' argmax), gt_boxes [N][G][4], gt_scores [N][G] or NULL.
 * bwd: scatters gout_cls[0] * gobj and gout_loc[0] * gdl (device scalars) to the sampled anchors of zero-filled grad_obj / grad_deltas
 * (same layouts; head = 1: both = the head-output gradient). */
int utv2_rpn_loss_fwd(const float* obj, const float* deltas, int head, int num_levels, const int* hw_host, int N, int A, int ch, int R,
                      const float* anchors, const int64_t* pos_idx, const unsigned char* pos_valid, int npos, const int64_t* neg_idx,
                      const unsigned char* neg_valid, int nneg, const int* matched, const unsigned char* has_gt, const float* gt_boxes,
                      const float* gt_scores, int G, const float* weights_host, float* sums, float* gobj, float* gdl, utv2_stream_t stream);
int utv2_rpn_loss_bwd(const float* gobj, const float* gdl, const float* gout_cls, const float* gout_loc, int head, int num_levels,
                      const int* hw_host, int N, int A, int ch, int R, const int64_t* pos_idx, const unsigned char* pos_valid, int npos,
                      const int64_t* neg_idx, const unsigned char* neg_valid, int nneg, float* grad_obj, float* grad_deltas,
                      utv2_stream_t stream);
/* _range (head = 1 only): the head output holds `batch` images per level and the N images of this call - the sampler arrays, gt and
 * has_gt are theirs alone - are images [img0, img0 + N) of it: one loss branch of a student pass that ran the labeled and the
 * pseudo-labeled images (reference trainer.py:838-866) as one batch.  The plain entries are batch = N, img0 = 0. */
int utv2_rpn_loss_fwd_range(const float* obj, const float* deltas, int head, int num_levels, const int* hw_host, int N, int batch, int img0,
                            int A, int ch, int R, const float* anchors, const int64_t* pos_idx, const unsigned char* pos_valid, int npos,
                            const int64_t* neg_idx, const unsigned char* neg_valid, int nneg, const int* matched,
                            const unsigned char* has_gt, const float* gt_boxes, const float* gt_scores, int G, const float* weights_host,
                            float* sums, float* gobj, float* gdl, utv2_stream_t stream);
int utv2_rpn_loss_bwd_range(const float* gobj, const float* gdl, const float* gout_cls, const float* gout_loc, int head, int num_levels,
                            const int* hw_host, int N, int batch, int img0, int A, int ch, int R, const int64_t* pos_idx,
                            const unsigned char* pos_valid, int npos, const int64_t* neg_idx, const unsigned char* neg_valid, int nneg,
                            float* grad_obj, float* grad_deltas, utv2_stream_t stream);
/* Labelling + random subsampling of anchors / proposals (rpn.py:112-148 -> D2 label_and_sample_anchors + subsample_labels;
 * roi_heads.py:141-270 -> D2 Matcher + _sample_proposals).  Random choice = the k smallest of one uniform key in [0, 1) per slot (keys
 * from the caller); equal keys are taken in slot order.
 * rpn_sample_keys: max_iou / lowq [N][R] (utv2_match_boxes / utv2_match_lowq), labels: IoU < lo negative, >= hi or lowq positive, image
 * without gt all negative; out [2N][R]: row n = positives of image n, row N + n its negatives, -1 = not a candidate, else a sortable key
 * (input of utv2_topk_rows_i64 with k = max(npos_max, nneg_max)).  rpn_sample_unpack: top [2N][k] selected keys -> pos_idx [N][npos_max]
 * / neg_idx [N][nneg_max] (ascending key order) and their valid flags; negatives are valid up to nneg_max - #positives; has_gt [N].
 * roi_sample: boxes [N][P][4] (proposals ++ gt, P <= 4096), valid [N][P], max_iou / argmax [N][P] against gt_* [N][G]; foreground = matched
 * IoU >= iou_thr in an image with gt; per image <= nfg_max foreground then background up to `batch` slots: out_boxes [N][batch][4],
 * out_classes (int64: gt class, num_classes = background, -1 = empty slot), out_gt_boxes, out_valid, out_idx (slot in [0, P)), and - when
 * gt_scores / gt_std [N][G] / [N][G][4] and the outputs are given - the matched pseudo box's score and std logits. */
int utv2_rpn_sample_keys(const float* max_iou, const unsigned char* lowq, const unsigned char* gt_valid, int G, const float* keys, int N,
                         int R, float lo, float hi, int64_t* out, utv2_stream_t stream);
int utv2_rpn_sample_unpack(const int64_t* top, int k, int N, int npos_max, int nneg_max, const unsigned char* gt_valid, int G,
                           int64_t* pos_idx, unsigned char* pos_valid, int64_t* neg_idx, unsigned char* neg_valid, unsigned char* has_gt,
                           utv2_stream_t stream);
int utv2_roi_sample(const float* boxes, const unsigned char* valid, const float* max_iou, const int* argmax, const float* keys, int N, int P,
                    const float* gt_boxes, const int* gt_classes, const unsigned char* gt_valid, const float* gt_scores, const float* gt_std,
                    int G, float iou_thr, int num_classes, int batch, int nfg_max, float* out_boxes, int64_t* out_classes,
                    float* out_gt_boxes, unsigned char* out_valid, int64_t* out_idx, float* out_conf, float* out_std, utv2_stream_t stream);
/* box_reg_loss / box_reg_pseudo_loss of the boundary-variance predictor (roi_heads/fast_rcnn.py:938-1090) on R sampled ROIs, summed:
 * deltas / stdl = the predicted boundary deltas and std logits (row pitch ld floats), cls [R] int64 (-1 = empty slot, foreground =
 * [0, num_classes)), prop / gtb [R][4] proposal and matched gt boxes, gstd [R][4] the pseudo boxes' std logits or NULL.
 * mode 0: nlloss (L1 + 0.05 sum NLL * IoU, gradient through the IoU), 1: smooth_l1 at beta 0, 2: tsbetter, 3: pseudo smooth_l1.
 * Writes sum[0] and the derivatives gdeltas / gstd_out [R][4] (Box2BoxXYXYTransform weights wx, wy and clamp). */
int utv2_roi_box_loss(const float* deltas, const float* stdl, int64_t ld, const int64_t* cls, const float* prop, const float* gtb,
                      const float* gstd, int R, int num_classes, int mode, float wx, float wy, float scale_clamp, float ts_better,
                      float t_cert, float* sum, float* gdeltas, float* gstd_out, utv2_stream_t stream);
/* roi_heads/fast_rcnn.py:925-936 + FocalLoss :1405-1429 (softmax CE focal, gamma 1.5), summed */
int utv2_softmax_focal_fwd(const float* logits, const int* target, int R, int C, float gamma, float* loss_sum, float* ws,
                           utv2_stream_t stream);
int utv2_softmax_focal_bwd(const float* logits, const int* target, int R, int C, float gamma, const float* coef,
                           float* dlogits, utv2_stream_t stream);
/* The scalar tail of the Faster-RCNN UTv2 losses in one launch (roi_heads/fast_rcnn.py:925-936 normalisation by the number of sampled
 * ROIs, proposal_generator/rpn.py:214-224 normalisation + loss weights, engine/trainer.py:880-893 weighting and sum): raw kernel sums of
 * the supervised and the pseudo branch -> rec[9] = {loss_cls, loss_box_reg, loss_rpn_cls, loss_rpn_loc} x {supervised, pseudo}, total;
 * coef[8] = d total / d raw sum.  wt_host: host float[8], the trainer's weight per loss in rec order. */
int utv2_rcnn_loss_combine(const float* rpn_sup, const float* rpn_uns, const float* focal_sup, const float* focal_uns, const float* box_sup,
                           const float* box_uns, const int* tgt_sup, int n_sup, const int* tgt_uns, int n_uns, float rpn_norm_sup,
                           float rpn_norm_uns, float w_rpn_cls, float w_rpn_loc, float w_box, const float* wt_host, float* rec, float* coef,
                           utv2_stream_t stream);

/* ---- the frozen ResNet stem as ONE kernel (csrc/stem_pool.hip): D2 BasicStem conv1 (7x7 stride 2 pad 3, 3 -> 64, FrozenBN as scale /
 * shift, ReLU) + max_pool2d(3, 2, 1), the first two layers of the R-50 the reference builds (backbone/fpn.py:21-22).  xpad16 / w16s as
 * for utv2_conv2d_stem_fwd_bf16 (W even); y: 16-bit [N][PH][PW][64], OH = (H - 1) / 2 + 1, PH = (OH - 1) / 2 + 1 (same for W).  The conv
 * output never leaves LDS. */
int utv2_stem_pool_fwd_bf16(const void* xpad16, const void* w16s, void* y, const float* scale, const float* shift, int N, int H, int W,
                            int K, utv2_stream_t stream);

/* ---- one frozen ResNet bottleneck as ONE kernel (csrc/bottleneck.hip): D2 BottleneckBlock conv1 1x1 -> conv2 3x3 -> conv3 1x1 + identity
 * (or + the 1x1 shortcut conv of a stage's first block), FrozenBN folded to scale / shift, of the R-50 the reference builds through
 * build_fcos_resnet_fpn_backbone (ubteacher/modeling/backbone/fpn.py:21-22; res2 under MODEL.BACKBONE.FREEZE_AT 2).
 * x: [N, H, W, C], y: [N, H, W, 256] of the library's 16-bit type, y != x; w1 [MID = 64][C], w2 [MID][3][3][MID], w3 [256][MID] 16-bit;
 * s* / b*: fp32 per output channel.  wsc NULL (C = 256): y = relu(conv3(relu(conv2(relu(conv1(x) s1 + b1)) s2 + b2)) s3 + b3 + x);
 * wsc [256][C] (C = 64): ... + h16(shortcut(x) ssc + bsc) instead of + x.  The 64-channel intermediates never leave LDS. */
int utv2_bottleneck_supported(int C, int MID, int has_shortcut);
int utv2_bottleneck_fwd_bf16(const void* x, void* y, const void* w1, const void* w2, const void* w3, const void* wsc, const float* s1,
                             const float* b1, const float* s2, const float* b2, const float* s3, const float* b3, const float* ssc,
                             const float* bsc, int N, int H, int W, int C, int MID, utv2_stream_t stream);

/* ---- two-crop data path: the pixel arithmetic of the reference's weak / strong views, on uint8 [H][W][3] images
 * in HBM, bit-exact to Pillow (which Detectron2 / torchvision / the reference call on the CPU) ------------------------------------- */
/* detectron2 ResizeTransform.apply_image (used by ubteacher/data/dataset_mapper.py:97-99) = PIL.Image.resize(BILINEAR): Resample.c
 * two-pass 8.22 fixed-point convolution; flip != 0 also applies HFlipTransform.  ws >= utv2_aug_resize_workspace_bytes bytes */
int64_t utv2_aug_resize_workspace_bytes(int H, int W, int OH, int OW);
int utv2_aug_resize_bilinear_u8(const unsigned char* src, int H, int W, unsigned char* dst, int OH, int OW, int flip, void* ws,
                                utv2_stream_t stream);
/* ubteacher/data/detection_utils.py:20-23 ColorJitter on PIL images = ImageEnhance.{Brightness,Contrast,Color} = Blend.c ImagingBlend
 * against a degenerate image: mode 0 black, 1 the constant *mean (from utv2_aug_gray_mean_u8: int(mean(L) + 0.5)), 2 the L image */
int utv2_aug_gray_mean_u8(const unsigned char* img, int64_t npix, void* sum_ws, int* mean_out, utv2_stream_t stream);
int utv2_aug_blend_u8(unsigned char* img, int64_t npix, int mode, float alpha, const int* mean, utv2_stream_t stream);
/* torchvision F_pil.adjust_hue: Convert.c rgb2hsv, H += shift (uint8 wrap), hsv2rgb */
int utv2_aug_hue_u8(unsigned char* img, int64_t npix, int shift, utv2_stream_t stream);
/* detection_utils.py:24 RandomGrayscale: convert("L") replicated to 3 channels */
int utv2_aug_grayscale_u8(unsigned char* img, int64_t npix, utv2_stream_t stream);
/* data/transforms/augmentation_impl.py:7-22 GaussianBlur = PIL.ImageFilter.GaussianBlur = 3 + 3 passes of BoxBlur.c; one pass per call
 * (radius / ww / fw = the pass constants of ImagingHorizontalBoxBlur), src != dst */
int utv2_aug_box_blur_u8(const unsigned char* src, unsigned char* dst, int H, int W, int vertical, int radius, int ww, int fw,
                         utv2_stream_t stream);
/* detection_utils.py:27-41 ToTensor -> RandomErasing(value="random") -> ToPILImage: rectangle <- (noise[3][h][w] * 255).byte() */
int utv2_aug_erase_u8(unsigned char* img, int H, int W, int i, int j, int h, int w, const float* noise, utv2_stream_t stream);
/* dataset_mapper.py:139-147 image_strong_aug.transpose(2, 0, 1) */
int utv2_aug_hwc_to_chw_u8(const unsigned char* src, unsigned char* dst, int64_t npix, utv2_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
