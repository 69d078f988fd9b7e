/*
 * seghiero_hip.h -- C ABI of libseghiero_hip.so: the MI355X (gfx950) kernels of the SegHiero
 * training hot path.
 *
 * The reference (Shadowfear36/SegHiero) has no native layer: its "operator API" for this path is
 * the chain of ATen ops issued by its Python modules.  Each entry point below replaces one such
 * chain; the reference site is cited per function (paths relative to the reference root).
 *
 * Conventions
 *   - every tensor is fp32 (labels: uint8), activations are NHWC ("channels_last"), weights OHWI
 *     (= a PyTorch [O,I,KH,KW] tensor in channels_last memory format);
 *   - `ld*` arguments are pixel (row) strides in ELEMENTS, so a kernel can read/write a channel
 *     slice of a wider concat buffer;
 *   - `stream` is a hipStream_t passed as void*; all functions are asynchronous on it, hold no
 *     global state, never allocate, never synchronise; the caller owns every buffer;
 *   - return value: 0 = launched, SH_EINVAL = rejected arguments (nothing launched),
 *     SH_ELAUNCH = HIP reported a launch error.  Nothing throws across this boundary.
 */
#ifndef SEGHIERO_HIP_H
#define SEGHIERO_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SH_OK 0
#define SH_EINVAL (-1)
#define SH_ELAUNCH (-2)
#define SH_EUNSUPPORTED (-3)   /* fused entry points only: no fused instantiation for this geometry, nothing launched */

/* library / device info ------------------------------------------------------------------- */
int sh_abi_version(void);                       /* bumps when a signature changes */
int sh_conv_tile_rows(void);                    /* rows per stat-partial of sh_conv_fprop (=64) */

/* layout --------------------------------------------------------------------------------- */
/* NCHW -> NHWC with channel padding to Cpad (zeros).  Replaces the implicit layout of
 * `img_t` entering `self.stem_conv` (models/backbone/resnet.py:65). */
int sh_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, void* stream);
/* NHWC (Cpad channels per pixel, first C used) -> NCHW. */
int sh_nhwc_to_nchw(const float* x, float* y, int N, int C, int H, int W, int Cpad, void* stream);

/* On-device input pipeline (dataset/dataloader.py:49-63 after PIL decoding / PIL resizing, which stay
 * on the host): per-sample horizontal flip (flip[n] != 0; NULL = none), ToTensor (u8/255) and Normalize((v-mean)/std)
 * from interleaved RGB u8 [N,H,W,3] straight into the stem's NHWC4 fp32 layout (4th channel 0). */
int sh_ingest_image_u8(const uint8_t* rgb, float* out, const uint8_t* flip, int N, int H, int W,
                       const float* mean3_host, const float* std3_host, void* stream);
/* PIL's antialiasing `img.resize(size, Image.BILINEAR)` of dataset/dataloader.py:50 on interleaved RGB u8, bit-identical to Pillow
 * (src/libImaging/Resample.c: separable triangle filter stretched by the downscale factor, horizontal pass first, 22-bit fixed-point
 * coefficients, each pass rounded to u8).  sh_resize_bilinear_coeffs is HOST arithmetic (double, Pillow's operation order): returns
 * ksize and fills bounds[out][2] = (first tap, tap count) and kk[out][ksize]; NULL tables = size query.  The caller uploads the
 * tables (int32) and passes tmp = N*H*Wo*3 bytes when both axes change. */
int sh_resize_bilinear_coeffs(int in_size, int out_size, int* bounds, int* kk, int kk_capacity);
int sh_resize_bilinear_u8(const uint8_t* src, uint8_t* tmp, uint8_t* dst, int N, int H, int W, int Ho, int Wo,
                          const int* bounds_x, const int* kk_x, int ksize_x, const int* bounds_y, const int* kk_y,
                          int ksize_y, void* stream);
/* Label maps: F.interpolate(mode="nearest") from [N,Hs,Ws] to [N,H,W] (ATen index rule), then the same flip, to u8.
 * mask: int64 (is_i64 = 1, as the reference holds them) or uint8. */
int sh_ingest_mask(const void* mask, int is_i64, uint8_t* out, const uint8_t* flip, int N, int Hs, int Ws, int H,
                   int W, void* stream);

/* dense convolution as implicit GEMM on fp32 MFMA ----------------------------------------- */
/* y[n,oh,ow,co] = bias[co] + sum x[n, oh*s-p+kh*d, ow*s-p+kw*d, ci] * w[co,kh,kw,ci]
 * Replaces nn.Conv2d forward: torchvision Bottleneck convs behind models/backbone/resnet.py:65-73,
 * and every 1x1 conv of models/head/sep_aspp_contrast_head.py (:15-22, :51, :79, :95, :181, :189, :207).
 * Cin % 4 == 0.  stat_partials (optional): [ceil(M/64)][2][Cout] floats receiving, per 64-row tile and channel,
 * sum(y) and M2 = sum((y - tile_mean)^2)  (centred partials of the train-mode BatchNorm statistics). */
int sh_conv_fprop(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                  float* stat_partials, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                  int stride, int pad, int dil, void* stream);
/* dx = conv_transpose(dy, w) (+ addend).  Autograd's grad_input of the same nn.Conv2d.
 * mode 0: gather form over the input pixels.  mode 1 (1x1, pad 0 only): rows of the output grid are
 * scattered to input pixel (oh*s, ow*s) and ACCUMULATED into dx (dx must hold the other summand). */
int sh_conv_dgrad(const float* dy, int lddy, const float* w, const float* addend, int ldadd,
                  float* dx, int lddx, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                  int stride, int pad, int dil, int mode, void* stream);
/* dw[co,kh,kw,ci] = sum_pixels dy * x.  Autograd's grad_weight of the same nn.Conv2d.
 * workspace: sh_conv_wgrad_workspace(...) bytes (split-K slabs, reduced deterministically). */
int64_t sh_conv_wgrad_workspace(int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                int stride, int pad, int dil);
int sh_conv_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw, float* workspace,
                  int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                  int dil, void* stream);

/* The same three GEMMs on the bf16 matrix cores with an exact 3-way bf16 split of every fp32 operand and six
 * v_mfma_f32_32x32x16_bf16 products per term (fp32-class accuracy, 2.7x fewer MFMA cycles than the f32 MFMA; see
 * csrc/conv_bf16x6.hip).  Same arguments, except that dgrad takes the transposed weight copy made by
 * sh_weight_transpose ([KH*KW][Cin][pad4(Cout)], zero padded) and needs lddy >= pad4(Cout) with zeroed padding lanes. */
/* fprop / dgrad take an optional split-K workspace (sh_conv_x6_workspace bytes; NULL / 0 = never split): shapes whose
 * 128x128 tiling gives < 1.5 blocks per CU but have a long K (layer3/4 3x3, the ASPP bottleneck, the aux head) run as
 * S K-slices into fp32 slabs [S][M][N] plus a deterministic reduce that applies bias / addend / the BN statistics. */
int64_t sh_conv_x6_workspace(int which /* 0 fprop, 1 dgrad */, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                             int stride, int pad, int dil, int dgrad_mode);
int sh_conv_fprop_x6(const float* x, int ldx, const float* w, const float* bias, float* y, int ldy,
                     float* stat_partials, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                     int stride, int pad, int dil, float* workspace, int64_t workspace_bytes, int act_flags, void* stream);
/* Inference forward with eval-mode BatchNorm (+ Bottleneck residual add) (+ ReLU) fused into the epilogue:
 * out = [relu]( conv(x, w) * scale[co] + shift[co] [+ residual] ), scale / shift from sh_bn_eval_coefs.  Same operation
 * order as sh_conv_fprop_x6 followed by sh_bn_act, so the result is bit-identical while y is never written or re-read. */
int sh_conv_fprop_x6_act(const float* x, int ldx, const float* w, const float* scale, const float* shift,
                         const float* residual, int ldr, int relu, float* out, int ldo, int N, int H, int W,
                         int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, void* stream);
/* Training forward reading its input THROUGH the producer's BatchNorm + ReLU: x is the RAW output of the previous convolution
 * and the loader applies relu(x * in_scale[c] + in_shift[c]) (sh_bn_act's own operation order; zero padding stays zero), so the
 * activated tensor of every conv -> BN -> ReLU -> conv chain (models/backbone/resnet.py:65-73,
 * models/head/sep_aspp_contrast_head.py:56-61, 180-184) is never written or re-read.  in_scale / in_shift: [Cin], 16-byte aligned.
 * SH_EUNSUPPORTED = no fused instantiation for this geometry (run sh_bn_act, then sh_conv_fprop_x6). */
int sh_conv_fprop_x6_aff(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* w,
                         const float* bias, float* y, int ldy, float* stat_partials, int N, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad, int dil, float* workspace, int64_t workspace_bytes, int act_flags,
                         void* stream);
/* ngroups (<= 6) pointwise convolutions of ONE geometry in one launch -- the ASPP branches of
 * models/head/sep_aspp_contrast_head.py:100-131 (the 1x1 branch + the pointwise convs of the three depthwise-separable branches):
 * group g: y[:, g*Cout:(g+1)*Cout] = conv1x1(in_g, w[g]), in_g = relu(x[g]*in_scale[g] + in_shift[g]) (the producer's BatchNorm +
 * ReLU in the loader) or plain x[g] when in_scale[g] is NULL.  x / ldx / in_scale / in_shift / w: HOST arrays of ngroups device
 * pointers / strides.  y = first group's column, ldy >= ngroups*Cout; stat_partials [ceil(N*H*W/64)][2][ngroups*Cout].
 * Cout % 128 == 0 and Cin % 16 == 0, else SH_EUNSUPPORTED. */
int sh_conv1x1_grouped_fprop_x6(int ngroups, const float* const* x, const int* ldx, const float* const* in_scale,
                                const float* const* in_shift, const float* const* w, float* y, int ldy,
                                float* stat_partials, int N, int H, int W, int Cin, int Cout, void* stream);
int sh_weight_transpose(const float* w, float* wt, int Cout, int KH, int KW, int Cin, void* stream);
/* ... for n <= SH_WT_MAX weights in one launch (taps = KH*KW); the training step prepares all dgrad operands at once. */
#define SH_WT_MAX 40
int sh_weight_transpose_multi(int n, const float* const* w, float* const* wt, const int* cout, const int* taps,
                              const int* cin, void* stream);
int sh_conv_dgrad_x6(const float* dy, int lddy, const float* wt, const float* addend, int ldadd,
                     float* dx, int lddx, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                     int stride, int pad, int dil, int mode, float* workspace, int64_t workspace_bytes,
                     void* stream);
/* Input gradient + the front half of the BatchNorm backward of the layer that produced the conv's input (conv -> BN [-> ReLU]
 * -> this conv): stores g = relumask(y_prev*scale + shift) * (dx [+ addend]) instead of dx and emits (sum g, sum g*xhat) per 64
 * rows into stat_partials[ceil(N*H*W/64)][2][Cin], i.e. what sh_bn_bwd_reduce would compute from dx and y_prev in a separate
 * pass (torch.nn.BatchNorm2d backward of resnet.py:65-73 / sep_aspp_contrast_head.py:56-61); finish with sh_bn_bwd_finalize and
 * sh_bn_bwd_apply(relu = 0) on g.  y_prev: raw output of the producer conv [N*H*W][ldyp]; mean / invstd / scale / shift: its
 * BatchNorm coefficients [Cin]; relu: 0 / 1.  out_prev (optional): the ReLU mask is out_prev > 0 instead -- residual blocks, where
 * out = relu(bn3(y) + identity) (models/backbone/resnet.py via torchvision Bottleneck): then g is also the identity path's
 * gradient; with act_flags bit 2, out_prev is that output's ReLU quad mask (sh_bn_act's relu_mask, ldop = bytes per pixel).
 * Stride-1 geometries only; SH_EUNSUPPORTED otherwise. */
int sh_conv_dgrad_x6_bnb(const float* dy, int lddy, const float* wt, const float* addend, int ldadd, float* g, int ldg,
                         const float* y_prev, int ldyp, const float* out_prev, int ldop, const float* mean,
                         const float* invstd, const float* scale, const float* shift, int relu, float* stat_partials,
                         int N, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad, int dil, float* workspace, int64_t workspace_bytes, int act_flags,
                         void* stream);
/* Deferred BatchNorm-backward apply (1x1, stride 1): the gradient operand dy of a conv -> BN layer's conv is NOT stored; the loaders
 * evaluate dy = lin[0][c]*g + lin[1][c]*(y - lin[2][c]) + lin[3][c] (= gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)), the second
 * half of torch.nn.BatchNorm2d's backward, coefficients from sh_bn_bwd_finalize) from the masked gradient g and the raw conv output
 * y.  Replaces sh_bn_bwd_apply + the dy tensor for Bottleneck conv1 / conv3 (resnet.py via torchvision) and the pointwise convs of
 * sep_aspp_contrast_head.py:47-61.  dgrad: optional BatchNorm-backward epilogue for the producer of the conv's input as in
 * sh_conv_dgrad_x6_bnb (y_prev == NULL: plain dx [+ addend]).  wgrad: x optionally through the producer's BatchNorm + ReLU.
 * SH_EUNSUPPORTED: no instantiation for the geometry -- materialise dy with sh_bn_bwd_apply(relu = 0) on g. */
int sh_conv_dgrad_x6_lin(const float* g, int ldg, const float* y, int ldy, const float* lin, const float* wt, const float* addend,
                         int ldadd, float* dx, int lddx, const float* y_prev, int ldyp, const float* mean, const float* invstd,
                         const float* scale, const float* shift, int relu, float* stat_partials, int N, int H, int W, int Cin,
                         int Cout, float* workspace, int64_t workspace_bytes, int act_flags, void* stream);
int sh_conv_wgrad_x6_lin(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* g, int ldg,
                         const float* y, int ldy, const float* lin, float* dw, float* workspace, int N, int H, int W, int Cin,
                         int Cout, int KH, int KW, int stride, int pad, int dil, int act_flags, void* stream);
int64_t sh_conv_wgrad_x6_workspace(int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                   int stride, int pad, int dil);
int sh_conv_wgrad_x6(const float* x, int ldx, const float* dy, int lddy, float* dw, float* workspace,
                     int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                     int dil, int act_flags, void* stream);
/* ... of a conv whose input is read through the producer's BatchNorm + ReLU (see sh_conv_fprop_x6_aff): x = raw output of
 * the previous convolution, loader applies relu(x * in_scale[c] + in_shift[c]).  SH_EUNSUPPORTED: output width < 16. */
int sh_conv_wgrad_x6_aff(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* dy, int lddy,
                         float* dw, float* workspace, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                         int stride, int pad, int dil, int act_flags, void* stream);

/* depthwise 3x3 (groups = C), stride 1, padding = dilation ---------------------------------- */
/* Replaces DepthwiseSeparableConv.depthwise (models/head/sep_aspp_contrast_head.py:43-46, :56).
 * w is [C][3][3].  stat_partials: [sh_dw_partials(N,H,W)][2][C]. */
int sh_dw_partials(int N, int H, int W);
int sh_dw_tile_rows(void);                      /* rows per stat-partial of sh_dwconv_fprop (=64) */
/* in_scale / in_shift (optional, both or neither, [C], 16-byte aligned): x is the RAW output of the producer convolution and is
 * read as relu(x * in_scale[c] + in_shift[c]) -- the producer's train-mode BatchNorm + ReLU in the loader (zero padding stays
 * zero), see sh_conv_fprop_x6_aff. */
int sh_dwconv_fprop(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* w, float* y, int ldy,
                    float* stat_partials, int N, int H, int W, int C, int dil, int act_flags, void* stream);
/* y_lin / lin (optional, both or neither; dgrad, dgrad_bnb, wgrad): deferred BatchNorm-backward apply of the depthwise conv's own
 * BatchNorm -- `dy` holds the masked gradient g and the loader evaluates dy = lin[0][c]*g + lin[1][c]*(y_lin - lin[2][c]) + lin[3][c]
 * (sh_bn_bwd_finalize's lin; y_lin = the conv's raw output), so sh_bn_bwd_apply and the dy tensor are skipped
 * (sep_aspp_contrast_head.py:43-61).  dil == 1 with H, W multiples of 8 only; SH_EUNSUPPORTED otherwise. */
int sh_dwconv_dgrad(const float* dy, int lddy, const float* y_lin, int ldyl, const float* lin, const float* w, float* dx, int lddx,
                    int N, int H, int W, int C, int dil, int accumulate, int act_flags, void* stream);
/* ... with the front half of the producer layer's BatchNorm backward in the epilogue (see sh_conv_dgrad_x6_bnb):
 * g <- relumask(y_prev*scale + shift) * dx, stat_partials[sh_dw_partials(N,H,W)][2][C] <- (sum g, sum g*xhat) per 64 pixels. */
int sh_dwconv_dgrad_bnb(const float* dy, int lddy, const float* y_lin, int ldyl, const float* lin, const float* w, float* g, int ldg,
                        const float* y_prev, int ldyp, const float* mean, const float* invstd, const float* scale, const float* shift,
                        float* stat_partials, int N, int H, int W, int C, int dil, int act_flags, void* stream);
/* dw_partials: [sh_dw_partials(N,H,W)][9][C] floats; dw: [C][9] */
int sh_dwconv_wgrad(const float* x, int ldx, const float* in_scale, const float* in_shift, const float* dy, int lddy,
                    const float* y_lin, int ldyl, const float* lin, float* dw_partials, float* dw, int N, int H, int W, int C, int dil,
                    int act_flags, void* stream);

/* bf16 ACTIVATION STORAGE (BASELINE configs[4]).  Raw conv outputs and block outputs may be stored as bf16 -- half the HBM bytes of the
 * streaming kernels and of the HBM-bound layer-1 convolutions; all arithmetic, the BatchNorm statistics (taken from the fp32
 * accumulators before the rounding) and every gradient stay fp32.  Entry points that can take such tensors have an `int act_flags`
 * bit mask (bit i set = the i-th activation tensor named in that entry point's comment is bf16; 0 = everything fp32); a pointer to
 * a bf16 tensor is passed as `const float*` / `float*`, its pixel stride counts ELEMENTS, 16-byte layouts only
 * (SH_EUNSUPPORTED otherwise).
 *   sh_bn_act: bit 0 y, 1 residual, 2 out.   sh_bn_bwd_reduce / sh_bn_bwd_apply: bit 0 y, 1 out.   sh_maxpool_fwd: bit 0 x, 1 y.
 *   sh_conv_fprop_x6 / _aff: bit 0 x, 1 y (statistics from the fp32 accumulators).   sh_conv_dgrad_x6_bnb: bit 0 y_prev, 1 out_prev.
 *   sh_conv_dgrad_x6_lin: bit 0 y, 1 y_prev.   sh_conv_wgrad_x6 / _aff: bit 0 x.   sh_conv_wgrad_x6_lin: bit 0 x, 1 y (x bf16 => y bf16).
 *   sh_dwconv_fprop: bit 0 x, 1 y.   sh_dwconv_dgrad: bit 0 y_lin.   sh_dwconv_dgrad_bnb: bit 0 y_lin, 1 y_prev.   sh_dwconv_wgrad: bit 0 x,
 *   1 y_lin (strip-walk geometries only: dilation 1, H and W multiples of 8).   sh_bilinear_fwd: bit 0 y.
 * Gradients (dy, g, dx, dW), weights, coefficient tables and statistics partials are always fp32. */
/* batch norm ------------------------------------------------------------------------------ */
/* Train-mode nn.BatchNorm2d (every BN of the path; math: SURVEY A.2).  Combines the centred stat partials
 * (sum, M2 over rows_per_partial rows each; count = total rows) in f64, writes mean/invstd/scale/shift and
 * updates the running statistics (momentum, unbiased var). */
int sh_bn_finalize(const float* partials, int n_partials, int C, double count, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean,
                   float* running_var, float* mean, float* invstd, float* scale, float* shift,
                   int rows_per_partial, int partials_ld, void* stream);
/* (partials_ld: row length of the partials when this layer's C columns are a slice of a wider set -- the grouped ASPP launch;
 * 0 = C.)
 * BatchNorm of y = chan_mul[c*chan_stride] * x from the statistics partials of x: depthwise conv + BN of an ASPP branch whose
 * dilation exceeds the feature map (sep_aspp_contrast_head.py:125-131 at stride 32: only the centre tap touches the image) without
 * forming y.  Coefficients come out in the x domain -- mean = mean_x, invstd = w*invstd_y, scale = w*gamma*invstd_y, shift as for y
 * -- so loaders / BatchNorm-backward kernels work on x itself and gamma*invstd*(...) is already d/dx; isy[C] <- invstd_y. */
int sh_bn_finalize_scaled(const float* partials, int n_partials, int C, double count, const float* chan_mul,
                          int chan_stride, const float* gamma, const float* beta, float eps, float momentum,
                          float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                          float* shift, float* isy, int rows_per_partial, void* stream);
/* k (<= 8) BatchNorm layers of C channels finalized in one launch (the grouped ASPP unit): layer i reads columns
 * [partials_col[i], +C) of `partials` (rows of partials_ld floats, 0 = C), optionally through a channel multiplier chan_mul[i]
 * (see sh_bn_finalize_scaled; chan_mul / isy may be NULL arrays or hold NULL entries).  All pointer arguments are HOST arrays of k
 * device pointers; eps / momentum are shared. */
int sh_bn_finalize_multi(int k, const float* partials, int n_partials, int C, double count, int rows_per_partial,
                         int partials_ld, const int* partials_col, const float* const* gamma, const float* const* beta,
                         float* const* running_mean, float* const* running_var, float* const* mean, float* const* invstd,
                         float* const* scale, float* const* shift, const float* const* chan_mul, int chan_stride,
                         float* const* isy, float eps, float momentum, void* stream);
/* Weight gradient [C][9] of such a centre-tap depthwise conv in closed form from the BatchNorm-backward sums:
 * dL/dw_centre = gamma * dgamma * eps * invstd_y^2 / w, off-centre taps exactly 0.  A centre tap that is exactly 0 leaves no trace in
 * dgamma; with g != NULL (the masked gradient w.r.t. the BatchNorm output [M][ldg], x = the conv input [M][ldx], mean_x[C]) that
 * channel's gradient is formed directly: gamma * invstd_y * sum g * (x - mean_x).  g == NULL: such channels get 0. */
int sh_dw_center_wgrad(const float* dgamma, const float* gamma, const float* isy, const float* w, float eps, float* dw,
                       int C, const float* g, int ldg, const float* x, int ldx, const float* mean_x, int64_t M, void* stream);
/* Fold a long list of statistics partials before finalizing it: `chunk` consecutive partials -> one partial of the same format
 * (centred (sum, M2) pairs of rows_per_partial rows each when rows_per_partial > 0 -- the last one short, count rows in all; plain
 * sums when 0), out = [ceil(n_partials / chunk)][2][C].  sh_bn_finalize / sh_bn_bwd_finalize then take `out` with
 * rows_per_partial * chunk.  Blocks own whole 128-byte lines of 32 channels over the whole chip; the one-stage finalize gathers 16
 * bytes per row on C / 4 CUs (PyTorch's nn.BatchNorm2d statistics, models/backbone/resnet.py:65-73 via torchvision; new kernel).
 * SH_EUNSUPPORTED unless C % 4 == 0 and both buffers are 16-byte aligned. */
int sh_bn_fold_partials(const float* partials, int n_partials, int C, double count, int rows_per_partial, int chunk,
                        float* out, void* stream);
/* SyncBN building blocks (cross-GPU BatchNorm is new functionality): reduce the partials to f64 per-channel
 * sums on each rank -- sq is double[2*C + 1] = {sum x [C], sum x^2 [C], local pixel count} (forward, from the centred
 * partials) or {sum g [C], sum g*xhat [C], count} (backward, rows_per_partial = 0) -- all-reduce the vector over RCCL on
 * the host side, then finalize on the global sums and the global count sq[2*C] (ranks may hold different pixel counts). */
int sh_bn_reduce_partials(const float* partials, int n_partials, int C, double count, int rows_per_partial, double* sq,
                          void* stream);
int sh_bn_finalize_sq(const double* sq, int C, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, float* mean, float* invstd,
                      float* scale, float* shift, void* stream);
int sh_bn_bwd_finalize_sq(const double* local_sq, const double* global_sq, int C, float* dgamma,
                          float* dbeta, float* c1, float* c2, const float* gamma, const float* invstd, const float* mean,
                          float* lin, void* stream);
/* Eval-mode coefficients from the running statistics. */
int sh_bn_eval_coefs(const float* gamma, const float* beta, const float* running_mean,
                     const float* running_var, float eps, int C, float* scale, float* shift, void* stream);
/* Per-channel statistics partials of an arbitrary tensor (used where no conv epilogue produced them). */
int sh_stats_partials_count(int64_t M);
int sh_stats_tile_rows(void);                   /* rows per stat-partial of sh_channel_stats (=256) */
int sh_channel_stats(const float* y, int ldy, int64_t M, int C, float* partials, void* stream);
/* out = [relu]( y*scale + shift [+ residual] ).  Replaces BN-apply + ReLU (+ the Bottleneck residual add).
 * res_scale / res_shift (optional, both or neither): `residual` is the RAW output of the block's downsample conv and its
 * BatchNorm is applied on the fly, residual*res_scale + res_shift (same operation order as applying it first: bit-identical),
 * so the downsample branch of torchvision's Bottleneck / BasicBlock never materialises its normalised output.
 * relu_mask (optional, 16-byte layouts only): the ReLU QUAD MASK of `out`, one byte per (pixel, 4 adjacent channels), bit j =
 * (out[c + j] > 0), M * C / 4 bytes -- what the backward passes of a residual block need of its output (1/16 of its bytes):
 * sh_bn_bwd_reduce / sh_bn_bwd_apply with relu = 3 and sh_conv_dgrad_x6_bnb with act_flags bit 2 take it in place of `out`. */
int sh_bn_act(const float* y, int ldy, const float* scale, const float* shift, const float* residual,
              int ldr, const float* res_scale, const float* res_shift, float* out, int ldo, int64_t M, int C, int relu,
              uint8_t* relu_mask, int act_flags, void* stream);
/* Backward of the above.  g = dout * mask.  relu = 0: no mask; 1: mask = out > 0 (needed when a residual was added);
 * 2: mask = y*scale+shift > 0, the forward's own arithmetic recomputed from y (no residual) -- `out` is not read, which
 * saves one activation-sized HBM read in each of the two passes; 3: `out` points to the ReLU quad mask sh_bn_act wrote (ldo = bytes
 * per pixel >= C / 4; 16-byte layouts only).  reduce: partials [n][2][C] of (sum g, sum g*xhat);
 * finalize: dgamma, dbeta; apply: dy = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)), optionally dres = g.
 * Deferred second half: reduce can also STORE g (g_out, 16-byte layouts only, else SH_EUNSUPPORTED) and finalize can emit
 * lin[4][C] = (A, B, mean, D) with dy = A*g + B*(y - mean) + D; the 1x1 consumers of dy (sh_conv_dgrad_x6_lin,
 * sh_conv_wgrad_x6_lin) then evaluate dy in their loaders and sh_bn_bwd_apply / the dy tensor are skipped.
 * act_flags (16-byte layouts): bit 0 y, 1 out stored as bf16; bf16 COMPUTE mode also stores gradients as bf16: bit 2 dout, bit 3 g_out
 * (reduce) / dy (apply), bit 4 dres (apply). */
int sh_bn_bwd_reduce(const float* dout, int lddo, const float* out, int ldo, const float* y, int ldy,
                     const float* mean, const float* invstd, const float* scale, const float* shift,
                     float* partials, int64_t M, int C, int relu, float* g_out, int ldg, int act_flags, void* stream);
int sh_bn_bwd_finalize(const float* partials, int n_partials, int C, const float* gamma,
                       const float* invstd, double count, float* dgamma, float* dbeta, float* c1,
                       float* c2, const float* mean, float* lin, void* stream);
int sh_bn_bwd_apply(const float* dout, int lddo, const float* out, int ldo, const float* y, int ldy,
                    const float* mean, const float* invstd, const float* scale, const float* shift,
                    const float* gamma, const float* c1, const float* c2, float* dy, int lddy, float* dres,
                    int lddres, int64_t M, int C, int relu, int act_flags, void* stream);

/* pooling / resampling -------------------------------------------------------------------- */
/* nn.MaxPool2d(3, 2, 1) (models/backbone/resnet.py:68) and its backward (first-max tie rule).  argmax: one byte per
 * output element [N,Ho,Wo,C] = window position kh*3+kw of the first maximum (NULL in fwd = not recorded); the backward
 * gathers from it and dy alone (x is not re-read, so the stem activation need not be kept).  in_scale / in_shift (optional):
 * x is the stem conv's RAW output, read as relu(x * in_scale[c] + in_shift[c]) -- stem_bn + stem_relu (resnet.py:65-67) in the
 * pooling kernel's loader, so the 64-channel half-resolution activation is never materialised. */
int sh_maxpool_fwd(const float* x, const float* in_scale, const float* in_shift, float* y, uint8_t* argmax, int N, int H,
                   int W, int C, int act_flags, void* stream);
int sh_maxpool_bwd(const uint8_t* argmax, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* nn.AdaptiveAvgPool2d(1) (sep_aspp_contrast_head.py:93,104): x [N,HW,C] -> y [N,C]; backward broadcasts. */
int sh_avgpool_fwd(const float* x, int ldx, float* y, int N, int HW, int C, void* stream);
int sh_avgpool_bwd(const float* dy, float* dx, int lddx, int N, int HW, int C, float scale, int accumulate, void* stream);
/* broadcast [N,C] -> [N,HW,C slice] (the 1x1 -> HxW bilinear of sep_aspp_contrast_head.py:106) and its
 * backward (sum over HW). */
int sh_broadcast_hw(const float* x, float* y, int ldy, int N, int HW, int C, void* stream);
int sh_sum_hw(const float* dy, int lddy, float* dx, int N, int HW, int C, void* stream);
/* F.interpolate(mode='bilinear', align_corners=False) NHWC (sep_aspp_contrast_head.py:235-238) and its backward. */
int sh_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int h, int w, int H, int W, int C, int act_flags, void* stream);
/* workspace (optional, sh_bilinear_bwd_workspace bytes, 16-byte aligned; used when H >= 2h and W >= 2w): backward that reads dy once
 * (row bands + two partial planes) instead of the gather form's ~4 reads per element. */
int64_t sh_bilinear_bwd_workspace(int N, int h, int w, int C);
int sh_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int h, int w, int H, int W, int C, float* workspace,
                    int64_t workspace_bytes, void* stream);
/* F.normalize(p=2, dim=1, eps=1e-12) over channels (sep_aspp_contrast_head.py:29). */
int sh_l2norm_fwd(const float* x, float* y, float* norm, int64_t M, int C, void* stream);
int sh_l2norm_bwd(const float* dy, const float* y, const float* norm, float* dx, int64_t M, int C, void* stream);

/* losses ---------------------------------------------------------------------------------- */
/* Fused: bilinear resize of logits [N,h,w,ldl] to the label grid [N,H,W] (train.py:282-284), coarse target by
 * bucket ranges (hiera_triplet_loss.py:11-38), sigmoid hierarchical BCE (:41-107) and the two all-pixel-mean
 * CE terms (:183-187; cross_entropy_loss.py:7-30, utils.py:6-55).  h==H && w==W means "already full resolution".
 *   buckets_host: HOST int32 [n_coarse][2] (start,end).   labels: uint8 [N,H,W] (255 = ignore).
 *   sums (device double[8]) <- {bce_fine, bce_coarse, ce_fine, ce_coarse, n_valid_fine, n_valid_coarse, n_pixels, 0}
 *   loss_out (device float[1]) <- 5*(bce_f/(max(nvf,1)*nf) + bce_c/(max(nvc,1)*nc)) + ce_f/npix + ce_c/npix
 *   partials: float [sh_hiera2_partials(N,H,W)][8] scratch.  coarse_out (optional): uint8 [N,H,W] coarse targets.
 *   grad_out (optional; sh_loss_bwd_workspace(N,H,W,ldg) bytes, 16-byte aligned, ldg % 4 == 0, upsampling only): the same pass
 *   also writes every full-resolution pixel's d(loss_out)/d(interpolated logits) [N*H*W][ldg] -- its normalisers are label counts,
 *   taken by a small pre-pass -- so the backward is only the adjoint of the resize (sh_hiera2_loss_bwd, workspace_has_grad = 1)
 *   and the per-pixel sigmoid / softmax arithmetic (VALU-bound, ~0.3 ms at 16x512x512) runs once per step instead of twice.
 *   norm_counts (optional, device int64[3], EXACT data-parallel mode): the all-reduced sh_label_counts vector {valid fine, valid coarse,
 *   pixels} of ALL shards -- the local numerators are then divided by these global denominators (loss_out, the emitted gradient and the
 *   counts left in `sums`), so the per-rank losses SUM to the full-batch loss and the summed gradients are the full-batch gradients. */
int sh_hiera2_partials(int N, int H, int W);
int sh_label_counts(const uint8_t* labels, const int* buckets_host, int n_fine, int n_coarse, int64_t total, int64_t* counts,
                    void* stream);
int sh_hiera2_loss_fwd(const float* logits, int ldl, const uint8_t* labels, const int* buckets_host, int n_fine,
                       int n_coarse, double* sums, float* loss_out, float* partials, uint8_t* coarse_out,
                       int N, int h, int w, int H, int W, float* grad_out, int64_t grad_out_bytes, int ldg,
                       const int64_t* norm_counts, void* stream);
/* d(loss_out)/d(logits) * gscale * gscale_dev[0] into dlogits [N,h,w,lddl] (gather form, deterministic; lanes
 * >= C of each row are zeroed).  Uses the counts left in `sums` by the forward.  workspace (optional, sh_loss_bwd_workspace
 * bytes, 16-byte aligned; needs lddl % 4 == 0): two streaming passes -- every full-resolution pixel's gradient once into the
 * workspace, then the adjoint of the bilinear resize as a gather -- instead of the LDS-tiled single kernel that recomputes
 * halo pixels (same arithmetic and summation order: bit-identical results, 4-5x faster at the x4 resize of train.py:282-284).
 * workspace_has_grad = 1: `workspace` is the grad_out of the forward (unit upstream gradient); only the gather runs, its result
 * scaled by gscale * gscale_dev[0]. */
int64_t sh_loss_bwd_workspace(int N, int H, int W, int lddl);
int sh_hiera2_loss_bwd(const float* logits, int ldl, const uint8_t* labels, const int* buckets_host, int n_fine,
                       int n_coarse, const double* sums, const float* gscale_dev, float gscale, float* dlogits,
                       int lddl, int N, int h, int w, int H, int W, float* workspace, int64_t workspace_bytes,
                       int workspace_has_grad, void* stream);
/* Fused bilinear resize + nn.CrossEntropyLoss(ignore_index=255) (valid-pixel mean) of the aux head
 * (train.py:309-313).  sums double[2] = {ce_sum, n_valid}; loss_out = ce_sum / n_valid.  norm_count (optional, device int64[1]):
 * the global valid-pixel count of the exact data-parallel mode (sh_label_counts counts[0], all-reduced). */
int sh_ce_loss_fwd(const float* logits, int ldl, const uint8_t* labels, int C, double* sums, float* loss_out,
                   float* partials, int N, int h, int w, int H, int W, float* grad_out, int64_t grad_out_bytes, int ldg,
                   const int64_t* norm_count, void* stream);
int sh_ce_loss_bwd(const float* logits, int ldl, const uint8_t* labels, int C, const double* sums,
                   const float* gscale_dev, float gscale, float* dlogits, int lddl, int N, int h, int w, int H,
                   int W, float* workspace, int64_t workspace_bytes, int workspace_has_grad,
                   void* stream);
/* int64 label map -> uint8 (the reference hands the loss i64 labels, train.py:262). */
int sh_labels_to_u8(const int64_t* in, uint8_t* out, int64_t n, void* stream);
/* Tree-triplet (tree_triplet_loss.py:15-65, rmi_tree_triplet_loss.py:14-70): nearest label resize to the
 * embedding grid, per anchor class the first <= max_triplet anchor / positive / negative rows in raster order,
 * hinge(d_ap - d_an + margin) mean, mean over the classes that produced triplets.
 *   emb: [N,h,w,D] unit-norm rows.  masks: device u64 [256][2][4] = per anchor class the 256-bit membership sets of
 *   positive and negative labels; anchor_ok: device u64 [4] = classes that may be anchors.
 *   out (device float[2]) <- {loss (0 if no class), class_count}.  workspace: sh_triplet_workspace(N*h*w) bytes,
 *   kept for the backward.  bwd ACCUMULATES gscale*gscale_dev[0]*d(loss)/d(emb) into demb (zero it first). */
int64_t sh_triplet_workspace(int64_t M);
int sh_triplet_fwd(const float* emb, const uint8_t* labels, const uint64_t* masks, const uint64_t* anchor_ok,
                   int max_triplet, float margin, float* out, void* workspace, int N, int h, int w, int D, int H,
                   int W, void* stream);
int sh_triplet_bwd(const float* emb, const void* workspace, const float* out, const float* gscale_dev, float gscale,
                   float* demb, int N, int h, int w, int D, void* stream);
/* out = (main + (count > 0 ? factor*trip_out[0] : 0)) * loss_weight, count = ready_count ? ready_count[0] : trip_out[1]
 * (hiera_triplet_loss.py:193-211). */
int sh_combine_loss(const float* main_loss, const float* trip_out, const float* ready_count, float factor,
                    float loss_weight, float* out, void* stream);
/* ---- 3-level loss with the RMI lower bound: RMIHieraTripletLoss (models/loss/rmi_hiera_triplet_loss.py:323-546) ---- */
/* Fused resize + mid/high targets by gather (:21-63) + 3-level sigmoid BCE (eps 1e-6, :352-470) + three all-pixel-mean CE
 * terms (:523-526).  f2m_host / f2h_host: HOST int32[n_fine] fine->mid / fine->high maps.
 *   sums (device double[8]) <- {bce_f, bce_m, bce_h, ce_f, ce_m, ce_h, n_valid, n_pixels}
 *   loss_out (device float[1]) <- 0.5*5*(bce_f/(nv*nf)+bce_m/(nv*nm)+bce_h/(nv*nh)) + (ce_f+ce_m+ce_h)/npix
 *   probs (optional): planar f32 [N][C][H][W] <- sigmoid(z)*valid + 1e-6 (the RMI input, :496).
 *   mid_out / high_out (optional, both or neither): uint8 [N][H][W] <- the target maps of _prepare_targets_three_level
 *   (:21-63): 255 where the fine label is 255, else fine_to_mid[f] / fine_to_high[f].
 *   grad_out (optional; ldg = 16 or 32 >= C, sh_loss_bwd_workspace(N,H,W,ldg) bytes): the same pass leaves d(loss_out)/d(interpolated
 *   logits) of every full-resolution pixel, [N*H*W][ldg], for sh_hiera3_loss_bwd(..., workspace_has_grad = 1). */
int sh_hiera3_loss_fwd(const float* logits, int ldl, const uint8_t* labels, const int* f2m_host, const int* f2h_host,
                       int n_fine, int n_mid, int n_high, double* sums, float* loss_out, float* partials, float* probs,
                       uint8_t* mid_out, uint8_t* high_out, int N, int h, int w, int H, int W, float* grad_out,
                       int64_t grad_out_bytes, int ldg, void* stream);
/* RMI lower bound (:292-317, :479-517): per (image, channel) f64 9x9 Gram matrices over the 3x3 windows, inverse, Schur
 * complement, Cholesky log-det; rmi_out (device float[1]) <- sum_c mean_b(0.5*logdet)/9.  dprob (optional, planar like
 * probs) <- d(0.5*logdet_{b,c})/dP (the 1/(9N) and lambda factors are applied by sh_hiera3_loss_bwd's rmi_coef). */
int64_t sh_rmi_workspace(int N, int C, int H, int W);
/* Byte offset inside that workspace of the f64 [N][C] per-(image, channel) values rmi_now = 0.5*logdet (:513) left by
 * sh_rmi_loss (parity tests read them back and compare with the reference's own f64 values). */
int64_t sh_rmi_values_offset(int N, int C, int H, int W);
int sh_rmi_loss(const float* probs, const uint8_t* labels, const int* f2m_host, const int* f2h_host, int n_fine, int n_mid,
                int n_high, void* workspace, float* rmi_out, float* dprob, int N, int H, int W, void* stream);
/* dlogits [N,h,w,lddl] <- gscale*gscale_dev[0] * d(loss_out + rmi_coef * <dprob, P>)/d(logits)  (tiled gather form; with a
 * workspace of sh_loss_bwd_workspace(N,H,W,lddl) bytes and upsampled logits: the two streaming passes of sh_hiera2_loss_bwd,
 * bit-identical).  workspace_has_grad: the workspace is sh_hiera3_loss_fwd's grad_out (row stride lddl); the RMI term is added to it in
 * one streaming pass (probs = that forward's planar probabilities, needed with dprob), then the scaled adjoint of the resize. */
int sh_hiera3_loss_bwd(const float* logits, int ldl, const uint8_t* labels, const int* f2m_host, const int* f2h_host,
                       int n_fine, int n_mid, int n_high, const double* sums, const float* dprob, float rmi_coef,
                       const float* gscale_dev, float gscale, float* dlogits, int lddl, int N, int h, int w, int H, int W,
                       float* workspace, int64_t workspace_bytes, int workspace_has_grad, const float* probs, void* stream);
/* out = (a + alpha*b[0]) -- device scalar combine used for lambda*rmi + rest. */
int sh_scalar_axpy(const float* a, const float* b, float alpha, float* out, void* stream);

/* fine argmax + pixel accuracy + confusion matrix (train.py:37-49, 381-385; mIoU is build-defined).
 * counts: int64 [2 + n_fine*n_fine] = {correct, valid, confusion[gt][pred]...} (accumulated; zero it first). */
int sh_pixel_metrics(const float* logits, int ldl, const uint8_t* labels, int n_fine, long long* counts,
                     int N, int h, int w, int H, int W, void* stream);

/* optimizer ------------------------------------------------------------------------------- */
/* torch.optim.SGD(momentum, weight_decay) step (train.py:239-246, 317) over up to SH_SGD_MAX tensors per call:
 * g += wd*w; v = first ? g : mom*v + g; w -= lr*v. */
#define SH_SGD_MAX 48
int sh_sgd_step(int n_tensors, float* const* w, const float* const* g, float* const* v, const int64_t* numel,
                float lr, float momentum, float weight_decay, int first_step, float gscale, void* stream);

/* misc ------------------------------------------------------------------------------------ */
int sh_fill(float* p, float v, int64_t n, void* stream);
int sh_axpy(float* y, const float* x, float a, int64_t n, void* stream);   /* y += a*x */
int sh_copy(void* dst, const void* src, int64_t bytes, void* stream);      /* async device-to-device copy */

/* ---------------------------------------------------------------------------------------------------------------------------
 * bf16 COMPUTE mode (BASELINE configs[4]: "bf16 storage / f32 accumulate"; seghiero_amd/csrc/conv_b16.hip).  Operands are rounded once
 * to bf16 on their way to LDS (after the producer's BatchNorm + ReLU where that runs in the loader), ONE MFMA product per tile instead
 * of six, fp32 accumulators, BatchNorm statistics from the fp32 accumulators, fp32 weight gradients / master weights / SGD.
 * Activations are bf16 tensors (pixel strides count elements, multiples of 8; 16-byte aligned rows); gradients w.r.t. activations are
 * bf16 or fp32 tensors.  Same reference op chains as the sh_conv_*_x6* entry points (models/backbone/resnet.py:65-73 via torchvision,
 * models/head/sep_aspp_contrast_head.py:43-62, 100-131, 172-191).  SH_EUNSUPPORTED = no instantiation for the geometry, nothing
 * launched: run the fp32-accurate entry point. */
/* bf16 copies of n (<= 40) dense conv weights [Cout][taps][Cin] fp32 (OHWI): wb[i] = the same layout in bf16 (forward operand),
 * wtb[i] = [taps][Cin][pad8(Cout)] bf16, zero padded (input-gradient operand); entries of wb / wtb may be NULL.  Host arrays of device pointers. */
int sh_weights_to_bf16_multi(int n, const float* const* w, void* const* wb, void* const* wtb, const int* cout, const int* taps,
                             const int* cin, void* stream);
/* act_flags bit 1: y stored as bf16 (else fp32).  in_scale / in_shift (both or neither): x read as relu(x * scale + shift).
 * Cout % 8 != 0 (the classifier): ldy >= pad8(Cout), no stat_partials, bias (if any) holds pad8(Cout) floats with zeros in the padding;
 * y's padding lanes are written (zeros). */
int sh_conv_fprop_b16(const void* x, int ldx, const float* in_scale, const float* in_shift, const void* w_bf16, const float* bias,
                      void* y, int ldy, float* stat_partials, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                      int pad, int dil, float* workspace, int64_t workspace_bytes, int act_flags, void* stream);
int sh_conv1x1_grouped_fprop_b16(int ngroups, const void* const* x, const int* ldx, const float* const* in_scale,
                                 const float* const* in_shift, const void* const* w_bf16, void* y, int ldy, float* stat_partials,
                                 int N, int H, int W, int Cin, int Cout, int act_flags, void* stream);
/* act_flags: bit 0 dy is bf16 (else fp32), 1 dx stored bf16, 2 addend bf16, 3 y_prev bf16, 4 out_prev bf16, 5 out_prev = ReLU quad mask.
 * y_lin / lin (both or neither, 1x1 only, dy = the masked gradient g in bf16): operand = lin(g, y_lin) as sh_conv_dgrad_x6_lin.
 * y_prev != NULL: BatchNorm-backward epilogue as sh_conv_dgrad_x6_bnb.
 * Strided convs (no hooks: y_lin, y_prev, addend NULL): stride-2 KxK runs by input-parity class; bit 6 = 1x1 strided conv whose result
 * is ADDED to dx at the strided pixels (sh_conv_dgrad_x6's mode 1: the downsample branch joins the block input's gradient). */
int sh_conv_dgrad_b16(const void* dy, int lddy, const void* y_lin, int ldyl, const float* lin, const void* wt_bf16, const void* addend,
                      int ldadd, void* dx, int lddx, const void* y_prev, int ldyp, const void* out_prev, int ldop, const float* mean,
                      const float* invstd, const float* scale, const float* shift, int relu, float* stat_partials, int N, int H,
                      int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dil, float* workspace,
                      int64_t workspace_bytes, int act_flags, void* stream);
/* act_flags bit 0: dy is bf16 (else fp32).  workspace: sh_conv_wgrad_x6_workspace(...) bytes. */
int sh_conv_wgrad_b16(const void* x, int ldx, const float* in_scale, const float* in_shift, const void* dy, int lddy, const void* y_lin,
                      int ldyl, const float* lin, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                      int stride, int pad, int dil, int act_flags, void* stream);

/* ---- RCCL communicator for a host without PyTorch (the Python package uses torch.distributed, backend "nccl" = RCCL) ----
 * One process per GPU.  Rank 0 makes a 128-byte id, the launcher hands it to every rank, every rank calls sh_comm_init on its current HIP
 * device (collective: returns when all ranks have joined).  Collectives are in place.  dtype: 0 f32, 1 f64, 2 i64; op: 0 sum, 1 min, 2 max.
 * sh_comm_all_reduce runs in order on `stream`.  sh_comm_all_reduce_async runs on the communicator's own side stream, which first waits
 * for everything `producer_stream` has queued (event hand-off) -- the gradient buckets of a backward that keeps running;
 * sh_comm_wait makes `consumer_stream` wait for everything issued on the side stream so far (before the optimizer step).
 * RCCL is loaded on first use (dlopen): SH_EUNSUPPORTED if librccl is not found. */
int sh_comm_unique_id(void* id128);
int sh_comm_init(const void* id128, int world, int rank, void** comm_out);
int sh_comm_destroy(void* comm);
int sh_comm_all_reduce(void* comm, void* buf, int64_t count, int dtype, int op, void* stream);
int sh_comm_all_reduce_async(void* comm, void* buf, int64_t count, int dtype, int op, void* producer_stream);
int sh_comm_wait(void* comm, void* consumer_stream);
int sh_comm_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif
