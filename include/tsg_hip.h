/*
 * tsg_hip.h — C-ABI of libtsg_hip.so, the MI355X (gfx950) hot-path kernels
 * behind TorchSeg's furnace.seg_opr / apex.parallel operator surface.
 *
 * Conventions (every entry point):
 *   - plain `extern "C"`, raw device pointers + sizes, no torch / C++ types;
 *   - returns 0 on success, <0 for an invalid argument (TSG_E_*), >0 = hipError_t;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*) and never
 *     synchronise; the caller owns every buffer including workspaces, whose size
 *     comes from the matching *_ws_bytes() query;
 *   - the library keeps no mutable global state; it is re-entrant across streams.
 *
 * Reference interfaces replaced are cited per group (paths relative to the
 * TorchSeg checkout).
 */
#ifndef TSG_HIP_H
#define TSG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSG_VERSION 100 /* major*10000 + minor*100 + patch */

/* element types of activations / logits */
enum { TSG_F32 = 0, TSG_BF16 = 1 };
/* memory layout of a 4-D activation */
enum { TSG_NCHW = 0, TSG_NHWC = 1 };
/* label element types */
enum { TSG_I64 = 0, TSG_U8 = 1 };

/* error codes (negative) */
#define TSG_E_DTYPE   (-1)
#define TSG_E_LAYOUT  (-2)
#define TSG_E_SHAPE   (-3)
#define TSG_E_ALIGN   (-4)
#define TSG_E_NULL    (-5)
#define TSG_E_WS      (-6)
#define TSG_E_COMM_LIB   (-7)    /* librccl.so missing or lacking a symbol */
#define TSG_E_COMM_BASE  (-100)  /* RCCL failure r is reported as TSG_E_COMM_BASE - r (tsg_comm_error_string) */

int tsg_version(void);

/* ------------------------------------------------------------------------
 * SyncBatchNorm — replaces apex.parallel.SyncBatchNorm (imported at
 * model/bisenet/cityscapes.bisenet.R18/train.py:24-25) whose in-tree statement
 * is furnace/legacy/sync_bn: sumsquare_forward / batchnorm_forward /
 * batchnorm_backward / sumsquare_backward (src/gpu/operator.h,
 * src/gpu/syncbn_kernel.cu:73-174) and _compute_mean_std (syncbn.py:86-98).
 *
 * x is [N, C, HW] (TSG_NCHW) or [N*HW, C] (TSG_NHWC), contiguous.
 * ---------------------------------------------------------------------- */

/* number of per-channel partial rows the stats / bwd-reduce kernels write */
int    tsg_bn_num_partials(int layout, int64_t N, int64_t C, int64_t HW);
/* bytes of the fp32 partial buffer: num_partials * 2 * C * 4 */
size_t tsg_bn_partial_ws_bytes(int layout, int64_t N, int64_t C, int64_t HW);

/* partial[s][0][c] = sum x, partial[s][1][c] = sum x^2 over slice s (fp32);
 * *rows (host int) receives the number of slices S actually written
 * (<= tsg_bn_num_partials, depends on pointer alignment).
 * Replaces Sum_Square_Forward_CUDA (syncbn_kernel.cu:142-157, 245-265). */
int tsg_bn_stats(const void* x, int dtype, int layout,
                 int64_t N, int64_t C, int64_t HW,
                 float* partial, int* rows, void* stream);

/* Collapse partial[S][2][C] to sums[2*C] (fp32, reduced in fp64, fixed order).
 * Used before the cross-GPU all-reduce (syncbn.py:75 ReduceAddCoalesced). */
int tsg_bn_collapse(const float* partial, int S, int64_t C, float* sums,
                    void* stream);
/* The forward exchange message in one launch: msg[0..2C) as tsg_bn_collapse, msg[2C] = count / 4096,
 * msg[2C+1] = count % 4096 (the local element count as two fp32 words that stay exact under SUM). */
int tsg_bn_collapse_count(const float* partial, int S, int64_t C, float* msg, int64_t count, void* stream);

/* Per-channel constants are folded into two small device arrays so the
 * streaming kernels touch 2-5 floats per channel:
 *   fwd_pack fp[3][C] = { a = gamma*invstd, b = beta - mean*a, mean }
 *   bwd_pack bp[5][C] = { a, b, mean, Bc = -a*k1*invstd, C2 = -a*k0 }
 * with k0 = sum dy'/n, k1 = sum dy'*xhat/n (n = GLOBAL count).  Both must be
 * 16-byte aligned. */

/* From (partial or all-reduced) sums and the global element count per channel:
 *   mean = sum/n ; sumvar = sumsq - sum*mean ; invstd = (sumvar/n + eps)^-1/2
 *   running_mean = (1-m)*running_mean + m*mean
 *   running_var  = (1-m)*running_var  + m*sumvar/(n-1)      (syncbn.py:86-98)
 * gamma/beta (may be NULL = 1/0), running_* and num_batches_tracked may be NULL.
 * count is the GLOBAL n.  When count_dev != NULL the count is read from the
 * device instead: n = count_dev[0]*4096 + count_dev[1] (two fp32 words that stay
 * exact under an all-reduce SUM; lets ranks with unequal batches agree without a
 * host sync).  Writes mean[C], invstd[C] and fwd_pack[3][C]. */
int tsg_bn_finalize(const float* partial, int S, int64_t C, double count,
                    const float* count_dev, float eps, float momentum,
                    const float* gamma, const float* beta,
                    float* running_mean, float* running_var,
                    int64_t* num_batches_tracked,
                    float* mean, float* invstd, float* fwd_pack, void* stream);

/* fwd_pack from given statistics (eval mode: running mean / rsqrt(var+eps)). */
int tsg_bn_affine(const float* mean, const float* invstd, const float* gamma,
                  const float* beta, int64_t C, float* fwd_pack, void* stream);

/* y = relu?( a*x + b (+ residual) )  ==  relu?(gamma*(x-mean)*invstd + beta (+ residual)).
 * Replaces BatchNorm_Forward_CUDA (syncbn_kernel.cu:73-89) fused with the
 * nn.ReLU / residual add that follow it (seg_oprs.py:39-46, resnet.py:33-53).
 * residual may be NULL; y may alias x. */
int tsg_bn_apply_fwd(const void* x, const void* residual, void* y,
                     int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                     const float* fwd_pack, int relu, void* stream);

/* Backward reduction: with dy' = dy * [pre-activation > 0] when relu, else dy:
 *   partial[s][0][c] = sum dy' ; partial[s][1][c] = sum dy' * (x-mean)
 * (GradOp, syncbn_kernel.cu:12-23,108-113).  When relu != 0 the mask is taken
 * from y > 0 if y != NULL (needed when a residual was fused), otherwise it is
 * recomputed from x with the forward's exact affine map (saves one read). */
int tsg_bn_bwd_reduce(const void* dy, const void* x, const void* y,
                      int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                      const float* fwd_pack, int relu,
                      float* partial, int* rows, void* stream);

/* From partials of the backward reduction:
 *   dgamma = sum dy'(x-mean) * invstd, dbeta = sum dy'   (syncbn_kernel.cu:130,135;
 *   pass the LOCAL sums: DDP averages parameter grads across ranks later), and/or
 *   bwd_pack from the GLOBAL (all-reduced) sums and count.  batch_stats == 0
 *   (eval mode) gives Bc = C2 = 0.  dgamma/dbeta/bwd_pack may each be NULL. */
int tsg_bn_bwd_coeffs(const float* partial, int S, int64_t C, double count,
                      const float* count_dev, int batch_stats,
                      const float* invstd, const float* fwd_pack,
                      float* dgamma, float* dbeta, float* bwd_pack, void* stream);

/* dx = a*dy' + Bc*(x-mean) + C2  ==  gamma*invstd*(dy' - k0 - xhat*k1); optionally
 * dres = dy' (gradient of the fused residual input).  Composition of
 * syncbn_kernel.cu:118-135 and Sum_Square_Backward (160-174) as autograd chains
 * them (functions.py:22-61). */
int tsg_bn_bwd_apply(const void* dy, const void* x, const void* y,
                     void* dx, void* dres,
                     int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                     const float* bwd_pack, int relu, void* stream);

/* The block tail BN -> (+identity) -> ReLU (resnet.py:44-51, 92-99) with its ReLU mask kept as ONE BIT per element
 * (round 6).  The two backward kernels of such a layer need "y > 0"; reading it from the stored output costs a whole
 * tensor per kernel.  tsg_bn_apply_fwd_maskbits is tsg_bn_apply_fwd(relu = 1) that also writes bits[(pixel * C + c) / V],
 * bit j = (y[pixel][c + j] > 0) of the ROUNDED output, V = 8 channels per byte for bf16 and 4 for fp32 (NHWC only, C % V
 * == 0, 16-byte aligned tensors: tsg_bn_maskbits_supported; N * HW * C / V bytes); the _maskbits backward entry points are
 * tsg_bn_bwd_reduce / tsg_bn_bwd_apply(relu = 1, y) reading that byte instead of y: same values, 2 of the 8 tensor passes
 * of the layer's backward gone.  residual may be NULL. */
int tsg_bn_maskbits_supported(int dtype, int layout, int64_t C, int64_t HW);
int tsg_bn_apply_fwd_maskbits(const void* x, const void* residual, void* y, void* bits, int dtype, int layout, int64_t N,
                              int64_t C, int64_t HW, const float* fwd_pack, void* stream);
int tsg_bn_bwd_reduce_maskbits(const void* dy, const void* x, const void* bits, int dtype, int layout, int64_t N, int64_t C,
                               int64_t HW, const float* fwd_pack, float* partial, int* rows, void* stream);
int tsg_bn_bwd_apply_maskbits(const void* dy, const void* x, const void* bits, void* dx, void* dres, int dtype, int layout,
                              int64_t N, int64_t C, int64_t HW, const float* bwd_pack, void* stream);

/* Mixed-layout passes for conv stems: x (conv output) / dx are NCHW [N, C, HW]
 * while y / dy are NHWC [N*HW, C]; a [C x 64-pixel] tile is transposed through
 * LDS inside the BN pass, so no separate layout-conversion copy is needed
 * between a C_in=3 stem (fastest in NCHW under MIOpen) and the NHWC network.
 * Supported when C % 8 == 0, C <= 128 and HW % (16 / elem_size) == 0; the ReLU
 * mask is recomputed from x (no fused residual). */
int tsg_bn_mixed_supported(int dtype, int64_t C, int64_t HW);
int tsg_bn_mixed_num_partials(int64_t N, int64_t C, int64_t HW);
int tsg_bn_apply_fwd_mixed(const void* x_nchw, void* y_nhwc, int dtype,
                           int64_t N, int64_t C, int64_t HW,
                           const float* fwd_pack, int relu, void* stream);
int tsg_bn_bwd_reduce_mixed(const void* dy_nhwc, const void* x_nchw, int dtype,
                            int64_t N, int64_t C, int64_t HW,
                            const float* fwd_pack, int relu,
                            float* partial, int* rows, void* stream);
int tsg_bn_bwd_apply_mixed(const void* dy_nhwc, const void* x_nchw, void* dx_nchw,
                           int dtype, int64_t N, int64_t C, int64_t HW,
                           const float* bwd_pack, int relu, void* stream);

/* ------------------------------------------------------------------------
 * Global average pooling [N,C,H,W] -> [N,C] — replaces nn.AdaptiveAvgPool2d(1)
 * inside AttentionRefinement / FeatureFusion (furnace/seg_opr/seg_oprs.py:200,
 * 224) and GlobalAvgPool2d (seg_oprs.py:97-107).  x is [N, C, HW] (TSG_NCHW)
 * or [N, HW, C] (TSG_NHWC); out / dout are [N, C] in the activation dtype.
 * ---------------------------------------------------------------------- */
size_t tsg_gap_ws_bytes(int layout, int64_t N, int64_t C, int64_t HW);
int tsg_gap_fwd(const void* x, void* out, int dtype, int layout,
                int64_t N, int64_t C, int64_t HW,
                void* ws, size_t ws_bytes, void* stream);
int tsg_gap_bwd(const void* dout, void* dx, int dtype, int layout,
                int64_t N, int64_t C, int64_t HW, void* stream);

/* Adaptive average pooling to a small grid, channels_last — nn.AdaptiveAvgPool2d(1 / 2 / 3 / 6) of PSPNet's pyramid
 * pooling (model/pspnet/ade.pspnet.R50_v1c/network.py:75-109) and PSANet's conv6 input.  x [N,H,W,C] -> out [N,OH,OW,C];
 * bin o of a dimension covers [floor(o * H / OH), ceil((o + 1) * H / OH)) (ATen's convention), OH <= H, OW <= W.  Forward
 * = a row-split partial-sum pass + a fold (fp32 workspace of tsg_adaptive_avgpool_nhwc_ws_bytes); backward: every input
 * pixel gathers the <= 4 bins that contain it.  Deterministic. */
size_t tsg_adaptive_avgpool_nhwc_ws_bytes(int dtype, int64_t N, int C, int H, int W, int OH, int OW);
int tsg_adaptive_avgpool_nhwc_fwd(const void* x, void* out, int dtype, int64_t N, int C, int H, int W, int OH, int OW,
                                  void* ws, size_t ws_bytes, void* stream);
int tsg_adaptive_avgpool_nhwc_bwd(const void* dout, void* dx, int dtype, int64_t N, int C, int H, int W, int OH, int OW,
                                  void* stream);

/* Channel concatenation of two channels_last maps — `torch.cat([x1, x2], dim=1)` in FeatureFusion.forward
 * (furnace/seg_opr/seg_oprs.py:233-235): out[m] = a[m] followed by b[m] for every row (pixel) m; row sizes in bytes,
 * multiples of 16.  The backward of a concatenation is two views of the gradient (no kernel). */
int tsg_cat2_rows(const void* a, const void* b, void* out, int64_t rows, int64_t row_bytes_a, int64_t row_bytes_b,
                  void* stream);

/* Channel gate y = x * s[n,c] (+ x when add_identity): the squeeze-excite
 * multiply of AttentionRefinement (`fm * fm_se`, seg_oprs.py:209-210) and
 * FeatureFusion (`fm + fm * fm_se`, seg_oprs.py:236-237).  s / ds are [N, C] in
 * the activation dtype.  Backward in one pass: dx = dy*s (+dy), ds = sum_p dy*x.
 * ws as for tsg_gap_fwd (tsg_gap_ws_bytes). */
int tsg_chanscale_fwd(const void* x, const void* s, void* y, int dtype, int layout,
                      int64_t N, int64_t C, int64_t HW, int add_identity, void* stream);
int tsg_chanscale_bwd(const void* dy, const void* x, const void* s, void* dx, void* ds,
                      int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                      int add_identity, void* ws, size_t ws_bytes, void* stream);
/* The two halves of tsg_chanscale_bwd for a gate whose scale was computed FROM the pooled map it gates (AttentionRefinement /
 * FeatureFusion, seg_oprs.py:192-238: `fm * se(gap(fm))`).  The map then has two gradients, the gate's dy s (+ dy) and the
 * pooled branch's g[n, c] / HW, and the second depends on ds: autograd wrote the first (tsg_chanscale_bwd), and added the second
 * in a pass of its own over the map.  _ds computes ds only (reads dy, x); _dx, called once g is known, writes
 * dx = round(round(dy s (+ dy)) + gadd[n, c]) — the same bits — in one pass (reads dy).  NHWC, C % (16 / elem size) == 0. */
int tsg_chanscale_bwd_ds(const void* dy, const void* x, void* ds, int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                         void* ws, size_t ws_bytes, void* stream);
int tsg_chanscale_bwd_dx(const void* dy, const void* s, const void* gadd, void* dx, int dtype, int layout, int64_t N, int64_t C,
                         int64_t HW, int add_identity, void* stream);

/* Max pooling, channels_last — replaces ResNet's nn.MaxPool2d(kernel_size=3,
 * stride=2, padding=1) (furnace/base_model/resnet.py:132).  x [N, IH, IW, C],
 * y [N, OH, OW, C] with OH = (IH + 2P - K)/S + 1; argmax_u8 [N, OH, OW, C] holds
 * the window position ky*K + kx of each maximum (first maximum in scan order, NaN
 * wins: at::native's rule).  C % (16 / elem_size) == 0. */
int tsg_maxpool_nhwc_fwd(const void* x, void* y, void* argmax_u8, int dtype,
                         int64_t N, int C, int IH, int IW, int OH, int OW,
                         int K, int S, int P, void* stream);
int tsg_maxpool_nhwc_bwd(const void* dy, const void* argmax_u8, void* dx, int dtype,
                         int64_t N, int C, int IH, int IW, int OH, int OW,
                         int K, int S, int P, void* stream);

/* Stem convolution — replaces the cuDNN call behind nn.Conv2d(3, 64, kernel_size=7,
 * stride=2, padding=3, bias=False): ResNet conv1 (furnace/base_model/resnet.py:96-97) and
 * BiSeNet's SpatialPath.conv_7x7 (model/bisenet/cityscapes.bisenet.R18/network.py:116,
 * through ConvBnRelu, furnace/seg_opr/seg_oprs.py:27-31).  The image needs no gradient, so
 * training is forward + weight gradient.  x [B,3,H,W] bf16 NCHW (W even); y and dy
 * [B,OH,OW,64] bf16 channels_last, OH = (H-1)/2+1; w and dw fp32 [64,3,7,7] (rounded to bf16
 * for the MFMA as autocast does; fp32 accumulation; dw summed in a fixed order).
 * ws: tsg_stem_conv_ws_bytes() bytes, 16-B aligned.  tsg_stem_conv_supported returns 1 when
 * a convolution with these hyper-parameters is handled here. */
int tsg_stem_conv_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad,
                            int dilation, int groups, int64_t H, int64_t W);
size_t tsg_stem_conv_ws_bytes(void);
int tsg_stem_conv_fwd(const void* x, const float* w, void* y, int64_t B, int64_t H, int64_t W,
                      void* ws, size_t ws_bytes, void* stream);
int tsg_stem_conv_wrw(const void* x, const void* dy, float* dw, int64_t B, int64_t H, int64_t W,
                      void* ws, size_t ws_bytes, void* stream);
/* The weight gradient with the BatchNorm + ReLU backward of the layer behind the stem folded into its staging: instead
 * of dy it takes da (gradient w.r.t. relu(bn(xc))), the stem's own output xc and the backward pack bp[5][64] of
 * tsg_bn_bwd_coeffs, and evaluates dy = a dv + Bc (xc - mean) + C2, dv = [a xc + b > 0] da (tsg_bn_bwd_apply's
 * arithmetic and bf16 rounding) while the tile is staged.  The image needs no gradient, so this kernel is the only
 * consumer of dy: the 0.5 GB tensor is never written (resnet.py:96-98 / bisenet network.py:116 with seg_oprs.py:27-46). */
int tsg_stem_conv_wrw_bn(const void* x, const void* da, const void* xc, const float* bp, float* dw, int64_t B,
                         int64_t H, int64_t W, void* ws, size_t ws_bytes, void* stream);
/* The same forward with the statistics pass of the BatchNorm that follows every such stem (resnet.py:98,
 * seg_oprs.py:27-31) folded into its epilogue: partial[S][2][64] fp32 = per-block sums / square sums of the
 * bf16-rounded outputs, S = tsg_stem_conv_stats_partials(B, H, W) — the layout tsg_bn_finalize / tsg_bn_collapse
 * take, so tsg_bn_stats (one more read of the 0.5 GB activation) is not run. */
int tsg_stem_conv_stats_partials(int64_t B, int64_t H, int64_t W);
int tsg_stem_conv_fwd_stats(const void* x, const float* w, void* y, float* partial, int64_t B, int64_t H,
                            int64_t W, void* ws, size_t ws_bytes, void* stream);

/* BatchNorm (statistics already folded into the packs of tsg_bn_finalize / tsg_bn_bwd_coeffs) + ReLU +
 * MaxPool2d(kernel 3, stride 2, padding 1) of the ResNet stem in one pass per direction — replaces
 * `x = self.maxpool(self.relu(self.bn1(x)))` (furnace/base_model/resnet.py:98-100,131-133) for channels_last x
 * [N, IH, IW, C], C % (16 / elem_size) == 0, OH = (IH - 1) / 2 + 1.  Forward: y [N, OH, OW, C] and the one-byte
 * window position of each maximum: y is bit-equal to tsg_maxpool_nhwc_fwd(tsg_bn_apply_fwd(x, relu)); the maximum is
 * ranked on the unrounded fp32 values, as the fp32 reference does, so in bf16 the position may differ from the unfused
 * pair's where two elements of a window round to the same bf16 value.  Backward: the
 * gradient of a stem pixel is gathered from dpool through argmax_u8, masked by the recomputed ReLU and consumed by
 * the BN reduction (partial[S][2][C], S = tsg_bn_relu_pool_bwd_num_partials; then tsg_bn_bwd_coeffs) and by
 * dx = a dy' + Bc (x - mean) + C2.  Neither relu(bn(x)) nor its gradient is ever written. */
int tsg_bn_relu_pool_fwd(const void* x, void* y, void* argmax_u8, int dtype, int64_t N, int C, int IH, int IW,
                         int OH, int OW, const float* fp, void* stream);
int tsg_bn_relu_pool_bwd_num_partials(int dtype, int64_t N, int C, int IH, int IW);
int tsg_bn_relu_pool_bwd_reduce(const void* dpool, const void* argmax_u8, const void* x, int dtype, int64_t N,
                                int C, int IH, int IW, int OH, int OW, const float* fp, float* partial,
                                void* stream);
int tsg_bn_relu_pool_bwd_apply(const void* dpool, const void* argmax_u8, const void* x, void* dx, int dtype,
                               int64_t N, int C, int IH, int IW, int OH, int OW, const float* bp, void* stream);

/* The ResNet stem WITHOUT its activation (round 6) — replaces `x = self.maxpool(self.relu(self.bn1(self.conv1(x))))`
 * (furnace/base_model/resnet.py:96-100,131-133) as a whole when conv1 is the 3 -> 64 7x7/2 convolution above: the
 * convolution costs 79 GFLOP over a 100 MB image while its output is 537 MB at 16 x 1024^2, so every pass RE-EVALUATES the
 * tile of y it needs (rounded to bf16 exactly as tsg_stem_conv_fwd stores it) instead of reading it:
 *   tsg_stem_conv_stats             partial[S][2][64] of tsg_stem_conv_fwd_stats (S = tsg_stem_conv_stats_partials), y not written
 *   tsg_stem_conv_bn_relu_pool_fwd  ypool [B,PH,PW,64] bf16 channels_last + argmax_u8 [B,PH,PW,64], PH = (OH-1)/2+1:
 *                                   tsg_bn_relu_pool_fwd(tsg_stem_conv_fwd(x, w), fp) bit for bit
 *   tsg_stem_conv_bn_relu_pool_bwd_reduce   partial[S][2][64] = {sum dy', sum dy' (y - mean)},
 *                                   S = tsg_stem_pool_bwd_num_partials (the sums of tsg_bn_relu_pool_bwd_reduce in another order)
 *   tsg_stem_conv_wrw_bn_pool       dw = tsg_stem_conv_wrw(x, tsg_bn_relu_pool_bwd_apply(dpool, argmax, y, bp)): the 537 MB
 *                                   gradient is staged tile by tile into the weight gradient's LDS image, never stored;
 *                                   xc = the stored stem output y [B,OH,OW,64] (read), or NULL (y re-evaluated)
 * x, w, ws as for tsg_stem_conv_fwd; fp / bp the packs of tsg_bn_finalize / tsg_bn_bwd_coeffs for C = 64. */
int tsg_stem_conv_stats(const void* x, const float* w, float* partial, int64_t B, int64_t H, int64_t W, void* ws,
                        size_t ws_bytes, void* stream);
int tsg_stem_conv_bn_relu_pool_fwd(const void* x, const float* w, const float* fp, void* ypool, void* argmax_u8,
                                   int64_t B, int64_t H, int64_t W, void* ws, size_t ws_bytes, void* stream);
int tsg_stem_pool_bwd_num_partials(int64_t B, int64_t H, int64_t W);
int tsg_stem_conv_bn_relu_pool_bwd_reduce(const void* x, const float* w, const void* dpool, const void* argmax_u8,
                                          const float* fp, float* partial, int64_t B, int64_t H, int64_t W, void* ws,
                                          size_t ws_bytes, void* stream);
int tsg_stem_conv_wrw_bn_pool(const void* x, const float* w, const void* xc, const void* dpool, const void* argmax_u8,
                              const float* bp, float* dw, int64_t B, int64_t H, int64_t W, void* ws, size_t ws_bytes,
                              void* stream);

/* Weight gradient of the 64 -> 64 channel 3x3 / stride 1 / padding 1 convolutions (ResNet-18 layer1:
 * BasicBlock.conv1 / conv2, furnace/base_model/resnet.py:24-29,36-53) — replaces the cuDNN
 * backward-filter call autograd makes for them.  x [B,H,W,64] and dy [B,H,W,64] bf16 channels_last;
 * dw fp32 in the channels_last weight layout [oc][kh][kw][ci] (fp32 accumulation, fixed summation
 * order).  ws: tsg_conv3x3_wrw_ws_bytes() bytes.  Forward and the data gradient stay on MIOpen. */
int tsg_conv3x3_wrw_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad,
                              int dilation, int groups);
size_t tsg_conv3x3_wrw_ws_bytes(void);
int tsg_conv3x3_wrw(const void* x, const void* dy, float* dw, int64_t B, int64_t H, int64_t W,
                    void* ws, size_t ws_bytes, void* stream);
/* Same result through the second kernel variant: tiles stay pixel-major in LDS and the K = pixel fragments
 * come from gfx950's transposing LDS read (ds_read_b64_tr_b16); bit-identical output is not guaranteed
 * between the variants (different summation order), each is deterministic run to run. */
int tsg_conv3x3_wrw_tr(const void* x, const void* dy, float* dw, int64_t B, int64_t H, int64_t W,
                       void* ws, size_t ws_bytes, void* stream);
/* Any C_in, C_out that are multiples of 64, stride 1 or 2 (every 3x3 convolution of the ResNet-18 context path, the
 * spatial path, the refines, the attention-refinement modules and the heads: resnet.py:24-29,36-53, bisenet
 * network.py:43-52,114-137,140-156): the kernel above over (oc tile, ci tile) pairs of 64 x 64 channels; for stride 2 the
 * x patch of a tile is 9 x 65 input pixels and the K fragments step two pixels.  x [B,Hin,Win,Cin], dy
 * [B,OH,OW,Cout] (OH = (Hin - 1) / stride + 1) bf16 channels_last; dw fp32 [Cout][3][3][Cin];
 * ws: tsg_conv3x3_wrw_gen_ws_bytes(...) bytes.  Deterministic. */
int tsg_conv3x3_wrw_gen_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad,
                                  int dilation, int groups);
size_t tsg_conv3x3_wrw_gen_ws_bytes(int64_t B, int64_t Hin, int64_t Win, int Cin, int Cout, int stride);
int tsg_conv3x3_wrw_gen(const void* x, const void* dy, float* dw, int64_t B, int64_t Hin, int64_t Win, int Cin, int Cout,
                        int stride, void* ws, size_t ws_bytes, void* stream);
/* out[ci][kh][kw][oc] (bf16) = w[oc][2-kh][2-kw][ci] (fp32 or bf16, the channels_last filter layout): the filter with
 * which the DATA gradient of a stride-1 / padding-1 3x3 convolution is itself a forward convolution of dy — what
 * autograd's cuDNN backward-data call computes for resnet.py:24-29 — so that it can run on the (faster) forward kernels. */
/* The same weight gradients when the convolution's input was relu(a x + b) of a tensor x that is still around — the
 * BatchNorm + ReLU in front of it, which tsg_conv3x3_c64_*_fwd(in_ab) applied on load without storing the result: x is
 * transformed the same way (same arithmetic and bf16 rounding as tsg_bn_apply_fwd) while its patch is staged.  in_ab:
 * fp32 [2][Cin] = rows a, b of the BatchNorm forward pack. */
int tsg_conv3x3_wrw_tr_norm(const void* x, const float* in_ab, const void* dy, float* dw, int64_t B, int64_t H, int64_t W,
                            void* ws, size_t ws_bytes, void* stream);
int tsg_conv3x3_wrw_gen_norm(const void* x, const float* in_ab, const void* dy, float* dw, int64_t B, int64_t Hin,
                             int64_t Win, int Cin, int Cout, int stride, void* ws, size_t ws_bytes, void* stream);

/* Forward of the 64 -> 64 channel 3x3 / stride 1 / padding 1 convolutions (ResNet-18 layer1, resnet.py:24-29,36-53) —
 * replaces the cuDNN forward call, and, fed dy and tsg_conv3x3_weight_rot180_t(w), the backward-data call.  x, y
 * [B,H,W,64] bf16 channels_last; w bf16 [oc][kh][kw][ci] (the channels_last filter layout); fp32 accumulation.
 * partial (may be NULL): [S][2][64] fp32 per-block sums / square sums of the bf16-rounded outputs, S =
 * tsg_conv3x3_c64_stats_partials(B, H, W) — the layout tsg_bn_finalize / tsg_bn_collapse take, so the BatchNorm that
 * follows does not re-read y for its statistics.  in_ab (may be NULL): [2][64] fp32 = the a, b rows of a BatchNorm
 * forward pack (tsg_bn_finalize): the convolution then reads relu(a x + b) of x — the BatchNorm + ReLU in front of it
 * (seg_oprs.py:39-46, resnet.py:36-46) applied while the input patch is staged, bit-equal to what tsg_bn_apply_fwd
 * would have written, without writing it; zero padding stays zero.  addend (may be NULL; stride 1, not together
 * with partial): [B,H,W,64] bf16, y = bf16(bf16(conv) + addend): the gradient of a residual block's skip connection
 * (resnet.py:48-52) added in the epilogue of the first convolution's data gradient. */
int tsg_conv3x3_c64_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad, int dilation,
                              int groups);
int tsg_conv3x3_c64_stats_partials(int64_t B, int64_t H, int64_t W);
int tsg_conv3x3_c64_fwd(const void* x, const void* w, void* y, float* partial, const float* in_ab, const void* addend,
                        int64_t B, int64_t H, int64_t W, void* stream);
/* The same for stride 2 (BiSeNet SpatialPath.conv_3x3_1 / conv_3x3_2, network.py:117-118): x [B,H,W,64] ->
 * y [B,OH,OW,64], OH = (H - 1) / 2 + 1; and its data gradient dx [B,H,W,64] from dy [B,OH,OW,64] and the transposed
 * filter wt = tsg_conv3x3_weight_rot180_t(w), evaluated per output parity (9 tap products per 2 x 2 block of dx, every
 * element written once, no atomics).  Both stream the large tensor once; fp32 accumulation, bf16 results. */
int tsg_conv3x3_c64_s2_stats_partials(int64_t B, int64_t H, int64_t W);
int tsg_conv3x3_c64_s2_fwd(const void* x, const void* w, void* y, float* partial, const float* in_ab, int64_t B,
                           int64_t H, int64_t W, void* stream);
int tsg_conv3x3_c64_s2_dgrad(const void* dy, const void* wt, void* dx, int64_t B, int64_t H, int64_t W,
                             void* stream);
/* The two data gradients with the BACKWARD SUMS of the BatchNorm -> ReLU in front of the convolution in their epilogue
 * (round 6): where the convolution's input was relu(bn(x)) (seg_oprs.py:39-46 followed by the next ConvBnRelu, bisenet
 * network.py:116-118; BasicBlock's bn1 -> relu -> conv2, resnet.py:36-46), the gradient dx computed here is the dy' that
 * SyncBN's backward starts from (syncbn_kernel.cu:160-174 with the ReLU mask folded in).  The thread that stores 16 bytes of
 * dx reads the 16 bytes of x beside them and accumulates sum dx m and sum dx m (x - mean), m = (a x + b > 0), from the
 * bf16-ROUNDED values it stores: partial [S][2][64] fp32 in the layout tsg_bn_bwd_coeffs takes, S = *_partials(B, H, W)
 * (0: shape not covered by this form, run tsg_bn_bwd_reduce instead).  tsg_bn_bwd_reduce(relu = 1, y = NULL) over the stored
 * dx computes the same sums in a pass of its own (one more read of dx and the same read of x); the two differ in the order
 * of the fp32 summation only.  bn_x [B,H,W,64] bf16: the BatchNorm's input; bn_fp: its forward pack [3][64] (a, b, mean). */
int tsg_conv3x3_c64_dgrad_bnsums_partials(int64_t B, int64_t H, int64_t W);
int tsg_conv3x3_c64_dgrad_bnsums(const void* dy, const void* wt, void* dx, const void* bn_x, const float* bn_fp,
                                 float* partial, int64_t B, int64_t H, int64_t W, void* stream);
int tsg_conv3x3_c64_s2_dgrad_partials(int64_t B, int64_t H, int64_t W);
int tsg_conv3x3_c64_s2_dgrad_bnsums(const void* dy, const void* wt, void* dx, const void* bn_x, const float* bn_fp,
                                    float* partial, int64_t B, int64_t H, int64_t W, void* stream);

int tsg_conv3x3_weight_rot180_t(const void* w, int dtype, void* out, int O, int I, void* stream);

/* ------------------------------------------------------------------------
 * General 3x3 / stride 1 / padding 1 convolution forward (csrc/conv3g.hip): C_in a multiple of 16, C_out a multiple
 * of 64 — ResNet layer2-4's conv3x3 (furnace/base_model/resnet.py:24-29,36-53), BiSeNet's attention-refinement 3x3,
 * refines and heads (bisenet network.py:43-52,140-156): the cuDNN forward calls of the reference and, fed dy and the
 * mode-1 filter, its cuDNN backward-data calls (the data gradient of a stride-1 3x3 convolution is the forward
 * convolution of dy with the rotated, transposed filter).
 *   tsg_conv3x3_gen_tile: output channels per block (64 or 128) chosen for a problem size; the prepared filter is laid
 *     out for that width, so pass the same value (BN) to the preparation, the partial count and the forward call.
 *   tsg_conv3x3_gen_prep_filter: master weight w [O][3][3][I] (fp32 or bf16, the channels_last filter layout) ->
 *     out: tsg_conv3x3_gen_filter_elems() bf16 in MFMA fragment order (csrc/conv3g.hip).  mode 0: the forward filter
 *     (the convolution then has C_out = O, C_in = I); mode 1: rot180 + transpose (C_out = I, C_in = O).
 *   tsg_conv3x3_gen_fwd: x [B,H,W,Cin] -> y [B,H,W,Cout], bf16 channels_last, fp32 accumulation.  partial (may be NULL):
 *     [S][2][Cout] fp32 sums / square sums of the bf16-rounded outputs, S = tsg_conv3x3_gen_stats_partials(...): the
 *     layout tsg_bn_finalize / tsg_bn_collapse take.  in_ab (may be NULL; Cin <= 512): [2][Cin] fp32 a, b rows of a
 *     BatchNorm forward pack: the convolution reads relu(a x + b) of x, bit-equal to tsg_bn_apply_fwd's output,
 *     zero padding stays zero.  addend (may be NULL; not together with partial): [B,H,W,Cout] bf16,
 *     y = bf16(bf16(conv) + addend) — used for the data gradient of a residual block's first convolution, where the
 *     gradient of the skip connection (resnet.py:48-52) is added in the epilogue instead of by a separate pass.
 *   tsg_conv3x3_gen_variant: which of the two kernels the forward call runs for a problem (0: 8-row pixel tiles; 1: the
 *     16-row tiles with all staging by LDS-DMA, taken when they fill the GPU and in_ab is not given) — informational, for
 *     tests and benchmarks; results and the partial's row count do not depend on it beyond fp32 summation order.
 * ---------------------------------------------------------------------- */
int tsg_conv3x3_gen_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad, int dilation,
                              int groups);
int64_t tsg_conv3x3_gen_filter_elems(int Cin, int Cout);
int tsg_conv3x3_gen_tile(int64_t B, int64_t H, int64_t W, int Cin, int Cout);
int tsg_conv3x3_gen_prep_filter(const void* w, int dtype, void* out, int O, int I, int mode, int BN, void* stream);
int tsg_conv3x3_gen_stats_partials(int64_t B, int64_t H, int64_t W, int Cin, int Cout, int BN);
int tsg_conv3x3_gen_variant(int64_t B, int64_t H, int64_t W, int Cin, int Cout, int BN, int with_in_ab);
/* Data gradient of the 3x3 / STRIDE 2 / padding 1 convolutions C_in -> C_out (ResNet's first block of layer2-4,
 * furnace/base_model/resnet.py:24-29,36-53): the cuDNN backward-data call of the reference plus, with `addend`, the
 * gradient accumulation of the shortcut branch (resnet.py:48-52).  dy [B,OH,OW,C_out] -> dx [B,H,W,C_in], bf16
 * channels_last, OH = (H - 1) / 2 + 1; wf = tsg_conv3x3_gen_prep_filter(w, ..., mode 1, BN 32); addend (may be NULL)
 * [B,H,W,C_in] bf16: dx = bf16(bf16(dgrad) + addend).  C_in, C_out multiples of 32.  Computed by output parity (1 / 2 / 2 /
 * 4 taps), no zero-stuffed operand; fp32 accumulation. */
int tsg_conv3x3_s2_dgrad_supported(int dtype, int Cin, int Cout);
int tsg_conv3x3_s2_dgrad(const void* dy, const void* wf, void* dx, const void* addend, int64_t B, int64_t H, int64_t W,
                         int Cin, int Cout, void* stream);
/* The same with a COMPACT addend (round 6): addend_sub [B,OH,OW,C_in] bf16 on dy's grid — the gradient of x[:, :, ::2, ::2],
 * i.e. of the block's 1x1 / stride-2 shortcut convolution (resnet.py:139-146) run as a stride-1 convolution of the
 * sub-sampled map — is added at the even pixels of dx; the zero-filled full-size gradient the cuDNN backward-data call of
 * that shortcut would write (and this epilogue read back) is never made. */
int tsg_conv3x3_s2_dgrad_subadd(const void* dy, const void* wf, void* dx, const void* addend_sub, int64_t B, int64_t H,
                                int64_t W, int Cin, int Cout, void* stream);
int tsg_conv3x3_gen_fwd(const void* x, const void* wf, void* y, float* partial, const float* in_ab, const void* addend,
                        int64_t B, int64_t H, int64_t W, int Cin, int Cout, int BN, void* stream);

/* ------------------------------------------------------------------------
 * OHEM 2-D cross entropy — replaces ProbOhemCrossEntropy2d.forward
 * (furnace/seg_opr/loss_opr.py:68-98) and the nn.CrossEntropyLoss it ends in.
 * logits are [B, C, HW] (NCHW planar), labels [B*HW].
 * ---------------------------------------------------------------------- */

typedef struct tsg_ohem_plan {
  int64_t P;          /* B*HW pixels                                   */
  int     C;
  int     grid;       /* blocks of the pixel passes                     */
  int     levels;     /* radix-select refinement levels (1..3)          */
  int     shift[3];   /* bit shift of each level                        */
  int     bins[3];    /* bins of each level                             */
  uint32_t thresh_bits;
  size_t  ws_bytes;   /* workspace the caller must provide (zeroed by us) */
} tsg_ohem_plan;

/* Fill `plan` for a problem size / threshold. */
int tsg_ohem_make_plan(int64_t B, int C, int64_t HW, float thresh,
                       tsg_ohem_plan* plan);

/* Forward.  Writes nll[P] (= lse - x_t, 0 for ignored pixels), lse[P],
 * loss[1] (fp32 mean over kept pixels, NaN when none) and
 * sel[8] = {thr (float bits), n_kept, num_valid, branch, denominator (float
 * bits), n_bad, 0, 0}; branch 0: thr == thresh, 1: thr == k-th smallest p_t,
 * 2: no hard-example mining applied (loss_opr.py:78-80, :85).  n_bad counts labels that are neither
 * ignore_label nor in [0, C): the reference device-asserts on them (prob[target], loss_opr.py:82-83); here they
 * are dropped like ignored pixels, never used as an index, and reported so the host can raise.
 * min_kept / thresh / ignore_label as in loss_opr.py:49-56.
 * weight (class weights, loss_opr.py:57-63) may be NULL. */
int tsg_ohem_fwd(const void* logits, int dtype, const void* labels, int ltype,
                 int64_t B, int C, int64_t HW,
                 int64_t ignore_label, float thresh, int64_t min_kept,
                 const float* weight,
                 float* nll, float* lse, float* loss, int32_t* sel,
                 void* ws, size_t ws_bytes, void* stream);

/* Backward: dlogits[b,c,p] = gscale * w_t * (softmax_c - [c==t]) / denom for
 * kept pixels, 0 elsewhere; gscale is a device scalar (upstream grad). */
int tsg_ohem_bwd(const void* logits, int dtype, const void* labels, int ltype,
                 int64_t B, int C, int64_t HW, int64_t ignore_label,
                 const float* weight,
                 const float* nll, const float* lse, const int32_t* sel,
                 const float* gscale, void* dlogits, void* ws, void* stream);

/* Fused head: the criterion applied to
 *   F.interpolate(z, size=(OH,OW), mode='bilinear', align_corners=True)
 * (bisenet network.py:104-106 with :164-166) WITHOUT materialising the
 * full-resolution logits: z [B, C, IH, IW] is interpolated inside the OHEM
 * kernels (forward: per-tile LDS window of z; backward: dz accumulated through
 * the transposed taps, deterministically).  Outputs as tsg_ohem_fwd / dz [B,C,IH,IW].
 * tsg_ohem_up_supported() != 0 when the kernels can take the shape (C <= 32,
 * genuine up-sampling by >= 2 and <= ~16 per axis).  The backward needs a
 * workspace of tsg_ohem_up_bwd_ws_bytes() (the vertically reduced gradient
 * V[B,C,IH,OW] in fp32). */
int tsg_ohem_up_supported(int C, int IH, int IW, int OH, int OW, float thresh);
size_t tsg_ohem_up_bwd_ws_bytes(int64_t B, int C, int IH, int OW);
int tsg_ohem_up_fwd(const void* z, int dtype, const void* labels, int ltype,
                    int64_t B, int C, int IH, int IW, int OH, int OW,
                    int64_t ignore_label, float thresh, int64_t min_kept,
                    const float* weight,
                    float* nll, float* lse, float* loss, int32_t* sel,
                    void* ws, size_t ws_bytes, void* stream);
int tsg_ohem_up_bwd(const void* z, int dtype, const void* labels, int ltype,
                    int64_t B, int C, int IH, int IW, int OH, int OW,
                    int64_t ignore_label, const float* weight,
                    const float* nll, const float* lse, const int32_t* sel,
                    const float* gscale, void* dz,
                    void* ws, size_t ws_bytes, void* stream);

/* prob[P] = the target-class probability the selection ranks (mask_prob of loss_opr.py:81-83): exp(-nll) evaluated
 * by the same device expression the OHEM kernels use, 1 for ignored pixels.  With it the selection contract is
 * checkable bit for bit: thr == sort(prob)[k-1] and kept == valid & (prob <= thr). */
int tsg_ohem_target_prob(const float* nll, const void* labels, int ltype, int64_t P, int C,
                         int64_t ignore_label, float* prob, void* stream);

/* Exact k-th order statistic of non-negative floats by radix select on their
 * IEEE bit patterns — what torch.sort(mask_prob)[k-1] returns
 * (loss_opr.py:86-88).  out[0] receives the value.  k is 1-based. */
size_t tsg_kth_ws_bytes(int64_t n);
int tsg_kth_value(const float* v, int64_t n, int64_t k, float* out,
                  void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Sigmoid focal loss — replaces SigmoidFocalLoss.forward
 * (furnace/seg_opr/loss_opr.py:23-45), including its use of the sigmoid where
 * logits were intended (line 32-39).  pred [B*HW], target [B*HW].
 * ---------------------------------------------------------------------- */
size_t tsg_focal_ws_bytes(int64_t P);
int tsg_focal_fwd(const void* pred, int dtype, const void* target, int ltype,
                  int64_t P, int64_t ignore_label, float gamma, float alpha,
                  float* loss, void* ws, size_t ws_bytes, void* stream);
int tsg_focal_bwd(const void* pred, int dtype, const void* target, int ltype,
                  int64_t P, int64_t ignore_label, float gamma, float alpha,
                  const float* gscale, void* dpred, void* stream);

/* ------------------------------------------------------------------------
 * Bilinear upsample, align_corners=True — replaces the
 * F.interpolate(..., mode='bilinear', align_corners=True) call sites
 * (bisenet network.py:82-84,93-94,164-166; pspnet network.py:46-49,103-105)
 * i.e. aten::upsample_bilinear2d / _backward.  x [NC, IH, IW] -> y [NC, OH, OW]
 * planar; `add` (may be NULL, shaped like y) is added to the result (dfn
 * network.py:130-133: `last_fm + F.interpolate(fm)`).
 * ---------------------------------------------------------------------- */
int tsg_upsample_bilinear_ac_fwd(const void* x, const void* add, void* y,
                                 int dtype, int64_t NC, int IH, int IW,
                                 int OH, int OW, void* stream);
int tsg_upsample_bilinear_ac_bwd(const void* dy, void* dx, int dtype,
                                 int64_t NC, int IH, int IW, int OH, int OW,
                                 void* stream);
/* channels_last variants for feature maps: x [N, IH, IW, C] -> y [N, OH, OW, C],
 * C % (16 / elem_size) == 0 (a thread owns one 16-byte channel vector). */
int tsg_upsample_bilinear_ac_nhwc_fwd(const void* x, const void* add, void* y,
                                      int dtype, int64_t N, int C, int IH, int IW,
                                      int OH, int OW, void* stream);
int tsg_upsample_bilinear_ac_nhwc_bwd(const void* dy, void* dx, int dtype,
                                      int64_t N, int C, int IH, int IW,
                                      int OH, int OW, void* stream);
/* "upsample+fuse" the way the reference orders it (bisenet network.py:91-95: `fm += last_fm`, then
 * F.interpolate(fm, ...)): y = up(round_T(x + x2)), x2 shaped like x.  The sum is formed while the taps are read and
 * never written to HBM; it is rounded to the element type exactly as the eager in-place add would have stored it.
 * Backward: both addends receive tsg_upsample_bilinear_ac{,_nhwc}_bwd(dy). */
int tsg_upsample_bilinear_ac_presum_fwd(const void* x, const void* x2, void* y,
                                        int dtype, int64_t NC, int IH, int IW,
                                        int OH, int OW, void* stream);
int tsg_upsample_bilinear_ac_nhwc_presum_fwd(const void* x, const void* x2, void* y,
                                             int dtype, int64_t N, int C, int IH, int IW,
                                             int OH, int OW, void* stream);
/* Half-pixel-centre bilinear resize of planar float / bf16 maps into fp32 (cv2.resize INTER_LINEAR on float data ==
 * F.interpolate(align_corners=False)): the score resize of the evaluator (furnace/engine/evaluator.py:250-252);
 * accumulate != 0 adds into y (the sum over scales, evaluator.py:196-199).  x [NC, IH, IW] -> y [NC, OH, OW]. */
int tsg_resize_bilinear_hp(const void* x, float* y, int dtype, int64_t NC, int IH, int IW, int OH, int OW,
                           int accumulate, void* stream);
/* nearest (floor(dst*in/out)) variant used for label maps. */
int tsg_upsample_nearest_fwd(const void* x, void* y, int elem_bytes,
                             int64_t NC, int IH, int IW, int OH, int OW,
                             void* stream);

/* ------------------------------------------------------------------------
 * PSANet collect / distribute attention — replaces
 *   torch.bmm(X, torch.softmax(A, dim=1))
 * (model/psanet/ade.psanet.R101_v1c/network.py:125-126,135-136).
 * X [B, Cx, K], A [B, K, N] (softmax over dim 1 = the K rows of each column),
 * out [B, Cx, N]; lse [B, N] (fp32 log-sum-exp of every column, saved for the
 * backward).  K, N, Cx must be multiples of 8 (3600 and 512 in PSANet).
 * dtype TSG_BF16: bf16 MFMA; TSG_F32: the same MFMA kernels with bf16 hi/lo
 * operand splitting (3 passes) to hold the 1e-4 fp32 parity bar.
 * ---------------------------------------------------------------------- */
size_t tsg_psa_ws_bytes(int dtype, int backward, int64_t B, int64_t Cx,
                        int64_t K, int64_t N);
int tsg_psa_fwd(const void* X, const void* A, void* out, float* lse,
                int dtype, int64_t B, int64_t Cx, int64_t K, int64_t N,
                void* ws, size_t ws_bytes, void* stream);
/* dX [B, Cx, K], dA [B, K, N] from dout [B, Cx, N]:
 *   dX = dout * P^T ; dA = P o (X^T dout - delta), delta_j = sum_c out*dout. */
int tsg_psa_bwd(const void* X, const void* A, const void* out, const void* dout,
                const float* lse, void* dX, void* dA,
                int dtype, int64_t B, int64_t Cx, int64_t K, int64_t N,
                void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * Evaluation metric — replaces hist_info (furnace/seg_opr/metric.py:9-19) as the
 * evaluators call it (model/bisenet/cityscapes.bisenet.R18/eval.py:31-33) on the
 * arg-max of the score map (furnace/engine/evaluator.py).  out is int64
 * [n_cl*n_cl + 3] = hist (row = ground truth, column = prediction), labeled,
 * correct, and the number of labelled pixels whose prediction is outside [0, n_cl)
 * (numpy would raise there); the call ADDS to out, so one zeroed buffer accumulates
 * a whole validation set.  gt values outside [0, n_cl) (255, -1) are skipped.
 * n_cl <= 192.  tsg_confusion_logits fuses the class arg-max over logits [B, C, HW]
 * (first maximum, NaN wins: numpy/torch rule).
 * ---------------------------------------------------------------------- */
int tsg_confusion_map(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int64_t P,
                      int n_cl, int64_t* out, void* stream);
int tsg_confusion_logits(const void* logits, int dtype, const void* gt, int gt_dtype, int64_t B,
                         int C, int64_t HW, int n_cl, int64_t* out, void* stream);

/* ------------------------------------------------------------------------
 * Fused SGD step over a flat parameter bucket (torch.optim.SGD semantics as
 * configured at train.py:86-89: momentum, weight decay, no nesterov).
 * ---------------------------------------------------------------------- */
int tsg_sgd_step(float* param, const float* grad, float* momentum_buf,
                 int64_t n, float lr, float momentum, float weight_decay,
                 float grad_scale, int first_step, void* stream);

/* Same update with lr = lr_dev[0] * lr_mult read on the device (momentum_buf must
 * start zeroed: buf = momentum*buf + g reproduces torch's first-step buf = g), so a
 * captured hipGraph follows a per-iteration schedule (train.py:133-139). */
int tsg_sgd_step_dev(float* param, const float* grad, float* momentum_buf,
                     int64_t n, const float* lr_dev, float lr_mult, float momentum,
                     float weight_decay, float grad_scale, void* stream);

/* Multi-tensor form of tsg_sgd_step_dev: one launch updates up to TSG_SGD_MAX_SEGS parameter
 * tensors (the ~113 of BiSeNet-R18 in one launch instead of one each; torch's foreach path
 * issues ~7 launches per parameter group).  params/grads/bufs: host arrays of device addresses;
 * numel, group: host arrays; lr_dev [ngroups] on the device; momentum / weight_decay: host
 * arrays per group.  blockmap_dev: device copy of the int pairs tsg_sgd_multi_blockmap writes
 * (static for a model); call it with map_host = NULL to get the block count. */
#define TSG_SGD_MAX_SEGS   128
#define TSG_SGD_MAX_GROUPS 24
int64_t tsg_sgd_multi_blockmap(const int64_t* numel, int nseg, int* map_host, int64_t cap_blocks);
int tsg_sgd_multi_step_dev(const uint64_t* params, const uint64_t* grads, const uint64_t* bufs,
                           const int64_t* numel, const int* group, int nseg, const float* lr_dev,
                           const float* momentum, const float* weight_decay, int ngroups,
                           const int* blockmap_dev, int64_t nblocks, float grad_scale, void* stream);

/* Multi-tensor copy dst[i][:] = scale * src[i][:] (fp32), one launch for up to TSG_SGD_MAX_SEGS
 * tensors: gathers the gradients autograd produced into the flat all-reduce bucket of the DDP
 * wrapper (apex.parallel.DistributedDataParallel flattens them with apex_C.flatten, reference
 * train.py:98-99).  Same block map as tsg_sgd_multi_step_dev (tsg_sgd_multi_blockmap). */
int tsg_multi_copy_f32(const uint64_t* src, const uint64_t* dst, const int64_t* numel, int nseg,
                       const int* blockmap_dev, int64_t nblocks, float scale, void* stream);

/* bf16 shadows of the fp32 master filters, all refreshed by ONE launch (after the optimizer step) instead of one cast
 * (+ one rotate/transpose) launch per convolution and step.  table_dev: device array of
 *   struct { const float* w; uint16_t* wb; uint16_t* wrt; int32_t n, O, I, pad; }   (tsg_weight_shadow_entry_bytes())
 * wb[i] = bf16(w[i]); for channels_last 3x3 filters (memory [O][kh][kw][I]) and wrt != NULL also
 * wrt[ci][2-kh][2-kw][o] = bf16(w[o][kh][kw][ci]) — what tsg_conv3x3_weight_rot180_t writes.  blockmap_dev / nblocks:
 * tsg_sgd_multi_blockmap over the entries' n. */
size_t tsg_weight_shadow_entry_bytes(void);
int tsg_weight_shadow_refresh(const void* table_dev, const int* blockmap_dev, int64_t nblocks, void* stream);

/* ------------------------------------------------------------------------
 * Training pre-processing on the GPU — replaces TrainPre.__call__ of the reference's dataloaders
 * (model/bisenet/cityscapes.bisenet.R18/dataloader.py:16-35) = furnace/utils/img_utils.py random_mirror (:138-143),
 * random_scale (:110-117: cv2 INTER_LINEAR image / INTER_NEAREST label), normalize (:174-180),
 * random_crop_pad_to_shape (:24-40, :60-75) and the transpose / float / long of datasets/BaseDataset.py:47-48,
 * as one kernel: each output pixel of the padded crop is sampled straight from the uint8 source image.
 *   imgs[i] uint8 [H][W][3], gts[i] uint8 [H][W] (device pointers, host array of n <= tsg_augment_max_samples());
 *   geom[7*i..] = {H, W, SH, SW, flip, crop_y, crop_x}: SH = int(H*scale), SW = int(W*scale), crop position in the
 *   scaled image (host-drawn with the reference's `random` call sequence);
 *   out_img float [n][3][CH][CW] = (v/255 - mean[c]) / std[c]; the padding holds pad_pixel < 0 ? 0 (TrainPre: the
 *   NORMALISED image is padded with 0) : (pad_pixel/255 - mean[c]) / std[c] (the evaluator pads the RAW image with 0,
 *   evaluator.py:217-218); out_gt [n][CH][CW] int64 (TSG_I64) or uint8 (TSG_U8), pad_label in the padding; gts / out_gt
 *   may be NULL (evaluation: image only).
 * ---------------------------------------------------------------------- */
int tsg_augment_max_samples(void);
int tsg_augment_crop(const void* const* imgs, const void* const* gts, const int32_t* geom, const double* inv_scale, int n,
                     int CH, int CW, const float* mean, const float* std, float pad_pixel, int pad_label, float* out_img,
                     void* out_gt, int gt_type, void* stream);

/* ------------------------------------------------------------------------
 * Collectives of the hot path — replace the exchange steps of the reference's SyncBN / DDP:
 * furnace/legacy/sync_bn/syncbn.py:75-78 (ReduceAddCoalesced / Broadcast of [sum x, sum x^2]),
 * furnace/legacy/sync_bn/comm.py:57-132 (the master/slave pipes carrying them), and the
 * torch.distributed all_reduce / all_gather / broadcast calls inside apex.parallel.SyncBatchNorm /
 * DistributedDataParallel (train.py:24-25,98-99).
 *
 * A tsg_comm is the one piece of state the library holds: an RCCL communicator (one rank per GPU) plus,
 * optionally, peer-mapped mailboxes for the one-shot small-message all-reduce.  Every collective is enqueued on
 * the CALLER's stream (no internal stream, no host synchronisation), in place, SUM.  librccl.so is resolved with
 * dlopen at first use — the copy the host framework already loaded if there is one.
 * ---------------------------------------------------------------------- */
typedef struct tsg_comm tsg_comm;

/* Optional: load librccl from an explicit path (NULL = default search). */
int tsg_comm_init_library(const char* librccl_path);
/* Bootstrap: rank 0 fills `id_out` (tsg_comm_unique_id_bytes() bytes) and hands it to the other ranks by any
 * means (the Python host uses the torch.distributed store); every rank then calls tsg_comm_create with it.
 * unique_id == NULL creates a communicator WITHOUT RCCL, usable only for the mailbox path below. */
int tsg_comm_unique_id_bytes(void);
int tsg_comm_get_unique_id(void* id_out);
int tsg_comm_create(const void* unique_id, int rank, int world, int device, tsg_comm** out);
int tsg_comm_destroy(tsg_comm* c);
int tsg_comm_rank(const tsg_comm* c);
int tsg_comm_world(const tsg_comm* c);
/* buf[count] <- sum over ranks (dtype TSG_F32 / TSG_BF16). */
int tsg_comm_allreduce(tsg_comm* c, void* buf, int64_t count, int dtype, void* stream);
/* recv[world * count_per_rank] <- concatenation of every rank's send[count_per_rank]. */
int tsg_comm_allgather(tsg_comm* c, const void* send, void* recv, int64_t count_per_rank, int dtype, void* stream);
/* recv[count_per_rank] <- this rank's slice of the sum over ranks of send[world * count_per_rank] (first half of the
 * reduce-scatter + all-gather form of the gradient-bucket exchange; recv may alias
 * send + rank * count_per_rank). */
int tsg_comm_reduce_scatter(tsg_comm* c, const void* send, void* recv, int64_t count_per_rank, int dtype, void* stream);
/* buf[count] <- root's buf (parameter broadcast at wrap time, apex DDP). */
int tsg_comm_broadcast(tsg_comm* c, void* buf, int64_t count, int dtype, int root, void* stream);
const char* tsg_comm_error_string(int code);

/* One-shot all-reduce for the SyncBN statistics messages (2C+2 floats, 105-390 per step):
 * every rank stores its vector straight into every peer's mailbox over xGMI, flags it, waits for the peers' flags
 * and sums the world slots in rank order — one kernel on the compute stream, one hop of latency, bit-identical
 * results on all ranks.  Setup: each rank calls tsg_comm_xgmi_export (allocates its mailbox for messages of up to
 * max_floats and returns an IPC handle of tsg_comm_xgmi_handle_bytes() bytes), the handles are all-gathered in rank
 * order by the host, then tsg_comm_xgmi_attach maps the peers.  tsg_xgmi_small_allreduce uses the mailboxes when they
 * are attached and count fits; otherwise it is ncclAllReduce on the same stream.  The call counter that selects the
 * mailbox parity and the flag value lives in device memory and is advanced by the kernel, so the launch may be
 * captured in a hipGraph and replayed. */
size_t tsg_comm_xgmi_handle_bytes(void);
int tsg_comm_xgmi_export(tsg_comm* c, int64_t max_floats, void* handle_out);
int tsg_comm_xgmi_attach(tsg_comm* c, const void* all_handles);
int tsg_xgmi_small_allreduce(tsg_comm* c, float* buf, int64_t count, void* stream);

/* ------------------------------------------------------------------------
 * Reference-accuracy fp32 convolution for the PARITY path (csrc/convf32.hip) — replaces the cuDNN calls behind every
 * `nn.Conv2d` of the reference (furnace/base_model/resnet.py:24-29,96-97; furnace/seg_opr/seg_oprs.py:27-31) when the
 * package computes in fp32 (the parity mode): products exact, accumulation in fp64, one rounding to fp32 at the store.
 * With them BiSeNet-R18's full-resolution fp32 logits are 1.3-1.9e-5 from the float64 evaluation of the network (the
 * reference's CPU path: 6.9-8.3e-5; the vendor library's fp32 kernels: 4.8-6.3e-5; tools/diag_fp64_truth.py), i.e. the
 * 1e-4 of north_star against the CPU path is the CPU's own rounding.  Any kernel size,
 * stride, padding, dilation; groups = 1; every tensor is addressed through four ELEMENT strides (x / dx: b, c, h, w;
 * w / dw: o, c, kh, kw; y / dy: b, o, oh, ow), so NCHW and channels_last need no copies.  H, W: input size.
 * Deterministic (fixed summation order).  Not a fast path: fp64 FMAs. */
int tsg_conv2d_f32_exact_fwd(const float* x, const float* w, float* y, int64_t B, int Cin, int H, int W, int Cout, int KH,
                             int KW, int sh, int sw, int ph, int pw, int dh, int dw, const int64_t* x_strides,
                             const int64_t* w_strides, const int64_t* y_strides, void* stream);
int tsg_conv2d_f32_exact_dgrad(const float* dy, const float* w, float* dx, int64_t B, int Cin, int H, int W, int Cout, int KH,
                               int KW, int sh, int sw, int ph, int pw, int dh, int dw, const int64_t* dx_strides,
                               const int64_t* w_strides, const int64_t* dy_strides, void* stream);
/* the weight gradient splits the pixels into slices (fp64 partials in `ws`, added in slice order): ws_bytes may be 0 */
size_t tsg_conv2d_f32_exact_wgrad_ws_bytes(int64_t B, int Cin, int H, int W, int Cout, int KH, int KW, int sh, int sw, int ph,
                                           int pw, int dh, int dw);
int tsg_conv2d_f32_exact_wgrad(const float* x, const float* dy, float* dw_out, int64_t B, int Cin, int H, int W, int Cout,
                               int KH, int KW, int sh, int sw, int ph, int pw, int dh, int dw, const int64_t* x_strides,
                               const int64_t* w_strides, const int64_t* dy_strides, void* ws, size_t ws_bytes,
                               void* stream);

/* ------------------------------------------------------------------------
 * Classifier convolution of a segmentation head (csrc/clshead.hip) — replaces the cuDNN calls behind
 * `nn.Conv2d(C_in, n_classes, kernel_size=1)` + bias of bisenet network.py:151-161 (BiSeNetHead.conv_1x1; the same layer
 * ends the 19-class heads of dfn network.py) and of its backward.  x / dx: channels_last bf16 [B, HW, C_in]; w: fp32
 * [n_classes, C_in] (rounded to bf16 inside, as autocast would); bias fp32 or NULL; z / dz: PLANAR bf16 [B, n_classes, HW] —
 * the layout tsg_ohem_up_fwd / tsg_ohem_fwd read, so no layout copy sits between the head and its criterion.
 * n_classes <= 32, C_in in {32, 64, 128, 256}, HW % 16 == 0.  fp32 accumulation; the weight / bias gradient is folded
 * from per-block partials in fp64 in a fixed order (deterministic).  ws: tsg_cls_head_wgrad_ws_bytes. */
int tsg_cls_head_supported(int dtype, int Cin, int n_classes, int64_t HW);
int tsg_cls_head_fwd(const void* x, const float* w, const float* bias, void* z, int64_t B, int64_t HW, int Cin, int n_classes,
                     void* stream);
int tsg_cls_head_dgrad(const void* dz, const float* w, void* dx, int64_t B, int64_t HW, int Cin, int n_classes, void* stream);
size_t tsg_cls_head_wgrad_ws_bytes(int64_t B, int Cin, int n_classes);
int tsg_cls_head_wgrad(const void* dz, const void* x, float* dw, float* dbias, int64_t B, int64_t HW, int Cin, int n_classes,
                       void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * 1x1 convolution of a globally pooled map (csrc/vecconv.hip) — replaces the cuDNN calls behind the bias-free
 * `ConvBnRelu(C_in, C_out, 1, 1, 0)` layers that follow `nn.AdaptiveAvgPool2d(1)`: furnace/seg_opr/seg_oprs.py:199-205
 * (AttentionRefinement.channel_attention), :222-231 (FeatureFusion.channel_attention), bisenet network.py:34-39
 * (global_context), and of their backward.  x / dx: bf16 [B, C_in]; y / dy: bf16 [B, C_out]; w: fp32 [C_out, C_in]
 * (rounded to bf16 inside, as autocast would), dw: fp32 [C_out, C_in] (not rounded).  B <= 32; C_in, C_out multiples
 * of 16.  fp32 accumulation, fixed summation order.  dx may be NULL (input needs no gradient). */
int tsg_conv1x1_vec_supported(int B, int Cin, int Cout);
int tsg_conv1x1_vec_fwd(const void* x, const float* w, void* y, int B, int Cin, int Cout, void* stream);
int tsg_conv1x1_vec_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, int B, int Cin, int Cout,
                        void* stream);
/* The whole pooled layer in one launch per direction (round 6): convolution -> [BatchNorm over the batch] -> [ReLU | sigmoid]
 * — `ConvBnRelu(C_in, C_out, 1, 1, 0, has_bn, has_relu)` on a [B, C, 1, 1] map and the `nn.Sigmoid()` that ends the attention
 * branches (seg_oprs.py:199-205, :222-231; bisenet network.py:34-39).  bnmode 0 none / 1 batch statistics (running_mean /
 * running_var / num_batches_tracked updated when given: momentum, unbiased variance, syncbn.py:86-98) / 2 running statistics;
 * act 0 none / 1 ReLU / 2 sigmoid.  out: the layer's output bf16 [B, C_out]; yc: the convolution's output (bf16, what the
 * unfused path stores between convolution and BatchNorm; bnmode != 0); stats fp32 [4][C_out] = {a = gamma invstd,
 * b = beta - mean a, mean, invstd}.  Arithmetic and rounding points of tsg_conv1x1_vec_fwd -> tsg_bn_stats -> tsg_bn_finalize
 * -> tsg_bn_apply_fwd -> at::sigmoid (the B-term sums directly in fp64).  Backward: dout = gradient w.r.t. out;
 * dx (may be NULL), dw fp32, dgamma / dbeta fp32 [C_out] (bnmode != 0; may be NULL). */
int tsg_conv1x1_vec_bnact_fwd(const void* x, const float* w, void* out, void* yc, float* stats, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                              float eps, float momentum, int bnmode, int act, int B, int Cin, int Cout, void* stream);
int tsg_conv1x1_vec_bnact_bwd(const void* dout, const void* out, const void* yc, const float* stats, const void* x,
                              const float* w, void* dx, float* dw, float* dgamma, float* dbeta, int bnmode, int act, int B,
                              int Cin, int Cout, void* stream);

/* ------------------------------------------------------------------------
 * DFN's border labels — replaces, on the GPU, lines 24-29 of model/dfn/cityscapes.dfn.R101_v1c/dataloader.py
 * (TrainPre.__call__: cv2.Canny(no255_gt, 5, 5, apertureSize=7) -> cv2.dilate(7 x 7) -> 255 becomes 1 ->
 * random_crop_pad_to_shape(..., 255)) for one sample: gt uint8 [H][W] on the device, geom = {H, W, SH, SW, flip, crop_y,
 * crop_x} as for tsg_augment_crop, out [CH][CW] int64 / uint8 with values {0, 1, pad_label}.  aperture must be 7 and
 * threshold1 == threshold2 (the reference's only use).  Arithmetic of torchseg_amd/shims_optional/cv2 (OpenCV's documented
 * Canny); parity with OpenCV's own rounding is unpinned (no cv2 in the build image). */
size_t tsg_edge_labels_ws_bytes(int SH, int SW);
int tsg_edge_labels(const void* gt, const int32_t* geom, const double* inv_scale, int CH, int CW, int ignore_label,
                    int threshold1, int threshold2, int aperture, int dilate_size, int pad_label, void* out, int out_type,
                    void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TSG_HIP_H */
