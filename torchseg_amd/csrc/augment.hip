// The training pre-processing of the reference's dataloaders as ONE pass on the GPU.
//
// Restates TrainPre.__call__ (model/bisenet/cityscapes.bisenet.R18/dataloader.py:16-35; same in pspnet / psanet / dfn)
// built from furnace/utils/img_utils.py:
//   random_mirror (:138-143)              horizontal flip of image and label
//   random_scale (:110-117)               sh = int(H * s), sw = int(W * s); image cv2.INTER_LINEAR, label cv2.INTER_NEAREST
//   normalize (:174-180)                  (img / 255 - mean) / std
//   random_crop_pad_to_shape (:24-40)     crop at crop_pos (clipped by the array), centred padding (pad_image_to_shape
//                                         :60-75: margin = pad // 2 before, pad // 2 + pad % 2 after), 0 for the
//                                         normalised image, 255 for the label
//   .transpose(2, 0, 1) and BaseDataset.py:47-48 (.float() / .long())
// The random draws (flip, scale, crop position) are made on the HOST with the reference's own call sequence on Python's
// `random`, so a seeded run selects the same crops; they arrive here as per-sample parameters.
//
// cv2 semantics restated (cv2 itself is not in this image: parity with cv2's 11-bit fixed-point rounding of the uint8
// resize is unpinned; the numpy oracle oracle/augment_ref.py follows the same formulas in float64):
//   INTER_LINEAR: src = (dst + 0.5) * (in / out) - 0.5; i0 = floor(src); w = src - i0; i0 < 0 -> (0, w = 0);
//                 i0 >= in - 1 -> (in - 1, w = 0); the uint8 result is the interpolated value rounded to nearest.
//   INTER_NEAREST: src = min(floor(dst * (in / out)), in - 1).
// One thread per output pixel: 4 taps x 3 bytes in, 3 floats + 1 label out; no intermediate image ever exists.
#include "tsg_common.h"

namespace tsg {

constexpr int kAugMax = 16;          // samples per launch (parameters travel by value in the kernel arguments)

struct AugSample {
  const uint8_t* img;                // [H][W][3], channel order as the network expects it (BaseDataset.py:45 flips BGR)
  const uint8_t* gt;                 // [H][W]
  int H, W, SH, SW;                  // source size, scaled size
  int flip;
  int crop_y, crop_x;                // crop position in the scaled image
  int top, left;                     // padding margins of the crop inside the output
  int ch, cw;                        // rows / columns of the scaled image that the crop actually covers
  double fy, fx;                     // source step per scaled pixel (OpenCV: 1 / inv_scale)
};

struct AugBatch {
  AugSample s[kAugMax];
  int n;
  int CH, CW;                        // output crop size
  float mean[3], inv_std[3], pad_img[3];
  int pad_label;
  int with_gt;
};

__device__ __forceinline__ void lin_index(int dst, double scale, int in, int& i0, int& i1, float& w) {
  const double src = ((double)dst + 0.5) * scale - 0.5;        // cv2 computes this in double, then narrows the weight
  int s = (int)floor(src);
  float f = (float)(src - (double)s);
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= in - 1) { s = in - 1; f = 0.f; }
  i0 = s; i1 = s + 1 < in ? s + 1 : in - 1; w = f;
}

template <typename LT>
__global__ __launch_bounds__(256) void augment_crop_k(AugBatch b, float* __restrict__ out_img, LT* __restrict__ out_gt) {
  const int sidx = blockIdx.z;
  const AugSample& p = b.s[sidx];
  const int ox = blockIdx.x * 256 + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= b.CW) return;
  const int64_t plane = (int64_t)b.CH * b.CW;
  float* oi = out_img + (int64_t)sidx * 3 * plane + (int64_t)oy * b.CW + ox;
  LT* og = b.with_gt ? out_gt + (int64_t)sidx * plane + (int64_t)oy * b.CW + ox : nullptr;
  const int iy = oy - p.top, ix = ox - p.left;
  if (iy < 0 || iy >= p.ch || ix < 0 || ix >= p.cw) {          // padding: 0 after normalisation, pad_label
    oi[0] = b.pad_img[0]; oi[plane] = b.pad_img[1]; oi[2 * plane] = b.pad_img[2];
    if (og) *og = (LT)b.pad_label;
    return;
  }
  const int sy = p.crop_y + iy, sx = p.crop_x + ix;              // pixel of the (mirrored, scaled) image
  const double fy = p.fy, fx = p.fx;
  int y0, y1, x0, x1; float wy, wx;
  lin_index(sy, fy, p.H, y0, y1, wy);
  lin_index(sx, fx, p.W, x0, x1, wx);
  int ny = (int)floor((double)sy * fy); if (ny > p.H - 1) ny = p.H - 1;
  int nx = (int)floor((double)sx * fx); if (nx > p.W - 1) nx = p.W - 1;
  if (p.flip) { x0 = p.W - 1 - x0; x1 = p.W - 1 - x1; nx = p.W - 1 - nx; }   // resize(flip(img)) == flip-indexed taps
  const uint8_t* r0 = p.img + ((int64_t)y0 * p.W) * 3;
  const uint8_t* r1 = p.img + ((int64_t)y1 * p.W) * 3;
  // interpolate in double with the float-rounded weights, operation for operation as the numpy oracle does: the uint8
  // rounding below is a step function, and with factors like 1.5 a sizeable share of the values sit exactly on x.5
  const double dwx = (double)wx, dwy = (double)wy;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double a = (double)r0[x0 * 3 + c], bb = (double)r0[x1 * 3 + c];
    const double cc = (double)r1[x0 * 3 + c], d = (double)r1[x1 * 3 + c];
    const double top = (1.0 - dwx) * a + dwx * bb, bot = (1.0 - dwx) * cc + dwx * d;
    double v = (1.0 - dwy) * top + dwy * bot;
    v = floor(v + 0.5);                                          // the resized image is uint8 again
    v = fmin(fmax(v, 0.0), 255.0);
    oi[c * plane] = ((float)v / 255.0f - b.mean[c]) * b.inv_std[c];
  }
  if (og) *og = (LT)p.gt[(int64_t)ny * p.W + nx];
}

}  // namespace tsg

using namespace tsg;

extern "C" {

int tsg_augment_max_samples(void) { return kAugMax; }

int tsg_augment_crop(const void* const* imgs, const void* const* gts, const int32_t* geom, const double* inv_scale, int n,
                     int CH, int CW, const float* mean, const float* std, float pad_pixel, int pad_label, float* out_img,
                     void* out_gt, int gt_type, void* stream) {
  if (!imgs || !geom || !mean || !std || !out_img) return TSG_E_NULL;
  if ((gts == nullptr) != (out_gt == nullptr)) return TSG_E_NULL;
  if (n < 1 || n > kAugMax || CH < 1 || CW < 1) return TSG_E_SHAPE;
  if (gts && gt_type != TSG_I64 && gt_type != TSG_U8) return TSG_E_DTYPE;
  AugBatch b;
  b.n = n; b.CH = CH; b.CW = CW; b.pad_label = pad_label;
  b.with_gt = gts != nullptr;
  for (int c = 0; c < 3; ++c) {
    b.mean[c] = mean[c]; b.inv_std[c] = 1.0f / std[c];
    b.pad_img[c] = pad_pixel < 0.f ? 0.f : (pad_pixel / 255.0f - mean[c]) * b.inv_std[c];
  }
  for (int i = 0; i < n; ++i) {
    const int32_t* g = geom + 7 * i;                             // H, W, SH, SW, flip, crop_y, crop_x
    AugSample& s = b.s[i];
    s.img = (const uint8_t*)imgs[i]; s.gt = gts ? (const uint8_t*)gts[i] : nullptr;
    if (!s.img || (gts && !s.gt)) return TSG_E_NULL;
    s.H = g[0]; s.W = g[1]; s.SH = g[2]; s.SW = g[3]; s.flip = g[4] != 0; s.crop_y = g[5]; s.crop_x = g[6];
    if (s.H < 1 || s.W < 1 || s.SH < 1 || s.SW < 1 || s.crop_y < 0 || s.crop_x < 0 || s.crop_y >= s.SH || s.crop_x >= s.SW)
      return TSG_E_SHAPE;                                        // img_utils.py:27-28 asserts the same
    s.ch = s.SH - s.crop_y < CH ? s.SH - s.crop_y : CH;          // numpy slicing clips the crop at the array border
    s.cw = s.SW - s.crop_x < CW ? s.SW - s.crop_x : CW;
    s.top = (CH - s.ch) / 2;                                     // pad_image_to_shape: margin[0] = pad // 2
    s.left = (CW - s.cw) / 2;
    // OpenCV's source step is 1 / inv_scale: inv_scale = dsize / ssize when resize() is given a size (random_scale,
    // img_utils.py:114), the fx / fy argument itself when it is given factors (evaluator.py:192-193)
    const double isy = inv_scale ? inv_scale[2 * i] : (double)s.SH / (double)s.H;
    const double isx = inv_scale ? inv_scale[2 * i + 1] : (double)s.SW / (double)s.W;
    if (!(isy > 0.0) || !(isx > 0.0)) return TSG_E_SHAPE;
    s.fy = 1.0 / isy; s.fx = 1.0 / isx;
  }
  dim3 grid((unsigned)((CW + 255) / 256), (unsigned)CH, (unsigned)n);
  if (gt_type == TSG_I64 || !gts)
    hipLaunchKernelGGL((augment_crop_k<int64_t>), grid, dim3(256), 0, (hipStream_t)stream, b, out_img, (int64_t*)out_gt);
  else
    hipLaunchKernelGGL((augment_crop_k<uint8_t>), grid, dim3(256), 0, (hipStream_t)stream, b, out_img, (uint8_t*)out_gt);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

namespace tsg {

// =====================================================================================================================
// DFN's border labels on the GPU (VERDICT r3 "missing" item 2) — restates, on the mirrored / scaled
// label image, the lines of model/dfn/cityscapes.dfn.R101_v1c/dataloader.py:24-29:
//     no255_gt = gt with 255 -> 0;  cgt = cv2.Canny(no255_gt, 5, 5, apertureSize=7);  cgt = cv2.dilate(cgt, 7 x 7 ones);
//     cgt[cgt == 255] = 1;  p_cgt = random_crop_pad_to_shape(cgt, crop_pos, crop_size, 255)
// cv2 is not in this image: the arithmetic is OpenCV's integer Canny as torchseg_amd/shims_optional/cv2 and
// oracle/edge_ref.py restate it (imgproc/src/canny.cpp of 3.x / 4.x; round 5): separable Sobel of aperture 7 with
// REPLICATED borders, scaled by 1 / 16 and rounded half to even into 16-bit gradients; L1 magnitude; thresholds / 16,
// floored (5 -> 0); sectors by the fixed-point test |dy| << 15 against |dx| TG22 and |dx| (TG22 + 2^16); non-maximum
// suppression `m > left && m >= right` / `m > up && m >= down` / strict on both sides along the diagonal picked by the
// sign of dx dy; zero magnitude outside the image; hysteresis degenerates to `m > low` when both thresholds are equal,
// the reference's only use.  Everything is exact in int32.  Parity with OpenCV's own Canny is UNPINNED, like the rest of
// row f3 (no OpenCV in the image to produce a vector).
//   k1 scaled label S (nearest, mirrored, 255 -> 0)   k2 Sobel -> magnitude + sector   k3 NMS + threshold -> edge map
//   k4 7 x 7 dilation, crop, centred padding -> aux label.   Workspace: 7 bytes per scaled pixel.
struct EdgeGeom { int H, W, SH, SW, flip, crop_y, crop_x, top, left, ch, cw, CH, CW; double fy, fx; int thr, rad, pad_label; };

__global__ __launch_bounds__(256) void edge_scale_k(const uint8_t* __restrict__ gt, uint8_t* __restrict__ S, EdgeGeom g,
                                                    int ignore_label) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= g.SW) return;
  int ny = (int)floor((double)y * g.fy); if (ny > g.H - 1) ny = g.H - 1;
  int nx = (int)floor((double)x * g.fx); if (nx > g.W - 1) nx = g.W - 1;
  if (g.flip) nx = g.W - 1 - nx;
  const uint8_t v = gt[(int64_t)ny * g.W + nx];
  S[(int64_t)y * g.SW + x] = v == (uint8_t)ignore_label ? (uint8_t)0 : v;
}

__device__ __forceinline__ int edge_clamp(int i, int n) { return i < 0 ? 0 : (i > n - 1 ? n - 1 : i); }   // BORDER_REPLICATE

__device__ __forceinline__ int edge_div16_half_even(int g) {         // cvRound(g / 16.0)
  const int q = g >> 4, r = g & 15;
  return q + ((r > 8 || (r == 8 && (q & 1))) ? 1 : 0);
}

__global__ __launch_bounds__(256) void edge_sobel_k(const uint8_t* __restrict__ S, int32_t* __restrict__ mag,
                                                    uint8_t* __restrict__ sec, EdgeGeom g) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= g.SW) return;
  const int sm[7] = {1, 6, 15, 20, 15, 6, 1}, df[7] = {1, 4, 5, 0, -5, -4, -1};
  int cx[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) cx[j] = edge_clamp(x + j - 3, g.SW);
  int gx = 0, gy = 0;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const uint8_t* row = S + (int64_t)edge_clamp(y + i - 3, g.SH) * g.SW;
    int rs = 0, rd = 0;                                              // row filtered with the smoothing / the difference taps
#pragma unroll
    for (int j = 0; j < 7; ++j) { const int v = row[cx[j]]; rs += sm[j] * v; rd += df[j] * v; }
    gx += sm[i] * rd;                                                // sep(ky = smooth, kx = diff)
    gy += df[i] * rs;                                                // sep(ky = diff, kx = smooth)
  }
  gx = edge_div16_half_even(gx);                                     // the 16-bit gradients of aperture 7
  gy = edge_div16_half_even(gy);
  const int a = gx < 0 ? -gx : gx, b = gy < 0 ? -gy : gy;
  const int TG22 = 13573;                                            // (int)(tan(22.5 deg) * 2^15 + 0.5)
  const int64_t yb = (int64_t)b << 15, tg22x = (int64_t)a * TG22, tg67x = tg22x + ((int64_t)a << 16);
  int q;
  if (yb < tg22x) q = 0;
  else if (yb > tg67x) q = 2;
  else q = ((gx ^ gy) < 0) ? 3 : 1;
  mag[(int64_t)y * g.SW + x] = a + b;
  sec[(int64_t)y * g.SW + x] = (uint8_t)q;
}

__global__ __launch_bounds__(256) void edge_nms_k(const int32_t* __restrict__ mag, const uint8_t* __restrict__ sec,
                                                  uint8_t* __restrict__ E, EdgeGeom g) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= g.SW) return;
  const int64_t i = (int64_t)y * g.SW + x;
  const int m = mag[i], q = sec[i];
  auto at = [&](int yy, int xx) { return (yy < 0 || yy >= g.SH || xx < 0 || xx >= g.SW) ? 0 : mag[(int64_t)yy * g.SW + xx]; };
  bool keep;
  if (q == 0) keep = m > at(y, x - 1) && m >= at(y, x + 1);          // previous neighbour strict, next one >=
  else if (q == 2) keep = m > at(y - 1, x) && m >= at(y + 1, x);
  else {
    const int s = q == 3 ? -1 : 1;                                   // dx, dy of opposite sign: the other diagonal
    keep = m > at(y - 1, x - s) && m > at(y + 1, x + s);             // strict on both sides
  }
  E[i] = (keep && m > g.thr) ? 1 : 0;
}

template <typename LT>
__global__ __launch_bounds__(256) void edge_dilate_crop_k(const uint8_t* __restrict__ E, LT* __restrict__ out, EdgeGeom g) {
  const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
  if (ox >= g.CW) return;
  const int iy = oy - g.top, ix = ox - g.left;
  LT v = (LT)g.pad_label;
  if (iy >= 0 && iy < g.ch && ix >= 0 && ix < g.cw) {
    const int sy = g.crop_y + iy, sx = g.crop_x + ix, lo = g.rad / 2, hi = g.rad - 1 - lo;
    int any = 0;
    for (int yy = sy - lo; yy <= sy + hi; ++yy) {
      if (yy < 0 || yy >= g.SH) continue;
      for (int xx = sx - lo; xx <= sx + hi; ++xx)
        if (xx >= 0 && xx < g.SW) any |= E[(int64_t)yy * g.SW + xx];
    }
    v = (LT)any;
  }
  out[(int64_t)oy * g.CW + ox] = v;
}

}  // namespace tsg

extern "C" {

size_t tsg_edge_labels_ws_bytes(int SH, int SW) {
  if (SH < 1 || SW < 1) return 0;
  const size_t n = (size_t)SH * SW;
  return ((n + 15) / 16 * 16) * 3 + n * 4 + 64;
}

int tsg_edge_labels(const void* gt, const int32_t* geom, const double* inv_scale, int CH, int CW, int ignore_label,
                    int threshold1, int threshold2, int aperture, int dilate_size, int pad_label, void* out, int out_type,
                    void* ws, size_t ws_bytes, void* stream) {
  using namespace tsg;
  if (!gt || !geom || !out || !ws) return TSG_E_NULL;
  if (out_type != TSG_I64 && out_type != TSG_U8) return TSG_E_DTYPE;
  if (aperture != 7 || threshold1 != threshold2 || dilate_size < 1 || dilate_size > 31 || CH < 1 || CW < 1) return TSG_E_SHAPE;
  EdgeGeom g;
  g.H = geom[0]; g.W = geom[1]; g.SH = geom[2]; g.SW = geom[3]; g.flip = geom[4] != 0; g.crop_y = geom[5]; g.crop_x = geom[6];
  if (g.H < 1 || g.W < 1 || g.SH < 1 || g.SW < 1 || g.crop_y < 0 || g.crop_x < 0 || g.crop_y >= g.SH || g.crop_x >= g.SW)
    return TSG_E_SHAPE;
  if (ws_bytes < tsg_edge_labels_ws_bytes(g.SH, g.SW)) return TSG_E_WS;
  g.CH = CH; g.CW = CW;
  g.ch = g.SH - g.crop_y < CH ? g.SH - g.crop_y : CH;
  g.cw = g.SW - g.crop_x < CW ? g.SW - g.crop_x : CW;
  g.top = (CH - g.ch) / 2; g.left = (CW - g.cw) / 2;
  const double isy = inv_scale ? inv_scale[0] : (double)g.SH / (double)g.H;
  const double isx = inv_scale ? inv_scale[1] : (double)g.SW / (double)g.W;
  if (!(isy > 0.0) || !(isx > 0.0)) return TSG_E_SHAPE;
  g.fy = 1.0 / isy; g.fx = 1.0 / isx;
  g.thr = threshold1 >> 4;                                           // aperture 7: floor(threshold / 16) against the scaled gradients
  g.rad = dilate_size; g.pad_label = pad_label;
  const size_t n = (size_t)g.SH * g.SW, na = (n + 15) / 16 * 16;
  uint8_t* S = (uint8_t*)ws;
  uint8_t* sec = S + na;
  uint8_t* E = sec + na;
  int32_t* mag = (int32_t*)(E + na);
  hipStream_t st = (hipStream_t)stream;
  dim3 gs((unsigned)((g.SW + 255) / 256), (unsigned)g.SH), gc((unsigned)((CW + 255) / 256), (unsigned)CH);
  hipLaunchKernelGGL(edge_scale_k, gs, dim3(256), 0, st, (const uint8_t*)gt, S, g, ignore_label);
  hipLaunchKernelGGL(edge_sobel_k, gs, dim3(256), 0, st, (const uint8_t*)S, mag, sec, g);
  hipLaunchKernelGGL(edge_nms_k, gs, dim3(256), 0, st, (const int32_t*)mag, (const uint8_t*)sec, E, g);
  if (out_type == TSG_I64) hipLaunchKernelGGL((edge_dilate_crop_k<int64_t>), gc, dim3(256), 0, st, (const uint8_t*)E, (int64_t*)out, g);
  else hipLaunchKernelGGL((edge_dilate_crop_k<uint8_t>), gc, dim3(256), 0, st, (const uint8_t*)E, (uint8_t*)out, g);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
