// Index / weight arithmetic of bilinear align_corners=True resampling, shared by
// upsample.hip and the fused upsample+OHEM kernels in ohem.hip so that forward,
// backward and the fused paths evaluate bit-identical taps.
//   src = dst * (in-1)/(out-1) (0 when out == 1); i0 = floor(src);
//   i1 = i0 + (i0 < in-1); lambda = src - i0        (aten::upsample_bilinear2d)
#pragma once
#include "tsg_common.h"

namespace tsg {

__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1,
                                          float& l1) {
  const float r = scale * (float)dst;
  i0 = (int)r;
  if (i0 > in_size - 1) i0 = in_size - 1;  // guards float round-up at the last index
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = r - (float)i0;
  if (l1 < 0.f) l1 = 0.f;
}

static inline float ac_scale(int in_size, int out_size) {
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

// footprint [lo, hi] of source index i: every dst whose taps may touch i
// (taps touch i  <=>  scale*dst in [i-1, i+1); +-1 absorbs float rounding)
__host__ __device__ __forceinline__ void footprint(float scale, int i, int out_size, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
  const float inv = 1.f / scale;
  int l = (int)ceilf((float)(i - 1) * inv) - 1;
  int h = (int)floorf((float)(i + 1) * inv) + 1;
  lo = l < 0 ? 0 : l;
  hi = h > out_size - 1 ? out_size - 1 : h;
}

__device__ __forceinline__ float tap_weight(float scale, int dst, int in_size, int i) {
  int i0, i1; float l1;
  src_index(scale, dst, in_size, i0, i1, l1);
  float w = 0.f;
  if (i0 == i) w += 1.f - l1;
  if (i1 == i) w += l1;
  return w;
}

}  // namespace tsg
