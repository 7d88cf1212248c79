// Bilinear (align_corners=True) and nearest up/down-sampling for gfx950.
//
// Restates what the reference gets from
//   F.interpolate(x, size|scale_factor, mode='bilinear', align_corners=True)
// (model/bisenet/cityscapes.bisenet.R18/network.py:82-84,93-94,164-166;
//  model/pspnet/ade.pspnet.R50_v1c/network.py:46-49,103-105), i.e.
// aten::upsample_bilinear2d{,_backward}:  src = dst * (in-1)/(out-1) (0 when
// out == 1), i0 = floor(src), i1 = i0 + (i0 < in-1), lambda = src - i0,
//   y = (1-ly)*((1-lx)*x[i0y][i0x] + lx*x[i0y][i1x]) + ly*((1-lx)*x[i1y][i0x] + lx*x[i1y][i1x]).
// Forward: the (small) source stays L2-resident, the kernel is bound by the
// 16-byte coalesced stores of the full-resolution result (+ optional fused add).
// Backward: a GATHER over every source pixel's footprint (no float atomics =>
// deterministic), the transposed operator evaluated with the forward's exact
// index/weight arithmetic so that <up(x), g> == <x, up^T(g)> to rounding.
#include "tsg_common.h"

namespace tsg {

constexpr int kT = 256;

__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1,
                                          float& l1) {
  const float r = scale * (float)dst;
  i0 = (int)r;
  if (i0 > in_size - 1) i0 = in_size - 1;  // guards float round-up at the last index
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = r - (float)i0;
  if (l1 < 0.f) l1 = 0.f;
}

static float ac_scale(int in_size, int out_size) {
  return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
}

template <typename T, int V> struct OutVec;
template <> struct OutVec<float, 4> : Vec<float> {};
template <> struct OutVec<bf16_t, 8> : Vec<bf16_t> {};
template <typename T> struct OutVec<T, 1> {
  float v[1];
  __device__ __forceinline__ void load(const T* p) { v[0] = ld1<T>(p); }
  __device__ __forceinline__ void store(T* p) const { st1<T>(p, v[0]); }
};

template <typename T, int V, bool ADD>
__global__ __launch_bounds__(kT) void up_fwd(const T* __restrict__ x, const T* __restrict__ add,
                                             T* __restrict__ y, int64_t NC, int IH, int IW, int OH,
                                             int OW, float sy, float sx) {
  const int vpr = OW / V;  // vectors per output row
  const int64_t total = NC * OH * (int64_t)vpr;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int vx = (int)(i % vpr);
    const int64_t row = i / vpr;
    const int oy = (int)(row % OH);
    const int64_t nc = row / OH;
    int y0, y1; float ly;
    src_index(sy, oy, IH, y0, y1, ly);
    const float hy = 1.f - ly;
    const T* r0 = x + (nc * IH + y0) * IW;
    const T* r1 = x + (nc * IH + y1) * IW;
    OutVec<T, V> o, a;
    const int64_t ooff = row * OW + (int64_t)vx * V;
    if (ADD) a.load(add + ooff);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      int x0, x1; float lx;
      src_index(sx, vx * V + j, IW, x0, x1, lx);
      const float hx = 1.f - lx;
      const float top = hx * ld1<T>(r0 + x0) + lx * ld1<T>(r0 + x1);
      const float bot = hx * ld1<T>(r1 + x0) + lx * ld1<T>(r1 + x1);
      float v = hy * top + ly * bot;
      if (ADD) v += a.v[j];
      o.v[j] = v;
    }
    o.store(y + ooff);
  }
}

// footprint [lo, hi] of source index i: every dst whose taps may touch i
__device__ __forceinline__ void footprint(float scale, int i, int out_size, int& lo, int& hi) {
  if (scale <= 0.f) { lo = 0; hi = out_size - 1; return; }
  const float inv = 1.f / scale;
  // taps touch i  <=>  scale*dst in [i-1, i+1); +-1 absorbs float rounding
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

// one thread per source pixel; MAXF bounds the x-footprint held in registers
template <typename T, int MAXF>
__global__ __launch_bounds__(kT) void up_bwd(const T* __restrict__ dy, T* __restrict__ dx,
                                             int64_t NC, int IH, int IW, int OH, int OW, float sy,
                                             float sx) {
  const int64_t total = NC * IH * (int64_t)IW;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int ix = (int)(i % IW);
    const int64_t row = i / IW;
    const int iy = (int)(row % IH);
    const int64_t nc = row / IH;
    int xlo, xhi, ylo, yhi;
    footprint(sx, ix, OW, xlo, xhi);
    footprint(sy, iy, OH, ylo, yhi);
    float acc = 0.f;
    const T* base = dy + nc * OH * (int64_t)OW;
    if (MAXF > 0) {
      float wx[MAXF > 0 ? MAXF : 1];
#pragma unroll
      for (int j = 0; j < MAXF; ++j) wx[j] = (xlo + j <= xhi) ? tap_weight(sx, xlo + j, IW, ix) : 0.f;
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = tap_weight(sy, oy, IH, iy);
        if (wy != 0.f) {
          const T* r = base + (int64_t)oy * OW + xlo;
          float racc = 0.f;
#pragma unroll
          for (int j = 0; j < MAXF; ++j)
            if (xlo + j <= xhi) racc += wx[j] * ld1<T>(r + j);
          acc += wy * racc;
        }
      }
    } else {
      for (int oy = ylo; oy <= yhi; ++oy) {
        const float wy = tap_weight(sy, oy, IH, iy);
        if (wy != 0.f) {
          const T* r = base + (int64_t)oy * OW;
          float racc = 0.f;
          for (int ox = xlo; ox <= xhi; ++ox) racc += tap_weight(sx, ox, IW, ix) * ld1<T>(r + ox);
          acc += wy * racc;
        }
      }
    }
    st1<T>(dx + i, acc);
  }
}

template <int EB>
__global__ __launch_bounds__(kT) void nearest_fwd(const void* __restrict__ x, void* __restrict__ y,
                                                  int64_t NC, int IH, int IW, int OH, int OW,
                                                  float sy, float sx) {
  typedef typename std::conditional<EB == 1, uint8_t,
          typename std::conditional<EB == 2, uint16_t,
          typename std::conditional<EB == 4, uint32_t, uint64_t>::type>::type>::type E;
  const int64_t total = NC * OH * (int64_t)OW;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int ox = (int)(i % OW);
    const int64_t row = i / OW;
    const int oy = (int)(row % OH);
    const int64_t nc = row / OH;
    int iy = (int)floorf((float)oy * sy); if (iy > IH - 1) iy = IH - 1;
    int ix = (int)floorf((float)ox * sx); if (ix > IW - 1) ix = IW - 1;
    ((E*)y)[i] = ((const E*)x)[(nc * IH + iy) * IW + ix];
  }
}

static int grid_for(int64_t work) {
  int64_t g = (work + kT - 1) / kT;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace tsg

using namespace tsg;

extern "C" {

int tsg_upsample_bilinear_ac_fwd(const void* x, const void* add, void* y, int dtype, int64_t NC,
                                 int IH, int IW, int OH, int OW, void* stream) {
  if (!x || !y) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (NC <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  const int native = dtype == TSG_BF16 ? 8 : 4;
  const bool vec = (OW % native == 0) && aligned16(y) && (!add || aligned16(add));
  const int V = vec ? native : 1;
  const int grid = grid_for(NC * OH * (int64_t)(OW / V));
#define GO(T, VV, A) hipLaunchKernelGGL((up_fwd<T, VV, A>), dim3(grid), dim3(kT), 0, st, (const T*)x, \
                                        (const T*)add, (T*)y, NC, IH, IW, OH, OW, sy, sx)
  if (dtype == TSG_F32) {
    if (vec) { if (add) GO(float, 4, true); else GO(float, 4, false); }
    else     { if (add) GO(float, 1, true); else GO(float, 1, false); }
  } else {
    if (vec) { if (add) GO(bf16_t, 8, true); else GO(bf16_t, 8, false); }
    else     { if (add) GO(bf16_t, 1, true); else GO(bf16_t, 1, false); }
  }
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_upsample_bilinear_ac_bwd(const void* dy, void* dx, int dtype, int64_t NC, int IH, int IW,
                                 int OH, int OW, void* stream) {
  if (!dy || !dx) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (NC <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  // widest x-footprint: floor(2/sx) + 4 candidates (see footprint())
  int need = sx > 0.f ? (int)floorf(2.f / sx) + 4 : OW;
  const int grid = grid_for(NC * IH * (int64_t)IW);
#define GO(T, F) hipLaunchKernelGGL((up_bwd<T, F>), dim3(grid), dim3(kT), 0, st, (const T*)dy, (T*)dx, \
                                    NC, IH, IW, OH, OW, sy, sx)
  if (dtype == TSG_F32) {
    if (need <= 9) GO(float, 9); else if (need <= 21) GO(float, 21); else if (need <= 37) GO(float, 37); else GO(float, 0);
  } else {
    if (need <= 9) GO(bf16_t, 9); else if (need <= 21) GO(bf16_t, 21); else if (need <= 37) GO(bf16_t, 37); else GO(bf16_t, 0);
  }
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_upsample_nearest_fwd(const void* x, void* y, int elem_bytes, int64_t NC, int IH, int IW,
                             int OH, int OW, void* stream) {
  if (!x || !y) return TSG_E_NULL;
  if (NC <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  // torch 'nearest': src = floor(dst * in/out)
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const int grid = grid_for(NC * OH * (int64_t)OW);
#define GO(EB) hipLaunchKernelGGL((nearest_fwd<EB>), dim3(grid), dim3(kT), 0, st, x, y, NC, IH, IW, OH, OW, sy, sx)
  switch (elem_bytes) {
    case 1: GO(1); break;
    case 2: GO(2); break;
    case 4: GO(4); break;
    case 8: GO(8); break;
    default: return TSG_E_DTYPE;
  }
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
