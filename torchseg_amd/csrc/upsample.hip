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
#include "tsg_resample.h"

namespace tsg {

constexpr int kT = 256;

template <typename T, int V> struct OutVec;
template <> struct OutVec<float, 4> : Vec<float> {};
template <> struct OutVec<bf16_t, 8> : Vec<bf16_t> {};
template <typename T> struct OutVec<T, 1> {
  float v[1];
  __device__ __forceinline__ void load(const T* p) { v[0] = ld1<T>(p); }
  __device__ __forceinline__ void store(T* p) const { st1<T>(p, v[0]); }
};

// value of an element of type T nearest to v (what an eager `fm += last_fm` would have stored)
template <typename T> __device__ __forceinline__ float round_as(float v);
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_as<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// ADD modes of the forward kernels: 0 none; 1 `add` has the OUTPUT shape, y = up(x) + add (dfn network.py:130-133);
// 2 `add` has the SOURCE shape, y = up(x + add) with the sum rounded to T exactly like the eager in-place add it
// replaces (bisenet network.py:91-95: `fm += last_fm` then F.interpolate) -- the sum is never written to HBM.
enum { kAddNone = 0, kAddOut = 1, kAddSrc = 2 };

// Forward, separable form.  One thread owns V consecutive output columns and a
// band of RB output rows.  H0/H1 hold the horizontally interpolated source rows
// y0 / y1 for its columns; they are refreshed only when y0 advances (every
// ~1/scale output rows), so the per-output work is 2 FMAs + the 16-B store:
//   top = hx*x[y0][x0] + lx*x[y0][x1] ; bot = (same on y1) ; y = hy*top + ly*bot
// (identical operations and order to the 4-tap formula above).
template <typename T, int V, int ADD, int RB>
__global__ __launch_bounds__(kT) void up_fwd(const T* __restrict__ x, const T* __restrict__ add,
                                             T* __restrict__ y, int64_t NC, int IH, int IW, int OH,
                                             int OW, float sy, float sx) {
  const int vpr = OW / V;                      // column groups per row
  const int bands = (OH + RB - 1) / RB;
  const int64_t total = NC * bands * (int64_t)vpr;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int vx = (int)(i % vpr);
    const int64_t t = i / vpr;
    const int band = (int)(t % bands);
    const int64_t nc = t / bands;
    int x0[V], x1[V];
    float lx[V];
#pragma unroll
    for (int j = 0; j < V; ++j) src_index(sx, vx * V + j, IW, x0[j], x1[j], lx[j]);
    const T* src = x + nc * IH * (int64_t)IW;
    const T* src2 = ADD == kAddSrc ? add + nc * IH * (int64_t)IW : nullptr;
    auto tap = [&](int64_t off) -> float {
      return ADD == kAddSrc ? round_as<T>(ld1<T>(src + off) + ld1<T>(src2 + off)) : ld1<T>(src + off);
    };
    float h0[V], h1[V];
    int cy0 = -1, cy1 = -1;
    const int oy_end = (band + 1) * RB < OH ? (band + 1) * RB : OH;
    for (int oy = band * RB; oy < oy_end; ++oy) {
      int y0, y1; float ly;
      src_index(sy, oy, IH, y0, y1, ly);
      if (y0 != cy0) {
        if (y0 == cy1) {
#pragma unroll
          for (int j = 0; j < V; ++j) h0[j] = h1[j];
        } else {
          const int64_t r = (int64_t)y0 * IW;
#pragma unroll
          for (int j = 0; j < V; ++j) h0[j] = (1.f - lx[j]) * tap(r + x0[j]) + lx[j] * tap(r + x1[j]);
        }
        cy0 = y0;
        cy1 = -1;
      }
      if (y1 != cy1) {
        if (y1 == cy0) {
#pragma unroll
          for (int j = 0; j < V; ++j) h1[j] = h0[j];
        } else {
          const int64_t r = (int64_t)y1 * IW;
#pragma unroll
          for (int j = 0; j < V; ++j) h1[j] = (1.f - lx[j]) * tap(r + x0[j]) + lx[j] * tap(r + x1[j]);
        }
        cy1 = y1;
      }
      const float hy = 1.f - ly;
      const int64_t ooff = (nc * OH + oy) * (int64_t)OW + (int64_t)vx * V;
      OutVec<T, V> o, a;
      if (ADD == kAddOut) a.load(add + ooff);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float v = hy * h0[j] + ly * h1[j];
        if (ADD == kAddOut) v += a.v[j];
        o.v[j] = v;
      }
      o.store(y + ooff);
    }
  }
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

// Backward, fused separable form (the fast path for up-sampling).  A block owns
// RB consecutive source rows of one (n,c) plane and the full row width:
//   1. vertical: every thread streams V output columns over the output rows whose
//      taps touch the owned source rows (16-B coalesced reads of dy, each read
//      once plus a one-row halo), keeping running sums for source rows y0 / y0+1
//      and flushing a finished row into LDS;
//   2. horizontal: thread ix gathers its x-footprint from the LDS rows with the
//      forward's exact tap weights and writes dx.
// No atomics; every source pixel is produced by exactly one thread in a fixed order.
template <typename T, int V, int MAXF>
__global__ void up_bwd_tiled(const T* __restrict__ dy, T* __restrict__ dx, int IH, int IW, int OH,
                             int OW, int RB, float sy, float sx) {
  extern __shared__ __attribute__((aligned(16))) float vrow[];    // [RB][OW]
  const int nthreads = blockDim.x;
  const int tid = threadIdx.x;
  const int vpr = OW / V;
  const int bands = (IH + RB - 1) / RB;
  const int band = blockIdx.x % bands;
  const int64_t nc = blockIdx.x / bands;
  const int r0 = band * RB;
  const int r1 = (r0 + RB < IH) ? r0 + RB : IH;
  const T* src = dy + nc * OH * (int64_t)OW;
  for (int i = tid; i < (r1 - r0) * OW; i += nthreads) vrow[i] = 0.f;
  __syncthreads();
  // output rows with y0 in [r0-1, r1-1]
  int oy_lo = 0, oy_hi = OH - 1;
  if (sy > 0.f) {
    const float inv = 1.f / sy;
    int l = (int)ceilf((float)(r0 - 1) * inv) - 1;
    int h = (int)floorf((float)r1 * inv) + 1;
    oy_lo = l < 0 ? 0 : l;
    oy_hi = h > OH - 1 ? OH - 1 : h;
  }
  if (tid < vpr) {
    float accA[V], accB[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { accA[j] = 0.f; accB[j] = 0.f; }
    int cur = -2;                                  // source row accA belongs to
    auto flush = [&](int row, const float (&acc)[V]) {
      if (row >= r0 && row < r1) {
        float* d = vrow + (row - r0) * OW + tid * V;
#pragma unroll
        for (int j = 0; j < V; ++j) d[j] = acc[j];
      }
    };
    auto absorb = [&](int oy, const OutVec<T, V>& g) {
      int y0, y1; float ly;
      src_index(sy, oy, IH, y0, y1, ly);
      if (y0 < r0 - 1 || y0 > r1 - 1) return;      // halo rows of the conservative range
      if (y0 != cur) {
        if (cur >= 0) {
          flush(cur, accA);
          if (y0 == cur + 1) {
#pragma unroll
            for (int j = 0; j < V; ++j) { accA[j] = accB[j]; accB[j] = 0.f; }
          } else {                                 // down-sampling jump: both rows are finished
            flush(cur + 1, accB);
#pragma unroll
            for (int j = 0; j < V; ++j) { accA[j] = 0.f; accB[j] = 0.f; }
          }
        }
        cur = y0;
      }
      const float hy = 1.f - ly;
      if (y1 == y0) {
#pragma unroll
        for (int j = 0; j < V; ++j) accA[j] += hy * g.v[j] + ly * g.v[j];
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) { accA[j] += hy * g.v[j]; accB[j] += ly * g.v[j]; }
      }
    };
    constexpr int U = 8;                           // rows in flight per thread
    int oy = oy_lo;
    for (; oy + U - 1 <= oy_hi; oy += U) {         // full batches: U independent 16-B loads, then the math
      OutVec<T, V> g[U];
#pragma unroll
      for (int u = 0; u < U; ++u) g[u].load(src + (int64_t)(oy + u) * OW + tid * V);
#pragma unroll
      for (int u = 0; u < U; ++u) absorb(oy + u, g[u]);
    }
    for (; oy <= oy_hi; ++oy) {
      OutVec<T, V> g;
      g.load(src + (int64_t)oy * OW + tid * V);
      absorb(oy, g);
    }
    if (cur >= 0) { flush(cur, accA); flush(cur + 1, accB); }
  }
  __syncthreads();
  for (int ix = tid; ix < IW; ix += nthreads) {
    int xlo, xhi;
    footprint(sx, ix, OW, xlo, xhi);
    float wx[MAXF];
#pragma unroll
    for (int j = 0; j < MAXF; ++j) wx[j] = (xlo + j <= xhi) ? tap_weight(sx, xlo + j, IW, ix) : 0.f;
    for (int r = r0; r < r1; ++r) {
      const float* v = vrow + (r - r0) * OW + xlo;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < MAXF; ++j)
        if (xlo + j <= xhi) acc += wx[j] * v[j];
      st1<T>(dx + (nc * IH + r) * (int64_t)IW + ix, acc);
    }
  }
}

// ---- channels_last (NHWC) variants for feature maps (C % V == 0) ---------------
// A thread owns V adjacent channels of one pixel: every tap is one 16-byte load,
// so neither direction needs LDS.  x [N, IH, IW, C] -> y [N, OH, OW, C].
template <typename T, int V, int ADD>
__global__ __launch_bounds__(kT) void up_fwd_nhwc(const T* __restrict__ x, const T* __restrict__ add,
                                                  T* __restrict__ y, int64_t N, int C, int IH, int IW,
                                                  int OH, int OW, float sy, float sx) {
  const int G = C / V;
  const int64_t total = N * OH * (int64_t)OW * G;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int g = (int)(i % G);
    int64_t t = i / G;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int64_t n = t / OH;
    int y0, y1, x0, x1; float ly, lx;
    src_index(sy, oy, IH, y0, y1, ly);
    src_index(sx, ox, IW, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const T* b = x + n * IH * (int64_t)IW * C + g * V;
    OutVec<T, V> p00, p01, p10, p11, o, a;
    p00.load(b + ((int64_t)y0 * IW + x0) * C);
    p01.load(b + ((int64_t)y0 * IW + x1) * C);
    p10.load(b + ((int64_t)y1 * IW + x0) * C);
    p11.load(b + ((int64_t)y1 * IW + x1) * C);
    if (ADD == kAddOut) a.load(add + i * V);
    if (ADD == kAddSrc) {
      const T* b2 = add + n * IH * (int64_t)IW * C + g * V;
      OutVec<T, V> q00, q01, q10, q11;
      q00.load(b2 + ((int64_t)y0 * IW + x0) * C);
      q01.load(b2 + ((int64_t)y0 * IW + x1) * C);
      q10.load(b2 + ((int64_t)y1 * IW + x0) * C);
      q11.load(b2 + ((int64_t)y1 * IW + x1) * C);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        p00.v[j] = round_as<T>(p00.v[j] + q00.v[j]); p01.v[j] = round_as<T>(p01.v[j] + q01.v[j]);
        p10.v[j] = round_as<T>(p10.v[j] + q10.v[j]); p11.v[j] = round_as<T>(p11.v[j] + q11.v[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float v = hy * (hx * p00.v[j] + lx * p01.v[j]) + ly * (hx * p10.v[j] + lx * p11.v[j]);
      if (ADD == kAddOut) v += a.v[j];
      o.v[j] = v;
    }
    o.store(y + i * V);
  }
}

template <typename T, int V>
__global__ __launch_bounds__(kT) void up_bwd_nhwc(const T* __restrict__ dy, T* __restrict__ dx,
                                                  int64_t N, int C, int IH, int IW, int OH, int OW,
                                                  float sy, float sx) {
  const int G = C / V;
  const int64_t total = N * IH * (int64_t)IW * G;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int g = (int)(i % G);
    int64_t t = i / G;
    const int ix = (int)(t % IW); t /= IW;
    const int iy = (int)(t % IH);
    const int64_t n = t / IH;
    int xlo, xhi, ylo, yhi;
    footprint(sx, ix, OW, xlo, xhi);
    footprint(sy, iy, OH, ylo, yhi);
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    const T* b = dy + n * OH * (int64_t)OW * C + g * V;
    for (int oy = ylo; oy <= yhi; ++oy) {
      const float wy = tap_weight(sy, oy, IH, iy);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const float w = wy * tap_weight(sx, ox, IW, ix);
        if (w != 0.f) {
          OutVec<T, V> p;
          p.load(b + ((int64_t)oy * OW + ox) * C);
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] += w * p.v[j];
        }
      }
    }
    OutVec<T, V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = acc[j];
    o.store(dx + i * V);
  }
}

// The same gather for SMALL sources with large footprints — the pyramid-pooling branches of PSPNet (pspnet
// network.py:101-106: 1x1 / 2x2 / 3x3 / 6x6 pooled maps interpolated to 90 x 90): one thread per source pixel and channel
// group would leave a few hundred threads walking thousands of output pixels each (1 ms per launch at 2 x 512 x 6 x 6 ->
// 90 x 90).  Here a block owns one source pixel (and up to 256 channel groups of V); its threads split the footprint's
// ROWS, and the row partials are folded through LDS in a fixed order: deterministic, no atomics.
template <typename T, int V>
__global__ __launch_bounds__(256) void up_bwd_nhwc_split(const T* __restrict__ dy, T* __restrict__ dx, int64_t N, int C,
                                                         int IH, int IW, int OH, int OW, float sy, float sx, int gpb) {
  __shared__ float red[256 * V];
  const int G = C / V, RS = 256 / gpb;                   // channel groups per block, row splitters
  const int tid = threadIdx.x, gl = tid % gpb, rs = tid / gpb;
  const int g = blockIdx.y * gpb + gl;
  int64_t t = blockIdx.x;
  const int ix = (int)(t % IW); t /= IW;
  const int iy = (int)(t % IH);
  const int64_t n = t / IH;
  int xlo, xhi, ylo, yhi;
  footprint(sx, ix, OW, xlo, xhi);
  footprint(sy, iy, OH, ylo, yhi);
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  if (rs < RS && g < G) {
    const T* b = dy + n * OH * (int64_t)OW * C + g * V;
    for (int oy = ylo + rs; oy <= yhi; oy += RS) {
      const float wy = tap_weight(sy, oy, IH, iy);
      if (wy == 0.f) continue;
      for (int ox = xlo; ox <= xhi; ++ox) {
        const float w = wy * tap_weight(sx, ox, IW, ix);
        if (w != 0.f) {
          OutVec<T, V> p;
          p.load(b + ((int64_t)oy * OW + ox) * C);
#pragma unroll
          for (int j = 0; j < V; ++j) acc[j] += w * p.v[j];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[tid * V + j] = acc[j];
  __syncthreads();
  if (rs == 0 && g < G) {
    OutVec<T, V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float s = 0.f;
      for (int q = 0; q < RS; ++q) s += red[(q * gpb + gl) * V + j];
      o.v[j] = s;
    }
    o.store(dx + (blockIdx.x * (int64_t)G + g) * V);
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

// Half-pixel-centre bilinear resize (cv2.resize INTER_LINEAR on float data == F.interpolate(align_corners=False), no
// antialiasing): what the evaluator applies to the class scores of every scale (furnace/engine/evaluator.py:250-252:
// cv2.resize(score, (ori_cols, ori_rows), interpolation=cv2.INTER_LINEAR)).  src = (dst + 0.5) * in / out - 0.5,
// i0 = floor(src), border taps clamped with weight 0 (OpenCV resize.cpp).  Planar [NC, IH, IW] -> [NC, OH, OW]; up- or
// down-sampling.  `accumulate` adds into y (the sum over scales of sliding_eval, evaluator.py:196-199).
__device__ __forceinline__ void hp_index(int dst, double scale, int in, int& i0, int& i1, float& w) {
  const double src = ((double)dst + 0.5) * scale - 0.5;
  int s = (int)floor(src);
  float f = (float)(src - (double)s);
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= in - 1) { s = in - 1; f = 0.f; }
  i0 = s; i1 = s + 1 < in ? s + 1 : in - 1; w = f;
}

template <typename T, bool ACC>
__global__ __launch_bounds__(kT) void resize_hp_k(const T* __restrict__ x, float* __restrict__ y, int64_t NC, int IH,
                                                  int IW, int OH, int OW) {
  const double sy = 1.0 / ((double)OH / (double)IH), sx = 1.0 / ((double)OW / (double)IW);   // as OpenCV forms it
  const int64_t total = NC * OH * (int64_t)OW;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int ox = (int)(i % OW);
    const int64_t t = i / OW;
    const int oy = (int)(t % OH);
    const int64_t nc = t / OH;
    int y0, y1, x0, x1; float wy, wx;
    hp_index(oy, sy, IH, y0, y1, wy);
    hp_index(ox, sx, IW, x0, x1, wx);
    const T* p = x + nc * IH * (int64_t)IW;
    const float a = ld1<T>(p + (int64_t)y0 * IW + x0), b = ld1<T>(p + (int64_t)y0 * IW + x1);
    const float c = ld1<T>(p + (int64_t)y1 * IW + x0), d = ld1<T>(p + (int64_t)y1 * IW + x1);
    const float v = (1.f - wy) * ((1.f - wx) * a + wx * b) + wy * ((1.f - wx) * c + wx * d);
    y[i] = ACC ? y[i] + v : v;
  }
}

extern "C" {

static int up_fwd_launch(const void* x, const void* add, int add_mode, void* y, int dtype, int64_t NC,
                         int IH, int IW, int OH, int OW, void* stream) {
  if (!x || !y || (add_mode != kAddNone && !add)) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (NC <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  const int native = dtype == TSG_BF16 ? 8 : 4;
  const bool vec = (OW % native == 0) && aligned16(y) && (add_mode != kAddOut || aligned16(add));
  const int V = vec ? native : 1;
  const int grid = grid_for(NC * ((OH + 15) / 16) * (int64_t)(OW / V));
#define GO(T, VV, A) hipLaunchKernelGGL((up_fwd<T, VV, A, 16>), dim3(grid), dim3(kT), 0, st, (const T*)x, \
                                        (const T*)add, (T*)y, NC, IH, IW, OH, OW, sy, sx)
#define GO3(T, VV) do { if (add_mode == kAddOut) GO(T, VV, kAddOut); else if (add_mode == kAddSrc) GO(T, VV, kAddSrc); \
                        else GO(T, VV, kAddNone); } while (0)
  if (dtype == TSG_F32) { if (vec) GO3(float, 4); else GO3(float, 1); }
  else                  { if (vec) GO3(bf16_t, 8); else GO3(bf16_t, 1); }
#undef GO3
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_upsample_bilinear_ac_fwd(const void* x, const void* add, void* y, int dtype, int64_t NC,
                                 int IH, int IW, int OH, int OW, void* stream) {
  return up_fwd_launch(x, add, add ? kAddOut : kAddNone, y, dtype, NC, IH, IW, OH, OW, stream);
}

int tsg_upsample_bilinear_ac_presum_fwd(const void* x, const void* x2, void* y, int dtype, int64_t NC,
                                        int IH, int IW, int OH, int OW, void* stream) {
  return up_fwd_launch(x, x2, kAddSrc, y, dtype, NC, IH, IW, OH, OW, stream);
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
  // fast path: up-sampling with a vectorisable row that one block can span
  const int native = dtype == TSG_BF16 ? 8 : 4;
  if (need <= 37 && OW % native == 0 && OW / native <= 1024 && OW >= IW && aligned16(dy)) {
    int threads = ((OW / native + 63) / 64) * 64;
    if (threads < 128) threads = 128;
    int RB = 8;
    while (RB > 1 && (size_t)RB * OW * sizeof(float) > 48 * 1024) RB >>= 1;
    if ((size_t)RB * OW * sizeof(float) <= 64 * 1024) {
      const int bands = (IH + RB - 1) / RB;
      const int64_t blocks = NC * bands;
      if (blocks <= 0x7fffffffLL) {
        const size_t sh = (size_t)RB * OW * sizeof(float);
#define GT_(T, VV, F) hipLaunchKernelGGL((up_bwd_tiled<T, VV, F>), dim3((unsigned)blocks), dim3(threads), sh, st, \
                                         (const T*)dy, (T*)dx, IH, IW, OH, OW, RB, sy, sx)
        if (dtype == TSG_F32) {
          if (need <= 9) GT_(float, 4, 9); else if (need <= 21) GT_(float, 4, 21); else GT_(float, 4, 37);
        } else {
          if (need <= 9) GT_(bf16_t, 8, 9); else if (need <= 21) GT_(bf16_t, 8, 21); else GT_(bf16_t, 8, 37);
        }
#undef GT_
        TSG_CHECK_LAUNCH();
        return 0;
      }
    }
  }
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

static int up_fwd_nhwc_launch(const void* x, const void* add, int add_mode, void* y, int dtype, int64_t N,
                              int C, int IH, int IW, int OH, int OW, void* stream) {
  if (!x || !y || (add_mode != kAddNone && !add)) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (N <= 0 || C <= 0 || C % V || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return TSG_E_SHAPE;
  if (!aligned16(x) || !aligned16(y) || (add && !aligned16(add))) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  const int grid = grid_for(N * OH * (int64_t)OW * (C / V));
#define GO(T, VV, A) hipLaunchKernelGGL((up_fwd_nhwc<T, VV, A>), dim3(grid), dim3(kT), 0, st, (const T*)x, \
                                        (const T*)add, (T*)y, N, C, IH, IW, OH, OW, sy, sx)
#define GO3(T, VV) do { if (add_mode == kAddOut) GO(T, VV, kAddOut); else if (add_mode == kAddSrc) GO(T, VV, kAddSrc); \
                        else GO(T, VV, kAddNone); } while (0)
  if (dtype == TSG_F32) GO3(float, 4); else GO3(bf16_t, 8);
#undef GO3
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_upsample_bilinear_ac_nhwc_fwd(const void* x, const void* add, void* y, int dtype, int64_t N,
                                      int C, int IH, int IW, int OH, int OW, void* stream) {
  return up_fwd_nhwc_launch(x, add, add ? kAddOut : kAddNone, y, dtype, N, C, IH, IW, OH, OW, stream);
}

int tsg_upsample_bilinear_ac_nhwc_presum_fwd(const void* x, const void* x2, void* y, int dtype, int64_t N,
                                             int C, int IH, int IW, int OH, int OW, void* stream) {
  return up_fwd_nhwc_launch(x, x2, kAddSrc, y, dtype, N, C, IH, IW, OH, OW, stream);
}

int tsg_upsample_bilinear_ac_nhwc_bwd(const void* dy, void* dx, int dtype, int64_t N, int C, int IH,
                                      int IW, int OH, int OW, void* stream) {
  if (!dy || !dx) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (N <= 0 || C <= 0 || C % V || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return TSG_E_SHAPE;
  if (!aligned16(dy) || !aligned16(dx)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  // few source pixels, each with a footprint of many rows: split the rows over the threads of a block
  const int G = C / V;
  if (N * IH * (int64_t)IW * G < 65536 && OH >= 8 * IH && N * IH * (int64_t)IW <= 0x7fffffffLL) {
    // channel groups per block: as few as 8 (= 32 row splitters per source pixel) until the launch has ~512 blocks — with
    // 64 groups and 4 splitters a thread walked 2000 footprint pixels one dependent load at a time (554 us at 2 x 512 x
    // 6 x 6 -> 90 x 90, profiles/r03_kernel_stats_pspnet.csv)
    int gpb = 256;
    while (gpb > G) gpb >>= 1;
    if (gpb < 1) gpb = 1;
    while (gpb > 8 && N * IH * (int64_t)IW * ((G + gpb - 1) / gpb) < 512) gpb >>= 1;
    const dim3 grid2((unsigned)(N * IH * IW), (unsigned)((G + gpb - 1) / gpb));
    if (dtype == TSG_F32)
      hipLaunchKernelGGL((up_bwd_nhwc_split<float, 4>), grid2, dim3(256), 0, st, (const float*)dy, (float*)dx, N, C, IH, IW,
                         OH, OW, sy, sx, gpb);
    else
      hipLaunchKernelGGL((up_bwd_nhwc_split<bf16_t, 8>), grid2, dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, C, IH,
                         IW, OH, OW, sy, sx, gpb);
    TSG_CHECK_LAUNCH();
    return 0;
  }
  const int grid = grid_for(N * IH * (int64_t)IW * (C / V));
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((up_bwd_nhwc<float, 4>), dim3(grid), dim3(kT), 0, st, (const float*)dy, (float*)dx, N, C,
                       IH, IW, OH, OW, sy, sx);
  else
    hipLaunchKernelGGL((up_bwd_nhwc<bf16_t, 8>), dim3(grid), dim3(kT), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, C,
                       IH, IW, OH, OW, sy, sx);
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

int tsg_resize_bilinear_hp(const void* x, float* y, int dtype, int64_t NC, int IH, int IW, int OH, int OW,
                           int accumulate, void* stream) {
  if (!x || !y) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (NC <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(NC * OH * (int64_t)OW);
#define GO(T, A) hipLaunchKernelGGL((resize_hp_k<T, A>), dim3(grid), dim3(kT), 0, st, (const T*)x, y, NC, IH, IW, OH, OW)
  if (dtype == TSG_F32) { if (accumulate) GO(float, true); else GO(float, false); }
  else { if (accumulate) GO(bf16_t, true); else GO(bf16_t, false); }
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
