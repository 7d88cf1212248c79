// Global average pooling ([N,C,H,W] -> [N,C,1,1]) for gfx950.
//
// The reference reaches it through nn.AdaptiveAvgPool2d(1) inside
// AttentionRefinement / FeatureFusion (furnace/seg_opr/seg_oprs.py:200,224) and
// GlobalAvgPool2d (seg_oprs.py:97-107).  On channels_last activations eager
// PyTorch runs it as a strided reduce_kernel at ~0.2 TB/s (318 us for the
// 16x256x128x128 FFM map, profiles/r01); here it is a streaming column reduce
// with 16-byte loads and fixed-order partials (deterministic), HBM-bound:
// algorithmic bytes = E*s forward, E*s backward (broadcast write).
#include "tsg_common.h"

namespace tsg {
constexpr int kT = 256;
constexpr int kU = 4;

template <typename T, int V> struct PV;
template <> struct PV<float, 4> : Vec<float> {};
template <> struct PV<bf16_t, 8> : Vec<bf16_t> {};
template <typename T> struct PV<T, 1> {
  float v[1];
  __device__ __forceinline__ void load(const T* p) { v[0] = ld1<T>(p); }
  __device__ __forceinline__ void store(T* p) const { st1<T>(p, v[0]); }
};

// fp32 maps (the parity path) are summed in fp64: the pooled value feeds the batch-2 BatchNorm of BiSeNet's global-context
// branch, which amplifies an fp32 summation error ~300x (csrc/bn.hip, RedAcc)
template <typename T> struct GapAcc { typedef float type; };
template <> struct GapAcc<float> { typedef double type; };

// NHWC: x [N, HW, C]; block (s, n): rows [s*rpb, ...) of image n -> partial[n][s][c]
template <typename T, int V>
__global__ __launch_bounds__(kT) void gap_partial_nhwc(const T* __restrict__ x, int64_t HW, int64_t C, int GT,
                                                       int R, int64_t rpb, int S, float* __restrict__ partial) {
  typedef typename GapAcc<T>::type AT;
  extern __shared__ __attribute__((aligned(16))) float sm_raw[];   // [R][C] of AT
  AT* sm = reinterpret_cast<AT*>(sm_raw);
  const int tid = threadIdx.x, gl = tid % GT, r = tid / GT;
  const int64_t n = blockIdx.y;
  const int64_t row0 = (int64_t)blockIdx.x * rpb;
  int64_t row1 = row0 + rpb;
  if (row1 > HW) row1 = HW;
  const int G = (int)(C / V);
  for (int gb = 0; gb < G; gb += GT) {                         // channel tiles when C/V > 256
    const int g = gb + gl;
    AT acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0;
    if (r < R && g < G) {
      const T* b = x + n * HW * C + (int64_t)g * V;
      for (int64_t row = row0 + r; row < row1; row += (int64_t)kU * R) {
#pragma unroll
        for (int k = 0; k < kU; ++k) {
          const int64_t rr = row + (int64_t)k * R;
          if (rr < row1) {
            PV<T, V> p;
            p.load(b + rr * C);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += p.v[j];
          }
        }
      }
    }
    __syncthreads();
    if (r < R && g < G) {
#pragma unroll
      for (int j = 0; j < V; ++j) sm[r * (GT * V) + gl * V + j] = acc[j];
    }
    __syncthreads();
    for (int t = tid; t < GT * V; t += kT) {
      const int64_t c = (int64_t)gb * V + t;
      if (c < C) {
        AT s = 0;
        for (int q = 0; q < R; ++q) s += sm[q * (GT * V) + t];
        partial[(n * S + blockIdx.x) * C + c] = (float)s;
      }
    }
  }
}

// NCHW: one block per (n, c) plane
template <typename T, int V>
__global__ __launch_bounds__(kT) void gap_plane_nchw(const T* __restrict__ x, int64_t HW, float inv,
                                                     T* __restrict__ out) {
  typedef typename GapAcc<T>::type AT;
  __shared__ AT sm[kT / 64];
  const T* b = x + (int64_t)blockIdx.x * HW;
  AT acc = 0;
  for (int64_t i = (int64_t)threadIdx.x * V; i < HW; i += (int64_t)kT * V) {
    PV<T, V> p;
    p.load(b + i);
#pragma unroll
    for (int j = 0; j < V; ++j) acc += p.v[j];
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    AT t = 0;
    for (int i = 0; i < kT / 64; ++i) t += sm[i];
    st1<T>(out + blockIdx.x, (float)(t * (AT)inv));
  }
}

template <typename T>
__global__ __launch_bounds__(kT) void gap_finish(const float* __restrict__ partial, int S, int64_t NC, int64_t C,
                                                 float inv, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;   // i = n*C + c
  if (i >= NC) return;
  const int64_t n = i / C, c = i - n * C;
  typename GapAcc<T>::type s = 0;
#pragma unroll 16
  for (int q = 0; q < S; ++q) s += partial[(n * S + q) * C + c];
  st1<T>(out + i, (float)(s * inv));
}

// backward: dx[n, p, c] = dout[n, c] * inv   (NHWC)   /   dx[n, c, p] (NCHW)
template <typename T, int V>
__global__ __launch_bounds__(kT) void gap_bwd_nhwc(const T* __restrict__ dout, int64_t HW, int64_t C, float inv,
                                                   T* __restrict__ dx) {
  const int64_t G = C / V;
  const int64_t n = blockIdx.y;
  const int64_t total = HW * G;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int64_t g = i % G;
    PV<T, V> p;
    p.load(dout + n * C + g * V);
#pragma unroll
    for (int j = 0; j < V; ++j) p.v[j] *= inv;
    p.store(dx + (n * HW) * C + i * V);
  }
}

template <typename T, int V>
__global__ __launch_bounds__(kT) void gap_bwd_nchw(const T* __restrict__ dout, int64_t HW, float inv,
                                                   T* __restrict__ dx) {
  const float v = ld1<T>(dout + blockIdx.x) * inv;
  PV<T, V> p;
#pragma unroll
  for (int j = 0; j < V; ++j) p.v[j] = v;
  T* b = dx + (int64_t)blockIdx.x * HW;
  for (int64_t i = (int64_t)threadIdx.x * V; i < HW; i += (int64_t)kT * V) p.store(b + i);
}

// ---- channel gate: y = x * s[n,c] (+ x)  -------------------------------------------
// The squeeze-excite multiply of AttentionRefinement (`fm * fm_se`, seg_oprs.py:209-210)
// and FeatureFusion (`fm + fm * fm_se`, seg_oprs.py:236-237).  Eager PyTorch runs the
// backward as 3 element-wise kernels plus a strided reduce (520 us for the FFM map);
// here backward is ONE pass: dx = dy*s (+dy) written while ds[n,c] = sum_p dy*x is
// accumulated (fixed-order partials).
template <typename T, int V, bool IDENT>
__global__ __launch_bounds__(kT) void cs_fwd_nhwc(const T* __restrict__ x, const T* __restrict__ s,
                                                  T* __restrict__ y, int64_t HW, int64_t C) {
  const int64_t G = C / V, n = blockIdx.y, total = HW * G;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int64_t g = i % G;
    PV<T, V> px, ps;
    px.load(x + n * HW * C + i * V);
    ps.load(s + n * C + g * V);
#pragma unroll
    for (int j = 0; j < V; ++j) px.v[j] = IDENT ? fmaf(px.v[j], ps.v[j], px.v[j]) : px.v[j] * ps.v[j];
    px.store(y + n * HW * C + i * V);
  }
}

// dx = round(round(dy s (+ dy)) + gadd[n, c]): the gate's data gradient plus the pooled branch's (already divided by HW and
// rounded to T on the host side) — bit for bit what tsg_chanscale_bwd's dx followed by the framework's `dx += expand(gadd)` gives
template <typename T, int V, bool IDENT>
__global__ __launch_bounds__(kT) void cs_dx_nhwc(const T* __restrict__ dy, const T* __restrict__ s, const T* __restrict__ gadd,
                                                 T* __restrict__ dx, int64_t HW, int64_t C) {
  const int64_t G = C / V, n = blockIdx.y, total = HW * G;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int64_t g = i % G;
    PV<T, V> pd, ps, pg;
    pd.load(dy + n * HW * C + i * V);
    ps.load(s + n * C + g * V);
    pg.load(gadd + n * C + g * V);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float t = IDENT ? fmaf(pd.v[j], ps.v[j], pd.v[j]) : pd.v[j] * ps.v[j];
      const float tr = sizeof(T) == 2 ? __uint_as_float((uint32_t)f32_to_bf16(t) << 16) : t;     // the rounding of the stored dx
      pd.v[j] = tr + pg.v[j];
    }
    pd.store(dx + n * HW * C + i * V);
  }
}

template <typename T, int V, bool IDENT>
__global__ __launch_bounds__(kT) void cs_fwd_nchw(const T* __restrict__ x, const T* __restrict__ s,
                                                  T* __restrict__ y, int64_t HW) {
  const float g = ld1<T>(s + blockIdx.x);
  const int64_t off = (int64_t)blockIdx.x * HW;
  for (int64_t i = (int64_t)threadIdx.x * V; i < HW; i += (int64_t)kT * V) {
    PV<T, V> px;
    px.load(x + off + i);
#pragma unroll
    for (int j = 0; j < V; ++j) px.v[j] = IDENT ? fmaf(px.v[j], g, px.v[j]) : px.v[j] * g;
    px.store(y + off + i);
  }
}

// DX = false (round 6): only ds — the gate's gradient; dx is then written by cs_dx_nhwc once the pooled branch's gradient
// is known (tsg_chanscale_bwd_ds / tsg_chanscale_bwd_dx), instead of being written here and re-read by a framework add
template <typename T, int V, bool IDENT, bool DX = true>
__global__ __launch_bounds__(kT) void cs_bwd_nhwc(const T* __restrict__ dy, const T* __restrict__ x,
                                                  const T* __restrict__ s, T* __restrict__ dx, int64_t HW,
                                                  int64_t C, int GT, int R, int64_t rpb, int S,
                                                  float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [R][GT*V]
  const int tid = threadIdx.x, gl = tid % GT, r = tid / GT;
  const int64_t n = blockIdx.y;
  const int64_t row0 = (int64_t)blockIdx.x * rpb;
  int64_t row1 = row0 + rpb;
  if (row1 > HW) row1 = HW;
  const int G = (int)(C / V);
  for (int gb = 0; gb < G; gb += GT) {
    const int g = gb + gl;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    if (r < R && g < G) {
      PV<T, V> ps;
      ps.load(s + n * C + (int64_t)g * V);
      const int64_t base = n * HW * C + (int64_t)g * V;
      for (int64_t row = row0 + r; row < row1; row += (int64_t)kU * R) {
#pragma unroll
        for (int k = 0; k < kU; ++k) {
          const int64_t rr = row + (int64_t)k * R;
          if (rr < row1) {
            PV<T, V> pd, px;
            pd.load(dy + base + rr * C);
            px.load(x + base + rr * C);
#pragma unroll
            for (int j = 0; j < V; ++j) {
              acc[j] = fmaf(pd.v[j], px.v[j], acc[j]);
              px.v[j] = IDENT ? fmaf(pd.v[j], ps.v[j], pd.v[j]) : pd.v[j] * ps.v[j];
            }
            if (DX) px.store(dx + base + rr * C);
          }
        }
      }
    }
    __syncthreads();
    if (r < R && g < G) {
#pragma unroll
      for (int j = 0; j < V; ++j) sm[r * (GT * V) + gl * V + j] = acc[j];
    }
    __syncthreads();
    for (int t = tid; t < GT * V; t += kT) {
      const int64_t c = (int64_t)gb * V + t;
      if (c < C) {
        float a = 0.f;
        for (int q = 0; q < R; ++q) a += sm[q * (GT * V) + t];
        partial[(n * S + blockIdx.x) * C + c] = a;
      }
    }
  }
}

template <typename T, int V, bool IDENT>
__global__ __launch_bounds__(kT) void cs_bwd_nchw(const T* __restrict__ dy, const T* __restrict__ x,
                                                  const T* __restrict__ s, T* __restrict__ dx,
                                                  T* __restrict__ ds, int64_t HW) {
  __shared__ float sm[2 * (kT / 64)];
  const float g = ld1<T>(s + blockIdx.x);
  const int64_t off = (int64_t)blockIdx.x * HW;
  float acc = 0.f, dummy = 0.f;
  for (int64_t i = (int64_t)threadIdx.x * V; i < HW; i += (int64_t)kT * V) {
    PV<T, V> pd, px;
    pd.load(dy + off + i);
    px.load(x + off + i);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      acc = fmaf(pd.v[j], px.v[j], acc);
      px.v[j] = IDENT ? fmaf(pd.v[j], g, pd.v[j]) : pd.v[j] * g;
    }
    px.store(dx + off + i);
  }
  block_sum2(acc, dummy, sm);
  if (threadIdx.x == 0) st1<T>(ds + blockIdx.x, acc);
}

// ---- max pooling (channels_last) ---------------------------------------------------
// ResNet's stem pool nn.MaxPool2d(3, stride 2, padding 1) (furnace/base_model/resnet.py:132).
// Eager PyTorch-ROCm keeps int64 argmax indices (8 B per output element) and spends
// 0.33 ms forward + 0.82 ms backward on the 16x64x512x512 map; here the argmax is one
// byte (position inside the window), the forward is one 16-byte-vector pass and the
// backward is a deterministic gather over the <= ceil(K/S)^2 windows covering a pixel.
// Tie / NaN rule as at::native max_pool2d: first maximum in (ky, kx) scan order, NaN wins.
template <typename T, int V>
__global__ __launch_bounds__(kT) void maxpool_fwd_nhwc(const T* __restrict__ x, T* __restrict__ y,
                                                       uint8_t* __restrict__ idx, int64_t N, int C, int IH,
                                                       int IW, int OH, int OW, int K, int S, int P) {
  const int G = C / V;
  const int64_t total = N * OH * (int64_t)OW * G;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int g = (int)(i % G);
    int64_t t = i / G;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int64_t n = t / OH;
    const int y0 = oy * S - P, x0 = ox * S - P;
    float m[V];
    int am[V];
    bool first = true;
    for (int ky = 0; ky < K; ++ky) {
      const int iy = y0 + ky;
      if (iy < 0 || iy >= IH) continue;
      for (int kx = 0; kx < K; ++kx) {
        const int ix = x0 + kx;
        if (ix < 0 || ix >= IW) continue;
        PV<T, V> p;
        p.load(x + ((n * IH + iy) * (int64_t)IW + ix) * C + g * V);
        const int pos = ky * K + kx;
#pragma unroll
        for (int j = 0; j < V; ++j) {
          if (first || p.v[j] > m[j] || p.v[j] != p.v[j]) { m[j] = p.v[j]; am[j] = pos; }
        }
        first = false;
      }
    }
    PV<T, V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = m[j];
    o.store(y + i * V);
    uint8_t* ip = idx + i * V;
#pragma unroll
    for (int j = 0; j < V; ++j) ip[j] = (uint8_t)am[j];
  }
}

// FIX = true: the ResNet pooling geometry K = 3, S = 2, P = 1 as compile-time constants (shifts instead of
// runtime integer divisions); the window positions of a vector arrive as one 4- or 8-byte load.
template <typename T, int V, bool FIX>
__global__ __launch_bounds__(kT) void maxpool_bwd_nhwc(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                       T* __restrict__ dx, int64_t N, int C, int IH, int IW,
                                                       int OH, int OW, int K_, int S_, int P_) {
  const int K = FIX ? 3 : K_, S = FIX ? 2 : S_, P = FIX ? 1 : P_;
  const int G = C / V;
  const int64_t total = N * IH * (int64_t)IW * G;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int g = (int)(i % G);
    int64_t t = i / G;
    const int ix = (int)(t % IW); t /= IW;
    const int iy = (int)(t % IH);
    const int64_t n = t / IH;
    // windows (oy, ox) with oy*S - P <= iy <= oy*S - P + K - 1
    int oy_lo = (iy + P - K + 1 + S - 1) / S; if (iy + P - K + 1 < 0) oy_lo = 0;
    int oy_hi = (iy + P) / S; if (oy_hi > OH - 1) oy_hi = OH - 1;
    int ox_lo = (ix + P - K + 1 + S - 1) / S; if (ix + P - K + 1 < 0) ox_lo = 0;
    int ox_hi = (ix + P) / S; if (ox_hi > OW - 1) ox_hi = OW - 1;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
      const int ky = iy - (oy * S - P);
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const int pos = ky * K + (ix - (ox * S - P));
        const int64_t o = ((n * OH + oy) * (int64_t)OW + ox) * C + g * V;
        PV<T, V> d;
        d.load(dy + o);
        uint32_t w[2];
        if (V == 8) {
          const uint2 u = *reinterpret_cast<const uint2*>(idx + o);
          w[0] = u.x; w[1] = u.y;
        } else {
          w[0] = *reinterpret_cast<const uint32_t*>(idx + o); w[1] = 0u;
        }
#pragma unroll
        for (int j = 0; j < V; ++j)
          if ((int)((w[j >> 2] >> (8 * (j & 3))) & 0xffu) == pos) acc[j] += d.v[j];
      }
    }
    PV<T, V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = acc[j];
    o.store(dx + i * V);
  }
}


// ---------------------------------------------------------------------------
// Adaptive average pooling to a SMALL grid, channels_last — the pyramid pooling of PSPNet / PSANet's conv6 input
// (pspnet network.py:75-109: nn.AdaptiveAvgPool2d(1 / 2 / 3 / 6) of a [2, 2048, 90, 90] map).  The framework's NHWC kernel
// takes 2.6 ms per call there (one thread per output element walking its 2000-pixel window; profiles/
// r03_kernel_stats_pspnet.csv: 20 % of the PSPNet step).  Here a window is split over blocks by rows and over the threads
// of a block by rows again, channels across lanes (16-byte loads), fixed-order folds: two launches, deterministic.
// Bin boundaries as ATen: [floor(o * H / OH), ceil((o + 1) * H / OH)).
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ int ap_start(int o, int in, int out) { return (int)(((int64_t)o * in) / out); }
__host__ __device__ __forceinline__ int ap_end(int o, int in, int out) { return (int)((((int64_t)o + 1) * in + out - 1) / out); }

// stage 1: block (bin, z, channel tile): rows z, z + RZ, ... of the bin's window -> partial[bin][z][C] (fp32 sums)
template <typename T, int V>
__global__ __launch_bounds__(kT) void apool_partial_nhwc(const T* __restrict__ x, int H, int W, int C, int OH, int OW,
                                                         int gpb, int RZ, float* __restrict__ partial) {
  __shared__ float red[kT * V];
  const int G = C / V, RS = kT / gpb;
  const int tid = threadIdx.x, gl = tid % gpb, rs = tid / gpb;
  const int g = blockIdx.z * gpb + gl;
  int64_t t = blockIdx.x;
  const int ow = (int)(t % OW); t /= OW;
  const int oh = (int)(t % OH);
  const int64_t n = t / OH;
  const int h0 = ap_start(oh, H, OH), h1 = ap_end(oh, H, OH), w0 = ap_start(ow, W, OW), w1 = ap_end(ow, W, OW);
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
  if (rs < RS && g < G) {
    const T* b = x + n * H * (int64_t)W * C + (int64_t)g * V;
    for (int h = h0 + blockIdx.y * RS + rs; h < h1; h += RZ * RS)
      for (int w = w0; w < w1; ++w) {
        PV<T, V> p;
        p.load(b + ((int64_t)h * W + w) * C);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += p.v[j];
      }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) red[tid * V + j] = acc[j];
  __syncthreads();
  if (rs == 0 && g < G) {
    float* o = partial + ((int64_t)blockIdx.x * RZ + blockIdx.y) * C + (int64_t)g * V;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float sum = 0.f;
      for (int q = 0; q < RS; ++q) sum += red[(q * gpb + gl) * V + j];
      o[j] = sum;
    }
  }
}

// stage 2: out[bin][c] = (sum over z of partial[bin][z][c]) / window size
template <typename T>
__global__ __launch_bounds__(kT) void apool_finish(const float* __restrict__ partial, int64_t bins, int C, int RZ, int H,
                                                   int W, int OH, int OW, T* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x;
  if (i >= bins * C) return;
  const int64_t bin = i / C;
  const int c = (int)(i % C);
  const int ow = (int)(bin % OW), oh = (int)((bin / OW) % OH);
  const float cnt = (float)((ap_end(oh, H, OH) - ap_start(oh, H, OH)) * (ap_end(ow, W, OW) - ap_start(ow, W, OW)));
  float sum = 0.f;
  for (int z = 0; z < RZ; ++z) sum += partial[(bin * RZ + z) * C + c];
  st1<T>(out + i, sum / cnt);
}

// backward: dx[n, h, w, :] = sum over the bins that contain (h, w) of dout[n, oh, ow, :] / window size
template <typename T, int V>
__global__ __launch_bounds__(kT) void apool_bwd_nhwc(const T* __restrict__ dout, int64_t N, int H, int W, int C, int OH,
                                                     int OW, T* __restrict__ dx) {
  const int64_t G = C / V, total = N * H * (int64_t)W * G;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int64_t g = i % G;
    int64_t t = i / G;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int64_t n = t / H;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    const int oh_c = (int)(((int64_t)h * OH) / H), ow_c = (int)(((int64_t)w * OW) / W);
    for (int oh = oh_c > 0 ? oh_c - 1 : 0; oh <= oh_c + 1 && oh < OH; ++oh) {
      const int h0 = ap_start(oh, H, OH), h1 = ap_end(oh, H, OH);
      if (h < h0 || h >= h1) continue;
      for (int ow = ow_c > 0 ? ow_c - 1 : 0; ow <= ow_c + 1 && ow < OW; ++ow) {
        const int w0 = ap_start(ow, W, OW), w1 = ap_end(ow, W, OW);
        if (w < w0 || w >= w1) continue;
        const float inv = 1.f / (float)((h1 - h0) * (w1 - w0));
        PV<T, V> p;
        p.load(dout + ((n * OH + oh) * OW + ow) * C + g * V);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += p.v[j] * inv;
      }
    }
    PV<T, V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = acc[j];
    o.store(dx + i * V);
  }
}

struct ApGeom { int gpb, RZ, ctiles; };
static ApGeom ap_geom(int64_t N, int C, int H, int OH, int OW, int V) {
  ApGeom g;
  const int G = C / V;
  g.gpb = kT;
  while (g.gpb > G) g.gpb >>= 1;
  if (g.gpb < 1) g.gpb = 1;
  g.ctiles = (G + g.gpb - 1) / g.gpb;
  const int RS = kT / g.gpb;
  const int rows = (H + OH - 1) / OH + 1;                            // longest window
  int64_t rz = (1024 + N * OH * OW * g.ctiles - 1) / (N * OH * OW * g.ctiles);   // ~1024 blocks in all
  const int64_t rz_max = (rows + RS - 1) / RS;                       // at least one row per splitter
  if (rz > rz_max) rz = rz_max;
  if (rz < 1) rz = 1;
  if (rz > 64) rz = 64;
  g.RZ = (int)rz;
  return g;
}

struct GapGeom { int gt, R, S; int64_t rpb; };
static GapGeom gap_geom(int64_t N, int64_t C, int64_t HW, int V) {
  GapGeom g;
  int64_t G = C / V;
  g.gt = (int)(G < kT ? G : kT);
  g.R = kT / g.gt;
  int64_t s = (1024 + N - 1) / N;
  int64_t min_rows = (int64_t)g.R * kU * 2;
  int64_t max_s = (HW + min_rows - 1) / min_rows;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 256) s = 256;
  g.rpb = (HW + s - 1) / s;
  g.rpb = (g.rpb + g.R - 1) / g.R * g.R;
  g.S = (int)((HW + g.rpb - 1) / g.rpb);
  return g;
}
// ---- channel concatenation of two channels_last maps: out[m][0 : Ca] = a[m], out[m][Ca :] = b[m] (rows = pixels) -------
// FeatureFusion.forward's `torch.cat([x1, x2], dim=1)` (furnace/seg_opr/seg_oprs.py:233-235): the framework's batched
// copy runs at 2.5 TB/s on [16, 128, 128, 128] x 2 (107 us per step); one 16-byte vector per thread and iteration here.
__global__ __launch_bounds__(256) void cat2_rows_k(const uint4* __restrict__ a, const uint4* __restrict__ b,
                                                   uint4* __restrict__ out, int64_t rows, int va, int vb) {
  const int vo = va + vb;
  const int64_t n = rows * vo;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / vo;
    const int c = (int)(i - m * vo);
    out[i] = c < va ? a[m * va + c] : b[m * vb + (c - va)];
  }
}

}  // namespace tsg

using namespace tsg;

extern "C" {

size_t tsg_gap_ws_bytes(int layout, int64_t N, int64_t C, int64_t HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  if (layout == TSG_NCHW) return 16;
  int smax = 1;
  const int vs[3] = {1, 4, 8};
  for (int i = 0; i < 3; ++i) if (C % vs[i] == 0) { int s = gap_geom(N, C, HW, vs[i]).S; if (s > smax) smax = s; }
  return (size_t)N * smax * C * sizeof(float);
}

int tsg_gap_fwd(const void* x, void* out, int dtype, int layout, int64_t N, int64_t C, int64_t HW, void* ws,
                size_t ws_bytes, void* stream) {
  if (!x || !out || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  if (ws_bytes < tsg_gap_ws_bytes(layout, N, C, HW)) return TSG_E_WS;
  hipStream_t st = (hipStream_t)stream;
  const int native = dtype == TSG_BF16 ? 8 : 4;
  const float inv = 1.f / (float)HW;
  if (layout == TSG_NCHW) {
    const bool vec = HW % native == 0 && aligned16(x);
    if (N * C > 0x7fffffffLL) return TSG_E_SHAPE;
    dim3 grid((unsigned)(N * C));
    if (dtype == TSG_F32) {
      if (vec) hipLaunchKernelGGL((gap_plane_nchw<float, 4>), grid, dim3(kT), 0, st, (const float*)x, HW, inv, (float*)out);
      else hipLaunchKernelGGL((gap_plane_nchw<float, 1>), grid, dim3(kT), 0, st, (const float*)x, HW, inv, (float*)out);
    } else {
      if (vec) hipLaunchKernelGGL((gap_plane_nchw<bf16_t, 8>), grid, dim3(kT), 0, st, (const bf16_t*)x, HW, inv, (bf16_t*)out);
      else hipLaunchKernelGGL((gap_plane_nchw<bf16_t, 1>), grid, dim3(kT), 0, st, (const bf16_t*)x, HW, inv, (bf16_t*)out);
    }
    TSG_CHECK_LAUNCH();
    return 0;
  }
  if (layout != TSG_NHWC) return TSG_E_LAYOUT;
  const int V = (C % native == 0 && aligned16(x)) ? native : 1;
  GapGeom g = gap_geom(N, C, HW, V);
  dim3 grid((unsigned)g.S, (unsigned)N);
  const size_t sh = (size_t)g.R * g.gt * V * (dtype == TSG_F32 ? sizeof(double) : sizeof(float));
#define GO(T, VV) hipLaunchKernelGGL((gap_partial_nhwc<T, VV>), grid, dim3(kT), sh, st, (const T*)x, HW, C, g.gt, \
                                     g.R, g.rpb, g.S, (float*)ws)
  if (dtype == TSG_F32) { if (V == 4) GO(float, 4); else GO(float, 1); }
  else { if (V == 8) GO(bf16_t, 8); else GO(bf16_t, 1); }
#undef GO
  TSG_CHECK_LAUNCH();
  const int64_t NC = N * C;
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((gap_finish<float>), dim3(ceil_div_i(NC, kT)), dim3(kT), 0, st, (const float*)ws, g.S, NC, C, inv, (float*)out);
  else
    hipLaunchKernelGGL((gap_finish<bf16_t>), dim3(ceil_div_i(NC, kT)), dim3(kT), 0, st, (const float*)ws, g.S, NC, C, inv, (bf16_t*)out);
  TSG_CHECK_LAUNCH();
  return 0;
}

size_t tsg_adaptive_avgpool_nhwc_ws_bytes(int dtype, int64_t N, int C, int H, int W, int OH, int OW) {
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (N <= 0 || C <= 0 || C % V || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return 0;
  const ApGeom g = ap_geom(N, C, H, OH, OW, V);
  return (size_t)N * OH * OW * g.RZ * C * sizeof(float);
}

int tsg_adaptive_avgpool_nhwc_fwd(const void* x, void* out, int dtype, int64_t N, int C, int H, int W, int OH, int OW,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (!x || !out || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (N <= 0 || C <= 0 || C % V || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || OH > H || OW > W) return TSG_E_SHAPE;
  if (N * OH * (int64_t)OW > 0x7fffffffLL) return TSG_E_SHAPE;
  if (ws_bytes < tsg_adaptive_avgpool_nhwc_ws_bytes(dtype, N, C, H, W, OH, OW)) return TSG_E_WS;
  if (!aligned16(x) || !aligned16(ws)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const ApGeom g = ap_geom(N, C, H, OH, OW, V);
  const int64_t bins = N * OH * OW;
  const dim3 grid((unsigned)bins, (unsigned)g.RZ, (unsigned)g.ctiles);
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((apool_partial_nhwc<float, 4>), grid, dim3(kT), 0, st, (const float*)x, H, W, C, OH, OW, g.gpb, g.RZ,
                       (float*)ws);
  else
    hipLaunchKernelGGL((apool_partial_nhwc<bf16_t, 8>), grid, dim3(kT), 0, st, (const bf16_t*)x, H, W, C, OH, OW, g.gpb, g.RZ,
                       (float*)ws);
  TSG_CHECK_LAUNCH();
  const dim3 fgrid((unsigned)ceil_div_i(bins * C, kT));
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((apool_finish<float>), fgrid, dim3(kT), 0, st, (const float*)ws, bins, C, g.RZ, H, W, OH, OW, (float*)out);
  else
    hipLaunchKernelGGL((apool_finish<bf16_t>), fgrid, dim3(kT), 0, st, (const float*)ws, bins, C, g.RZ, H, W, OH, OW,
                       (bf16_t*)out);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_adaptive_avgpool_nhwc_bwd(const void* dout, void* dx, int dtype, int64_t N, int C, int H, int W, int OH, int OW,
                                  void* stream) {
  if (!dout || !dx) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (N <= 0 || C <= 0 || C % V || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || OH > H || OW > W) return TSG_E_SHAPE;
  if (!aligned16(dout) || !aligned16(dx)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (N * H * (int64_t)W * (C / V) + kT - 1) / kT;
  if (blocks > 8192) blocks = 8192;
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((apool_bwd_nhwc<float, 4>), dim3((unsigned)blocks), dim3(kT), 0, st, (const float*)dout, N, H, W, C, OH,
                       OW, (float*)dx);
  else
    hipLaunchKernelGGL((apool_bwd_nhwc<bf16_t, 8>), dim3((unsigned)blocks), dim3(kT), 0, st, (const bf16_t*)dout, N, H, W, C,
                       OH, OW, (bf16_t*)dx);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_gap_bwd(const void* dout, void* dx, int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                void* stream) {
  if (!dout || !dx) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int native = dtype == TSG_BF16 ? 8 : 4;
  const float inv = 1.f / (float)HW;
  if (layout == TSG_NCHW) {
    const bool vec = HW % native == 0 && aligned16(dx);
    dim3 grid((unsigned)(N * C));
    if (dtype == TSG_F32) {
      if (vec) hipLaunchKernelGGL((gap_bwd_nchw<float, 4>), grid, dim3(kT), 0, st, (const float*)dout, HW, inv, (float*)dx);
      else hipLaunchKernelGGL((gap_bwd_nchw<float, 1>), grid, dim3(kT), 0, st, (const float*)dout, HW, inv, (float*)dx);
    } else {
      if (vec) hipLaunchKernelGGL((gap_bwd_nchw<bf16_t, 8>), grid, dim3(kT), 0, st, (const bf16_t*)dout, HW, inv, (bf16_t*)dx);
      else hipLaunchKernelGGL((gap_bwd_nchw<bf16_t, 1>), grid, dim3(kT), 0, st, (const bf16_t*)dout, HW, inv, (bf16_t*)dx);
    }
    TSG_CHECK_LAUNCH();
    return 0;
  }
  if (layout != TSG_NHWC) return TSG_E_LAYOUT;
  const int V = (C % native == 0 && aligned16(dx) && aligned16(dout)) ? native : 1;
  int64_t gx = (HW * (C / V) + kT - 1) / kT;
  if (gx > 2048) gx = 2048;
  dim3 grid((unsigned)gx, (unsigned)N);
#define GO(T, VV) hipLaunchKernelGGL((gap_bwd_nhwc<T, VV>), grid, dim3(kT), 0, st, (const T*)dout, HW, C, inv, (T*)dx)
  if (dtype == TSG_F32) { if (V == 4) GO(float, 4); else GO(float, 1); }
  else { if (V == 8) GO(bf16_t, 8); else GO(bf16_t, 1); }
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}


int tsg_chanscale_fwd(const void* x, const void* s, void* y, int dtype, int layout, int64_t N, int64_t C,
                      int64_t HW, int add_identity, void* stream) {
  if (!x || !s || !y) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int native = dtype == TSG_BF16 ? 8 : 4;
  if (layout == TSG_NCHW) {
    const bool vec = HW % native == 0 && aligned16(x) && aligned16(y);
    dim3 grid((unsigned)(N * C));
#define GO(T, VV, I) hipLaunchKernelGGL((cs_fwd_nchw<T, VV, I>), grid, dim3(kT), 0, st, (const T*)x, (const T*)s, (T*)y, HW)
    if (dtype == TSG_F32) { if (vec) { if (add_identity) GO(float, 4, true); else GO(float, 4, false); } else { if (add_identity) GO(float, 1, true); else GO(float, 1, false); } }
    else { if (vec) { if (add_identity) GO(bf16_t, 8, true); else GO(bf16_t, 8, false); } else { if (add_identity) GO(bf16_t, 1, true); else GO(bf16_t, 1, false); } }
#undef GO
    TSG_CHECK_LAUNCH();
    return 0;
  }
  if (layout != TSG_NHWC) return TSG_E_LAYOUT;
  const bool vec = C % native == 0 && aligned16(x) && aligned16(y) && aligned16(s);
  const int V = vec ? native : 1;
  int64_t gx = (HW * (C / V) + kT - 1) / kT;
  if (gx > 2048) gx = 2048;
  dim3 grid((unsigned)gx, (unsigned)N);
#define GO(T, VV, I) hipLaunchKernelGGL((cs_fwd_nhwc<T, VV, I>), grid, dim3(kT), 0, st, (const T*)x, (const T*)s, (T*)y, HW, C)
  if (dtype == TSG_F32) { if (vec) { if (add_identity) GO(float, 4, true); else GO(float, 4, false); } else { if (add_identity) GO(float, 1, true); else GO(float, 1, false); } }
  else { if (vec) { if (add_identity) GO(bf16_t, 8, true); else GO(bf16_t, 8, false); } else { if (add_identity) GO(bf16_t, 1, true); else GO(bf16_t, 1, false); } }
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_chanscale_bwd(const void* dy, const void* x, const void* s, void* dx, void* ds, int dtype, int layout,
                      int64_t N, int64_t C, int64_t HW, int add_identity, void* ws, size_t ws_bytes,
                      void* stream) {
  if (!dy || !x || !s || !dx || !ds || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  if (ws_bytes < tsg_gap_ws_bytes(layout, N, C, HW)) return TSG_E_WS;
  hipStream_t st = (hipStream_t)stream;
  const int native = dtype == TSG_BF16 ? 8 : 4;
  if (layout == TSG_NCHW) {
    const bool vec = HW % native == 0 && aligned16(x) && aligned16(dy) && aligned16(dx);
    dim3 grid((unsigned)(N * C));
#define GO(T, VV, I) hipLaunchKernelGGL((cs_bwd_nchw<T, VV, I>), grid, dim3(kT), 0, st, (const T*)dy, (const T*)x, (const T*)s, (T*)dx, (T*)ds, HW)
    if (dtype == TSG_F32) { if (vec) { if (add_identity) GO(float, 4, true); else GO(float, 4, false); } else { if (add_identity) GO(float, 1, true); else GO(float, 1, false); } }
    else { if (vec) { if (add_identity) GO(bf16_t, 8, true); else GO(bf16_t, 8, false); } else { if (add_identity) GO(bf16_t, 1, true); else GO(bf16_t, 1, false); } }
#undef GO
    TSG_CHECK_LAUNCH();
    return 0;
  }
  if (layout != TSG_NHWC) return TSG_E_LAYOUT;
  const bool vec = C % native == 0 && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(s);
  const int V = vec ? native : 1;
  GapGeom g = gap_geom(N, C, HW, V);
  dim3 grid((unsigned)g.S, (unsigned)N);
  const size_t sh = (size_t)g.R * g.gt * V * sizeof(float);
#define GO(T, VV, I) hipLaunchKernelGGL((cs_bwd_nhwc<T, VV, I>), grid, dim3(kT), sh, st, (const T*)dy, (const T*)x, (const T*)s, \
                                        (T*)dx, HW, C, g.gt, g.R, g.rpb, g.S, (float*)ws)
  if (dtype == TSG_F32) { if (vec) { if (add_identity) GO(float, 4, true); else GO(float, 4, false); } else { if (add_identity) GO(float, 1, true); else GO(float, 1, false); } }
  else { if (vec) { if (add_identity) GO(bf16_t, 8, true); else GO(bf16_t, 8, false); } else { if (add_identity) GO(bf16_t, 1, true); else GO(bf16_t, 1, false); } }
#undef GO
  TSG_CHECK_LAUNCH();
  const int64_t NC = N * C;
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((gap_finish<float>), dim3(ceil_div_i(NC, kT)), dim3(kT), 0, st, (const float*)ws, g.S, NC, C, 1.f, (float*)ds);
  else
    hipLaunchKernelGGL((gap_finish<bf16_t>), dim3(ceil_div_i(NC, kT)), dim3(kT), 0, st, (const float*)ws, g.S, NC, C, 1.f, (bf16_t*)ds);
  TSG_CHECK_LAUNCH();
  return 0;
}


/* The two halves of tsg_chanscale_bwd for a gate whose scale was computed FROM the pooled map it gates (seg_oprs.py:192-238):
 * _ds: ds[n, c] = sum_p dy x only (NHWC, vector path: C % (16 / elem) == 0, 16-byte aligned — TSG_E_LAYOUT / TSG_E_ALIGN
 * otherwise); _dx: dx = dy s (+ dy) + gadd[n, c], gadd = the pooled branch's gradient per pixel, [N, C] of the tensor dtype. */
int tsg_chanscale_bwd_ds(const void* dy, const void* x, void* ds, int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                         void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !x || !ds || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  if (layout != TSG_NHWC) return TSG_E_LAYOUT;
  if (ws_bytes < tsg_gap_ws_bytes(layout, N, C, HW)) return TSG_E_WS;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (C % V) return TSG_E_SHAPE;
  if (!aligned16(x) || !aligned16(dy)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  GapGeom g = gap_geom(N, C, HW, V);
  dim3 grid((unsigned)g.S, (unsigned)N);
  const size_t sh = (size_t)g.R * g.gt * V * sizeof(float);
  // (s is only read for dx: any valid pointer — x — stands in; IDENT does not matter)
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((cs_bwd_nhwc<float, 4, false, false>), grid, dim3(kT), sh, st, (const float*)dy, (const float*)x,
                       (const float*)x, (float*)nullptr, HW, C, g.gt, g.R, g.rpb, g.S, (float*)ws);
  else
    hipLaunchKernelGGL((cs_bwd_nhwc<bf16_t, 8, false, false>), grid, dim3(kT), sh, st, (const bf16_t*)dy, (const bf16_t*)x,
                       (const bf16_t*)x, (bf16_t*)nullptr, HW, C, g.gt, g.R, g.rpb, g.S, (float*)ws);
  TSG_CHECK_LAUNCH();
  const int64_t NC = N * C;
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((gap_finish<float>), dim3(ceil_div_i(NC, kT)), dim3(kT), 0, st, (const float*)ws, g.S, NC, C, 1.f, (float*)ds);
  else
    hipLaunchKernelGGL((gap_finish<bf16_t>), dim3(ceil_div_i(NC, kT)), dim3(kT), 0, st, (const float*)ws, g.S, NC, C, 1.f, (bf16_t*)ds);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_chanscale_bwd_dx(const void* dy, const void* s, const void* gadd, void* dx, int dtype, int layout, int64_t N, int64_t C,
                         int64_t HW, int add_identity, void* stream) {
  if (!dy || !s || !gadd || !dx) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  if (layout != TSG_NHWC) return TSG_E_LAYOUT;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (C % V) return TSG_E_SHAPE;
  if (!aligned16(dy) || !aligned16(dx) || !aligned16(s) || !aligned16(gadd)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  int64_t gx = (HW * (C / V) + kT - 1) / kT;
  if (gx > 2048) gx = 2048;
  dim3 grid((unsigned)gx, (unsigned)N);
#define GO(T, VV, I) hipLaunchKernelGGL((cs_dx_nhwc<T, VV, I>), grid, dim3(kT), 0, st, (const T*)dy, (const T*)s, (const T*)gadd, (T*)dx, HW, C)
  if (dtype == TSG_F32) { if (add_identity) GO(float, 4, true); else GO(float, 4, false); }
  else { if (add_identity) GO(bf16_t, 8, true); else GO(bf16_t, 8, false); }
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_maxpool_nhwc_fwd(const void* x, void* y, void* argmax_u8, int dtype, int64_t N, int C, int IH, int IW,
                         int OH, int OW, int K, int S, int P, void* stream) {
  if (!x || !y || !argmax_u8) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (N <= 0 || C <= 0 || C % V || K <= 0 || K > 15 || S <= 0 || P < 0 || 2 * P > K) return TSG_E_SHAPE;
  if (OH != (IH + 2 * P - K) / S + 1 || OW != (IW + 2 * P - K) / S + 1) return TSG_E_SHAPE;
  if (!aligned16(x) || !aligned16(y)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  int64_t g = (N * OH * (int64_t)OW * (C / V) + kT - 1) / kT;
  if (g > 16384) g = 16384;
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((maxpool_fwd_nhwc<float, 4>), dim3((unsigned)g), dim3(kT), 0, st, (const float*)x, (float*)y,
                       (uint8_t*)argmax_u8, N, C, IH, IW, OH, OW, K, S, P);
  else
    hipLaunchKernelGGL((maxpool_fwd_nhwc<bf16_t, 8>), dim3((unsigned)g), dim3(kT), 0, st, (const bf16_t*)x,
                       (bf16_t*)y, (uint8_t*)argmax_u8, N, C, IH, IW, OH, OW, K, S, P);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_maxpool_nhwc_bwd(const void* dy, const void* argmax_u8, void* dx, int dtype, int64_t N, int C, int IH,
                         int IW, int OH, int OW, int K, int S, int P, void* stream) {
  if (!dy || !dx || !argmax_u8) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (N <= 0 || C <= 0 || C % V || K <= 0 || K > 15 || S <= 0 || P < 0) return TSG_E_SHAPE;
  if (!aligned16(dy) || !aligned16(dx)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  int64_t g = (N * IH * (int64_t)IW * (C / V) + kT - 1) / kT;
  if (g > 16384) g = 16384;
  const bool fix = K == 3 && S == 2 && P == 1;
  if (dtype == TSG_F32) {
    if (fix)
      hipLaunchKernelGGL((maxpool_bwd_nhwc<float, 4, true>), dim3((unsigned)g), dim3(kT), 0, st, (const float*)dy,
                         (const uint8_t*)argmax_u8, (float*)dx, N, C, IH, IW, OH, OW, K, S, P);
    else
      hipLaunchKernelGGL((maxpool_bwd_nhwc<float, 4, false>), dim3((unsigned)g), dim3(kT), 0, st, (const float*)dy,
                         (const uint8_t*)argmax_u8, (float*)dx, N, C, IH, IW, OH, OW, K, S, P);
  } else {
    if (fix)
      hipLaunchKernelGGL((maxpool_bwd_nhwc<bf16_t, 8, true>), dim3((unsigned)g), dim3(kT), 0, st, (const bf16_t*)dy,
                         (const uint8_t*)argmax_u8, (bf16_t*)dx, N, C, IH, IW, OH, OW, K, S, P);
    else
      hipLaunchKernelGGL((maxpool_bwd_nhwc<bf16_t, 8, false>), dim3((unsigned)g), dim3(kT), 0, st, (const bf16_t*)dy,
                         (const uint8_t*)argmax_u8, (bf16_t*)dx, N, C, IH, IW, OH, OW, K, S, P);
  }
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_cat2_rows(const void* a, const void* b, void* out, int64_t rows, int64_t row_bytes_a, int64_t row_bytes_b,
                  void* stream) {
  if (!a || !b || !out) return TSG_E_NULL;
  if (rows <= 0 || row_bytes_a <= 0 || row_bytes_b <= 0 || row_bytes_a % 16 || row_bytes_b % 16 ||
      row_bytes_a + row_bytes_b > 0x7fffffffLL)
    return TSG_E_SHAPE;
  if (!aligned16(a) || !aligned16(b) || !aligned16(out)) return TSG_E_ALIGN;
  const int va = (int)(row_bytes_a / 16), vb = (int)(row_bytes_b / 16);
  int64_t g = (rows * (va + vb) + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(cat2_rows_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, (const uint4*)a, (const uint4*)b,
                     (uint4*)out, rows, va, vb);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
