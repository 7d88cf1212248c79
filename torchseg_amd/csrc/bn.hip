// SyncBatchNorm kernels for gfx950 (wave64, 256 CUs / 8 XCDs).
//
// Arithmetic follows the in-tree statement of the reference's SyncBN
// (furnace/legacy/sync_bn/syncbn.py:86-98, src/gpu/syncbn_kernel.cu:12-23,
// 73-174) but none of its launch shape: the reference runs ONE block per
// channel (<= C blocks, <= 512 threads, strided NCHW reads).  Here every pass
// is a streaming kernel with 16-byte loads, >= ~2048 workgroups, fp32
// per-lane accumulators, a fixed-order (deterministic) two-stage reduction
// whose second stage runs in fp64, and BN+ReLU(+residual) fused so an
// activation is read once and written once per pass.
//
// All passes are HBM-bound: algorithmic bytes per element (s = element size)
//   stats        1 read            = s
//   apply_fwd    1 read + 1 write  = 2s   (+s with residual)
//   bwd_reduce   2 reads           = 2s   (+s when the ReLU mask comes from y)
//   bwd_apply    2 reads + 1 write = 3s   (+s mask-from-y, +s dres)
#include "tsg_common.h"

namespace tsg {

constexpr int kThreads = 256;
constexpr int kUnroll = 4;
constexpr int kTargetBlocks = 2048;  // 8 per CU
constexpr int kMaxSplit = 256;

// ---- element packs: V elements per lane-access, unpacked to fp32 ----------
template <typename T, int V> struct Pack;
template <> struct Pack<float, 4> : Vec<float> {};
template <> struct Pack<bf16_t, 8> : Vec<bf16_t> {};
template <typename T> struct Pack<T, 1> {
  float v[1];
  __device__ __forceinline__ void load(const T* p) { v[0] = ld1<T>(p); }
  __device__ __forceinline__ void store(T* p) const { st1<T>(p, v[0]); }
};

// scale/shift of the affine map y = a*x + b for channel c; shared by forward
// and the backward mask recompute so both see bit-identical pre-activations.
__device__ __forceinline__ void bn_coef(const float* mean, const float* invstd,
                                        const float* gamma, const float* beta,
                                        int64_t c, float& a, float& b) {
  const float g = gamma ? gamma[c] : 1.f;
  const float be = beta ? beta[c] : 0.f;
  a = g * invstd[c];
  b = fmaf(-mean[c], a, be);
}

struct NchwGeom {
  int seg;    // elements of one plane segment handled by one block-iteration
  int segs;   // segments per plane
  int split;  // S: slices per channel
};

static NchwGeom nchw_geom(int64_t N, int64_t C, int64_t HW, int V) {
  NchwGeom g;
  g.seg = kThreads * V * kUnroll;
  g.segs = ceil_div_i(HW, g.seg);
  int64_t units = N * g.segs;
  int64_t want = (kTargetBlocks + C - 1) / C;
  int64_t s = want < 1 ? 1 : want;
  if (s > units) s = units;
  if (s > kMaxSplit) s = kMaxSplit;
  g.split = (int)s;
  return g;
}

struct NhwcGeom {
  int gt;              // channel groups handled per block (<= 256)
  int rows_per_iter;   // R
  int ytiles;          // gridDim.y
  int split;           // S = gridDim.x
  int64_t rows_per_block;
};

static NhwcGeom nhwc_geom(int64_t M, int64_t C, int V) {
  NhwcGeom g;
  int64_t G = C / V;
  g.gt = (int)(G < kThreads ? G : kThreads);
  g.rows_per_iter = kThreads / g.gt;
  g.ytiles = ceil_div_i(G, g.gt);
  int64_t min_rows = (int64_t)g.rows_per_iter * kUnroll;
  int64_t s = ceil_div_i(kTargetBlocks, g.ytiles);
  int64_t max_s = (M + min_rows - 1) / min_rows;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  if (s > 4096) s = 4096;
  g.rows_per_block = (M + s - 1) / s;
  // keep each block's row range a multiple of R so that lanes stay row-aligned
  g.rows_per_block = ((g.rows_per_block + g.rows_per_iter - 1) / g.rows_per_iter) * g.rows_per_iter;
  g.split = ceil_div_i(M, g.rows_per_block);
  return g;
}

static int pick_vec(int dtype, int layout, int64_t C, int64_t HW, const void* p0,
                    const void* p1, const void* p2, const void* p3, const void* p4) {
  const int native = (dtype == TSG_BF16) ? 8 : 4;
  const int64_t inner = (layout == TSG_NCHW) ? HW : C;
  if (inner % native != 0) return 1;
  const void* ps[5] = {p0, p1, p2, p3, p4};
  for (int i = 0; i < 5; ++i)
    if (ps[i] && !aligned16(ps[i])) return 1;
  return native;
}

// =========================================================================
// reductions (stats and bwd_reduce share one skeleton)
//   MODE 0: (x)          -> sum x, sum x^2
//   MODE 1: (dy, x)      -> sum dy', sum dy' * xhat   (MASK: 0 none, 1 y>0, 2 recompute)
// =========================================================================
template <typename T, int V, int MODE, int MASK>
__global__ __launch_bounds__(kThreads) void bn_reduce_nchw(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
    int64_t N, int64_t C, int64_t HW, int seg, int segs, int S,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ partial) {
  __shared__ float sm[2 * (kThreads / 64)];
  const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  float mu = 0.f, is = 0.f, ca = 0.f, cb = 0.f;
  if (MODE == 1) {
    mu = mean[c]; is = invstd[c];
    if (MASK == 2) bn_coef(mean, invstd, gamma, beta, c, ca, cb);
  }
  float a1[V], a2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a1[j] = 0.f; a2[j] = 0.f; }
  const int64_t U = N * segs;
  for (int64_t u = s; u < U; u += S) {
    const int64_t n = u / segs;
    const int sg = (int)(u - n * segs);
    const int64_t off = (n * C + c) * HW + (int64_t)sg * seg;
    const int64_t rem = HW - (int64_t)sg * seg;
    const int len = rem < seg ? (int)rem : seg;
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int i = (k * kThreads + tid) * V;
      if (i < len) {
        Pack<T, V> px;
        px.load(x + off + i);
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < V; ++j) { a1[j] += px.v[j]; a2[j] = fmaf(px.v[j], px.v[j], a2[j]); }
        } else {
          Pack<T, V> pd;
          pd.load(dy + off + i);
          if (MASK == 1) {
            Pack<T, V> py;
            py.load(y + off + i);
#pragma unroll
            for (int j = 0; j < V; ++j) pd.v[j] = py.v[j] > 0.f ? pd.v[j] : 0.f;
          } else if (MASK == 2) {
#pragma unroll
            for (int j = 0; j < V; ++j) pd.v[j] = fmaf(px.v[j], ca, cb) > 0.f ? pd.v[j] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < V; ++j) {
            a1[j] += pd.v[j];
            a2[j] = fmaf(pd.v[j], (px.v[j] - mu) * is, a2[j]);
          }
        }
      }
    }
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int j = 0; j < V; ++j) { s1 += a1[j]; s2 += a2[j]; }
  block_sum2(s1, s2, sm);
  if (tid == 0) {
    partial[((int64_t)s * 2 + 0) * C + c] = s1;
    partial[((int64_t)s * 2 + 1) * C + c] = s2;
  }
}

template <typename T, int V, int MODE, int MASK>
__global__ __launch_bounds__(kThreads) void bn_reduce_nhwc(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
    int64_t M, int64_t C, int GT, int R, int64_t rows_per_block,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][R][GT*V]
  const int tid = threadIdx.x;
  const int gl = tid % GT, r = tid / GT;
  const int64_t G = C / V;
  const int64_t g = (int64_t)blockIdx.y * GT + gl;
  const bool live = (r < R) && (g < G);
  const int64_t c0 = g * V;
  float mu[V], is[V], ca[V], cb[V];
  float a1[V], a2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a1[j] = 0.f; a2[j] = 0.f; mu[j] = 0.f; is[j] = 0.f; ca[j] = 0.f; cb[j] = 0.f; }
  if (MODE == 1 && live) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j];
      if (MASK == 2) bn_coef(mean, invstd, gamma, beta, c0 + j, ca[j], cb[j]);
    }
  }
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t row1 = row0 + rows_per_block;
  if (row1 > M) row1 = M;
  if (live) {
    for (int64_t row = row0 + r; row < row1; row += (int64_t)kUnroll * R) {
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {
        const int64_t rr = row + (int64_t)k * R;
        if (rr < row1) {
          const int64_t off = rr * C + c0;
          Pack<T, V> px;
          px.load(x + off);
          if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < V; ++j) { a1[j] += px.v[j]; a2[j] = fmaf(px.v[j], px.v[j], a2[j]); }
          } else {
            Pack<T, V> pd;
            pd.load(dy + off);
            if (MASK == 1) {
              Pack<T, V> py;
              py.load(y + off);
#pragma unroll
              for (int j = 0; j < V; ++j) pd.v[j] = py.v[j] > 0.f ? pd.v[j] : 0.f;
            } else if (MASK == 2) {
#pragma unroll
              for (int j = 0; j < V; ++j) pd.v[j] = fmaf(px.v[j], ca[j], cb[j]) > 0.f ? pd.v[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
              a1[j] += pd.v[j];
              a2[j] = fmaf(pd.v[j], (px.v[j] - mu[j]) * is[j], a2[j]);
            }
          }
        }
      }
    }
  }
  // cross-row reduction through LDS, fixed order
  const int W = GT * V;
  float* s1 = smem;
  float* s2 = smem + (size_t)R * W;
  if (r < R) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      s1[r * W + gl * V + j] = a1[j];
      s2[r * W + gl * V + j] = a2[j];
    }
  }
  __syncthreads();
  for (int t = tid; t < W; t += kThreads) {
    const int64_t c = (int64_t)blockIdx.y * W + t;
    if (c < C) {
      float t1 = 0.f, t2 = 0.f;
      for (int q = 0; q < R; ++q) { t1 += s1[q * W + t]; t2 += s2[q * W + t]; }
      partial[((int64_t)blockIdx.x * 2 + 0) * C + c] = t1;
      partial[((int64_t)blockIdx.x * 2 + 1) * C + c] = t2;
    }
  }
}

// =========================================================================
// element-wise passes
//   FWD:  y  = act(a*x + b (+res))
//   BWD:  dx = a * (dy' - k0 - xhat*k1) ; dres = dy'
// =========================================================================
template <typename T, int V, bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void bn_fwd_nchw(
    const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
    int64_t C, int64_t HW, int seg, int segs,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta) {
  const int64_t bid = blockIdx.x;
  const int64_t plane = bid / segs;
  const int sg = (int)(bid - plane * segs);
  const int64_t c = plane % C;
  float a, b;
  bn_coef(mean, invstd, gamma, beta, c, a, b);
  const int64_t off = plane * HW + (int64_t)sg * seg;
  const int64_t rem = HW - (int64_t)sg * seg;
  const int len = rem < seg ? (int)rem : seg;
#pragma unroll
  for (int k = 0; k < kUnroll; ++k) {
    const int i = (k * kThreads + threadIdx.x) * V;
    if (i < len) {
      Pack<T, V> px, pr;
      px.load(x + off + i);
      if (RES) pr.load(res + off + i);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float t = fmaf(px.v[j], a, b);
        if (RES) t += pr.v[j];
        if (RELU) t = t > 0.f ? t : 0.f;
        px.v[j] = t;
      }
      px.store(y + off + i);
    }
  }
}

template <typename T, int V, bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void bn_fwd_nhwc(
    const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
    int64_t M, int64_t C, int GT, int R, int64_t rows_per_block,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta) {
  const int tid = threadIdx.x;
  const int gl = tid % GT, r = tid / GT;
  const int64_t g = (int64_t)blockIdx.y * GT + gl;
  if (r >= R || g >= C / V) return;
  const int64_t c0 = g * V;
  float a[V], b[V];
#pragma unroll
  for (int j = 0; j < V; ++j) bn_coef(mean, invstd, gamma, beta, c0 + j, a[j], b[j]);
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t row1 = row0 + rows_per_block;
  if (row1 > M) row1 = M;
  for (int64_t row = row0 + r; row < row1; row += (int64_t)kUnroll * R) {
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int64_t rr = row + (int64_t)k * R;
      if (rr < row1) {
        const int64_t off = rr * C + c0;
        Pack<T, V> px, pr;
        px.load(x + off);
        if (RES) pr.load(res + off);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          float t = fmaf(px.v[j], a[j], b[j]);
          if (RES) t += pr.v[j];
          if (RELU) t = t > 0.f ? t : 0.f;
          px.v[j] = t;
        }
        px.store(y + off);
      }
    }
  }
}

template <typename T, int V, int MASK, bool DRES>
__global__ __launch_bounds__(kThreads) void bn_bwd_nchw(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
    T* __restrict__ dx, T* __restrict__ dres,
    int64_t C, int64_t HW, int seg, int segs,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ kc) {
  const int64_t bid = blockIdx.x;
  const int64_t plane = bid / segs;
  const int sg = (int)(bid - plane * segs);
  const int64_t c = plane % C;
  float a, b;
  bn_coef(mean, invstd, gamma, beta, c, a, b);
  const float mu = mean[c], is = invstd[c];
  const float k0 = kc[c], k1 = kc[C + c];
  const int64_t off = plane * HW + (int64_t)sg * seg;
  const int64_t rem = HW - (int64_t)sg * seg;
  const int len = rem < seg ? (int)rem : seg;
#pragma unroll
  for (int k = 0; k < kUnroll; ++k) {
    const int i = (k * kThreads + threadIdx.x) * V;
    if (i < len) {
      Pack<T, V> pd, px;
      pd.load(dy + off + i);
      px.load(x + off + i);
      if (MASK == 1) {
        Pack<T, V> py;
        py.load(y + off + i);
#pragma unroll
        for (int j = 0; j < V; ++j) pd.v[j] = py.v[j] > 0.f ? pd.v[j] : 0.f;
      } else if (MASK == 2) {
#pragma unroll
        for (int j = 0; j < V; ++j) pd.v[j] = fmaf(px.v[j], a, b) > 0.f ? pd.v[j] : 0.f;
      }
      if (DRES) pd.store(dres + off + i);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float xh = (px.v[j] - mu) * is;
        px.v[j] = a * (pd.v[j] - k0 - xh * k1);
      }
      px.store(dx + off + i);
    }
  }
}

template <typename T, int V, int MASK, bool DRES>
__global__ __launch_bounds__(kThreads) void bn_bwd_nhwc(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
    T* __restrict__ dx, T* __restrict__ dres,
    int64_t M, int64_t C, int GT, int R, int64_t rows_per_block,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ kc) {
  const int tid = threadIdx.x;
  const int gl = tid % GT, r = tid / GT;
  const int64_t g = (int64_t)blockIdx.y * GT + gl;
  if (r >= R || g >= C / V) return;
  const int64_t c0 = g * V;
  float a[V], b[V], mu[V], is[V], k0[V], k1[V];
#pragma unroll
  for (int j = 0; j < V; ++j) {
    bn_coef(mean, invstd, gamma, beta, c0 + j, a[j], b[j]);
    mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j];
    k0[j] = kc[c0 + j]; k1[j] = kc[C + c0 + j];
  }
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t row1 = row0 + rows_per_block;
  if (row1 > M) row1 = M;
  for (int64_t row = row0 + r; row < row1; row += (int64_t)kUnroll * R) {
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int64_t rr = row + (int64_t)k * R;
      if (rr < row1) {
        const int64_t off = rr * C + c0;
        Pack<T, V> pd, px;
        pd.load(dy + off);
        px.load(x + off);
        if (MASK == 1) {
          Pack<T, V> py;
          py.load(y + off);
#pragma unroll
          for (int j = 0; j < V; ++j) pd.v[j] = py.v[j] > 0.f ? pd.v[j] : 0.f;
        } else if (MASK == 2) {
#pragma unroll
          for (int j = 0; j < V; ++j) pd.v[j] = fmaf(px.v[j], a[j], b[j]) > 0.f ? pd.v[j] : 0.f;
        }
        if (DRES) pd.store(dres + off);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float xh = (px.v[j] - mu[j]) * is[j];
          px.v[j] = a[j] * (pd.v[j] - k0[j] - xh * k1[j]);
        }
        px.store(dx + off);
      }
    }
  }
}

// =========================================================================
// per-channel tail kernels (C threads, trivially small)
// =========================================================================
__global__ void bn_collapse_k(const float* __restrict__ partial, int S, int64_t C,
                              float* __restrict__ sums) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double t1 = 0.0, t2 = 0.0;
  for (int s = 0; s < S; ++s) {
    t1 += (double)partial[((int64_t)s * 2 + 0) * C + c];
    t2 += (double)partial[((int64_t)s * 2 + 1) * C + c];
  }
  sums[c] = (float)t1;
  sums[C + c] = (float)t2;
}

// global element count: host double, or (hi, lo) floats on the device with
// count = hi*4096 + lo (both exact in fp32, so an all-reduce SUM stays exact).
__device__ __forceinline__ double bn_count(double count, const float* cd) {
  return cd ? (double)cd[0] * 4096.0 + (double)cd[1] : count;
}

__global__ void bn_finalize_k(const float* __restrict__ partial, int S, int64_t C,
                              double count, const float* __restrict__ count_dev,
                              float eps, float momentum,
                              float* __restrict__ rmean, float* __restrict__ rvar,
                              int64_t* __restrict__ nbt,
                              float* __restrict__ mean, float* __restrict__ invstd) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double t1 = 0.0, t2 = 0.0;
  for (int s = 0; s < S; ++s) {
    t1 += (double)partial[((int64_t)s * 2 + 0) * C + c];
    t2 += (double)partial[((int64_t)s * 2 + 1) * C + c];
  }
  count = bn_count(count, count_dev);
  const double m = t1 / count;
  double sumvar = t2 - t1 * m;          // syncbn.py:91
  if (sumvar < 0.0) sumvar = 0.0;
  const double bias_var = sumvar / count;
  mean[c] = (float)m;
  invstd[c] = (float)(1.0 / sqrt(bias_var + (double)eps));
  if (rmean) rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
  if (rvar) {
    const double unbias = sumvar / (count - 1.0);  // syncbn.py:92 (inf/nan when count==1, as the reference asserts)
    rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unbias);
  }
}

__global__ void bn_bwd_coeffs_k(const float* __restrict__ partial, int S, int64_t C,
                                double count, const float* __restrict__ count_dev,
                                float* __restrict__ dgamma,
                                float* __restrict__ dbeta, float* __restrict__ kc) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double t1 = 0.0, t2 = 0.0;
  for (int s = 0; s < S; ++s) {
    t1 += (double)partial[((int64_t)s * 2 + 0) * C + c];
    t2 += (double)partial[((int64_t)s * 2 + 1) * C + c];
  }
  if (dbeta) dbeta[c] = (float)t1;
  if (dgamma) dgamma[c] = (float)t2;
  if (kc) {
    count = bn_count(count, count_dev);
    kc[c] = (float)(t1 / count);
    kc[C + c] = (float)(t2 / count);
  }
}

// ---- host-side dispatch helpers ------------------------------------------
template <typename T, int V, int MODE>
static int launch_reduce(const T* x, const T* dy, const T* y, int layout, int64_t N,
                         int64_t C, int64_t HW, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, int mask,
                         float* partial, hipStream_t st) {
  if (layout == TSG_NCHW) {
    NchwGeom g = nchw_geom(N, C, HW, V);
    dim3 grid((unsigned)C, (unsigned)g.split);
#define L_(MK) hipLaunchKernelGGL((bn_reduce_nchw<T, V, MODE, MK>), grid, dim3(kThreads), 0, st, \
      x, dy, y, N, C, HW, g.seg, g.segs, g.split, mean, invstd, gamma, beta, partial)
    if (MODE == 0 || mask == 0) L_(0); else if (mask == 1) L_(1); else L_(2);
#undef L_
  } else {
    const int64_t M = N * HW;
    NhwcGeom g = nhwc_geom(M, C, V);
    dim3 grid((unsigned)g.split, (unsigned)g.ytiles);
    const size_t sh = (size_t)2 * g.rows_per_iter * g.gt * V * sizeof(float);
#define L_(MK) hipLaunchKernelGGL((bn_reduce_nhwc<T, V, MODE, MK>), grid, dim3(kThreads), sh, st, \
      x, dy, y, M, C, g.gt, g.rows_per_iter, g.rows_per_block, mean, invstd, gamma, beta, partial)
    if (MODE == 0 || mask == 0) L_(0); else if (mask == 1) L_(1); else L_(2);
#undef L_
  }
  TSG_CHECK_LAUNCH();
  return 0;
}

template <typename T, int V>
static int launch_fwd(const T* x, const T* res, T* y, int layout, int64_t N, int64_t C,
                      int64_t HW, const float* mean, const float* invstd,
                      const float* gamma, const float* beta, int relu, hipStream_t st) {
  const bool R_ = relu != 0, S_ = res != nullptr;
  if (layout == TSG_NCHW) {
    NchwGeom g = nchw_geom(N, C, HW, V);
    const int64_t blocks = N * C * g.segs;
    if (blocks > 0x7fffffffLL) return TSG_E_SHAPE;
#define L_(A, B) hipLaunchKernelGGL((bn_fwd_nchw<T, V, A, B>), dim3((unsigned)blocks), dim3(kThreads), 0, st, \
      x, res, y, C, HW, g.seg, g.segs, mean, invstd, gamma, beta)
    if (R_ && S_) L_(true, true); else if (R_) L_(true, false); else if (S_) L_(false, true); else L_(false, false);
#undef L_
  } else {
    const int64_t M = N * HW;
    NhwcGeom g = nhwc_geom(M, C, V);
    dim3 grid((unsigned)g.split, (unsigned)g.ytiles);
#define L_(A, B) hipLaunchKernelGGL((bn_fwd_nhwc<T, V, A, B>), grid, dim3(kThreads), 0, st, \
      x, res, y, M, C, g.gt, g.rows_per_iter, g.rows_per_block, mean, invstd, gamma, beta)
    if (R_ && S_) L_(true, true); else if (R_) L_(true, false); else if (S_) L_(false, true); else L_(false, false);
#undef L_
  }
  TSG_CHECK_LAUNCH();
  return 0;
}

template <typename T, int V>
static int launch_bwd(const T* dy, const T* x, const T* y, T* dx, T* dres, int layout,
                      int64_t N, int64_t C, int64_t HW, const float* mean,
                      const float* invstd, const float* gamma, const float* beta,
                      const float* kc, int mask, hipStream_t st) {
  const bool D_ = dres != nullptr;
  if (layout == TSG_NCHW) {
    NchwGeom g = nchw_geom(N, C, HW, V);
    const int64_t blocks = N * C * g.segs;
    if (blocks > 0x7fffffffLL) return TSG_E_SHAPE;
#define L_(MK, D) hipLaunchKernelGGL((bn_bwd_nchw<T, V, MK, D>), dim3((unsigned)blocks), dim3(kThreads), 0, st, \
      dy, x, y, dx, dres, C, HW, g.seg, g.segs, mean, invstd, gamma, beta, kc)
    if (mask == 0) { if (D_) L_(0, true); else L_(0, false); }
    else if (mask == 1) { if (D_) L_(1, true); else L_(1, false); }
    else { if (D_) L_(2, true); else L_(2, false); }
#undef L_
  } else {
    const int64_t M = N * HW;
    NhwcGeom g = nhwc_geom(M, C, V);
    dim3 grid((unsigned)g.split, (unsigned)g.ytiles);
#define L_(MK, D) hipLaunchKernelGGL((bn_bwd_nhwc<T, V, MK, D>), grid, dim3(kThreads), 0, st, \
      dy, x, y, dx, dres, M, C, g.gt, g.rows_per_iter, g.rows_per_block, mean, invstd, gamma, beta, kc)
    if (mask == 0) { if (D_) L_(0, true); else L_(0, false); }
    else if (mask == 1) { if (D_) L_(1, true); else L_(1, false); }
    else { if (D_) L_(2, true); else L_(2, false); }
#undef L_
  }
  TSG_CHECK_LAUNCH();
  return 0;
}

static int check_dims(int dtype, int layout, int64_t N, int64_t C, int64_t HW) {
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (layout != TSG_NCHW && layout != TSG_NHWC) return TSG_E_LAYOUT;
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  return 0;
}

}  // namespace tsg

using namespace tsg;

extern "C" {

// The partial count must not depend on pointer alignment, so it is computed for
// the scalar and the vector geometry and the larger one is reported.
int tsg_bn_num_partials(int layout, int64_t N, int64_t C, int64_t HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  int best = 1;
  const int vs[3] = {1, 4, 8};
  for (int i = 0; i < 3; ++i) {
    const int V = vs[i];
    int s;
    if (layout == TSG_NCHW) s = nchw_geom(N, C, HW, V).split;
    else { if (C % V) continue; s = nhwc_geom(N * HW, C, V).split; }
    if (s > best) best = s;
  }
  return best;
}

size_t tsg_bn_partial_ws_bytes(int layout, int64_t N, int64_t C, int64_t HW) {
  const int s = tsg_bn_num_partials(layout, N, C, HW);
  if (s < 0) return 0;
  return (size_t)s * 2 * (size_t)C * sizeof(float);
}

static int partial_rows(int layout, int64_t N, int64_t C, int64_t HW, int V) {
  return layout == TSG_NCHW ? nchw_geom(N, C, HW, V).split : nhwc_geom(N * HW, C, V).split;
}

// returns the number of partial rows actually written (>0) or an error (<0 / hipError)
static int bn_reduce_dispatch(int mode, const void* x, const void* dy, const void* y,
                              int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                              const float* mean, const float* invstd, const float* gamma,
                              const float* beta, int mask, float* partial, void* stream,
                              int* rows) {
  int e = check_dims(dtype, layout, N, C, HW);
  if (e) return e;
  if (!x || !partial) return TSG_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  const int V = pick_vec(dtype, layout, C, HW, x, dy, mask == 1 ? y : nullptr, nullptr, nullptr);
  *rows = partial_rows(layout, N, C, HW, V);
#define GO(T, VV)                                                                          \
  (mode == 0 ? launch_reduce<T, VV, 0>((const T*)x, nullptr, nullptr, layout, N, C, HW,    \
                                       mean, invstd, gamma, beta, 0, partial, st)          \
             : launch_reduce<T, VV, 1>((const T*)x, (const T*)dy, (const T*)y, layout, N,  \
                                       C, HW, mean, invstd, gamma, beta, mask, partial, st))
  if (dtype == TSG_F32) return V == 4 ? GO(float, 4) : GO(float, 1);
  return V == 8 ? GO(bf16_t, 8) : GO(bf16_t, 1);
#undef GO
}

int tsg_bn_stats(const void* x, int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                 float* partial, int* rows, void* stream) {
  int r = 0;
  int e = bn_reduce_dispatch(0, x, nullptr, nullptr, dtype, layout, N, C, HW, nullptr,
                             nullptr, nullptr, nullptr, 0, partial, stream, &r);
  if (rows) *rows = r;
  return e;
}

int tsg_bn_collapse(const float* partial, int S, int64_t C, float* sums, void* stream) {
  if (!partial || !sums) return TSG_E_NULL;
  if (S <= 0 || C <= 0) return TSG_E_SHAPE;
  hipLaunchKernelGGL(bn_collapse_k, dim3(ceil_div_i(C, 128)), dim3(128), 0, (hipStream_t)stream,
                     partial, S, C, sums);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_finalize(const float* partial, int S, int64_t C, double count,
                    const float* count_dev, float eps,
                    float momentum, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, float* mean, float* invstd, void* stream) {
  if (!partial || !mean || !invstd) return TSG_E_NULL;
  if (S <= 0 || C <= 0 || (!count_dev && !(count > 0.0))) return TSG_E_SHAPE;
  hipLaunchKernelGGL(bn_finalize_k, dim3(ceil_div_i(C, 128)), dim3(128), 0, (hipStream_t)stream,
                     partial, S, C, count, count_dev, eps, momentum, running_mean, running_var,
                     num_batches_tracked, mean, invstd);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_apply_fwd(const void* x, const void* residual, void* y, int dtype, int layout,
                     int64_t N, int64_t C, int64_t HW, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, int relu, void* stream) {
  int e = check_dims(dtype, layout, N, C, HW);
  if (e) return e;
  if (!x || !y || !mean || !invstd) return TSG_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  const int V = pick_vec(dtype, layout, C, HW, x, residual, y, nullptr, nullptr);
#define GO(T, VV) launch_fwd<T, VV>((const T*)x, (const T*)residual, (T*)y, layout, N, C, HW, \
                                    mean, invstd, gamma, beta, relu, st)
  if (dtype == TSG_F32) return V == 4 ? GO(float, 4) : GO(float, 1);
  return V == 8 ? GO(bf16_t, 8) : GO(bf16_t, 1);
#undef GO
}

int tsg_bn_bwd_reduce(const void* dy, const void* x, const void* y, int dtype, int layout,
                      int64_t N, int64_t C, int64_t HW, const float* mean,
                      const float* invstd, const float* gamma, const float* beta, int relu,
                      float* partial, int* rows, void* stream) {
  if (!dy || !mean || !invstd) return TSG_E_NULL;
  const int mask = relu ? (y ? 1 : 2) : 0;
  int r = 0;
  int e = bn_reduce_dispatch(1, x, dy, y, dtype, layout, N, C, HW, mean, invstd, gamma, beta,
                             mask, partial, stream, &r);
  if (rows) *rows = r;
  return e;
}

int tsg_bn_bwd_coeffs(const float* partial, int S, int64_t C, double count,
                      const float* count_dev, float* dgamma,
                      float* dbeta, float* k, void* stream) {
  if (!partial) return TSG_E_NULL;
  if (S <= 0 || C <= 0 || (k && !count_dev && !(count > 0.0))) return TSG_E_SHAPE;
  hipLaunchKernelGGL(bn_bwd_coeffs_k, dim3(ceil_div_i(C, 128)), dim3(128), 0, (hipStream_t)stream,
                     partial, S, C, count, count_dev, dgamma, dbeta, k);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_bwd_apply(const void* dy, const void* x, const void* y, void* dx, void* dres,
                     int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                     const float* mean, const float* invstd, const float* gamma,
                     const float* beta, const float* k, int relu, void* stream) {
  int e = check_dims(dtype, layout, N, C, HW);
  if (e) return e;
  if (!dy || !x || !dx || !mean || !invstd || !k) return TSG_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  const int mask = relu ? (y ? 1 : 2) : 0;
  const int V = pick_vec(dtype, layout, C, HW, x, dy, mask == 1 ? y : nullptr, dx, dres);
#define GO(T, VV) launch_bwd<T, VV>((const T*)dy, (const T*)x, (const T*)y, (T*)dx, (T*)dres, \
                                    layout, N, C, HW, mean, invstd, gamma, beta, k, mask, st)
  if (dtype == TSG_F32) return V == 4 ? GO(float, 4) : GO(float, 1);
  return V == 8 ? GO(bf16_t, 8) : GO(bf16_t, 1);
#undef GO
}

}  // extern "C"
