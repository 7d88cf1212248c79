// SyncBatchNorm kernels for gfx950 (wave64, 256 CUs / 8 XCDs).
//
// Arithmetic follows the in-tree statement of the reference's SyncBN
// (furnace/legacy/sync_bn/syncbn.py:86-98, src/gpu/syncbn_kernel.cu:12-23,
// 73-174) but none of its launch shape: the reference runs ONE block per
// channel (<= C blocks, <= 512 threads, strided NCHW reads).  Here every pass
// is a streaming kernel with 16-byte loads, ~1-2k workgroups, fp32 per-lane
// accumulators, a fixed-order (deterministic) two-stage reduction whose second
// stage runs in fp64, and BN+ReLU(+residual) fused so an activation is read
// once and written once per pass.  Per-channel constants are folded once into
// small "packs" so the streaming kernels load 2-5 floats per channel:
//   fwd pack  fp[3][C] = { a = gamma*invstd, b = beta - mean*a, mean }
//   bwd pack  bp[5][C] = { a, b, mean, Bc = -a*k1*invstd, C2 = -a*k0 }
//   y  = relu?(a*x + b (+res))                     dx = a*dy' + Bc*(x-mean) + C2
//
// All passes are HBM-bound: algorithmic bytes per element (s = element size)
//   stats        1 read            = s
//   apply_fwd    1 read + 1 write  = 2s   (+s with residual)
//   bwd_reduce   2 reads           = 2s   (+s when the ReLU mask comes from y)
//   bwd_apply    2 reads + 1 write = 3s   (+s mask-from-y, +s dres)
#include "tsg_common.h"
#include <stdlib.h>

namespace tsg {

constexpr int kThreads = 256;
constexpr int kUnroll = 4;
constexpr int kTargetBlocks = 2048;       // NCHW: 8 per CU
constexpr int kTargetBlocksNhwc = 1024;   // NHWC: 4 per CU, more rows per block
constexpr int kMaxSplit = 256;

// ---- element packs: V elements per lane-access, unpacked to fp32 ----------
template <typename T, int V> struct Pack;
template <> struct Pack<float, 4> : Vec<float> {};
template <> struct Pack<bf16_t, 8> : Vec<bf16_t> {};
template <typename T> struct Pack<T, 1> {
  float v[1];
  typedef T Raw;
  static __device__ __forceinline__ Raw ldraw(const T* p) { return *p; }
  static __device__ __forceinline__ Raw ldraw_dead(const T* p) { return *p; }
  __device__ __forceinline__ void unpack(const Raw& t) { v[0] = ld1<T>(&t); }
  __device__ __forceinline__ void load(const T* p) { v[0] = ld1<T>(p); }
  __device__ __forceinline__ void store(T* p) const { st1<T>(p, v[0]); }
};

// V consecutive per-channel floats starting at c0 (c0 % V == 0 => 16-B aligned)
template <int V>
__device__ __forceinline__ void ldc(const float* __restrict__ p, int64_t c0, float (&o)[V]) {
  if (V == 1) {
    o[0] = p[c0];
  } else {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(p + c0 + 4 * q);
      o[4 * q + 0] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
    }
  }
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
  // at least two unrolled iterations per thread so the per-channel constants amortise
  int64_t min_rows = (int64_t)g.rows_per_iter * kUnroll * 2;
  int64_t s = ceil_div_i(kTargetBlocksNhwc, g.ytiles);
  int64_t max_s = (M + min_rows - 1) / min_rows;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  g.rows_per_block = (M + s - 1) / s;
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

// Accumulator of a reduction pass.  The STATISTICS of an fp32 tensor (the parity path: north_star's fp32 logits within 1e-4
// of the reference's CPU path) are accumulated in fp64 and handed on as hi / lo fp32 row pairs: with sum x / sum x^2 in
// fp32 — the reference's own formulation, syncbn_kernel.cu:12-23,73-89 — var = E[x^2] - mean^2 cancels, and the batch-2
// BatchNorm of BiSeNet's global-context branch ([2, 128, 1, 1]: var = ((a - b) / 2)^2 against a^2) amplifies the 6e-8 of an
// fp32 partial into 3-6e-4 of the logits: exactly the distance EVERY GPU path kept from the CPU (round 4,
// tools/diag_fp64_truth.py; reproduced on the CPU by rounding the two sums to fp32).  bf16 tensors (the bench path) keep
// fp32 accumulators: their values carry 2^-9 of rounding already.
// Round 5: the BACKWARD sums of an fp32 tensor (sum dy', sum dy' (x - mean)) take the same accumulators: dx = a dy' + Bc (x -
// mean) + C2 removes dy's components along 1 and xhat, a difference of sums, and the parity mode should not depend on how
// benign a layer's gradient happens to be.  (What tests/test_headline_gpu.py::test_fp32_gradients_per_parameter_against_
// float64 measures at 4 x 256^2 — 6.9e-6 global, 7.8e-3 at one parameter — is NOT this: it is ONE ReLU whose argument is
// 7e-8 in float64 and 0 in fp32, tools/r5/debug_ffm_grad.py.)
template <typename T, int MODE> struct RedAcc { typedef float type; };
template <> struct RedAcc<float, 0> { typedef double type; };
template <> struct RedAcc<float, 1> { typedef double type; };

// Block-wide sum of two doubles, fixed order; result valid in thread 0.  `sm` needs 2 * (blockDim.x / 64) doubles.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sm) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { sm[2 * w] = a; sm[2 * w + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0.0, tb = 0.0;
    for (int i = 0; i < nw; ++i) { ta += sm[2 * i]; tb += sm[2 * i + 1]; }
    a = ta; b = tb;
  }
}

// one partial entry: fp32 accumulators store the value; fp64 accumulators store hi in row s and lo in row S + s (the
// collapse / finalize kernels sum ALL rows in fp64, so the pair adds back to ~48 bits of the block's sum)
__device__ __forceinline__ void put_partial(float* partial, int64_t C, int S, int s, int which, int64_t c, float v) {
  partial[((int64_t)s * 2 + which) * C + c] = v;
}
__device__ __forceinline__ void put_partial(float* partial, int64_t C, int S, int s, int which, int64_t c, double v) {
  const float hi = (float)v;
  partial[((int64_t)s * 2 + which) * C + c] = hi;
  partial[((int64_t)(S + s) * 2 + which) * C + c] = (float)(v - (double)hi);
}

// =========================================================================
// reductions (stats and bwd_reduce share one skeleton)
//   MODE 0: (x)          -> sum x, sum x^2
//   MODE 1: (dy, x)      -> sum dy', sum dy' * (x - mean)   (MASK: 0 none, 1 y>0, 2 recompute, 3 bit mask: NHWC only —
//                           `y` then points at one byte per V channels of a pixel, bit j = (y[.. + j] > 0), written by
//                           bn_fwd_nhwc<., ., true, true, true>: the block tail BN -> (+identity) -> ReLU (resnet.py:44-51)
//                           reads 1/16 of a tensor for its ReLU mask instead of the whole output)
// fp[0]=a, fp[1]=b, fp[2]=mean
// =========================================================================
template <typename T, int V, int MODE, int MASK>
__global__ __launch_bounds__(kThreads) void bn_reduce_nchw(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
    int64_t N, int64_t C, int64_t HW, int seg, int segs, int S,
    const float* __restrict__ fp, float* __restrict__ partial) {
  typedef typename RedAcc<T, MODE>::type AT;
  __shared__ AT sm[2 * (kThreads / 64)];
  const int c = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
  float mu = 0.f, ca = 0.f, cb = 0.f;
  if (MODE == 1) {
    mu = fp[2 * C + c];
    if (MASK == 2) { ca = fp[c]; cb = fp[C + c]; }
  }
  AT a1[V], a2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a1[j] = 0; a2[j] = 0; }
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
          for (int j = 0; j < V; ++j) { const AT xv = px.v[j]; a1[j] += xv; a2[j] = fma(xv, xv, a2[j]); }
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
            a1[j] += (AT)pd.v[j];
            a2[j] = fma((AT)pd.v[j], (AT)(px.v[j] - mu), a2[j]);
          }
        }
      }
    }
  }
  AT s1 = 0, s2 = 0;
#pragma unroll
  for (int j = 0; j < V; ++j) { s1 += a1[j]; s2 += a2[j]; }
  block_sum2(s1, s2, sm);
  if (tid == 0) {
    put_partial(partial, C, S, s, 0, c, s1);
    put_partial(partial, C, S, s, 1, c, s2);
  }
}

template <typename T, int V, int MODE, int MASK>
__global__ __launch_bounds__(kThreads) void bn_reduce_nhwc(
    const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
    int64_t M, int64_t C, int GT, int R, int64_t rows_per_block,
    const float* __restrict__ fp, float* __restrict__ partial) {
  typedef typename RedAcc<T, MODE>::type AT;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][R][GT*V] of AT
  const int tid = threadIdx.x;
  const int gl = tid % GT, r = tid / GT;
  const int64_t G = C / V;
  const int64_t g = (int64_t)blockIdx.y * GT + gl;
  const bool live = (r < R) && (g < G);
  const int64_t c0 = g * V;
  float mu[V], ca[V], cb[V];
  AT a1[V], a2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a1[j] = 0; a2[j] = 0; mu[j] = 0.f; ca[j] = 0.f; cb[j] = 0.f; }
  if (MODE == 1 && live) {
    ldc<V>(fp + 2 * C, c0, mu);
    if (MASK == 2) { ldc<V>(fp, c0, ca); ldc<V>(fp + C, c0, cb); }
  }
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t row1 = row0 + rows_per_block;
  if (row1 > M) row1 = M;
  if (live) {
    // Full groups of kUnroll rows issue ALL their 16-byte loads before any arithmetic (kUnroll, x 2 / x 3 in MODE 1, in
    // flight per wave); round 4's `if (rr < row1)` around each unrolled row put every load in a divergent branch of its
    // own, and a load in a branch is waited for at the end of that branch: one load in flight per wave, s_waitcnt vmcnt(0)
    // after each (profiles/r05_bn_isa.txt).  Rows are accumulated in the same order as before: bit-identical sums.
    typedef typename Pack<T, V>::Raw Raw;
    auto accumulate = [&](const Raw& rx, const Raw& rd, const Raw& ry, unsigned int mb) {
      Pack<T, V> px;
      px.unpack(rx);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < V; ++j) { const AT xv = px.v[j]; a1[j] += xv; a2[j] = fma(xv, xv, a2[j]); }
      } else {
        Pack<T, V> pd;
        pd.unpack(rd);
        if (MASK == 1) {
          Pack<T, V> py;
          py.unpack(ry);
#pragma unroll
          for (int j = 0; j < V; ++j) pd.v[j] = py.v[j] > 0.f ? pd.v[j] : 0.f;
        } else if (MASK == 2) {
#pragma unroll
          for (int j = 0; j < V; ++j) pd.v[j] = fmaf(px.v[j], ca[j], cb[j]) > 0.f ? pd.v[j] : 0.f;
        } else if (MASK == 3) {
#pragma unroll
          for (int j = 0; j < V; ++j) pd.v[j] = ((mb >> j) & 1u) ? pd.v[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
          a1[j] += (AT)pd.v[j];
          a2[j] = fma((AT)pd.v[j], (AT)(px.v[j] - mu[j]), a2[j]);
        }
      }
    };
    const unsigned char* bits = reinterpret_cast<const unsigned char*>(y);
    int64_t row = row0 + r;
    for (; row + (int64_t)(kUnroll - 1) * R < row1; row += (int64_t)kUnroll * R) {
      Raw rx[kUnroll], rd[kUnroll], ry[kUnroll];
      unsigned int rb[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) {                 // every load of the group first ...
        const int64_t off = (row + (int64_t)k * R) * C + c0;
        rx[k] = Pack<T, V>::ldraw(x + off);
        if (MODE == 1) rd[k] = Pack<T, V>::ldraw(dy + off);
        if (MODE == 1 && MASK == 1) ry[k] = Pack<T, V>::ldraw(y + off);
        rb[k] = (MODE == 1 && MASK == 3) ? bits[off / V] : 0u;
      }
#pragma unroll
      for (int k = 0; k < kUnroll; ++k) accumulate(rx[k], rd[k], ry[k], rb[k]);   // ... then the arithmetic, in row order
    }
    for (; row < row1; row += R) {
      const int64_t off = row * C + c0;
      Raw rx = Pack<T, V>::ldraw(x + off), rd = rx, ry = rx;
      if (MODE == 1) rd = Pack<T, V>::ldraw(dy + off);
      if (MODE == 1 && MASK == 1) ry = Pack<T, V>::ldraw(y + off);
      accumulate(rx, rd, ry, (MODE == 1 && MASK == 3) ? bits[off / V] : 0u);
    }
  }
  // cross-row reduction through LDS, fixed order
  const int W = GT * V;
  AT* s1 = reinterpret_cast<AT*>(smem);
  AT* s2 = s1 + (size_t)R * W;
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
      AT t1 = 0, t2 = 0;
      for (int q = 0; q < R; ++q) { t1 += s1[q * W + t]; t2 += s2[q * W + t]; }
      put_partial(partial, C, (int)gridDim.x, (int)blockIdx.x, 0, c, t1);
      put_partial(partial, C, (int)gridDim.x, (int)blockIdx.x, 1, c, t2);
    }
  }
}

// =========================================================================
// element-wise passes
// =========================================================================
// TSG_BN_REVERSE=1|0 (default set by measurement): the apply kernels traverse the tensor end-first
static int bn_reverse() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("TSG_BN_REVERSE"); v = e ? (e[0] != '0') : 0; }
  return v;
}

template <typename T, int V, bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void bn_fwd_nchw(
    const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
    int64_t C, int64_t HW, int seg, int segs, const float* __restrict__ fp) {
  const int64_t bid = blockIdx.x;
  const int64_t plane = bid / segs;
  const int sg = (int)(bid - plane * segs);
  const int64_t c = plane % C;
  const float a = fp[c], b = fp[C + c];
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

template <typename T, int V, bool RELU, bool RES, bool BITS = false>
__global__ __launch_bounds__(kThreads) void bn_fwd_nhwc(
    const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
    int64_t M, int64_t C, int GT, int R, int64_t rows_per_block, const float* __restrict__ fp, int rev,
    unsigned char* __restrict__ bits = nullptr) {
  const int tid = threadIdx.x;
  const int gl = tid % GT, r = tid / GT;
  const int64_t g = (int64_t)blockIdx.y * GT + gl;
  if (r >= R || g >= C / V) return;
  const int64_t c0 = g * V;
  float a[V], b[V];
  ldc<V>(fp, c0, a);
  ldc<V>(fp + C, c0, b);
  // rev: walk the tensor END first.  The statistics pass that ran just before read x front to back, so the tail of x is
  // what the 256 MB Infinity Cache still holds; blocks are dispatched in index order, so the first ones take the tail.
  const int64_t row0 = (int64_t)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rows_per_block;
  int64_t row1 = row0 + rows_per_block;
  if (row1 > M) row1 = M;
  typedef typename Pack<T, V>::Raw Raw;                   // see bn_reduce_nhwc: a group's loads first, then arithmetic + stores
  auto finish = [&](const Raw& rx, const Raw& rr_, int64_t off) {
    Pack<T, V> px, pr;
    px.unpack(rx);
    if (RES) pr.unpack(rr_);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t = fmaf(px.v[j], a[j], b[j]);
      if (RES) t += pr.v[j];
      if (RELU) t = t > 0.f ? t : 0.f;
      px.v[j] = t;
    }
    px.store(y + off);
    if (BITS) {                                         // from the ROUNDED value that was stored: what `y > 0` reads back
      unsigned int mb = 0u;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float q = sizeof(T) == 2 ? __uint_as_float((uint32_t)f32_to_bf16(px.v[j]) << 16) : px.v[j];
        mb |= (q > 0.f ? 1u : 0u) << j;
      }
      bits[off / V] = (unsigned char)mb;
    }
  };
  int64_t row = row0 + r;
  for (; row + (int64_t)(kUnroll - 1) * R < row1; row += (int64_t)kUnroll * R) {
    Raw rx[kUnroll], rs[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int64_t off = (row + (int64_t)k * R) * C + c0;
      rx[k] = Pack<T, V>::ldraw_dead(x + off);          // the convolution's output: next read in the backward pass
      if (RES) rs[k] = Pack<T, V>::ldraw(res + off);    // (non-temporal here and for the fold's partials: step + 0.07 ms)
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) finish(rx[k], rs[k], (row + (int64_t)k * R) * C + c0);
  }
  for (; row < row1; row += R) {
    const int64_t off = row * C + c0;
    Raw rx = Pack<T, V>::ldraw_dead(x + off), rs = rx;
    if (RES) rs = Pack<T, V>::ldraw(res + off);
    finish(rx, rs, off);
  }
}

// bp[0]=a, bp[1]=b, bp[2]=mean, bp[3]=Bc, bp[4]=C2 ; dx = a*dy' + Bc*(x-mean) + C2
template <typename T, int V, int MASK, bool DRES>
__global__ __launch_bounds__(kThreads) void bn_bwd_nchw(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
    T* __restrict__ dx, T* __restrict__ dres,
    int64_t C, int64_t HW, int seg, int segs, const float* __restrict__ bp) {
  const int64_t bid = blockIdx.x;
  const int64_t plane = bid / segs;
  const int sg = (int)(bid - plane * segs);
  const int64_t c = plane % C;
  const float a = bp[c], b = bp[C + c], mu = bp[2 * C + c], bc = bp[3 * C + c], c2 = bp[4 * C + c];
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
      for (int j = 0; j < V; ++j) px.v[j] = fmaf(a, pd.v[j], fmaf(bc, px.v[j] - mu, c2));
      px.store(dx + off + i);
    }
  }
}

template <typename T, int V, int MASK, bool DRES>
__global__ __launch_bounds__(kThreads) void bn_bwd_nhwc(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
    T* __restrict__ dx, T* __restrict__ dres,
    int64_t M, int64_t C, int GT, int R, int64_t rows_per_block, const float* __restrict__ bp, int rev) {
  const int tid = threadIdx.x;
  const int gl = tid % GT, r = tid / GT;
  const int64_t g = (int64_t)blockIdx.y * GT + gl;
  if (r >= R || g >= C / V) return;
  const int64_t c0 = g * V;
  float a[V], b[V], mu[V], bc[V], c2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) b[j] = 0.f;
  ldc<V>(bp, c0, a);
  if (MASK == 2) ldc<V>(bp + C, c0, b);
  ldc<V>(bp + 2 * C, c0, mu);
  ldc<V>(bp + 3 * C, c0, bc);
  ldc<V>(bp + 4 * C, c0, c2);
  const int64_t row0 = (int64_t)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * rows_per_block;
  int64_t row1 = row0 + rows_per_block;
  if (row1 > M) row1 = M;
  typedef typename Pack<T, V>::Raw Raw;                   // see bn_reduce_nhwc: a group's loads first, then arithmetic + stores
  auto finish = [&](const Raw& rd, const Raw& rx, const Raw& ry, unsigned int mb, int64_t off) {
    Pack<T, V> pd, px;
    pd.unpack(rd);
    px.unpack(rx);
    if (MASK == 1) {
      Pack<T, V> py;
      py.unpack(ry);
#pragma unroll
      for (int j = 0; j < V; ++j) pd.v[j] = py.v[j] > 0.f ? pd.v[j] : 0.f;
    } else if (MASK == 2) {
#pragma unroll
      for (int j = 0; j < V; ++j) pd.v[j] = fmaf(px.v[j], a[j], b[j]) > 0.f ? pd.v[j] : 0.f;
    } else if (MASK == 3) {
#pragma unroll
      for (int j = 0; j < V; ++j) pd.v[j] = ((mb >> j) & 1u) ? pd.v[j] : 0.f;
    }
    if (DRES) pd.store(dres + off);
#pragma unroll
    for (int j = 0; j < V; ++j) px.v[j] = fmaf(a[j], pd.v[j], fmaf(bc[j], px.v[j] - mu[j], c2[j]));
    px.store(dx + off);
  };
  const unsigned char* bits = reinterpret_cast<const unsigned char*>(y);
  int64_t row = row0 + r;
  for (; row + (int64_t)(kUnroll - 1) * R < row1; row += (int64_t)kUnroll * R) {
    Raw rd[kUnroll], rx[kUnroll], ry[kUnroll];
    unsigned int rb[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
      const int64_t off = (row + (int64_t)k * R) * C + c0;
      rd[k] = Pack<T, V>::ldraw_dead(dy + off);         // gradient and BatchNorm input: this pass is their last reader
      rx[k] = Pack<T, V>::ldraw_dead(x + off);
      if (MASK == 1) ry[k] = Pack<T, V>::ldraw(y + off);
      rb[k] = MASK == 3 ? bits[off / V] : 0u;
    }
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) finish(rd[k], rx[k], ry[k], rb[k], (row + (int64_t)k * R) * C + c0);
  }
  for (; row < row1; row += R) {
    const int64_t off = row * C + c0;
    Raw rd = Pack<T, V>::ldraw_dead(dy + off), rx = Pack<T, V>::ldraw_dead(x + off), ry = rx;
    if (MASK == 1) ry = Pack<T, V>::ldraw(y + off);
    finish(rd, rx, ry, MASK == 3 ? bits[off / V] : 0u, off);
  }
}

// =========================================================================
// mixed-layout passes for conv stems: x (the conv output) is NCHW because MIOpen's
// NHWC kernels lose badly at C_in = 3, while everything downstream is NHWC.
// These kernels read/write x-side tensors as [N, C, HW] and y-side tensors
// (y, dy) as [N*HW, C], transposing a [C x TP] tile through LDS, so no separate
// layout-conversion copy (a ~0.6 TB/s strided copy in eager PyTorch) is needed.
// Block = one image n, TP = 64 consecutive pixels, all C channels (C % 8 == 0, C <= 128).
// =========================================================================
constexpr int kTP = 64;

template <typename T, bool RELU>
__global__ __launch_bounds__(kThreads) void bn_fwd_mixed(
    const T* __restrict__ x, T* __restrict__ y, int64_t C, int64_t HW, int tiles_per_img,
    const float* __restrict__ fp) {
  constexpr int V = Pack<T, (sizeof(T) == 2 ? 8 : 4)>::N;
  extern __shared__ __attribute__((aligned(16))) float tile[];     // [kTP][C + 1]
  const int ldt = (int)C + 1;
  const int64_t n = blockIdx.x / tiles_per_img;
  const int64_t p0 = (int64_t)(blockIdx.x % tiles_per_img) * kTP;
  const int npx = (HW - p0) < kTP ? (int)(HW - p0) : kTP;
  const int tid = threadIdx.x;
  // phase 1: NCHW rows -> LDS[p][c]
  const int vpr = kTP / V;                       // vectors per channel row
  for (int it = tid; it < (int)C * vpr; it += kThreads) {
    const int c = it / vpr, v = it - c * vpr;
    const int p = v * V;
    if (p < npx) {
      Pack<T, V> px;
      px.load(x + (n * C + c) * HW + p0 + p);   // HW % V == 0 (host check) => full vector in range
      const float a = fp[c], b = fp[C + c];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float t = fmaf(px.v[j], a, b);
        if (RELU) t = t > 0.f ? t : 0.f;
        tile[(p + j) * ldt + c] = t;
      }
    }
  }
  __syncthreads();
  // phase 2: LDS[p][c-vector] -> NHWC
  const int gpr = (int)C / V;                    // channel groups per pixel
  for (int it = tid; it < npx * gpr; it += kThreads) {
    const int p = it / gpr, g = it - p * gpr;
    Pack<T, V> o;
#pragma unroll
    for (int j = 0; j < V; ++j) o.v[j] = tile[p * ldt + g * V + j];
    o.store(y + ((n * HW + p0 + p) * C) + g * V);
  }
}

// Backward reduction: dy is NHWC, x is NCHW; the ReLU mask is recomputed from x
// (stems have no fused residual).  A block walks tiles blockIdx.x, +gridDim.x, ...
// keeping per-(channel, pixel-vector) sums in registers, then folds them once.
// ITEMS = ceil(C * (kTP / V) / kThreads): (channel, pixel-vector) pairs per thread
template <typename T, bool RELU, int ITEMS>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_mixed(
    const T* __restrict__ dy, const T* __restrict__ x, int64_t C, int64_t HW, int tiles_per_img,
    int64_t total_tiles, const float* __restrict__ pk, float* __restrict__ partial) {
  constexpr int V = Pack<T, (sizeof(T) == 2 ? 8 : 4)>::N;
  extern __shared__ __attribute__((aligned(16))) float tile[];     // [kTP][C + 1] + [2][C][vpr]
  const int ldt = (int)C + 1;
  const int tid = threadIdx.x;
  const int gpr = (int)C / V, vpr = kTP / V;
  const int items = (int)C * vpr;
  float s1[ITEMS], s2[ITEMS];
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) { s1[k] = 0.f; s2[k] = 0.f; }
  for (int64_t t = blockIdx.x; t < total_tiles; t += gridDim.x) {
    const int64_t n = t / tiles_per_img;
    const int64_t p0 = (t % tiles_per_img) * kTP;
    const int npx = (HW - p0) < kTP ? (int)(HW - p0) : kTP;
    __syncthreads();                                            // previous tile fully consumed
    for (int it = tid; it < npx * gpr; it += kThreads) {
      const int p = it / gpr, g = it - p * gpr;
      Pack<T, V> pd;
      pd.load(dy + ((n * HW + p0 + p) * C) + g * V);
#pragma unroll
      for (int j = 0; j < V; ++j) tile[p * ldt + g * V + j] = pd.v[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const int it = tid + k * kThreads;
      if (it < items) {
        const int c = it / vpr, v = it - c * vpr;
        const int p = v * V;
        if (p < npx) {
          Pack<T, V> px;
          px.load(x + (n * C + c) * HW + p0 + p);
          const float a = pk[c], b = pk[C + c], mu = pk[2 * C + c];
#pragma unroll
          for (int j = 0; j < V; ++j) {
            float d = tile[(p + j) * ldt + c];
            if (RELU) d = fmaf(px.v[j], a, b) > 0.f ? d : 0.f;
            s1[k] += d;
            s2[k] = fmaf(d, px.v[j] - mu, s2[k]);
          }
        }
      }
    }
  }
  __syncthreads();
  float* red = tile;                                            // reuse: [2][C][vpr]
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const int it = tid + k * kThreads;
    if (it < items) { red[it] = s1[k]; red[items + it] = s2[k]; }
  }
  __syncthreads();
  for (int c = tid; c < (int)C; c += kThreads) {
    float t1 = 0.f, t2 = 0.f;
    for (int v = 0; v < vpr; ++v) { t1 += red[c * vpr + v]; t2 += red[items + c * vpr + v]; }
    partial[((int64_t)blockIdx.x * 2 + 0) * C + c] = t1;
    partial[((int64_t)blockIdx.x * 2 + 1) * C + c] = t2;
  }
}

// dx (NCHW) = a*dy' + Bc*(x-mean) + C2 with dy NHWC
template <typename T, bool RELU>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_mixed(
    const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx, int64_t C, int64_t HW,
    int tiles_per_img, const float* __restrict__ pk) {
  constexpr int V = Pack<T, (sizeof(T) == 2 ? 8 : 4)>::N;
  extern __shared__ __attribute__((aligned(16))) float tile[];     // [kTP][C + 1]
  const int ldt = (int)C + 1;
  const int64_t n = blockIdx.x / tiles_per_img;
  const int64_t p0 = (int64_t)(blockIdx.x % tiles_per_img) * kTP;
  const int npx = (HW - p0) < kTP ? (int)(HW - p0) : kTP;
  const int tid = threadIdx.x;
  const int gpr = (int)C / V, vpr = kTP / V;
  for (int it = tid; it < npx * gpr; it += kThreads) {
    const int p = it / gpr, g = it - p * gpr;
    Pack<T, V> pd;
    pd.load(dy + ((n * HW + p0 + p) * C) + g * V);
#pragma unroll
    for (int j = 0; j < V; ++j) tile[p * ldt + g * V + j] = pd.v[j];
  }
  __syncthreads();
  for (int it = tid; it < (int)C * vpr; it += kThreads) {
    const int c = it / vpr, v = it - c * vpr;
    const int p = v * V;
    if (p < npx) {
      Pack<T, V> px;
      px.load(x + (n * C + c) * HW + p0 + p);
      const float a = pk[c], b = pk[C + c], mu = pk[2 * C + c], bc = pk[3 * C + c], c2 = pk[4 * C + c];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float d = tile[(p + j) * ldt + c];
        if (RELU) d = fmaf(px.v[j], a, b) > 0.f ? d : 0.f;
        px.v[j] = fmaf(a, d, fmaf(bc, px.v[j] - mu, c2));
      }
      px.store(dx + (n * C + c) * HW + p0 + p);
    }
  }
}

// =========================================================================
// per-channel tail kernels: block = 8 channels x 64 slice-lanes, fp64, fixed order
// =========================================================================
// Latency-bound (S x 2C floats, S <= ~1024): few channels per block so that even C = 64 spreads over 8 CUs, many slice
// lanes so that a thread has at most two rounds of 8 predicated loads.  512 threads: bn_finalize_k's fp64 tail needs > 128 registers.
constexpr int kTc = 8, kTs = 64;

__device__ __forceinline__ void tail_sums(const float* __restrict__ partial, int S, int64_t C,
                                          double& t1, double& t2, bool& owner, int64_t& c) {
  __shared__ double sh[2][kTs][kTc];
  const int tc = threadIdx.x % kTc, ts = threadIdx.x / kTc;
  c = (int64_t)blockIdx.x * kTc + tc;
  double p1 = 0.0, p2 = 0.0;
  if (c < C) {
    // these kernels are pure load latency: 8 row pairs in flight per thread, predicated (not a remainder loop: the
    // usual trip count is 2 .. 32), summed in row order
    for (int s0 = ts; s0 < S; s0 += kTs * 8) {
      float v1[8], v2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = s0 + u * kTs;
        const bool ok = s < S;
        const int64_t row = ok ? s : ts;
        v1[u] = partial[(row * 2 + 0) * C + c];
        v2[u] = partial[(row * 2 + 1) * C + c];
        if (!ok) { v1[u] = 0.f; v2[u] = 0.f; }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { p1 += (double)v1[u]; p2 += (double)v2[u]; }
    }
  }
  sh[0][ts][tc] = p1;
  sh[1][ts][tc] = p2;
  __syncthreads();
  // fixed two-level order: 8 slice-lanes fold 8 consecutive lanes each, the owner folds those 8 (a 64-long serial chain
  // of dependent LDS reads + fp64 adds was ~1/3 of these kernels' 6 us)
  if (ts < 8) {                                          // slots ts * 8 .. + 7 of column tc are read by this thread only
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { a1 += sh[0][ts * 8 + q][tc]; a2 += sh[1][ts * 8 + q][tc]; }
    sh[0][ts * 8][tc] = a1; sh[1][ts * 8][tc] = a2;
  }
  __syncthreads();
  owner = (ts == 0) && (c < C);
  t1 = 0.0; t2 = 0.0;
  if (owner) {
#pragma unroll
    for (int q = 0; q < 8; ++q) { t1 += sh[0][q * 8][tc]; t2 += sh[1][q * 8][tc]; }
  }
}

// global element count: host double, or (hi, lo) floats on the device with
// count = hi*4096 + lo (both exact in fp32, so an all-reduce SUM stays exact).
__device__ __forceinline__ double bn_count(double count, const float* cd) {
  return cd ? (double)cd[0] * 4096.0 + (double)cd[1] : count;
}

__global__ __launch_bounds__(kTc * kTs) void bn_collapse_k(const float* __restrict__ partial, int S,
                                                          int64_t C, float* __restrict__ sums, int with_count,
                                                          float count_hi, float count_lo) {
  if (with_count && blockIdx.x == 0 && threadIdx.x == 0) {      // the exactly summable element count of the message
    sums[2 * C] = count_hi;
    sums[2 * C + 1] = count_lo;
  }
  double t1, t2; bool owner; int64_t c;
  tail_sums(partial, S, C, t1, t2, owner, c);
  if (!owner) return;
  sums[c] = (float)t1;
  sums[C + c] = (float)t2;
}

__global__ __launch_bounds__(kTc * kTs) void bn_finalize_k(
    const float* __restrict__ partial, int S, int64_t C, double count,
    const float* __restrict__ count_dev, float eps, float momentum,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ nbt,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ fp) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  double t1, t2; bool owner; int64_t c;
  tail_sums(partial, S, C, t1, t2, owner, c);
  if (!owner) return;
  count = bn_count(count, count_dev);
  const double m = t1 / count;
  double sumvar = t2 - t1 * m;          // syncbn.py:91
  if (sumvar < 0.0) sumvar = 0.0;
  const double bias_var = sumvar / count;
  const float mf = (float)m;
  const float isf = (float)(1.0 / sqrt(bias_var + (double)eps));
  mean[c] = mf;
  invstd[c] = isf;
  const float a = (gamma ? gamma[c] : 1.f) * isf;
  fp[c] = a;
  fp[C + c] = fmaf(-mf, a, beta ? beta[c] : 0.f);
  fp[2 * C + c] = mf;
  if (rmean) rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
  if (rvar) {
    const double unbias = sumvar / (count - 1.0);  // syncbn.py:92
    rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unbias);
  }
}

__global__ void bn_affine_k(const float* __restrict__ mean, const float* __restrict__ invstd,
                            const float* __restrict__ gamma, const float* __restrict__ beta, int64_t C,
                            float* __restrict__ fp) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float a = (gamma ? gamma[c] : 1.f) * invstd[c];
  fp[c] = a;
  fp[C + c] = fmaf(-mean[c], a, beta ? beta[c] : 0.f);
  fp[2 * C + c] = mean[c];
}

__global__ __launch_bounds__(kTc * kTs) void bn_bwd_coeffs_k(
    const float* __restrict__ partial, int S, int64_t C, double count,
    const float* __restrict__ count_dev, int batch_stats, const float* __restrict__ invstd,
    const float* __restrict__ fp, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ bp) {
  double t1, t2; bool owner; int64_t c;
  tail_sums(partial, S, C, t1, t2, owner, c);
  if (!owner) return;
  const double is = (double)invstd[c];
  if (dbeta) dbeta[c] = (float)t1;
  if (dgamma) dgamma[c] = (float)(t2 * is);      // sum dy' * xhat   (syncbn_kernel.cu:130)
  if (bp) {
    const float a = fp[c];
    float bc = 0.f, c2 = 0.f;
    if (batch_stats) {
      count = bn_count(count, count_dev);
      const double k0 = t1 / count, k1 = t2 * is / count;
      bc = (float)(-(double)a * k1 * is);
      c2 = (float)(-(double)a * k0);
    }
    bp[c] = a;
    bp[C + c] = fp[C + c];
    bp[2 * C + c] = fp[2 * C + c];
    bp[3 * C + c] = bc;
    bp[4 * C + c] = c2;
  }
}

// ---- host-side dispatch helpers ------------------------------------------
template <typename T, int V, int MODE>
static int launch_reduce(const T* x, const T* dy, const T* y, int layout, int64_t N,
                         int64_t C, int64_t HW, const float* fp, int mask,
                         float* partial, hipStream_t st) {
  if (layout == TSG_NCHW) {
    NchwGeom g = nchw_geom(N, C, HW, V);
    dim3 grid((unsigned)C, (unsigned)g.split);
#define L_(MK) hipLaunchKernelGGL((bn_reduce_nchw<T, V, MODE, MK>), grid, dim3(kThreads), 0, st, \
      x, dy, y, N, C, HW, g.seg, g.segs, g.split, fp, partial)
    if (MODE == 0 || mask == 0) L_(0); else if (mask == 1) L_(1); else L_(2);
#undef L_
  } else {
    const int64_t M = N * HW;
    NhwcGeom g = nhwc_geom(M, C, V);
    dim3 grid((unsigned)g.split, (unsigned)g.ytiles);
    const size_t sh = (size_t)2 * g.rows_per_iter * g.gt * V * sizeof(typename RedAcc<T, MODE>::type);
#define L_(MK) hipLaunchKernelGGL((bn_reduce_nhwc<T, V, MODE, MK>), grid, dim3(kThreads), sh, st, \
      x, dy, y, M, C, g.gt, g.rows_per_iter, g.rows_per_block, fp, partial)
    if (MODE == 0 || mask == 0) L_(0); else if (mask == 1) L_(1); else L_(2);
#undef L_
  }
  TSG_CHECK_LAUNCH();
  return 0;
}

template <typename T, int V>
static int launch_fwd(const T* x, const T* res, T* y, int layout, int64_t N, int64_t C,
                      int64_t HW, const float* fp, int relu, hipStream_t st) {
  const bool R_ = relu != 0, S_ = res != nullptr;
  if (layout == TSG_NCHW) {
    NchwGeom g = nchw_geom(N, C, HW, V);
    const int64_t blocks = N * C * g.segs;
    if (blocks > 0x7fffffffLL) return TSG_E_SHAPE;
#define L_(A, B) hipLaunchKernelGGL((bn_fwd_nchw<T, V, A, B>), dim3((unsigned)blocks), dim3(kThreads), 0, st, \
      x, res, y, C, HW, g.seg, g.segs, fp)
    if (R_ && S_) L_(true, true); else if (R_) L_(true, false); else if (S_) L_(false, true); else L_(false, false);
#undef L_
  } else {
    const int64_t M = N * HW;
    NhwcGeom g = nhwc_geom(M, C, V);
    dim3 grid((unsigned)g.split, (unsigned)g.ytiles);
#define L_(A, B) hipLaunchKernelGGL((bn_fwd_nhwc<T, V, A, B>), grid, dim3(kThreads), 0, st, \
      x, res, y, M, C, g.gt, g.rows_per_iter, g.rows_per_block, fp, bn_reverse())
    if (R_ && S_) L_(true, true); else if (R_) L_(true, false); else if (S_) L_(false, true); else L_(false, false);
#undef L_
  }
  TSG_CHECK_LAUNCH();
  return 0;
}

// the bit-mask forms (NHWC, V = 8 bf16 / 4 fp32 channels per byte): forward of BN -> (+identity) -> ReLU that also writes
// the mask, and the two backward kernels that read it instead of y
template <typename T, int V>
static int launch_fwd_bits(const T* x, const T* res, T* y, unsigned char* bits, int64_t N, int64_t C, int64_t HW,
                           const float* fp, hipStream_t st) {
  const int64_t M = N * HW;
  NhwcGeom g = nhwc_geom(M, C, V);
  dim3 grid((unsigned)g.split, (unsigned)g.ytiles);
  if (res)
    hipLaunchKernelGGL((bn_fwd_nhwc<T, V, true, true, true>), grid, dim3(kThreads), 0, st, x, res, y, M, C, g.gt,
                       g.rows_per_iter, g.rows_per_block, fp, bn_reverse(), bits);
  else
    hipLaunchKernelGGL((bn_fwd_nhwc<T, V, true, false, true>), grid, dim3(kThreads), 0, st, x, res, y, M, C, g.gt,
                       g.rows_per_iter, g.rows_per_block, fp, bn_reverse(), bits);
  TSG_CHECK_LAUNCH();
  return 0;
}

template <typename T, int V>
static int launch_reduce_bits(const T* x, const T* dy, const unsigned char* bits, int64_t N, int64_t C, int64_t HW,
                              const float* fp, float* partial, hipStream_t st) {
  const int64_t M = N * HW;
  NhwcGeom g = nhwc_geom(M, C, V);
  dim3 grid((unsigned)g.split, (unsigned)g.ytiles);
  const size_t sh = (size_t)2 * g.rows_per_iter * g.gt * V * sizeof(typename RedAcc<T, 1>::type);
  hipLaunchKernelGGL((bn_reduce_nhwc<T, V, 1, 3>), grid, dim3(kThreads), sh, st, x, dy, reinterpret_cast<const T*>(bits), M, C,
                     g.gt, g.rows_per_iter, g.rows_per_block, fp, partial);
  TSG_CHECK_LAUNCH();
  return 0;
}

template <typename T, int V>
static int launch_bwd_bits(const T* dy, const T* x, const unsigned char* bits, T* dx, T* dres, int64_t N, int64_t C,
                           int64_t HW, const float* bp, hipStream_t st) {
  const int64_t M = N * HW;
  NhwcGeom g = nhwc_geom(M, C, V);
  dim3 grid((unsigned)g.split, (unsigned)g.ytiles);
#define L_(D) hipLaunchKernelGGL((bn_bwd_nhwc<T, V, 3, D>), grid, dim3(kThreads), 0, st, dy, x, reinterpret_cast<const T*>(bits), \
      dx, dres, M, C, g.gt, g.rows_per_iter, g.rows_per_block, bp, bn_reverse())
  if (dres) L_(true); else L_(false);
#undef L_
  TSG_CHECK_LAUNCH();
  return 0;
}

template <typename T, int V>
static int launch_bwd(const T* dy, const T* x, const T* y, T* dx, T* dres, int layout,
                      int64_t N, int64_t C, int64_t HW, const float* bp, int mask, hipStream_t st) {
  const bool D_ = dres != nullptr;
  if (layout == TSG_NCHW) {
    NchwGeom g = nchw_geom(N, C, HW, V);
    const int64_t blocks = N * C * g.segs;
    if (blocks > 0x7fffffffLL) return TSG_E_SHAPE;
#define L_(MK, D) hipLaunchKernelGGL((bn_bwd_nchw<T, V, MK, D>), dim3((unsigned)blocks), dim3(kThreads), 0, st, \
      dy, x, y, dx, dres, C, HW, g.seg, g.segs, bp)
    if (mask == 0) { if (D_) L_(0, true); else L_(0, false); }
    else if (mask == 1) { if (D_) L_(1, true); else L_(1, false); }
    else { if (D_) L_(2, true); else L_(2, false); }
#undef L_
  } else {
    const int64_t M = N * HW;
    NhwcGeom g = nhwc_geom(M, C, V);
    dim3 grid((unsigned)g.split, (unsigned)g.ytiles);
#define L_(MK, D) hipLaunchKernelGGL((bn_bwd_nhwc<T, V, MK, D>), grid, dim3(kThreads), 0, st, \
      dy, x, y, dx, dres, M, C, g.gt, g.rows_per_iter, g.rows_per_block, bp, bn_reverse())
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

static int partial_rows(int layout, int64_t N, int64_t C, int64_t HW, int V) {
  return layout == TSG_NCHW ? nchw_geom(N, C, HW, V).split : nhwc_geom(N * HW, C, V).split;
}

static int bn_reduce_dispatch(int mode, const void* x, const void* dy, const void* y,
                              int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                              const float* fp, int mask, float* partial, void* stream, int* rows) {
  int e = check_dims(dtype, layout, N, C, HW);
  if (e) return e;
  if (!x || !partial) return TSG_E_NULL;
  hipStream_t st = (hipStream_t)stream;
  const int V = pick_vec(dtype, layout, C, HW, x, dy, mask == 1 ? y : nullptr, nullptr, nullptr);
  *rows = partial_rows(layout, N, C, HW, V) * (dtype == TSG_F32 ? 2 : 1);   // fp32 tensors: fp64 sums as hi + lo rows
#define GO(T, VV)                                                                          \
  (mode == 0 ? launch_reduce<T, VV, 0>((const T*)x, nullptr, nullptr, layout, N, C, HW,    \
                                       fp, 0, partial, st)                                 \
             : launch_reduce<T, VV, 1>((const T*)x, (const T*)dy, (const T*)y, layout, N,  \
                                       C, HW, fp, mask, partial, st))
  if (dtype == TSG_F32) return V == 4 ? GO(float, 4) : GO(float, 1);
  return V == 8 ? GO(bf16_t, 8) : GO(bf16_t, 1);
#undef GO
}

}  // namespace tsg

using namespace tsg;

extern "C" {

// The partial count must not depend on pointer alignment, so it is computed for
// the scalar and the vector geometries and the largest one is reported.
int tsg_bn_num_partials(int layout, int64_t N, int64_t C, int64_t HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  if (layout != TSG_NCHW && layout != TSG_NHWC) return TSG_E_LAYOUT;
  int best = 1;
  const int vs[3] = {1, 4, 8};
  for (int i = 0; i < 3; ++i) {
    const int V = vs[i];
    if (layout == TSG_NHWC && C % V) continue;
    const int s = partial_rows(layout, N, C, HW, V);
    if (s > best) best = s;
  }
  return 2 * best;                                          // fp32 statistics write a hi and a lo row per slice
}

size_t tsg_bn_partial_ws_bytes(int layout, int64_t N, int64_t C, int64_t HW) {
  const int s = tsg_bn_num_partials(layout, N, C, HW);
  if (s < 0) return 0;
  return (size_t)s * 2 * (size_t)C * sizeof(float);
}

int tsg_bn_stats(const void* x, int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                 float* partial, int* rows, void* stream) {
  int r = 0;
  int e = bn_reduce_dispatch(0, x, nullptr, nullptr, dtype, layout, N, C, HW, nullptr, 0, partial,
                             stream, &r);
  if (rows) *rows = r;
  return e;
}

int tsg_bn_collapse(const float* partial, int S, int64_t C, float* sums, void* stream) {
  if (!partial || !sums) return TSG_E_NULL;
  if (S <= 0 || C <= 0) return TSG_E_SHAPE;
  hipLaunchKernelGGL(bn_collapse_k, dim3(ceil_div_i(C, kTc)), dim3(kTc * kTs), 0, (hipStream_t)stream,
                     partial, S, C, sums, 0, 0.f, 0.f);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_collapse_count(const float* partial, int S, int64_t C, float* msg, int64_t count, void* stream) {
  if (!partial || !msg) return TSG_E_NULL;
  if (S <= 0 || C <= 0 || count < 0) return TSG_E_SHAPE;
  hipLaunchKernelGGL(bn_collapse_k, dim3(ceil_div_i(C, kTc)), dim3(kTc * kTs), 0, (hipStream_t)stream,
                     partial, S, C, msg, 1, (float)(count / 4096), (float)(count % 4096));
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_finalize(const float* partial, int S, int64_t C, double count, const float* count_dev,
                    float eps, float momentum, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float* mean, float* invstd, float* fwd_pack, void* stream) {
  if (!partial || !mean || !invstd || !fwd_pack) return TSG_E_NULL;
  if (S <= 0 || C <= 0 || (!count_dev && !(count > 0.0))) return TSG_E_SHAPE;
  hipLaunchKernelGGL(bn_finalize_k, dim3(ceil_div_i(C, kTc)), dim3(kTc * kTs), 0, (hipStream_t)stream,
                     partial, S, C, count, count_dev, eps, momentum, gamma, beta, running_mean,
                     running_var, num_batches_tracked, mean, invstd, fwd_pack);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_affine(const float* mean, const float* invstd, const float* gamma, const float* beta,
                  int64_t C, float* fwd_pack, void* stream) {
  if (!mean || !invstd || !fwd_pack) return TSG_E_NULL;
  if (C <= 0) return TSG_E_SHAPE;
  hipLaunchKernelGGL(bn_affine_k, dim3(ceil_div_i(C, 128)), dim3(128), 0, (hipStream_t)stream, mean,
                     invstd, gamma, beta, C, fwd_pack);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_apply_fwd(const void* x, const void* residual, void* y, int dtype, int layout,
                     int64_t N, int64_t C, int64_t HW, const float* fwd_pack, int relu,
                     void* stream) {
  int e = check_dims(dtype, layout, N, C, HW);
  if (e) return e;
  if (!x || !y || !fwd_pack) return TSG_E_NULL;
  if (!aligned16(fwd_pack)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int V = pick_vec(dtype, layout, C, HW, x, residual, y, nullptr, nullptr);
#define GO(T, VV) launch_fwd<T, VV>((const T*)x, (const T*)residual, (T*)y, layout, N, C, HW, fwd_pack, relu, st)
  if (dtype == TSG_F32) return V == 4 ? GO(float, 4) : GO(float, 1);
  return V == 8 ? GO(bf16_t, 8) : GO(bf16_t, 1);
#undef GO
}

int tsg_bn_bwd_reduce(const void* dy, const void* x, const void* y, int dtype, int layout,
                      int64_t N, int64_t C, int64_t HW, const float* fwd_pack, int relu,
                      float* partial, int* rows, void* stream) {
  if (!dy || !fwd_pack) return TSG_E_NULL;
  if (!aligned16(fwd_pack)) return TSG_E_ALIGN;
  const int mask = relu ? (y ? 1 : 2) : 0;
  int r = 0;
  int e = bn_reduce_dispatch(1, x, dy, y, dtype, layout, N, C, HW, fwd_pack, mask, partial, stream, &r);
  if (rows) *rows = r;
  return e;
}

int tsg_bn_bwd_coeffs(const float* partial, int S, int64_t C, double count, const float* count_dev,
                      int batch_stats, const float* invstd, const float* fwd_pack, float* dgamma,
                      float* dbeta, float* bwd_pack, void* stream) {
  if (!partial || !invstd || (bwd_pack && !fwd_pack)) return TSG_E_NULL;
  if (S <= 0 || C <= 0 || (bwd_pack && batch_stats && !count_dev && !(count > 0.0))) return TSG_E_SHAPE;
  hipLaunchKernelGGL(bn_bwd_coeffs_k, dim3(ceil_div_i(C, kTc)), dim3(kTc * kTs), 0, (hipStream_t)stream,
                     partial, S, C, count, count_dev, batch_stats, invstd, fwd_pack, dgamma, dbeta, bwd_pack);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_bwd_apply(const void* dy, const void* x, const void* y, void* dx, void* dres,
                     int dtype, int layout, int64_t N, int64_t C, int64_t HW,
                     const float* bwd_pack, int relu, void* stream) {
  int e = check_dims(dtype, layout, N, C, HW);
  if (e) return e;
  if (!dy || !x || !dx || !bwd_pack) return TSG_E_NULL;
  if (!aligned16(bwd_pack)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int mask = relu ? (y ? 1 : 2) : 0;
  const int V = pick_vec(dtype, layout, C, HW, x, dy, mask == 1 ? y : nullptr, dx, dres);
#define GO(T, VV) launch_bwd<T, VV>((const T*)dy, (const T*)x, (const T*)y, (T*)dx, (T*)dres, \
                                    layout, N, C, HW, bwd_pack, mask, st)
  if (dtype == TSG_F32) return V == 4 ? GO(float, 4) : GO(float, 1);
  return V == 8 ? GO(bf16_t, 8) : GO(bf16_t, 1);
#undef GO
}


// ---- ReLU mask as one bit per element (block tails: BN -> (+identity) -> ReLU) --------------------------------
static int bits_vec(int dtype, int layout, int64_t C, int64_t HW, const void* a, const void* b, const void* c, const void* d,
                    const void* e) {
  if (layout != TSG_NHWC) return 0;
  const int want = dtype == TSG_BF16 ? 8 : dtype == TSG_F32 ? 4 : 0;
  if (!want || C % want) return 0;
  return pick_vec(dtype, layout, C, HW, a, b, c, d, e) == want ? want : 0;
}

int tsg_bn_maskbits_supported(int dtype, int layout, int64_t C, int64_t HW) {
  return bits_vec(dtype, layout, C, HW, nullptr, nullptr, nullptr, nullptr, nullptr) != 0;
}

int tsg_bn_apply_fwd_maskbits(const void* x, const void* residual, void* y, void* bits, int dtype, int layout, int64_t N,
                              int64_t C, int64_t HW, const float* fwd_pack, void* stream) {
  int e = check_dims(dtype, layout, N, C, HW);
  if (e) return e;
  if (!x || !y || !bits || !fwd_pack) return TSG_E_NULL;
  if (!aligned16(fwd_pack)) return TSG_E_ALIGN;
  const int V = bits_vec(dtype, layout, C, HW, x, residual, y, nullptr, nullptr);
  if (!V) return layout != TSG_NHWC ? TSG_E_LAYOUT : TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TSG_F32)
    return launch_fwd_bits<float, 4>((const float*)x, (const float*)residual, (float*)y, (unsigned char*)bits, N, C, HW, fwd_pack, st);
  return launch_fwd_bits<bf16_t, 8>((const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, (unsigned char*)bits, N, C, HW, fwd_pack, st);
}

int tsg_bn_bwd_reduce_maskbits(const void* dy, const void* x, const void* bits, int dtype, int layout, int64_t N, int64_t C,
                               int64_t HW, const float* fwd_pack, float* partial, int* rows, void* stream) {
  int e = check_dims(dtype, layout, N, C, HW);
  if (e) return e;
  if (!dy || !x || !bits || !fwd_pack || !partial) return TSG_E_NULL;
  if (!aligned16(fwd_pack)) return TSG_E_ALIGN;
  const int V = bits_vec(dtype, layout, C, HW, x, dy, nullptr, nullptr, nullptr);
  if (!V) return layout != TSG_NHWC ? TSG_E_LAYOUT : TSG_E_ALIGN;
  if (rows) *rows = partial_rows(layout, N, C, HW, V) * (dtype == TSG_F32 ? 2 : 1);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TSG_F32)
    return launch_reduce_bits<float, 4>((const float*)x, (const float*)dy, (const unsigned char*)bits, N, C, HW, fwd_pack, partial, st);
  return launch_reduce_bits<bf16_t, 8>((const bf16_t*)x, (const bf16_t*)dy, (const unsigned char*)bits, N, C, HW, fwd_pack, partial, st);
}

int tsg_bn_bwd_apply_maskbits(const void* dy, const void* x, const void* bits, void* dx, void* dres, int dtype, int layout,
                              int64_t N, int64_t C, int64_t HW, const float* bwd_pack, void* stream) {
  int e = check_dims(dtype, layout, N, C, HW);
  if (e) return e;
  if (!dy || !x || !bits || !dx || !bwd_pack) return TSG_E_NULL;
  if (!aligned16(bwd_pack)) return TSG_E_ALIGN;
  const int V = bits_vec(dtype, layout, C, HW, x, dy, nullptr, dx, dres);
  if (!V) return layout != TSG_NHWC ? TSG_E_LAYOUT : TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == TSG_F32)
    return launch_bwd_bits<float, 4>((const float*)dy, (const float*)x, (const unsigned char*)bits, (float*)dx, (float*)dres, N, C, HW, bwd_pack, st);
  return launch_bwd_bits<bf16_t, 8>((const bf16_t*)dy, (const bf16_t*)x, (const unsigned char*)bits, (bf16_t*)dx, (bf16_t*)dres, N, C, HW, bwd_pack, st);
}

// ---- mixed layout (x NCHW, y/dy NHWC) --------------------------------------------
static int mixed_ok(int dtype, int64_t C, int64_t HW) {
  const int V = dtype == TSG_BF16 ? 8 : 4;
  return (dtype == TSG_F32 || dtype == TSG_BF16) && C % 8 == 0 && C <= 128 && HW % V == 0;
}

int tsg_bn_mixed_supported(int dtype, int64_t C, int64_t HW) { return mixed_ok(dtype, C, HW); }

static int mixed_blocks(int64_t N, int64_t HW) {
  int64_t t = N * ((HW + kTP - 1) / kTP);
  if (t > kTargetBlocksNhwc) t = kTargetBlocksNhwc;
  return (int)(t < 1 ? 1 : t);
}

int tsg_bn_mixed_num_partials(int64_t N, int64_t C, int64_t HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  return mixed_blocks(N, HW);
}

int tsg_bn_apply_fwd_mixed(const void* x_nchw, void* y_nhwc, int dtype, int64_t N, int64_t C,
                           int64_t HW, const float* fwd_pack, int relu, void* stream) {
  if (!x_nchw || !y_nhwc || !fwd_pack) return TSG_E_NULL;
  if (N <= 0 || !mixed_ok(dtype, C, HW)) return TSG_E_SHAPE;
  if (!aligned16(x_nchw) || !aligned16(y_nhwc)) return TSG_E_ALIGN;
  const int tpi = (int)((HW + kTP - 1) / kTP);
  const size_t sh = (size_t)kTP * (C + 1) * sizeof(float);
  dim3 grid((unsigned)(N * tpi));
  hipStream_t st = (hipStream_t)stream;
#define GO(T, R) hipLaunchKernelGGL((bn_fwd_mixed<T, R>), grid, dim3(kThreads), sh, st, (const T*)x_nchw, \
                                    (T*)y_nhwc, C, HW, tpi, fwd_pack)
  if (dtype == TSG_F32) { if (relu) GO(float, true); else GO(float, false); }
  else { if (relu) GO(bf16_t, true); else GO(bf16_t, false); }
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_bwd_reduce_mixed(const void* dy_nhwc, const void* x_nchw, int dtype, int64_t N, int64_t C,
                            int64_t HW, const float* fwd_pack, int relu, float* partial, int* rows,
                            void* stream) {
  if (!dy_nhwc || !x_nchw || !fwd_pack || !partial) return TSG_E_NULL;
  if (N <= 0 || !mixed_ok(dtype, C, HW)) return TSG_E_SHAPE;
  if (!aligned16(x_nchw) || !aligned16(dy_nhwc)) return TSG_E_ALIGN;
  const int tpi = (int)((HW + kTP - 1) / kTP);
  const int64_t total = N * tpi;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  size_t sh = (size_t)kTP * (C + 1) * sizeof(float);
  const size_t red = (size_t)2 * C * (kTP / V) * sizeof(float);
  if (red > sh) sh = red;
  const int blocks = mixed_blocks(N, HW);
  if (rows) *rows = blocks;
  hipStream_t st = (hipStream_t)stream;
  const int items = (int)((C * (kTP / V) + kThreads - 1) / kThreads);   // 1..8 for C <= 128
#define GO(T, R, I) hipLaunchKernelGGL((bn_bwd_reduce_mixed<T, R, I>), dim3(blocks), dim3(kThreads), sh, st, \
                                       (const T*)dy_nhwc, (const T*)x_nchw, C, HW, tpi, total, fwd_pack, partial)
#define GI(T, R) do { if (items <= 1) GO(T, R, 1); else if (items <= 2) GO(T, R, 2); else if (items <= 4) GO(T, R, 4); \
                      else GO(T, R, 8); } while (0)
  if (dtype == TSG_F32) { if (relu) GI(float, true); else GI(float, false); }
  else { if (relu) GI(bf16_t, true); else GI(bf16_t, false); }
#undef GI
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_bwd_apply_mixed(const void* dy_nhwc, const void* x_nchw, void* dx_nchw, int dtype, int64_t N,
                           int64_t C, int64_t HW, const float* bwd_pack, int relu, void* stream) {
  if (!dy_nhwc || !x_nchw || !dx_nchw || !bwd_pack) return TSG_E_NULL;
  if (N <= 0 || !mixed_ok(dtype, C, HW)) return TSG_E_SHAPE;
  if (!aligned16(x_nchw) || !aligned16(dy_nhwc) || !aligned16(dx_nchw)) return TSG_E_ALIGN;
  const int tpi = (int)((HW + kTP - 1) / kTP);
  const size_t sh = (size_t)kTP * (C + 1) * sizeof(float);
  dim3 grid((unsigned)(N * tpi));
  hipStream_t st = (hipStream_t)stream;
#define GO(T, R) hipLaunchKernelGGL((bn_bwd_apply_mixed<T, R>), grid, dim3(kThreads), sh, st, \
                                    (const T*)dy_nhwc, (const T*)x_nchw, (T*)dx_nchw, C, HW, tpi, bwd_pack)
  if (dtype == TSG_F32) { if (relu) GO(float, true); else GO(float, false); }
  else { if (relu) GO(bf16_t, true); else GO(bf16_t, false); }
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
