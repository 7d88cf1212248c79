// BatchNorm + ReLU + MaxPool2d(3, stride 2, padding 1) of the ResNet stem as ONE pass per direction
// (furnace/base_model/resnet.py:98-100,131-133: x = maxpool(relu(bn1(conv1(x))))), channels_last.
//
// Unfused, the 16 x 64 x 512 x 512 stem activation (537 MB in bf16) makes five trips through HBM forward (stats read,
// normalise read + write, pool read) and eight backward (pool-gradient write, BN reduce 2 reads, BN apply 2 reads + 1
// write, ...).  Here the normalised activation and its gradient are never materialised:
//   forward   y_pool = max over the window of relu(a*x + b), argmax as ONE byte per element (first maximum in scan order:
//             at::native max_pool2d's rule) — the same values as tsg_maxpool_nhwc_fwd(tsg_bn_apply_fwd(x));
//   backward  the gradient of a stem pixel is GATHERED from the <= 4 windows that cover it (dpool where the argmax byte
//             names this pixel), masked by the recomputed ReLU, and consumed on the spot by the BN backward reduction
//             (sum dy', sum dy' (x - mean)) and, in the second pass, by dx = a dy' + Bc (x - mean) + C2.
// HBM per element of x (s = element size): forward s + (s + 1) / 4; backward 2 s + s (+ the pooled side arrays, 1/4 size).
// Packs as in bn.hip: fp[3][C] = {a, b, mean}, bp[5][C] = {a, b, mean, Bc, C2}; partial[S][2][C] fp32.
#include "tsg_common.h"

namespace tsg {

constexpr int kBpT = 256;

template <int V>
__device__ __forceinline__ void bp_ldc(const float* __restrict__ p, int c0, float (&o)[V]) {
#pragma unroll
  for (int q = 0; q < V / 4; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(p + c0 + 4 * q);
    o[4 * q + 0] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
  }
}

template <typename T> __device__ __forceinline__ float round_as(float v);
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_as<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

// Index arithmetic: blockIdx.z = image, blockIdx.y = a chunk of rows, blockIdx.x = a segment of columns; a thread keeps
// ONE (column, channel group) and walks down the rows of its chunk, so the per-channel constants are loaded once and
// there is not a single integer division per element (64-bit divisions were 2/3 of the first version's run time).
struct BpMap { int GT, PW; };                                  // channel groups / pixels per block row
static BpMap bp_map(int C, int V) {
  BpMap m;
  const int G = C / V;
  m.GT = G < kBpT ? G : kBpT;
  m.PW = kBpT / m.GT;
  return m;
}
constexpr int kFwdRows = 8;                                    // pooled rows per forward block
constexpr int kAppRows = 8;                                    // stem row PAIRS per backward-apply block

// ---------------------------------------------------------------- forward
// One row of a window: running maximum of relu(a x + b) over its (up to) three columns and the kx of the FIRST maximum.
// Post-ReLU values are >= 0, so -1 marks "no column yet" / "row outside the image".
// Loads are unconditional (clamped addresses, validity applied to the values): the three columns of both new rows of a step
// are in flight together instead of one load per bounds-check branch.
template <typename T>
__device__ __forceinline__ void pool_row(const T* __restrict__ xn, int iy, int IH, int x0, int IW, int C,
                                         const float (&a)[Vec<T>::N], const float (&b)[Vec<T>::N],
                                         float (&rm)[Vec<T>::N], int (&rk)[Vec<T>::N]) {
  constexpr int V = Vec<T>::N;
  const bool row_ok = iy >= 0 && iy < IH;
  const int iyc = iy < 0 ? 0 : (iy >= IH ? IH - 1 : iy);
  const T* row = xn + (int64_t)iyc * IW * C;
  Vec<T> p[3];
  bool ok[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int ix = x0 + kx;
    ok[kx] = row_ok && ix >= 0 && ix < IW;
    const int ixc = ix < 0 ? 0 : (ix >= IW ? IW - 1 : ix);
    p[kx].load(row + (int64_t)ixc * C);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) { rm[j] = -1.f; rk[j] = 0; }
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float v = fmaxf(fmaf(p[kx].v[j], a[j], b[j]), 0.f);      // the ReLU of tsg_bn_apply_fwd (NaN -> 0 as there)
      v = ok[kx] ? v : -1.f;
      if (v > rm[j]) { rm[j] = v; rk[j] = kx; }
    }
  }
}

// A thread owns one (pooled column, channel group) and walks down kFwdRows pooled rows; window row 2 oy + 1 is window
// row 2 (oy + 1) - 1 of the next output, so its running maximum is carried over: 6 loads per output, not 9.
// The maximum is taken over the UNROUNDED fp32 values (the reference pools fp32 activations); the stored value is the
// rounded maximum = the maximum of the rounded values, so y equals the unfused kernels' bit for bit, and the argmax byte
// differs from theirs only where two window elements round to the same bf16 value (they see a tie, fp32 does not).
template <typename T>
__global__ __launch_bounds__(kBpT) void bn_relu_pool_fwd_k(const T* __restrict__ x, T* __restrict__ y,
                                                           uint8_t* __restrict__ idx, int C, int IH, int IW,
                                                           int OH, int OW, int GT, int PW, int ytiles,
                                                           const float* __restrict__ fp) {
  constexpr int V = Vec<T>::N;
  const int tid = threadIdx.x;
  const int gl = tid % GT, pl = tid / GT;
  const int g = (blockIdx.y % ytiles) * GT + gl;
  const int ox = blockIdx.x * PW + pl;
  if (pl >= PW || g * V >= C || ox >= OW) return;
  const int oy0 = (blockIdx.y / ytiles) * kFwdRows;
  const int oy1 = oy0 + kFwdRows < OH ? oy0 + kFwdRows : OH;
  const int64_t n = blockIdx.z;
  float a[V], b[V];
  bp_ldc<V>(fp, g * V, a);
  bp_ldc<V>(fp + C, g * V, b);
  const int x0 = 2 * ox - 1;
  const T* xn = x + n * IH * (int64_t)IW * C + g * V;
  float m0[V], m1[V], m2[V];
  int k0[V], k1[V], k2[V];
  pool_row<T>(xn, 2 * oy0 - 1, IH, x0, IW, C, a, b, m0, k0);
  for (int oy = oy0; oy < oy1; ++oy) {
    pool_row<T>(xn, 2 * oy, IH, x0, IW, C, a, b, m1, k1);
    pool_row<T>(xn, 2 * oy + 1, IH, x0, IW, C, a, b, m2, k2);
    Vec<T> ov;
    uint32_t w[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float m = m0[j];
      int am = k0[j];
      if (m1[j] > m) { m = m1[j]; am = 3 + k1[j]; }
      if (m2[j] > m) { m = m2[j]; am = 6 + k2[j]; }
      ov.v[j] = m;
      w[j >> 2] |= (uint32_t)am << (8 * (j & 3));
      m0[j] = m2[j]; k0[j] = k2[j];
    }
    const int64_t o = ((n * OH + oy) * (int64_t)OW + ox) * C + g * V;
    ov.store(y + o);
    if (V == 8) *reinterpret_cast<uint2*>(idx + o) = make_uint2(w[0], w[1]);
    else *reinterpret_cast<uint32_t*>(idx + o) = w[0];
  }
}

// ---------------------------------------------------------------- backward
// Both passes walk 2 x 2 blocks of stem pixels: rows 2k, 2k+1 and columns 2m, 2m+1 lie in exactly the windows
// (k..k+1, m..m+1), and the window position each of the four pixels has in each of them is a constant:
//   window (k, m)    : (2k,2m) -> 4   (2k,2m+1) -> 5   (2k+1,2m) -> 7   (2k+1,2m+1) -> 8
//   window (k, m+1)  :                (2k,2m+1) -> 3                    (2k+1,2m+1) -> 6
//   window (k+1, m)  :                                 (2k+1,2m) -> 1   (2k+1,2m+1) -> 2
//   window (k+1, m+1):                                                  (2k+1,2m+1) -> 0
// so a thread that owns one (column pair, channel group) and walks down the row pairs loads every window once (the lower
// two are the upper two of its next step), extracts each argmax byte once and has no data-dependent control flow.
template <typename T> struct PoolWin {
  float d[Vec<T>::N];
  uint32_t w[2];
  __device__ __forceinline__ void load(const T* __restrict__ dpn, const uint8_t* __restrict__ idn, int oy, int ox,
                                       int OH, int OW, int C) {
    constexpr int V = Vec<T>::N;
    const bool ok = oy < OH && ox < OW;                // outside: clamped (valid) address, no position matches
    const int64_t o = ((int64_t)(oy < OH ? oy : OH - 1) * OW + (ox < OW ? ox : OW - 1)) * C;
    Vec<T> v;
    v.load(dpn + o);
#pragma unroll
    for (int j = 0; j < V; ++j) d[j] = v.v[j];
    if (V == 8) {
      const uint2 u = *reinterpret_cast<const uint2*>(idn + o);
      w[0] = ok ? u.x : 0xffffffffu; w[1] = ok ? u.y : 0xffffffffu;
    } else {
      const uint32_t u = *reinterpret_cast<const uint32_t*>(idn + o);
      w[0] = ok ? u : 0xffffffffu; w[1] = 0xffffffffu;
    }
  }
};

// gradients of the four pixels of block (k, m): g00 (2k,2m), g01 (2k,2m+1), g10 (2k+1,2m), g11 (2k+1,2m+1)
template <typename T>
__device__ __forceinline__ void block_grads(const PoolWin<T>& w00, const PoolWin<T>& w01, const PoolWin<T>& w10,
                                            const PoolWin<T>& w11, float (&g00)[Vec<T>::N], float (&g01)[Vec<T>::N],
                                            float (&g10)[Vec<T>::N], float (&g11)[Vec<T>::N]) {
  constexpr int V = Vec<T>::N;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    const int sh = 8 * (j & 3);
    const uint32_t b00 = (w00.w[j >> 2] >> sh) & 0xffu, b01 = (w01.w[j >> 2] >> sh) & 0xffu;
    const uint32_t b10 = (w10.w[j >> 2] >> sh) & 0xffu, b11 = (w11.w[j >> 2] >> sh) & 0xffu;
    g00[j] = b00 == 4u ? w00.d[j] : 0.f;
    g01[j] = (b00 == 5u ? w00.d[j] : 0.f) + (b01 == 3u ? w01.d[j] : 0.f);
    g10[j] = (b00 == 7u ? w00.d[j] : 0.f) + (b10 == 1u ? w10.d[j] : 0.f);
    g11[j] = ((b00 == 8u ? w00.d[j] : 0.f) + (b01 == 6u ? w01.d[j] : 0.f)) +
             ((b10 == 2u ? w10.d[j] : 0.f) + (b11 == 0u ? w11.d[j] : 0.f));
  }
}

// row pairs per block of the reduction: ~2048 blocks (= partial rows) in total
static int bp_red_rows(int64_t N, int KH, int KW, int PW, int ytiles) {
  const int64_t xseg = (KW + PW - 1) / PW;
  int64_t chunks = 2048 / (N * xseg * ytiles);                // row chunks per image
  if (chunks < 1) chunks = 1;
  if (chunks > KH) chunks = KH;
  return (int)((KH + chunks - 1) / chunks);
}

// ---------------------------------------------------------------- backward, reduction
template <typename T>
__global__ __launch_bounds__(kBpT) void bn_relu_pool_bwd_reduce_k(
    const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ x, int C, int IH,
    int IW, int OH, int OW, int GT, int PW, int ytiles, int rows, const float* __restrict__ fp,
    float* __restrict__ partial) {
  constexpr int V = Vec<T>::N;
  extern __shared__ __attribute__((aligned(16))) float smem[];        // [2][PW][GT * V]
  const int tid = threadIdx.x;
  const int gl = tid % GT, pl = tid / GT;
  const int yt = blockIdx.y % ytiles;
  const int g = yt * GT + gl;
  const int m = blockIdx.x * PW + pl;                                 // column pair
  const int KH = (IH + 1) >> 1;
  const bool live = pl < PW && g * V < C && 2 * m < IW;
  const int k0 = (blockIdx.y / ytiles) * rows;
  const int k1 = k0 + rows < KH ? k0 + rows : KH;
  const int64_t n = blockIdx.z;
  float a[V], b[V], mu[V], a1[V], a2[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { a1[j] = 0.f; a2[j] = 0.f; a[j] = 0.f; b[j] = 0.f; mu[j] = 0.f; }
  if (live) {
    bp_ldc<V>(fp, g * V, a); bp_ldc<V>(fp + C, g * V, b); bp_ldc<V>(fp + 2 * C, g * V, mu);
    const T* xn = x + n * IH * (int64_t)IW * C + g * V;
    const T* dpn = dpool + n * OH * (int64_t)OW * C + g * V;
    const uint8_t* idn = idx + n * OH * (int64_t)OW * C + g * V;
    const bool col1 = 2 * m + 1 < IW;
    PoolWin<T> w00, w01, w10, w11;
    w10.load(dpn, idn, k0, m, OH, OW, C);
    w11.load(dpn, idn, k0, m + 1, OH, OW, C);
    for (int k = k0; k < k1; ++k) {
      w00 = w10; w01 = w11;
      w10.load(dpn, idn, k + 1, m, OH, OW, C);
      w11.load(dpn, idn, k + 1, m + 1, OH, OW, C);
      Vec<T> px[4];
      bool ok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {                   // unconditional loads: a pixel outside the image re-reads (2k, 2m)
        const int iy = 2 * k + (q >> 1);
        ok[q] = iy < IH && (!(q & 1) || col1);
        px[q].load(xn + ((int64_t)(ok[q] ? iy : 2 * k) * IW + (ok[q] ? 2 * m + (q & 1) : 2 * m)) * C);
      }
      float gr[4][V];
      block_grads<T>(w00, w01, w10, w11, gr[0], gr[1], gr[2], gr[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float dv = (ok[q] && fmaf(px[q].v[j], a[j], b[j]) > 0.f) ? gr[q][j] : 0.f;
          a1[j] += dv;
          a2[j] = fmaf(dv, px[q].v[j] - mu[j], a2[j]);
        }
      }
    }
  }
  const int W = GT * V;
  float* s1 = smem;
  float* s2 = smem + (size_t)PW * W;
  if (pl < PW) {
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[pl * W + gl * V + j] = a1[j]; s2[pl * W + gl * V + j] = a2[j]; }
  }
  __syncthreads();
  const int64_t prow = ((int64_t)blockIdx.z * (gridDim.y / ytiles) + blockIdx.y / ytiles) * gridDim.x + blockIdx.x;
  for (int t = tid; t < W; t += kBpT) {
    const int c = yt * W + t;
    if (c < C) {
      float t1 = 0.f, t2 = 0.f;
      for (int q = 0; q < PW; ++q) { t1 += s1[q * W + t]; t2 += s2[q * W + t]; }
      partial[(prow * 2 + 0) * C + c] = t1;
      partial[(prow * 2 + 1) * C + c] = t2;
    }
  }
}

// ---------------------------------------------------------------- backward, apply
template <typename T>
__global__ __launch_bounds__(kBpT) void bn_relu_pool_bwd_apply_k(
    const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ x, T* __restrict__ dx,
    int C, int IH, int IW, int OH, int OW, int GT, int PW, int ytiles, const float* __restrict__ bp) {
  constexpr int V = Vec<T>::N;
  const int tid = threadIdx.x;
  const int gl = tid % GT, pl = tid / GT;
  const int g = (blockIdx.y % ytiles) * GT + gl;
  const int m = blockIdx.x * PW + pl;
  if (pl >= PW || g * V >= C || 2 * m >= IW) return;
  const int KH = (IH + 1) >> 1;
  const int k0 = (blockIdx.y / ytiles) * kAppRows;
  const int k1 = k0 + kAppRows < KH ? k0 + kAppRows : KH;
  const int64_t n = blockIdx.z;
  float a[V], b[V], mu[V], bc[V], c2[V];
  bp_ldc<V>(bp, g * V, a);
  bp_ldc<V>(bp + C, g * V, b);
  bp_ldc<V>(bp + 2 * C, g * V, mu);
  bp_ldc<V>(bp + 3 * C, g * V, bc);
  bp_ldc<V>(bp + 4 * C, g * V, c2);
  const T* xn = x + n * IH * (int64_t)IW * C + g * V;
  T* dxn = dx + n * IH * (int64_t)IW * C + g * V;
  const T* dpn = dpool + n * OH * (int64_t)OW * C + g * V;
  const uint8_t* idn = idx + n * OH * (int64_t)OW * C + g * V;
  const bool col1 = 2 * m + 1 < IW;
  PoolWin<T> w00, w01, w10, w11;
  w10.load(dpn, idn, k0, m, OH, OW, C);
  w11.load(dpn, idn, k0, m + 1, OH, OW, C);
  for (int k = k0; k < k1; ++k) {
    w00 = w10; w01 = w11;
    w10.load(dpn, idn, k + 1, m, OH, OW, C);
    w11.load(dpn, idn, k + 1, m + 1, OH, OW, C);
    Vec<T> px[4];
    bool ok[4];
    int64_t off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {                     // unconditional loads: a pixel outside the image re-reads (2k, 2m)
      const int iy = 2 * k + (q >> 1);
      ok[q] = iy < IH && (!(q & 1) || col1);
      off[q] = ((int64_t)(ok[q] ? iy : 2 * k) * IW + (ok[q] ? 2 * m + (q & 1) : 2 * m)) * C;
      px[q].load(xn + off[q]);
    }
    float gr[4][V];
    block_grads<T>(w00, w01, w10, w11, gr[0], gr[1], gr[2], gr[3]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float dv = fmaf(px[q].v[j], a[j], b[j]) > 0.f ? gr[q][j] : 0.f;
        px[q].v[j] = fmaf(a[j], dv, fmaf(bc[j], px[q].v[j] - mu[j], c2[j]));
      }
      if (ok[q]) px[q].store(dxn + off[q]);
    }
  }
}

static int bp_check(int dtype, int64_t N, int C, int IH, int IW, int OH, int OW) {
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (N <= 0 || C <= 0 || C % V || IH <= 0 || IW <= 0) return TSG_E_SHAPE;
  if (OH != (IH - 1) / 2 + 1 || OW != (IW - 1) / 2 + 1) return TSG_E_SHAPE;       // K = 3, S = 2, P = 1
  return 0;
}

}  // namespace tsg

using namespace tsg;

extern "C" {

int tsg_bn_relu_pool_fwd(const void* x, void* y, void* argmax_u8, int dtype, int64_t N, int C, int IH, int IW,
                         int OH, int OW, const float* fp, void* stream) {
  if (!x || !y || !argmax_u8 || !fp) return TSG_E_NULL;
  int e = bp_check(dtype, N, C, IH, IW, OH, OW);
  if (e) return e;
  if (!aligned16(x) || !aligned16(y) || !aligned16(fp) || (((uintptr_t)argmax_u8) & 7u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  const BpMap m = bp_map(C, V);
  const int ytiles = (C / V + m.GT - 1) / m.GT;
  const int64_t gy = (int64_t)((OH + kFwdRows - 1) / kFwdRows) * ytiles;
  if (gy > 65535 || N > 65535) return TSG_E_SHAPE;
  dim3 grid((unsigned)((OW + m.PW - 1) / m.PW), (unsigned)gy, (unsigned)N);
  if (dtype == TSG_BF16)
    hipLaunchKernelGGL((bn_relu_pool_fwd_k<bf16_t>), grid, dim3(kBpT), 0, st, (const bf16_t*)x, (bf16_t*)y,
                       (uint8_t*)argmax_u8, C, IH, IW, OH, OW, m.GT, m.PW, ytiles, fp);
  else
    hipLaunchKernelGGL((bn_relu_pool_fwd_k<float>), grid, dim3(kBpT), 0, st, (const float*)x, (float*)y,
                       (uint8_t*)argmax_u8, C, IH, IW, OH, OW, m.GT, m.PW, ytiles, fp);
  TSG_CHECK_LAUNCH();
  return 0;
}

struct BpRed { BpMap m; int ytiles, rows, chunks; int64_t S; };
static BpRed bp_red(int64_t N, int C, int IH, int IW, int V) {
  BpRed r;
  r.m = bp_map(C, V);
  r.ytiles = (C / V + r.m.GT - 1) / r.m.GT;
  const int KH = (IH + 1) / 2, KW = (IW + 1) / 2;               // 2 x 2 pixel blocks
  r.rows = bp_red_rows(N, KH, KW, r.m.PW, r.ytiles);
  r.chunks = (KH + r.rows - 1) / r.rows;
  r.S = N * r.chunks * (int64_t)((KW + r.m.PW - 1) / r.m.PW);
  return r;
}

int tsg_bn_relu_pool_bwd_num_partials(int dtype, int64_t N, int C, int IH, int IW) {
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  if (N <= 0 || C <= 0 || C % V || IH <= 0 || IW <= 0) return TSG_E_SHAPE;
  const BpRed r = bp_red(N, C, IH, IW, V);
  if (r.S > 0x7fffffffLL) return TSG_E_SHAPE;
  return (int)r.S;
}

int tsg_bn_relu_pool_bwd_reduce(const void* dpool, const void* argmax_u8, const void* x, int dtype, int64_t N, int C,
                                int IH, int IW, int OH, int OW, const float* fp, float* partial, void* stream) {
  if (!dpool || !argmax_u8 || !x || !fp || !partial) return TSG_E_NULL;
  int e = bp_check(dtype, N, C, IH, IW, OH, OW);
  if (e) return e;
  if (!aligned16(x) || !aligned16(dpool) || !aligned16(fp) || (((uintptr_t)argmax_u8) & 7u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  const BpRed r = bp_red(N, C, IH, IW, V);
  const int64_t gy = (int64_t)r.chunks * r.ytiles;
  if (gy > 65535 || N > 65535) return TSG_E_SHAPE;
  dim3 grid((unsigned)(((IW + 1) / 2 + r.m.PW - 1) / r.m.PW), (unsigned)gy, (unsigned)N);
  const size_t sh = (size_t)2 * r.m.PW * r.m.GT * V * sizeof(float);
  if (dtype == TSG_BF16)
    hipLaunchKernelGGL((bn_relu_pool_bwd_reduce_k<bf16_t>), grid, dim3(kBpT), sh, st, (const bf16_t*)dpool,
                       (const uint8_t*)argmax_u8, (const bf16_t*)x, C, IH, IW, OH, OW, r.m.GT, r.m.PW, r.ytiles, r.rows,
                       fp, partial);
  else
    hipLaunchKernelGGL((bn_relu_pool_bwd_reduce_k<float>), grid, dim3(kBpT), sh, st, (const float*)dpool,
                       (const uint8_t*)argmax_u8, (const float*)x, C, IH, IW, OH, OW, r.m.GT, r.m.PW, r.ytiles, r.rows,
                       fp, partial);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_bn_relu_pool_bwd_apply(const void* dpool, const void* argmax_u8, const void* x, void* dx, int dtype, int64_t N,
                               int C, int IH, int IW, int OH, int OW, const float* bp, void* stream) {
  if (!dpool || !argmax_u8 || !x || !dx || !bp) return TSG_E_NULL;
  int e = bp_check(dtype, N, C, IH, IW, OH, OW);
  if (e) return e;
  if (!aligned16(x) || !aligned16(dx) || !aligned16(dpool) || !aligned16(bp) || (((uintptr_t)argmax_u8) & 7u))
    return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int V = dtype == TSG_BF16 ? 8 : 4;
  const BpMap m = bp_map(C, V);
  const int ytiles = (C / V + m.GT - 1) / m.GT;
  const int64_t gy = (int64_t)(((IH + 1) / 2 + kAppRows - 1) / kAppRows) * ytiles;
  if (gy > 65535 || N > 65535) return TSG_E_SHAPE;
  dim3 grid((unsigned)(((IW + 1) / 2 + m.PW - 1) / m.PW), (unsigned)gy, (unsigned)N);
  if (dtype == TSG_BF16)
    hipLaunchKernelGGL((bn_relu_pool_bwd_apply_k<bf16_t>), grid, dim3(kBpT), 0, st, (const bf16_t*)dpool,
                       (const uint8_t*)argmax_u8, (const bf16_t*)x, (bf16_t*)dx, C, IH, IW, OH, OW, m.GT, m.PW, ytiles, bp);
  else
    hipLaunchKernelGGL((bn_relu_pool_bwd_apply_k<float>), grid, dim3(kBpT), 0, st, (const float*)dpool,
                       (const uint8_t*)argmax_u8, (const float*)x, (float*)dx, C, IH, IW, OH, OW, m.GT, m.PW, ytiles, bp);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
