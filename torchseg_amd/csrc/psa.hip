// PSANet collect / distribute attention for gfx950 — the one MFMA kernel.
//
// Restates  out = torch.bmm(X, torch.softmax(A, dim=1))
// (model/psanet/ade.psanet.R101_v1c/network.py:125-126,135-136):
//   X [B, Cx, K], A [B, K, N] (softmax over the K rows of every column j),
//   out[b,c,j] = sum_i X[b,c,i] * P[b,i,j],  P = exp(A - lse_j).
// The reference materialises the fp32 softmax (51.8 MB/sample/branch) and calls
// a library bmm; backward runs softmax-backward plus two more GEMMs.  Here:
//   colstat   one streaming read of A -> lse[b,j]                      (HBM-bound)
//   prob      P (or P^T) in bf16 with the K index contiguous             (HBM-bound)
//   gemm_nt   C[M,N] (+)= sum_k Aop[M,k] * Bop[N,k] on v_mfma_f32_32x32x16_bf16,
//             128x128x64 block tiles, 4 waves x (64x64), LDS rows padded to
//             144 B so ds_read_b128 fragment reads are conflict-free,
//             register-prefetched double buffering                        (MFMA-bound)
//   backward  dX = dOut * P^T (gemm_nt), delta_j = sum_c out*dOut,
//             dA = P o (X^T dOut - delta_j) fused into the gemm epilogue.
// fp32 inputs take the same kernels with bf16 hi/lo operand splitting
// (a*b ~= ah*bh + ah*bl + al*bh, three accumulating passes, ~2^-16 relative),
// which keeps the 1e-4 parity bar without an fp32 tensor-core path.
#include "tsg_common.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace tsg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_ROW = 72;                 // bf16 elements per padded tile row (144 B)
constexpr int GT = 256;                     // threads per gemm block (4 waves)

// ---------------------------------------------------------------------------
// column statistics of A [K, N]: lse[j] = log sum_i exp(A[i][j])
// stage 1: block (jt, chunk, b): 256 columns x rows [chunk*RC, ...): running (m, l)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void psa_colstat1(const T* __restrict__ A, int64_t K, int64_t N,
                                                    int rows_per_chunk, float* __restrict__ pm,
                                                    float* __restrict__ pl, const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int chunk = blockIdx.y, nchunk = gridDim.y;
  const int64_t b = blockIdx.z;
  if (j >= N) return;
  const T* a = A + b * K * N;
  int64_t r0 = (int64_t)chunk * rows_per_chunk, r1 = r0 + rows_per_chunk;
  if (r1 > K) r1 = K;
  float m = -INFINITY, l = 0.f;
  for (int64_t i = r0; i < r1; ++i) {
    const float v = ld1<T>(a + i * N + j);
    const float mn = fmaxf(m, v);
    l = l * __expf(m - mn) + __expf(v - mn);
    m = mn;
  }
  pm[(b * nchunk + chunk) * N + j] = m;
  pl[(b * nchunk + chunk) * N + j] = l;
}

__global__ __launch_bounds__(256) void psa_colstat2(const float* __restrict__ pm, const float* __restrict__ pl,
                                                    int nchunk, int64_t N, float* __restrict__ lse,
                                                    const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t b = blockIdx.y;
  if (j >= N) return;
  float m = -INFINITY;
  for (int c = 0; c < nchunk; ++c) m = fmaxf(m, pm[(b * nchunk + c) * N + j]);
  float l = 0.f;
  for (int c = 0; c < nchunk; ++c) l += pl[(b * nchunk + c) * N + j] * __expf(pm[(b * nchunk + c) * N + j] - m);
  lse[b * N + j] = m + logf(l);
}

// Vectorised stage 1 (N % V == 0): a thread owns V adjacent columns (one 16-byte load per row) and a chunk of rows,
// GR rows in flight at a time; per group one max pass and ONE exp per element (the running sum is rescaled per
// group, not per element).  120 row chunks x N / V threads keep > 10^5 independent 16-byte loads in flight, which
// the one-column-per-thread kernel above (1.2 TB/s at 3600^2) could not.
template <typename T, int V> struct ColVec;
template <> struct ColVec<bf16_t, 8> : Vec<bf16_t> {};
template <> struct ColVec<float, 4> : Vec<float> {};

template <typename T, int V, int GR>
__global__ __launch_bounds__(128) void psa_colstat1_vec(const T* __restrict__ A, int64_t K, int64_t N,
                                                        int rows_per_chunk, float* __restrict__ pm,
                                                        float* __restrict__ pl, const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  const int64_t j = ((int64_t)blockIdx.x * 128 + threadIdx.x) * V;
  const int chunk = blockIdx.y, nchunk = gridDim.y;
  const int64_t b = blockIdx.z;
  if (j >= N) return;
  const T* a = A + b * K * N + j;
  int64_t r0 = (int64_t)chunk * rows_per_chunk, r1 = r0 + rows_per_chunk;
  if (r1 > K) r1 = K;
  float m[V], l[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { m[e] = -INFINITY; l[e] = 0.f; }
  for (int64_t i = r0; i < r1; i += GR) {
    ColVec<T, V> v[GR];
#pragma unroll
    for (int u = 0; u < GR; ++u)
      if (i + u < r1) v[u].load(a + (i + u) * N);
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float mn = m[e];
#pragma unroll
      for (int u = 0; u < GR; ++u)
        if (i + u < r1) mn = fmaxf(mn, v[u].v[e]);
      float acc = l[e] * __expf(m[e] - mn);             // exp(-inf - finite) = 0 on the first group
#pragma unroll
      for (int u = 0; u < GR; ++u)
        if (i + u < r1) acc += __expf(v[u].v[e] - mn);
      m[e] = mn; l[e] = acc;
    }
  }
  float* om = pm + (b * nchunk + chunk) * N + j;
  float* ol = pl + (b * nchunk + chunk) * N + j;
#pragma unroll
  for (int e = 0; e < V; ++e) { om[e] = m[e]; ol[e] = l[e]; }
}

// Stage 2 for many chunks: block = 64 columns x 4 chunk groups; the per-thread loads are independent, the four
// group results meet in LDS (fixed order).
__global__ __launch_bounds__(256) void psa_colstat2_wide(const float* __restrict__ pm, const float* __restrict__ pl,
                                                         int nchunk, int64_t N, float* __restrict__ lse,
                                                         const int* __restrict__ run_if) {
  if (run_if && *run_if == 0) return;
  __shared__ float sm[4][64], sl[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + col;
  const int64_t b = blockIdx.y;
  const int per = (nchunk + 3) / 4;
  const int c0 = grp * per, c1 = (c0 + per < nchunk) ? c0 + per : nchunk;
  float m = -INFINITY, l = 0.f;
  if (j < N) {
    // batches of 16 independent loads (a rolled loop waits for every load before issuing the next: 30 x ~600 ns)
    for (int cb = c0; cb < c1; cb += 16) {
      float vm[16], vl[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const bool in = cb + u < c1;
        vm[u] = in ? pm[(b * nchunk + cb + u) * N + j] : -INFINITY;
        vl[u] = in ? pl[(b * nchunk + cb + u) * N + j] : 0.f;
      }
      float mn = m;
#pragma unroll
      for (int u = 0; u < 16; ++u) mn = fmaxf(mn, vm[u]);
      float acc = (l > 0.f) ? l * __expf(m - mn) : 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) acc += (vl[u] > 0.f) ? vl[u] * __expf(vm[u] - mn) : 0.f;
      m = mn; l = acc;
    }
  }
  sm[grp][col] = m; sl[grp][col] = l;
  __syncthreads();
  if (grp == 0 && j < N) {
    float mm = fmaxf(fmaxf(sm[0][col], sm[1][col]), fmaxf(sm[2][col], sm[3][col]));
    float ll = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) ll += (sl[q][col] > 0.f) ? sl[q][col] * __expf(sm[q][col] - mm) : 0.f;
    lse[b * N + j] = mm + logf(ll);
  }
}

__device__ __forceinline__ void split_bf16(float v, bf16_t& hi, bf16_t& lo) {
  hi = f32_to_bf16(v);
  lo = f32_to_bf16(v - bf16_to_f32(hi));
}

// P[i][j] = exp(A[i][j] - lse[j]) as bf16 (hi [+ lo]), same layout as A
template <typename T, bool SPLIT>
__global__ __launch_bounds__(256) void psa_prob(const T* __restrict__ A, const float* __restrict__ lse,
                                                int64_t K, int64_t N, bf16_t* __restrict__ Phi,
                                                bf16_t* __restrict__ Plo) {
  const int64_t b = blockIdx.y;
  const int64_t total = K * N;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t j = e % N;
    const float p = __expf(ld1<T>(A + b * total + e) - lse[b * N + j]);
    bf16_t h, l;
    split_bf16(p, h, l);
    Phi[b * total + e] = h;
    if (SPLIT) Plo[b * total + e] = l;
  }
}

// generic tiled transpose with optional exp(. - lse[col]) : in [R, C] -> out [C, R] bf16 hi/lo
// MODE 0: plain value; MODE 1: exp(v - lse[c])
template <typename T, int MODE, bool SPLIT>
__global__ __launch_bounds__(256) void psa_transpose(const T* __restrict__ in, const float* __restrict__ lse,
                                                     int64_t R, int64_t Cn, bf16_t* __restrict__ ohi,
                                                     bf16_t* __restrict__ olo) {
  __shared__ float tile[64][65];
  const int64_t b = blockIdx.z;
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const T* src = in + b * R * Cn;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int rr = ty; rr < 64; rr += 4) {
    const int64_t r = r0 + rr, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < Cn) {
      v = ld1<T>(src + r * Cn + c);
      if (MODE == 1) v = __expf(v - lse[b * Cn + c]);
    }
    tile[rr][tx] = v;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4) {
    const int64_t c = c0 + cc, r = r0 + tx;
    if (c < Cn && r < R) {
      bf16_t h, l;
      split_bf16(tile[tx][cc], h, l);
      ohi[b * R * Cn + c * R + r] = h;
      if (SPLIT) olo[b * R * Cn + c * R + r] = l;
    }
  }
}

// fp32 -> bf16 hi/lo, same layout
__global__ __launch_bounds__(256) void psa_split_k(const float* __restrict__ in, int64_t n,
                                                   bf16_t* __restrict__ hi, bf16_t* __restrict__ lo) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
    bf16_t h, l;
    split_bf16(in[e], h, l);
    hi[e] = h; lo[e] = l;
  }
}

// delta[b][j] = sum_c out[b][c][j] * dout[b][c][j]; stage 1 splits the Cx rows over
// gridDim.y chunks (a single pass had only N/256 blocks in flight), stage 2 folds them.
constexpr int kDeltaChunks = 16;

template <typename T>
__global__ __launch_bounds__(256) void psa_delta(const T* __restrict__ out, const T* __restrict__ dout,
                                                 int64_t Cx, int64_t N, float* __restrict__ part) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int chunk = blockIdx.y;
  const int64_t b = blockIdx.z;
  if (j >= N) return;
  const int64_t per = (Cx + kDeltaChunks - 1) / kDeltaChunks;
  int64_t c0 = chunk * per, c1 = c0 + per;
  if (c1 > Cx) c1 = Cx;
  float acc = 0.f;
  for (int64_t c = c0; c < c1; ++c)
    acc += ld1<T>(out + (b * Cx + c) * N + j) * ld1<T>(dout + (b * Cx + c) * N + j);
  part[(b * kDeltaChunks + chunk) * N + j] = acc;
}

__global__ __launch_bounds__(256) void psa_delta_fold(const float* __restrict__ part, int64_t N,
                                                      float* __restrict__ delta) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t b = blockIdx.y;
  if (j >= N) return;
  float v[kDeltaChunks];
#pragma unroll
  for (int c = 0; c < kDeltaChunks; ++c) v[c] = part[(b * kDeltaChunks + c) * N + j];   // independent loads, fixed order sum
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < kDeltaChunks; ++c) acc += v[c];
  delta[b * N + j] = acc;
}

// Vectorised stage 1 (N % V == 0): block = 4 row groups x 64 column vectors; a thread keeps 8 rows of both tensors in
// flight (16 independent 16-byte loads; the scalar kernel above issued 2-byte loads one row at a time: 11.6 us for the
// 15 MB of PSANet's shape), the four groups meet in LDS in a fixed order.
template <typename T, int V>
__global__ __launch_bounds__(256) void psa_delta_vec(const T* __restrict__ out, const T* __restrict__ dout,
                                                     int64_t Cx, int64_t N, float* __restrict__ part) {
  __shared__ float red[3][64][V];
  const int vc = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t j = ((int64_t)blockIdx.x * 64 + vc) * V;
  const int chunk = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t per = (Cx + kDeltaChunks - 1) / kDeltaChunks, sub = (per + 3) / 4;
  int64_t c0 = chunk * per + grp * sub, c1 = c0 + sub, cend = (chunk + 1) * per < Cx ? (chunk + 1) * per : Cx;
  if (c1 > cend) c1 = cend;
  float acc[V];
#pragma unroll
  for (int e = 0; e < V; ++e) acc[e] = 0.f;
  if (j < N) {
    for (int64_t c = c0; c < c1; c += 8) {
      ColVec<T, V> a[8], d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c + u < c1) { a[u].load(out + (b * Cx + c + u) * N + j); d[u].load(dout + (b * Cx + c + u) * N + j); }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c + u < c1) {
#pragma unroll
          for (int e = 0; e < V; ++e) acc[e] += a[u].v[e] * d[u].v[e];
        }
    }
  }
  if (grp) {
#pragma unroll
    for (int e = 0; e < V; ++e) red[grp - 1][vc][e] = acc[e];
  }
  __syncthreads();
  if (grp == 0 && j < N) {
#pragma unroll
    for (int e = 0; e < V; ++e) part[(b * kDeltaChunks + chunk) * N + j + e] = ((acc[e] + red[0][vc][e]) + red[1][vc][e]) + red[2][vc][e];
  }
}

// fp32 path: dA = exp(A - lse_j) * (dP - delta_j)
__global__ __launch_bounds__(256) void psa_da_f32(const float* __restrict__ A, const float* __restrict__ lse,
                                                  const float* __restrict__ dP, const float* __restrict__ delta,
                                                  int64_t K, int64_t N, float* __restrict__ dA) {
  const int64_t b = blockIdx.y;
  const int64_t total = K * N;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t j = e % N;
    const float p = __expf(A[b * total + e] - lse[b * N + j]);
    dA[b * total + e] = p * (dP[b * total + e] - delta[b * N + j]);
  }
}

// ---------------------------------------------------------------------------
// C[M,N] (+)= Aop[M,K] * Bop[N,K]^T   (bf16 operands, K contiguous, fp32 accumulate)
// EPI 0: store (ACC: read-modify-write of an fp32 C)      TO = float | bf16_t
// EPI 1: C[m][n] = P[m][n] * (acc - delta[n])             (dA epilogue, P bf16 [M,N])
// ---------------------------------------------------------------------------
struct GemmArgs {
  const bf16_t* A; const bf16_t* B; void* C;
  int64_t M, N, K;
  int64_t sA, sB, sC;          // batch strides (elements)
  const bf16_t* P; const float* delta; int64_t sP, sD;
  int accumulate;
};

__device__ __forceinline__ uint4 ld16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

template <typename TO, int EPI>
__global__ __launch_bounds__(GT) void gemm_nt_bf16(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];   // 2 stages x (A tile + B tile)
  constexpr int TILE = BM * LDS_ROW;                             // elements per operand tile (BM == BN)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;                       // 2 x 2 waves, 64 x 64 each
  const int64_t b = blockIdx.z;
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;
  const bf16_t* Ag = g.A + b * g.sA;
  const bf16_t* Bg = g.B + b * g.sB;

  // global -> register staging: each tile is 128 rows x 8 chunks of 16 B; 4 chunks per thread per operand
  const int lrow = tid >> 3, lchk = tid & 7;                     // rows lrow + 32*q, q = 0..3
  uint4 ra[4], rb[4];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t r = lrow + 32 * q;
      const int64_t k = k0 + lchk * 8;
      const bool kin = k < g.K;                                  // K % 8 == 0 (host check)
      ra[q] = (kin && m0 + r < g.M) ? ld16(Ag + (m0 + r) * g.K + k) : make_uint4(0, 0, 0, 0);
      rb[q] = (kin && n0 + r < g.N) ? ld16(Bg + (n0 + r) * g.K + k) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stash = [&](int stage) {
    bf16_t* sa = lds + (size_t)stage * 2 * TILE;
    bf16_t* sb = sa + TILE;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = lrow + 32 * q;
      *reinterpret_cast<uint4*>(sa + r * LDS_ROW + lchk * 8) = ra[q];
      *reinterpret_cast<uint4*>(sb + r * LDS_ROW + lchk * 8) = rb[q];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (int)((g.K + BK - 1) / BK);
  fetch(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) fetch((int64_t)(kt + 1) * BK);             // in flight during the MFMAs below
    const bf16_t* sa = lds + (size_t)cur * 2 * TILE + (wm * 64) * LDS_ROW;
    const bf16_t* sb = lds + (size_t)cur * 2 * TILE + TILE + (wn * 64) * LDS_ROW;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int kof = ks * 16 + (lane >> 5) * 8;
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = *reinterpret_cast<const bf16x8*>(sa + (i * 32 + (lane & 31)) * LDS_ROW + kof);
        fb[i] = *reinterpret_cast<const bf16x8*>(sb + (i * 32 + (lane & 31)) * LDS_ROW + kof);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) stash(cur ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  TO* Cg = reinterpret_cast<TO*>(g.C) + b * g.sC;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t n = n0 + wn * 64 + j * 32 + (lane & 31);
      float dl = 0.f;
      if (EPI == 1 && n < g.N) dl = g.delta[b * g.sD + n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < g.M && n < g.N) {
          float v = acc[i][j][r];
          if (EPI == 1) v = bf16_to_f32(g.P[b * g.sP + m * g.N + n]) * (v - dl);
          if (EPI == 0 && g.accumulate) v += ld1<TO>(Cg + m * g.N + n);
          st1<TO>(Cg + m * g.N + n, v);
        }
      }
    }
}

template <typename TO, int EPI>
static int launch_gemm(const GemmArgs& g, int64_t batch, hipStream_t st) {
  if (g.K % 8 != 0) return TSG_E_SHAPE;
  if (!aligned16(g.A) || !aligned16(g.B)) return TSG_E_ALIGN;
  dim3 grid((unsigned)((g.N + BN - 1) / BN), (unsigned)((g.M + BM - 1) / BM), (unsigned)batch);
  const size_t sh = (size_t)2 * 2 * BM * LDS_ROW * sizeof(bf16_t);   // 73,728 B (> the 64 KiB default cap)
  TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16<TO, EPI>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
  hipLaunchKernelGGL((gemm_nt_bf16<TO, EPI>), grid, dim3(GT), sh, st, g);
  TSG_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------
// psa_mm: the bf16 contraction of all three PSA products with the softmax fused in (no P workspace):
//   C[M,N] = sum_k Aop[m,k] * Bop[k,n],   256 x 64 x 64 block tiles, 4 waves stacked along M (64 x 64 each),
//   v_mfma_f32_32x32x16_bf16, double-buffered LDS, one barrier per K tile (tile t+1 is written after the barrier
//   from registers whose loads were issued one iteration earlier, then the loads of tile t+2 are re-issued).
// Operand images in LDS:
//   "NT"  global [rows][K], K contiguous  -> LDS [rows][72] (144-B rows: ds_read_b128 fragments, conflict-free)
//   "TR"  global [K][cols], cols contiguous (the layout A [K,N] and X [Cx,K] / dOut [Cx,N] have for the products that
//         contract over their ROW index) -> LDS [64 k][cols + 32] exactly as it comes from HBM (16-B copies); the
//         k-major MFMA fragments come out of ds_read_b64_tr_b16 (lane mapping as in conv3wrw.hip).  Row strides
//         (cols * 2 + 64) B are odd multiples of 64 B: the 4 rows x 64 B a half-wave touches fall on distinct banks.
// B transforms (the fused softmax):  EXPB 1: B is a TR tile of A, element -> exp(a - lse[n])   (forward)
//                                    EXPB 2: B is an NT tile of A, element -> exp(a - lse[k])   (dX)
//                                    EXPB 3 (round 4, "optimistic" forward): B is a TR tile of A, element -> exp(a), and the
//                                    staging threads keep the COLUMN SUMS of the (unrounded) exponentials: no column
//                                    statistics pass in front of the contraction
// Epilogues (through LDS, 16-B global accesses): EPI 0 store bf16; EPI 1 dA = exp(Araw[m][n] - lse[n]) * (acc - delta[n]);
//   EPI 2 (with EXPB 3): out = acc / colsum[n], lse[n] = log(colsum[n]) written by the tm = 0 tiles, and *flag |= 1 when a
//   column sum left [1e-20, 1e20] (logits beyond ~+-46: exp without the max subtraction is no longer safe) — the host
//   then has the guarded three-kernel path (column statistics + EXPB 1) recompute the call.
//   forward  out[c][j] = sum_i X[c][i] P[i][j]       A = X    NT   B = A     TR  EXPB 1
//   dX       dX[c][i]  = sum_j dOut[c][j] P[i][j]    A = dOut NT   B = A     NT  EXPB 2
//   dA       dP[i][j]  = sum_c X[c][i] dOut[c][j]    A = X    TR   B = dOut  TR  EPI 1
// ---------------------------------------------------------------------------
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef v4i16 __attribute__((address_space(3))) lds_v4i16;

constexpr int MM_BK = 64, MM_T = 256;
constexpr int MM_NT_ROW = 72;                       // elements per NT row (64 + 8 pad)

// BM = rows of C per block (256: 64 per wave, 1 block/CU; 128: 32 per wave, 2 blocks/CU), PF = K tiles whose global
// loads are in flight in registers beyond the one being written to LDS (HBM/L2 latency is ~2 us: with one wave per
// SIMD a single tile of lead time leaves every iteration waiting for its loads)
// BN = columns of C per block: 64 (the four waves stacked along M, each BM / 4 rows x 64 columns) or 128 (round 3: the
// waves form a 2 x 2 grid, each BM / 2 rows x 64 columns).  With M = Cx = 512 rows only, the forward / dX products have
// 2 x 512 x 3600 outputs for 256 CUs: 128 x 128 tiles are one per CU (225 blocks), every staged (and, for B,
// exponentiated) K tile feeds 16 MFMAs per wave instead of 8, and X / dOut are re-read by 29 instead of 57 column tiles.
template <int BM, int BN> struct MmGeom {
  static constexpr int WN = BN / 64, WM = 4 / WN;                              // wave grid
  static constexpr int WROWS = BM / WM;                                        // C rows per wave
  static constexpr int A_TR_ROW = BM + 32;                                     // (2 BM + 64) B: odd multiple of 64 B
  static constexpr int A_ELEMS = (BM * MM_NT_ROW > MM_BK * A_TR_ROW) ? BM * MM_NT_ROW : MM_BK * A_TR_ROW;
  static constexpr int B_TR_ROW = BN + 32;                                     // 96 / 160 elements: odd multiples of 64 B
  static constexpr int B_ELEMS = (BN * MM_NT_ROW > MM_BK * B_TR_ROW) ? BN * MM_NT_ROW : MM_BK * B_TR_ROW;
  static constexpr int STAGE = A_ELEMS + B_ELEMS;
  static constexpr size_t LDS = (size_t)2 * STAGE * sizeof(bf16_t);            // 98,304 B (256 x 64) / 65,536 B (128 x 64) / 81,920 B (128 x 128)
  static constexpr int ACH = BM / 32;                                          // 16-byte A chunks per thread and tile
  static constexpr int BCH = BN / 32;                                          // 16-byte B chunks per thread and tile
  static constexpr int MI = WROWS / 32;                                        // 32-row MFMA tiles per wave along M
  static constexpr int EPI_ROW = BN + 4;                                       // fp32 words per epilogue row
  static constexpr int CPR = BN / 8;                                           // 8-column chunks per epilogue row
  static_assert((size_t)BM * EPI_ROW * 4 <= LDS, "epilogue image must fit the tile buffers");
};

struct MmArgs {
  const bf16_t* A; const bf16_t* B; bf16_t* C;
  int64_t M, N, K;                  // C is [M, N] row-major; NT operands are [rows, K], TR operands [K, cols]
  int64_t sA, sB, sC;               // batch strides (elements)
  const float* lse; int64_t sL;     // EXPB / EPI 1: log-sum-exp per softmax column
  const bf16_t* Araw; int64_t sR;   // EPI 1: raw attention logits [M, N]
  const float* delta; int64_t sD;   // EPI 1
  int tiles_m, tiles_n;             // tile grid per batch
  int per_xcd;                      // ceil(tiles / 8): every XCD gets a contiguous run of tiles
  int m_fastest;                    // tile order inside the run (0: N fastest, the default)
  int ablate;                       // diagnosis only (TSG_PSA_ABLATE): 1 no A reloads, 2 no B reloads, 4 no exp, 8 no MFMA
  const bf16_t* Af; int64_t sAf;    // AF: the NT A operand in MFMA fragment order (psa_frag_k), batch stride in elements
  int MB, KS;                       // AF: 32-row blocks of A, 16-wide k steps (4 per K tile, zero padded)
  int64_t batch;
  float* lse_out;                   // EPI 2: log of the column sums (batch stride sL)
  int* flag;                        // EPI 2: set when a column sum is out of the safe range
  const int* run_if;                // when given: the kernel returns at once unless *run_if != 0 (guarded fallback)
};

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// AF (round 3): the NT A operand (X for the forward, dOut for dX) never goes through LDS.  psa_frag_k lays it out once per
// call in MFMA fragment order, Xf[b][32-row block][k step][lane][8] with lane l = row (l & 31), k = 16 ks + 8 (l >> 5) ..
// (zero padded to whole K tiles), so that an MFMA wave loads an A fragment as ONE coalesced 1 KB read straight into
// registers.  The ablations (profiles/r03_psa_ablations.txt) showed psa_mm bound by its register -> LDS write path
// (768 B per MFMA at 128 x 64 tiles): without the A tile only the exponentiated B tile is left on it (8 KB per K tile).
__global__ __launch_bounds__(256) void psa_frag_k(const bf16_t* __restrict__ X, int64_t M, int64_t K, int MB, int KS,
                                                  bf16_t* __restrict__ Xf, int* __restrict__ clear_flag) {
  if (clear_flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *clear_flag = 0;   // EPI 2's flag, fresh per call
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one 16-byte vector per thread
  const int64_t per_b = (int64_t)MB * KS * 64;
  const int64_t b = blockIdx.y;
  if (v >= per_b) return;
  const int l = (int)(v % 64);
  const int ks = (int)((v / 64) % KS);
  const int mb = (int)(v / (64 * (int64_t)KS));
  const int64_t m = (int64_t)mb * 32 + (l & 31), k = (int64_t)ks * 16 + (l >> 5) * 8;
  uint4 o = make_uint4(0u, 0u, 0u, 0u);
  if (m < M && k + 8 <= K) o = *reinterpret_cast<const uint4*>(X + (b * M + m) * K + k);
  *reinterpret_cast<uint4*>(Xf + (b * per_b + v) * 8) = o;
}

// SPLIT (round 3): 8 waves, two per SIMD with different jobs.  Waves 0-3 only read fragments and issue MFMAs; waves 4-7
// only fetch, transform (the fused softmax: 32 v_exp_f32 + ~130 other VALU instructions per thread and K tile) and
// write the LDS images of the next K tile.  The round-2 kernel did both in every wave, one wave per SIMD, in order:
// its counters (profiles/r03_psa_sq_counters.txt) show SQ_ACTIVE_INST_ANY at 48 % of the wave cycles with the MFMA pipe
// busy 17 % of the time — issue-bound on the staging code.  The matrix pipe and the VALU are separate pipes, so a
// staging wave and an MFMA wave on one SIMD run concurrently (MI355X_MICROARCH.md, wave scheduling).
// UT (round 3): the staging waves prefetch with untracked loads (all AF kernels; and, without AF, the products whose A
// operand is staged too -- dA with its 8 K tiles per output tile is pure load latency, PF = 3 keeps three tiles in flight)
template <int BM, int BN, int PF, bool A_TR, bool B_TR, int EXPB, int EPI, bool SPLIT, bool AF = false, bool UT = AF>
__global__ __launch_bounds__(SPLIT ? 2 * MM_T : MM_T,
                             (UT && !AF && BM == 128 && PF == 2) ? 4 : ((SPLIT && !(AF && BM == 256)) || (BM == 128 && BN == 64) ? 2 : 1))
void psa_mm(MmArgs g) {
  typedef MmGeom<BM, BN> G;
  // AF is instantiated with PF = 2 only: a deeper pipeline spills at 256 VGPRs, and a spill of a register an untracked
  // load is still writing stores garbage (af256x64x3 failed its parity test exactly so)
  static_assert(!AF || (!A_TR && SPLIT && PF == 2), "AF: NT A operand, MFMA / staging wave split, two register sets");
  static_assert((!AF || UT) && (!UT || (SPLIT && PF >= 2)), "UT: staging waves of their own, at least two register sets");
  extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
  static_assert((EXPB == 3) == (EPI == 2), "the optimistic forward is EXPB 3 with EPI 2");
  static_assert(EPI != 2 || (BN == 64 && B_TR && (size_t)BM * (BN + 4) * 4 + 33 * 64 * 4 <= MmGeom<BM, BN>::LDS),
                "EPI 2: 64-column tiles, column-sum image behind the epilogue image");
  if (g.run_if && *g.run_if == 0) return;                // guarded fallback launch: nothing to redo
  const bool producer = SPLIT && threadIdx.x >= MM_T;    // wave-uniform
  const bool stages = !SPLIT || producer, computes = !SPLIT || !producer;
  const int tid = threadIdx.x & (MM_T - 1), lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, sub = (lane >> 4) & 1, i16 = lane & 15;
  constexpr float kLog2e = 1.4426950408889634f;
  constexpr int WROWS = G::WROWS;                        // C rows per wave
  const int wm = wave / G::WN, wn = wave % G::WN;        // wave grid: WM x WN

  // block -> (batch, tile): consecutive block ids go round-robin over the 8 XCDs; give every XCD a contiguous run of tiles
  const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n * g.batch;
  const int64_t t = (int64_t)(blockIdx.x & 7) * g.per_xcd + (blockIdx.x >> 3);
  if (t >= tiles) return;
  // N fastest: an XCD's run of tiles shares ONE A slab (BM rows x K, <= 1.8 MB: it stays in that XCD's 4 MB L2 while
  // the B operand streams past), instead of every XCD cycling through all of A (measured: section 4a of DESIGN.md)
  int tm, tn;
  if (g.m_fastest) { tm = (int)(t % g.tiles_m); tn = (int)((t / g.tiles_m) % g.tiles_n); }
  else             { tn = (int)(t % g.tiles_n); tm = (int)((t / g.tiles_n) % g.tiles_m); }
  const int64_t b = t / ((int64_t)g.tiles_m * g.tiles_n);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const bf16_t* Ag = g.A + b * g.sA;
  const bf16_t* Bg = g.B + b * g.sB;
  const float* lse = (EXPB == 1 || EXPB == 2 || EPI == 1) ? g.lse + b * g.sL : nullptr;

  // ---- staging maps: 16-byte chunks.  A tile = BM * 8 chunks (ACH per thread), B tile = BN * 8 chunks (BCH per thread)
  //   NT image [rows][8 chunks]: chunk id c -> row c >> 3, k-chunk c & 7
  //   TR image [64 k][cols / 8 chunks]: chunk id c -> k row c / (cols / 8), column chunk c % (cols / 8)
  // Everything that does not depend on the K tile is computed once per thread here (the K loop is issue-bound:
  // 64-bit index arithmetic and software bf16 rounding in it cost more cycles than its MFMAs).
  // Loads are UNCONDITIONAL (per-lane predicates compile to divergent branches, ~40 of them per K tile): rows / columns /
  // k positions outside the problem are clamped to the last valid chunk, i.e. they read real tensor data; the K tail is
  // made exact by zeroing the B chunk (finite x 0), rows / columns beyond M / N are never stored.
  uint4 ra[PF][G::ACH], rb[PF][G::BCH];
  const bf16_t* pa[G::ACH]; int ka[G::ACH], oa[G::ACH];                  // base pointer (k = 0), k inside the tile, LDS offset
  const bf16_t* pb[G::BCH]; int kb[G::BCH], ob[G::BCH]; bool vb[G::BCH]; // vb: column / row of B inside the problem
  const int64_t sa_k = A_TR ? g.M : 1, sb_k = B_TR ? g.N : 1;            // elements per unit of k
  const int Kd = (int)g.K;
  const int ka_max = A_TR ? Kd - 1 : Kd - 8, kb_max = B_TR ? Kd - 1 : Kd - 8;
#pragma unroll
  for (int q = 0; q < G::ACH; ++q) {
    const int c = tid + MM_T * q;
    if (A_TR) {
      const int kr = c / (BM / 8), mc = (c % (BM / 8)) * 8;
      const int64_t m = (m0 + mc < g.M) ? m0 + mc : g.M - 8;
      ka[q] = kr; pa[q] = Ag + m; oa[q] = kr * G::A_TR_ROW + mc;
    } else {
      const int mr = c >> 3, kc = (c & 7) * 8;
      const int64_t m = (m0 + mr < g.M) ? m0 + mr : g.M - 1;
      ka[q] = kc; pa[q] = Ag + m * g.K; oa[q] = mr * MM_NT_ROW + kc;
    }
  }
#pragma unroll
  for (int q = 0; q < G::BCH; ++q) {
    const int c = tid + MM_T * q;
    if (B_TR) {
      const int kr = c / G::CPR, nc = (c % G::CPR) * 8;
      vb[q] = n0 + nc < g.N;
      kb[q] = kr; pb[q] = Bg + (vb[q] ? n0 + nc : g.N - 8); ob[q] = kr * G::B_TR_ROW + nc;
    } else {
      const int nr = c >> 3, kc = (c & 7) * 8;
      vb[q] = n0 + nr < g.N;
      kb[q] = kc; pb[q] = Bg + (vb[q] ? n0 + nr : g.N - 1) * g.K; ob[q] = nr * MM_NT_ROW + kc;
    }
  }
  float csum[8];                                        // EXPB 3: sum over k of exp(a) for the thread's 8 columns
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  float bl[EXPB == 1 ? G::BCH : 1][8];                  // EXPB 1: lse * log2e of the thread's B columns (fixed per block)
  if (EXPB == 1) {
#pragma unroll
    for (int q = 0; q < G::BCH; ++q) {
      const int64_t n = n0 + ((tid + MM_T * q) % G::CPR) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) bl[q][e] = (n + e < g.N) ? lse[n + e] * kLog2e : 0.f;
    }
  }
  // Tiles are fetched in increasing order, so every chunk's pointer just advances by one tile (a 64-bit add); only the
  // last, partial K tile takes the clamped path (K % 64 != 0: 3600 = 56 * 64 + 16).
  const int64_t step_a = (int64_t)MM_BK * sa_k, step_b = (int64_t)MM_BK * sb_k;
#pragma unroll
  for (int q = 0; q < G::ACH; ++q) pa[q] += (int64_t)ka[q] * sa_k;
#pragma unroll
  for (int q = 0; q < G::BCH; ++q) pb[q] += (int64_t)kb[q] * sb_k;
  const int ablate = g.ablate;
  auto fetch = [&](uint4* qa, uint4* qb, int k0) {
    if (k0 + MM_BK <= Kd) {
      if (!AF && (!(ablate & 1) || k0 == 0)) {
#pragma unroll
        for (int q = 0; q < G::ACH; ++q) { qa[q] = ld16(pa[q]); pa[q] += step_a; }
      }
      if (!(ablate & 2) || k0 == 0) {
#pragma unroll
        for (int q = 0; q < G::BCH; ++q) { qb[q] = ld16(pb[q]); pb[q] += step_b; }
      }
    } else {
      if (!AF) {
#pragma unroll
        for (int q = 0; q < G::ACH; ++q) {
          const int over = k0 + ka[q] - ka_max;                // > 0: this chunk lies beyond K, read the last valid one
          qa[q] = ld16(pa[q] - (over > 0 ? (int64_t)over * sa_k : 0));
        }
      }
#pragma unroll
      for (int q = 0; q < G::BCH; ++q) {
        const int over = k0 + kb[q] - kb_max;
        qb[q] = ld16(pb[q] - (over > 0 ? (int64_t)over * sb_k : 0));
      }
    }
  };
  // exp(a - lse) of one 16-byte chunk, repacked to bf16 (round to nearest even, like the P the reference's bf16 path holds)
  auto expchunk = [&](uint4 v, const float* l2) -> uint4 {
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float lo = exp2_fast(fmaf(__uint_as_float(w[i] << 16), kLog2e, -l2[2 * i]));
      const float hi = exp2_fast(fmaf(__uint_as_float(w[i] & 0xffff0000u), kLog2e, -l2[2 * i + 1]));
      w[i] = pack2_bf16(lo, hi);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  };
  auto stash = [&](const uint4* qa, const uint4* qb, int stage, int k0, const u32x4* lpre = nullptr) {
    bf16_t* sa = lds + (size_t)stage * G::STAGE;
    bf16_t* sb = sa + G::A_ELEMS;
    if (!AF) {
#pragma unroll
      for (int q = 0; q < G::ACH; ++q) *reinterpret_cast<uint4*>(sa + oa[q]) = qa[q];
    }
#pragma unroll
    for (int q = 0; q < G::BCH; ++q) {
      uint4 v = qb[q];
      const bool in = vb[q] && k0 + kb[q] < Kd;            // the K tail (and the padding) must be exact zeros, not exp(-lse)
      if (ablate & 4) {
      } else if (EXPB == 3) {
        // exp(a) itself; the thread's chunks all lie in the same 8 columns (256 % CPR == 0), so one register row of sums
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float lo = exp2_fast(__uint_as_float(w[i] << 16) * kLog2e);
          const float hi = exp2_fast(__uint_as_float(w[i] & 0xffff0000u) * kLog2e);
          if (in) { csum[2 * i] += lo; csum[2 * i + 1] += hi; }
          w[i] = pack2_bf16(lo, hi);
        }
        v = make_uint4(w[0], w[1], w[2], w[3]);
      } else if (EXPB == 1) {
        v = expchunk(v, bl[EXPB == 1 ? q : 0]);
      } else if (EXPB == 2 && UT) {                        // lse of the chunk's 8 k positions came with the tile (TSG_UT_FETCH)
        const u32x4 l0 = lpre[2 * q], l1 = lpre[2 * q + 1];
        const float l2[8] = {__uint_as_float(l0.x) * kLog2e, __uint_as_float(l0.y) * kLog2e, __uint_as_float(l0.z) * kLog2e,
                             __uint_as_float(l0.w) * kLog2e, __uint_as_float(l1.x) * kLog2e, __uint_as_float(l1.y) * kLog2e,
                             __uint_as_float(l1.z) * kLog2e, __uint_as_float(l1.w) * kLog2e};
        v = expchunk(v, l2);
      } else if (EXPB == 2) {
        const int k = k0 + kb[q] < kb_max ? k0 + kb[q] : kb_max;
        const float4 l0 = *reinterpret_cast<const float4*>(lse + k), l1 = *reinterpret_cast<const float4*>(lse + k + 4);
        const float l2[8] = {l0.x * kLog2e, l0.y * kLog2e, l0.z * kLog2e, l0.w * kLog2e,
                             l1.x * kLog2e, l1.y * kLog2e, l1.z * kLog2e, l1.w * kLog2e};
        v = expchunk(v, l2);
      }
      v.x = in ? v.x : 0u; v.y = in ? v.y : 0u; v.z = in ? v.z : 0u; v.w = in ? v.w : 0u;
      *reinterpret_cast<uint4*>(sb + ob[q]) = v;
    }
  };

  f32x16 acc[G::MI][2];
#pragma unroll
  for (int i = 0; i < G::MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment bases (elements).  NT: row = 32 i + (lane & 31), k = 16 ks + 8 half .. +7 (one ds_read_b128).
  // TR: source row k = 16 ks + 8 half + (i16 >> 2) (+4 for the second read), column = 32 i + 16 sub + 4 (i16 & 3).
  const int a_nt = (wm * WROWS + (lane & 31)) * MM_NT_ROW + half * 8;
  const int a_tr = (8 * half + (i16 >> 2)) * G::A_TR_ROW + wm * WROWS + 16 * sub + 4 * (i16 & 3);
  const int b_nt = (wn * 64 + (lane & 31)) * MM_NT_ROW + half * 8;
  const int b_tr = (8 * half + (i16 >> 2)) * G::B_TR_ROW + wn * 64 + 16 * sub + 4 * (i16 & 3);

  // tile t travels in register set t % PF: fetched PF iterations before it is written to LDS
  // AF: the MFMA waves fetch their own A fragments (1 KB coalesced reads) PF - 1 K tiles ahead, PF register sets
  u32x4 xa[AF ? PF : 1][AF ? G::MI : 1][4];
  const bf16_t* afp[AF ? G::MI : 1];                      // next K tile to fetch (tiles are fetched in order)
  if (AF) {
#pragma unroll
    for (int i = 0; i < G::MI; ++i) {
      int mb = (int)((m0 + wm * WROWS) / 32) + i;
      if (mb > g.MB - 1) mb = g.MB - 1;                    // rows beyond M are never stored
      afp[i] = g.Af + b * g.sAf + ((int64_t)mb * g.KS * 64 + lane) * 8;
    }
  }
  // ADV: pointer step after the load, 0 once the last tile has been requested (the tail re-reads it: the number of loads
  // in flight never changes, so ONE s_waitcnt site with a constant serves every iteration -- a branch around the wait
  // makes the compiler merge register copies of the set, and it may place them BEFORE the wait)
#define TSG_AF_LOAD(SET, ADV)                                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < (AF ? G::MI : 1); ++i_) {                                       \
    TSG_ASM_LD16(xa[SET][i_][0], afp[i_], 0);    TSG_ASM_LD16(xa[SET][i_][1], afp[i_], 1024);             \
    TSG_ASM_LD16(xa[SET][i_][2], afp[i_], 2048); TSG_ASM_LD16(xa[SET][i_][3], afp[i_], 3072);             \
    afp[i_] += (ADV);                                                                                     \
  }
  // UT, staging waves: the B tile (EXPB 2: and the lse of its k positions; without AF: and the A tile) by the same
  // untracked loads, PF tiles ahead; ONE load site per register (the pointer is selected, not the load) and one wait site,
  // as for the fragments.  Tiles beyond the last one re-read the last one, so (PF - 1) sets are in flight behind the one
  // being written to LDS.
  constexpr bool UTA = UT && !AF;                          // the A tile is staged as well
  constexpr int LPC = UT ? (EXPB == 2 ? 3 : 1) : 1;        // loads per B chunk
  constexpr int LPS = UT ? (UTA ? G::ACH : 0) + LPC * G::BCH : 1;   // loads per set
  u32x4 ya[UTA ? PF : 1][UTA ? G::ACH : 1];
  u32x4 xb[UT ? PF : 1][UT ? G::BCH : 1], xl[UT && EXPB == 2 ? PF : 1][UT && EXPB == 2 ? 2 * G::BCH : 1];
#define TSG_UT_FETCH(SET, K0, ADV_A, ADV_B)                                                               \
  if (UTA) {                                                                                              \
    _Pragma("unroll") for (int q_ = 0; q_ < (UTA ? G::ACH : 1); ++q_) {                                   \
      const int over_ = (K0) + ka[q_] - ka_max;        /* > 0 only in the partial last tile */            \
      const bf16_t* p_ = pa[q_] - (over_ > 0 ? (int64_t)over_ * sa_k : 0);                                \
      TSG_ASM_LD16(ya[UTA ? SET : 0][UTA ? q_ : 0], p_, 0);                                               \
      pa[q_] += (ADV_A);                                                                                  \
    }                                                                                                     \
  }                                                                                                       \
  _Pragma("unroll") for (int q_ = 0; q_ < (UT ? G::BCH : 1); ++q_) {                                      \
    const int over_ = (K0) + kb[q_] - kb_max;                                                             \
    const bf16_t* p_ = pb[q_] - (over_ > 0 ? (int64_t)over_ * sb_k : 0);                                  \
    TSG_ASM_LD16(xb[SET][q_], p_, 0);                                                                     \
    if (EXPB == 2) {                                                                                      \
      const float* l_ = lse + ((K0) + kb[q_] < kb_max ? (K0) + kb[q_] : kb_max);                          \
      TSG_ASM_LD16(xl[EXPB == 2 ? SET : 0][EXPB == 2 ? 2 * q_ : 0], l_, 0);                               \
      TSG_ASM_LD16(xl[EXPB == 2 ? SET : 0][EXPB == 2 ? 2 * q_ + 1 : 0], l_, 16);                          \
    }                                                                                                     \
    pb[q_] += (ADV_B);                                                                                    \
  }
#define TSG_UT_STASH(SET, STAGE_, K0)                                                                     \
  {                                                                                                       \
    uint4 ta_[UTA ? G::ACH : 1], tb_[UT ? G::BCH : 1];                                                    \
    if (UTA) { _Pragma("unroll") for (int q_ = 0; q_ < (UTA ? G::ACH : 1); ++q_) {                        \
      const u32x4 v_ = ya[UTA ? SET : 0][UTA ? q_ : 0]; ta_[q_] = make_uint4(v_.x, v_.y, v_.z, v_.w); } } \
    _Pragma("unroll") for (int q_ = 0; q_ < (UT ? G::BCH : 1); ++q_)                                      \
      tb_[q_] = make_uint4(xb[SET][q_].x, xb[SET][q_].y, xb[SET][q_].z, xb[SET][q_].w);                   \
    stash(ta_, tb_, STAGE_, K0, xl[EXPB == 2 ? SET : 0]);                                                 \
  }
#define TSG_UT_WAIT(SET, N_)                                                                              \
  {                                                                                                       \
    vm_wait<N_>(xb[SET][0]);                                                                              \
    _Pragma("unroll") for (int q_ = 1; q_ < (UT ? G::BCH : 1); ++q_) vm_tie(xb[SET][q_]);                 \
    if (UTA) { _Pragma("unroll") for (int q_ = 0; q_ < (UTA ? G::ACH : 1); ++q_)                          \
      vm_tie(ya[UTA ? SET : 0][UTA ? q_ : 0]); }                                                          \
    if (EXPB == 2) { _Pragma("unroll") for (int q_ = 0; q_ < 2 * (UT ? G::BCH : 1); ++q_)                 \
      vm_tie(xl[EXPB == 2 ? SET : 0][EXPB == 2 ? q_ : 0]); }                                              \
  }
  const int nk = (int)((g.K + MM_BK - 1) / MM_BK);
  if (AF && computes) {
#pragma unroll
    for (int u = 0; u + 1 < PF; ++u) { TSG_AF_LOAD(AF ? u : 0, u + 1 < nk ? 4 * 512 : 0) }
  }
  if (UT && stages) {
    TSG_UT_FETCH(0, 0, nk > 1 ? step_a : 0, nk > 1 ? step_b : 0)
    TSG_UT_WAIT(0, 0)
    TSG_UT_STASH(0, 0, 0)
#pragma unroll
    for (int u = 1; u <= PF; ++u) {                       // tile min(u, nk - 1) into set u % PF
      const int tu = u < nk ? u : nk - 1;
      TSG_UT_FETCH(UT ? u % PF : 0, tu * MM_BK, u + 1 < nk ? step_a : 0, u + 1 < nk ? step_b : 0)
    }
  } else if (stages) {
    fetch(ra[0], rb[0], 0);
    stash(ra[0], rb[0], 0, 0);
#pragma unroll
    for (int u = 1; u <= PF; ++u)
      if (u < nk) fetch(ra[u % PF], rb[u % PF], u * MM_BK);
  }
  for (int kt0 = 0; kt0 < nk; kt0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int kt = kt0 + u;
      if (kt >= nk) break;
      const int s = (u + 1) % PF;                        // == (kt + 1) % PF: kt0 is a multiple of PF
      __syncthreads();                                   // tile kt is complete in LDS; tile kt-1's reads are done
      if (UT && stages) {
        TSG_UT_WAIT(UT ? s : 0, (PF - 1) * LPS)            // set s = tile kt + 1 (or a re-read of the last one)
        if (kt + 1 < nk) TSG_UT_STASH(UT ? s : 0, (kt + 1) & 1, (kt + 1) * MM_BK)
        const int tn_ = kt + 1 + PF < nk ? kt + 1 + PF : nk - 1;
        TSG_UT_FETCH(UT ? s : 0, tn_ * MM_BK, kt + 2 + PF < nk ? step_a : 0, kt + 2 + PF < nk ? step_b : 0)
      } else if (stages) {
        if (kt + 1 < nk) stash(ra[s], rb[s], (kt + 1) & 1, (kt + 1) * MM_BK);
        if (kt + 1 + PF < nk) fetch(ra[s], rb[s], (kt + 1 + PF) * MM_BK);
      }
      if (!computes || (ablate & 8)) continue;
      if (AF) {                                                        // set u holds tile kt (kt0 is a multiple of PF)
        TSG_AF_LOAD(AF ? (u + PF - 1) % PF : 0, kt + PF < nk ? 4 * 512 : 0)
        vm_wait<(PF - 1) * (AF ? G::MI : 1) * 4>(xa[AF ? u : 0][0][0]);   // all but the PF - 1 later sets have landed
#pragma unroll
        for (int i = 0; i < (AF ? G::MI : 1); ++i)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            if (i + ks) vm_tie(xa[AF ? u : 0][i][ks]);
      }
      const bf16_t* sa = lds + (size_t)(kt & 1) * G::STAGE;
      const bf16_t* sb = sa + G::A_ELEMS;
      // fragments of k step ks + 1 are read while the MFMAs of step ks run (two register sets: with one, every step
      // exposed an LDS latency -- the MFMA waves have nothing else to issue)
      union Frag { v4i16 q[2]; bf16x8 v; };
      Frag fa[2][G::MI], fb[2][2];
      auto read_frags = [&](Frag* fa_, Frag* fb_, int ks) {
#pragma unroll
        for (int i = 0; i < G::MI; ++i) {
          if (AF) {
            fa_[i].v = __builtin_bit_cast(bf16x8, xa[AF ? u : 0][AF ? i : 0][ks]);
          } else if (A_TR) {
            const lds_v4i16* p = (const lds_v4i16*)(sa + a_tr + ks * 16 * G::A_TR_ROW + i * 32);
            fa_[i].q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)p);
            fa_[i].q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(p + 4 * (G::A_TR_ROW / 4)));
          } else {
            fa_[i].v = *reinterpret_cast<const bf16x8*>(sa + a_nt + i * 32 * MM_NT_ROW + ks * 16);
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (B_TR) {
            const lds_v4i16* p = (const lds_v4i16*)(sb + b_tr + ks * 16 * G::B_TR_ROW + j * 32);
            fb_[j].q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)p);
            fb_[j].q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(p + 4 * (G::B_TR_ROW / 4)));
          } else {
            fb_[j].v = *reinterpret_cast<const bf16x8*>(sb + b_nt + j * 32 * MM_NT_ROW + ks * 16);
          }
        }
      };
      read_frags(fa[0], fb[0], 0);
#pragma unroll
      for (int ks = 0; ks < MM_BK / 16; ++ks) {
        if (ks + 1 < MM_BK / 16) read_frags(fa[(ks + 1) & 1], fb[(ks + 1) & 1], ks + 1);
#pragma unroll
        for (int i = 0; i < G::MI; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i].v, fb[ks & 1][j].v, acc[i][j], 0, 0, 0);
      }
    }
  }

  if (UT) asm volatile("s_waitcnt vmcnt(0)");             // the tail's re-reads still target xa / xb / ya: land them before reuse
  // ---- epilogue through LDS: acc (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 half) -> fp32 image
  // [BM][BN + 4], then every thread owns 8 consecutive columns of a row: 16-byte global accesses
  __syncthreads();
  float* ep = reinterpret_cast<float*>(lds);
  float* cs = ep + BM * G::EPI_ROW;                       // EPI 2: [32 staging rows][64 columns] partial column sums
  if (EPI == 2 && stages) {
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[(tid / G::CPR) * 64 + (tid % G::CPR) * 8 + e] = csum[e];
  }
  if (computes) {
#pragma unroll
    for (int i = 0; i < G::MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          ep[row * G::EPI_ROW + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
        }
  }
  __syncthreads();
  // SPLIT: both halves of the block store, the staging waves take the odd passes
  bf16_t* Cg = g.C + b * g.sC;
  constexpr int RPP = MM_T / G::CPR;                     // rows per pass of the block: 32 / 16
  const int cchunk = tid % G::CPR;
  const int64_t n = n0 + cchunk * 8;
  float dl[8], l2[8];
  if (EPI == 1 && n < g.N) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { dl[e] = g.delta[b * g.sD + n + e]; l2[e] = lse[n + e] * kLog2e; }
  }
  if (EPI == 2) {
    // column sums: the 32 staging rows in a fixed order, one column per thread of wave 0 (the tile of the other M half
    // gets the same values in the same order); reciprocals shared through LDS so that no thread carries them in registers
    float* cinv = cs + (MM_T / G::CPR) * 64;
    if (threadIdx.x < 64) {
      float tot = 0.f;
      for (int q = 0; q < MM_T / G::CPR; ++q) tot += cs[q * 64 + threadIdx.x];
      cinv[threadIdx.x] = 1.f / tot;
      const int64_t nc = n0 + threadIdx.x;
      if (tm == 0 && nc < g.N) {
        g.lse_out[b * g.sL + nc] = logf(tot);
        if (!(tot >= 1e-20f && tot <= 1e20f)) atomicOr(g.flag, 1);
      }
    }
    __syncthreads();
  }
#pragma unroll 4
  for (int q = 0; q < BM / RPP; ++q) {
    if (SPLIT && (q & 1) != (producer ? 1 : 0)) continue;
    const int row = tid / G::CPR + RPP * q;
    const int64_t m = m0 + row;
    if (m >= g.M || n >= g.N) continue;
    const float4 v0 = *reinterpret_cast<const float4*>(ep + row * G::EPI_ROW + cchunk * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(ep + row * G::EPI_ROW + cchunk * 8 + 4);
    float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    if (EPI == 1) {
      const uint4 a = ld16(g.Araw + b * g.sR + m * g.N + n);
      const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // P exactly as the contraction kernels see it: the bf16-rounded exp(a - lse) (v_cvt_pk_bf16_f32: the same round to
        // nearest even as f32_to_bf16, one instruction per pair instead of ~10)
        const uint32_t pp = pack2_bf16(exp2_fast(fmaf(__uint_as_float(w[e] << 16), kLog2e, -l2[2 * e])),
                                       exp2_fast(fmaf(__uint_as_float(w[e] & 0xffff0000u), kLog2e, -l2[2 * e + 1])));
        const float p0 = __uint_as_float(pp << 16), p1 = __uint_as_float(pp & 0xffff0000u);
        v[2 * e] = p0 * (v[2 * e] - dl[2 * e]);
        v[2 * e + 1] = p1 * (v[2 * e + 1] - dl[2 * e + 1]);
      }
    }
    if (EPI == 2) {
      const float* cinv = cs + (MM_T / G::CPR) * 64 + cchunk * 8;
      const float4 i0 = *reinterpret_cast<const float4*>(cinv), i1 = *reinterpret_cast<const float4*>(cinv + 4);
      v[0] *= i0.x; v[1] *= i0.y; v[2] *= i0.z; v[3] *= i0.w; v[4] *= i1.x; v[5] *= i1.y; v[6] *= i1.z; v[7] *= i1.w;
    }
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2_bf16(v[2 * e], v[2 * e + 1]);
    *reinterpret_cast<uint4*>(Cg + m * g.N + n) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

template <int BM, int BN, int PF, bool A_TR, bool B_TR, int EXPB, int EPI, bool SPLIT = false, bool AF = false, bool UT = AF>
static int launch_mm_cfg(MmArgs g, hipStream_t st) {
  g.tiles_m = (int)((g.M + BM - 1) / BM);
  g.tiles_n = (int)((g.N + BN - 1) / BN);
  const int64_t tiles = (int64_t)g.tiles_m * g.tiles_n * g.batch;
  g.per_xcd = (int)((tiles + 7) / 8);
  { const char* o = getenv("TSG_PSA_ORDER"); g.m_fastest = (o && o[0] == 'm') ? 1 : 0; }
  { const char* o = getenv("TSG_PSA_ABLATE"); g.ablate = o ? atoi(o) : 0; }
  constexpr size_t lds_bytes = MmGeom<BM, BN>::LDS;
  TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&psa_mm<BM, BN, PF, A_TR, B_TR, EXPB, EPI, SPLIT, AF, UT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL((psa_mm<BM, BN, PF, A_TR, B_TR, EXPB, EPI, SPLIT, AF, UT>), dim3((unsigned)(8 * g.per_xcd)),
                     dim3(SPLIT ? 2 * MM_T : MM_T), lds_bytes, st, g);
  TSG_CHECK_LAUNCH();
  return 0;
}

// tile configuration: TSG_PSA_CFG = "<BM>x<PF>" (64-column tiles, round 2), "128x128x<PF>" (round 3: 2 x 2 wave grid),
// "split<BM>x<BN>x<PF>" (round 3: 4 MFMA waves + 4 staging waves) or "af<BM>x<BN>" (round 3: split, and the NT A operand
// comes from global memory in fragment order with untracked prefetches -- the products with a transposed A operand, dA,
// run split128x64x1 then); bring-up / tuning knob, read once; default chosen by measurement
static int mm_cfg() {
  static int cfg = -1;
  if (cfg < 0) {
    const char* e = getenv("TSG_PSA_CFG");
    cfg = 72560642;      // af256x64: fwd 76 / bwd 136 us at B = 2, 512 x 3600^2 (split128x64x1: 95 / 180; 128x1 of round 2: 94 / 178)
    if (e) {
      if (!strcmp(e, "256x1")) cfg = 2561; else if (!strcmp(e, "256x2")) cfg = 2562;
      else if (!strcmp(e, "128x1")) cfg = 1281; else if (!strcmp(e, "128x2")) cfg = 1282;
      else if (!strcmp(e, "128x128x1")) cfg = 1281281; else if (!strcmp(e, "128x128x2")) cfg = 1281282;
      else if (!strcmp(e, "split128x128x1")) cfg = 91281281; else if (!strcmp(e, "split128x128x2")) cfg = 91281282;
      else if (!strcmp(e, "split128x64x1")) cfg = 9128641; else if (!strcmp(e, "split256x64x1")) cfg = 9256641;
      else if (!strncmp(e, "af128x64", 8)) cfg = 71280642;   // A fragments from global, one K tile ahead
      else if (!strncmp(e, "af256x64", 8)) cfg = 72560642;
    }
  }
  return cfg;
}

// AF configurations apply to the products whose A operand is NT and for which the caller prepared the fragment image
template <bool A_TR, bool B_TR, int EXPB, int EPI>
static int launch_mm_af(MmArgs g, hipStream_t st, int cfg) {
  if constexpr (!A_TR && EPI == 2) {
    // the optimistic forward exists for the 256 x 64 tile only: at 128 x 64 (two blocks per CU, 128 VGPRs) its column sums
    // spill, and a kernel with untracked loads must not spill (tests/test_isa_guards_cpu.py)
    return launch_mm_cfg<256, 64, 2, A_TR, B_TR, EXPB, EPI, true, true>(g, st);
  } else if constexpr (!A_TR) {
    switch (cfg) {
      case 71280642: return launch_mm_cfg<128, 64, 2, A_TR, B_TR, EXPB, EPI, true, true>(g, st);
      default:       return launch_mm_cfg<256, 64, 2, A_TR, B_TR, EXPB, EPI, true, true>(g, st);
    }
  }
  return TSG_E_SHAPE;
}

template <bool A_TR, bool B_TR, int EXPB, int EPI>
static int launch_mm(MmArgs g, hipStream_t st) {
  if (g.M % 8 || g.N % 8 || g.K % 8) return TSG_E_SHAPE;
  if (!aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C)) return TSG_E_ALIGN;
  const int cfg = mm_cfg();
  if (cfg / 10000000 == 7) {
    if (!A_TR && g.Af) return launch_mm_af<A_TR, B_TR, EXPB, EPI>(g, st, cfg);
    // products without a fragment image (dA): both tiles staged by tracked loads, one tile ahead.  TSG_PSA_UT_PF=2 runs
    // them with two sets of untracked prefetch instead: measured 151 vs 141-147 us for the backward call (three sets at one
    // block per CU: 193 us) -- with 8 K tiles per output tile dA is not bound by the depth of its K pipeline
    if constexpr (A_TR) {
      static const int ut_pf = [] { const char* e = getenv("TSG_PSA_UT_PF"); return e ? atoi(e) : 1; }();
      if (ut_pf == 2) return launch_mm_cfg<128, 64, 2, A_TR, B_TR, EXPB, EPI, true, false, true>(g, st);
    }
    return launch_mm_cfg<128, 64, 1, A_TR, B_TR, EXPB, EPI, true>(g, st);
  }
  switch (cfg) {
    case 2561: return launch_mm_cfg<256, 64, 1, A_TR, B_TR, EXPB, EPI>(g, st);
    case 2562: return launch_mm_cfg<256, 64, 2, A_TR, B_TR, EXPB, EPI>(g, st);
    case 1282: return launch_mm_cfg<128, 64, 2, A_TR, B_TR, EXPB, EPI>(g, st);
    case 1281: return launch_mm_cfg<128, 64, 1, A_TR, B_TR, EXPB, EPI>(g, st);
    case 1281282: return launch_mm_cfg<128, 128, 2, A_TR, B_TR, EXPB, EPI>(g, st);
    case 1281281: return launch_mm_cfg<128, 128, 1, A_TR, B_TR, EXPB, EPI>(g, st);
    case 91281282: return launch_mm_cfg<128, 128, 2, A_TR, B_TR, EXPB, EPI, true>(g, st);
    case 91281281: return launch_mm_cfg<128, 128, 1, A_TR, B_TR, EXPB, EPI, true>(g, st);
    case 9256641: return launch_mm_cfg<256, 64, 1, A_TR, B_TR, EXPB, EPI, true>(g, st);
    default:   return launch_mm_cfg<128, 64, 1, A_TR, B_TR, EXPB, EPI, true>(g, st);
  }
}

static size_t au(size_t v) { return (v + 255) / 256 * 256; }

struct PsaWs {
  // common
  float* pm; float* pl; float* lse_tmp;
  bf16_t* P_hi; bf16_t* P_lo;        // [B, K, N] or transposed [B, N, K]
  bf16_t* X_hi; bf16_t* X_lo;        // fp32 path fwd: split X; bwd: X^T (hi/lo)
  bf16_t* D_hi; bf16_t* D_lo;        // bwd: dOut (split) and dOut^T
  bf16_t* Dt_hi; bf16_t* Dt_lo;
  float* delta;
  float* delta_part;                 // [B, kDeltaChunks, N]
  float* dP;                         // fp32 path bwd: [B, K, N]
  bf16_t* Af;                        // bf16 path: the NT A operand (X fwd, dOut bwd) in MFMA fragment order
  int* flag;                         // bf16 forward: "a column sum left the safe range" (optimistic contraction, EPI 2)
  size_t total;
};

constexpr int kChunks = 240;          // row chunks of the column statistics (workspace [B][kChunks][N] x 2)

static inline int af_mb(int64_t M) { return (int)((M + 31) / 32); }
static inline int af_ks(int64_t K) { return 4 * (int)((K + MM_BK - 1) / MM_BK); }
static inline size_t af_elems(int64_t B, int64_t M, int64_t K) { return (size_t)B * af_mb(M) * af_ks(K) * 512; }

// lay the NT A operand out in fragment order when the configured tile wants it (TSG_PSA_CFG=af...)
static int af_prepare(MmArgs& m, bf16_t* Af, hipStream_t st, int* clear_flag = nullptr) {
  if (mm_cfg() / 10000000 != 7 || !Af) return 0;
  m.MB = af_mb(m.M); m.KS = af_ks(m.K);
  const int64_t per_b = (int64_t)m.MB * m.KS * 64;
  hipLaunchKernelGGL(psa_frag_k, dim3((unsigned)((per_b + 255) / 256), (unsigned)m.batch), dim3(256), 0, st, m.A,
                     m.M, m.K, m.MB, m.KS, Af, clear_flag);
  TSG_CHECK_LAUNCH();
  m.Af = Af; m.sAf = per_b * 8;
  return 0;
}

static PsaWs psa_carve(void* base, int64_t B, int64_t Cx, int64_t K, int64_t N, bool f32, bool bwd) {
  PsaWs w;
  char* p = (char*)base;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* q = p + off; off += au(bytes); return q; };
  w.pm = (float*)take((size_t)B * kChunks * N * 4);
  w.pl = (float*)take((size_t)B * kChunks * N * 4);
  w.lse_tmp = (float*)take((size_t)B * N * 4);
  w.flag = (int*)take(256);
  // the bf16 path (psa_mm) fuses the softmax into the contractions and needs no operand images at all
  w.P_hi = f32 ? (bf16_t*)take((size_t)B * K * N * 2) : nullptr;
  w.P_lo = f32 ? (bf16_t*)take((size_t)B * K * N * 2) : nullptr;
  w.X_hi = f32 ? (bf16_t*)take((size_t)B * Cx * K * 2) : nullptr;
  w.X_lo = f32 ? (bf16_t*)take((size_t)B * Cx * K * 2) : nullptr;
  w.D_hi = w.D_lo = w.Dt_hi = w.Dt_lo = nullptr; w.delta = nullptr; w.delta_part = nullptr; w.dP = nullptr;
  w.Af = f32 ? nullptr : (bf16_t*)take(af_elems(B, Cx, bwd ? N : K) * 2);
  if (bwd) {
    w.D_hi = f32 ? (bf16_t*)take((size_t)B * Cx * N * 2) : nullptr;
    w.D_lo = f32 ? (bf16_t*)take((size_t)B * Cx * N * 2) : nullptr;
    w.Dt_hi = f32 ? (bf16_t*)take((size_t)B * Cx * N * 2) : nullptr;
    w.Dt_lo = f32 ? (bf16_t*)take((size_t)B * Cx * N * 2) : nullptr;
    w.delta = (float*)take((size_t)B * N * 4);
    w.delta_part = (float*)take((size_t)B * kDeltaChunks * N * 4);
    w.dP = f32 ? (float*)take((size_t)B * K * N * 4) : nullptr;
  }
  w.total = off;
  return w;
}

static int egrid(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

template <typename T>
static int colstat(const T* A, int64_t B, int64_t K, int64_t N, PsaWs& w, float* lse, hipStream_t st,
                   const int* run_if = nullptr) {
  constexpr int V = sizeof(T) == 2 ? 8 : 4;
  if (N % V == 0 && aligned16(A)) {
    static const int want = [] { const char* e = getenv("TSG_PSA_COLCHUNKS"); const int v = e ? atoi(e) : kChunks;
                                 return v < 1 ? 1 : (v > kChunks ? kChunks : v); }();
    const int rpc = (int)((K + want - 1) / want);
    const int nch = (int)((K + rpc - 1) / rpc);                        // chunks that actually hold rows
    const int64_t nv = N / V;
    hipLaunchKernelGGL((psa_colstat1_vec<T, V, 15>), dim3((unsigned)((nv + 127) / 128), (unsigned)nch, (unsigned)B),
                       dim3(128), 0, st, A, K, N, rpc, w.pm, w.pl, run_if);
    TSG_CHECK_LAUNCH();
    hipLaunchKernelGGL(psa_colstat2_wide, dim3((unsigned)((N + 63) / 64), (unsigned)B), dim3(256), 0, st, w.pm, w.pl,
                       nch, N, lse, run_if);
    TSG_CHECK_LAUNCH();
    return 0;
  }
  const int nch = 30;
  const int rpc = (int)((K + nch - 1) / nch);
  hipLaunchKernelGGL((psa_colstat1<T>), dim3((unsigned)((N + 255) / 256), nch, (unsigned)B), dim3(256), 0, st,
                     A, K, N, rpc, w.pm, w.pl, run_if);
  TSG_CHECK_LAUNCH();
  hipLaunchKernelGGL(psa_colstat2, dim3((unsigned)((N + 255) / 256), (unsigned)B), dim3(256), 0, st, w.pm, w.pl,
                     nch, N, lse, run_if);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // namespace tsg

using namespace tsg;

extern "C" {

size_t tsg_psa_ws_bytes(int dtype, int backward, int64_t B, int64_t Cx, int64_t K, int64_t N) {
  if (B <= 0 || Cx <= 0 || K <= 0 || N <= 0) return 0;
  return psa_carve(nullptr, B, Cx, K, N, dtype == TSG_F32, backward != 0).total;
}

int tsg_psa_fwd(const void* X, const void* A, void* out, float* lse, int dtype, int64_t B, int64_t Cx,
                int64_t K, int64_t N, void* ws, size_t ws_bytes, void* stream) {
  if (!X || !A || !out || !lse || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (B <= 0 || Cx <= 0 || K <= 0 || N <= 0 || K % 8 != 0) return TSG_E_SHAPE;
  const bool f32 = dtype == TSG_F32;
  if (ws_bytes < tsg_psa_ws_bytes(dtype, 0, B, Cx, K, N)) return TSG_E_WS;
  if (!aligned16(ws) || !aligned16(X)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  PsaWs w = psa_carve(ws, B, Cx, K, N, f32, false);
  int e;
  dim3 tgrid((unsigned)((N + 63) / 64), (unsigned)((K + 63) / 64), (unsigned)B);
  GemmArgs g = {};
  g.M = Cx; g.N = N; g.K = K; g.sA = Cx * K; g.sB = N * K; g.sC = Cx * N; g.C = out;
  if (f32) {
    if ((e = colstat<float>((const float*)A, B, K, N, w, lse, st))) return e;
    // P^T [N, K] hi/lo and X hi/lo
    hipLaunchKernelGGL((psa_transpose<float, 1, true>), tgrid, dim3(256), 0, st, (const float*)A, lse, K, N,
                       w.P_hi, w.P_lo);
    TSG_CHECK_LAUNCH();
    hipLaunchKernelGGL(psa_split_k, dim3(egrid(B * Cx * K)), dim3(256), 0, st, (const float*)X, B * Cx * K,
                       w.X_hi, w.X_lo);
    TSG_CHECK_LAUNCH();
    g.A = w.X_hi; g.B = w.P_hi; g.accumulate = 0;
    if ((e = launch_gemm<float, 0>(g, B, st))) return e;
    g.accumulate = 1;
    g.A = w.X_hi; g.B = w.P_lo;
    if ((e = launch_gemm<float, 0>(g, B, st))) return e;
    g.A = w.X_lo; g.B = w.P_hi;
    if ((e = launch_gemm<float, 0>(g, B, st))) return e;
  } else {
    if (Cx % 8 != 0 || N % 8 != 0) return TSG_E_SHAPE;
    MmArgs m = {};
    m.A = (const bf16_t*)X; m.B = (const bf16_t*)A; m.C = (bf16_t*)out;
    m.M = Cx; m.N = N; m.K = K; m.sA = Cx * K; m.sB = K * N; m.sC = Cx * N; m.lse = lse; m.sL = N; m.batch = B;
    // TSG_PSA_OPTIMISTIC=1|0 (default 1; fragment-order tiles only).  Round 3's forward spent 35 % of its time on the
    // column statistics (psa_colstat1_vec + psa_colstat2_wide: a full pass over A before the contraction could start).
    // The optimistic form contracts X with exp(A) directly, sums the columns of exp(A) on the way (in the staging waves,
    // which hold fixed columns) and normalises in the epilogue: one pass over A.  exp without the max subtraction is only
    // safe while the column sums stay in [1e-20, 1e20] (|logit| up to ~46); a tile that sees anything else raises a device
    // flag, and the three launches of the classic path that follow — each returns at once unless the flag is set —
    // recompute the whole call exactly as round 3 did.  No host synchronisation either way.
    static const bool optimistic = [] { const char* o = getenv("TSG_PSA_OPTIMISTIC"); return !(o && o[0] == '0'); }();
    if (optimistic && mm_cfg() / 10000000 == 7 && mm_cfg() != 71280642 && w.Af) {
      if ((e = af_prepare(m, w.Af, st, w.flag))) return e;             // psa_frag_k also clears the flag
      MmArgs o = m;
      o.lse = nullptr; o.lse_out = lse; o.flag = w.flag;
      if ((e = launch_mm_af<false, true, 3, 2>(o, st, mm_cfg()))) return e;
      if ((e = colstat<bf16_t>((const bf16_t*)A, B, K, N, w, lse, st, w.flag))) return e;
      m.run_if = w.flag;
      if ((e = launch_mm<false, true, 1, 0>(m, st))) return e;
      return 0;
    }
    if ((e = colstat<bf16_t>((const bf16_t*)A, B, K, N, w, lse, st))) return e;
    if ((e = af_prepare(m, w.Af, st))) return e;
    if ((e = launch_mm<false, true, 1, 0>(m, st))) return e;
  }
  return 0;
}

int tsg_psa_bwd(const void* X, const void* A, const void* out, const void* dout, const float* lse,
                void* dX, void* dA, int dtype, int64_t B, int64_t Cx, int64_t K, int64_t N, void* ws,
                size_t ws_bytes, void* stream) {
  if (!X || !A || !out || !dout || !lse || !dX || !dA || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (B <= 0 || Cx <= 0 || K <= 0 || N <= 0 || K % 8 != 0 || N % 8 != 0 || Cx % 8 != 0) return TSG_E_SHAPE;
  const bool f32 = dtype == TSG_F32;
  if (ws_bytes < tsg_psa_ws_bytes(dtype, 1, B, Cx, K, N)) return TSG_E_WS;
  if (!aligned16(ws) || !aligned16(dout)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  PsaWs w = psa_carve(ws, B, Cx, K, N, f32, true);
  int e;
  dim3 pgrid((unsigned)egrid(K * N), (unsigned)B);
  dim3 xgrid((unsigned)((K + 63) / 64), (unsigned)((Cx + 63) / 64), (unsigned)B);   // X [Cx, K] -> [K, Cx]
  dim3 dgrid((unsigned)((N + 63) / 64), (unsigned)((Cx + 63) / 64), (unsigned)B);   // dOut [Cx, N] -> [N, Cx]
  dim3 jgrid((unsigned)((N + 255) / 256), (unsigned)B);
  dim3 dgrid3((unsigned)((N + 255) / 256), (unsigned)kDeltaChunks, (unsigned)B);
  GemmArgs gx = {};   // dX[c][i] = sum_j dOut[c][j] * P[i][j]
  gx.M = Cx; gx.N = K; gx.K = N; gx.sA = Cx * N; gx.sB = K * N; gx.sC = Cx * K; gx.C = dX;
  GemmArgs ga = {};   // dP[i][j] = sum_c Xt[i][c] * dOt[j][c]
  ga.M = K; ga.N = N; ga.K = Cx; ga.sA = K * Cx; ga.sB = N * Cx; ga.sC = K * N;
  if (f32) {
    hipLaunchKernelGGL((psa_prob<float, true>), pgrid, dim3(256), 0, st, (const float*)A, lse, K, N, w.P_hi, w.P_lo);
    TSG_CHECK_LAUNCH();
    if (N % 4 == 0 && aligned16(out) && aligned16(dout))
      hipLaunchKernelGGL((psa_delta_vec<float, 4>), dim3((unsigned)((N / 4 + 63) / 64), (unsigned)kDeltaChunks, (unsigned)B),
                         dim3(256), 0, st, (const float*)out, (const float*)dout, Cx, N, w.delta_part);
    else
      hipLaunchKernelGGL((psa_delta<float>), dgrid3, dim3(256), 0, st, (const float*)out, (const float*)dout, Cx, N, w.delta_part);
    TSG_CHECK_LAUNCH();
    hipLaunchKernelGGL(psa_delta_fold, jgrid, dim3(256), 0, st, w.delta_part, N, w.delta);
    TSG_CHECK_LAUNCH();
    hipLaunchKernelGGL(psa_split_k, dim3(egrid(B * Cx * N)), dim3(256), 0, st, (const float*)dout, B * Cx * N, w.D_hi, w.D_lo);
    TSG_CHECK_LAUNCH();
    hipLaunchKernelGGL((psa_transpose<float, 0, true>), xgrid, dim3(256), 0, st, (const float*)X, (const float*)nullptr,
                       Cx, K, w.X_hi, w.X_lo);
    TSG_CHECK_LAUNCH();
    hipLaunchKernelGGL((psa_transpose<float, 0, true>), dgrid, dim3(256), 0, st, (const float*)dout,
                       (const float*)nullptr, Cx, N, w.Dt_hi, w.Dt_lo);
    TSG_CHECK_LAUNCH();
    gx.A = w.D_hi; gx.B = w.P_hi; gx.accumulate = 0;
    if ((e = launch_gemm<float, 0>(gx, B, st))) return e;
    gx.accumulate = 1;
    gx.A = w.D_hi; gx.B = w.P_lo;
    if ((e = launch_gemm<float, 0>(gx, B, st))) return e;
    gx.A = w.D_lo; gx.B = w.P_hi;
    if ((e = launch_gemm<float, 0>(gx, B, st))) return e;
    ga.C = w.dP;
    ga.A = w.X_hi; ga.B = w.Dt_hi; ga.accumulate = 0;
    if ((e = launch_gemm<float, 0>(ga, B, st))) return e;
    ga.accumulate = 1;
    ga.A = w.X_hi; ga.B = w.Dt_lo;
    if ((e = launch_gemm<float, 0>(ga, B, st))) return e;
    ga.A = w.X_lo; ga.B = w.Dt_hi;
    if ((e = launch_gemm<float, 0>(ga, B, st))) return e;
    hipLaunchKernelGGL(psa_da_f32, pgrid, dim3(256), 0, st, (const float*)A, lse, w.dP, w.delta, K, N, (float*)dA);
    TSG_CHECK_LAUNCH();
  } else {
    if (N % 8 == 0 && aligned16(out) && aligned16(dout))
      hipLaunchKernelGGL((psa_delta_vec<bf16_t, 8>), dim3((unsigned)((N / 8 + 63) / 64), (unsigned)kDeltaChunks, (unsigned)B),
                         dim3(256), 0, st, (const bf16_t*)out, (const bf16_t*)dout, Cx, N, w.delta_part);
    else
      hipLaunchKernelGGL((psa_delta<bf16_t>), dgrid3, dim3(256), 0, st, (const bf16_t*)out, (const bf16_t*)dout, Cx, N, w.delta_part);
    TSG_CHECK_LAUNCH();
    hipLaunchKernelGGL(psa_delta_fold, jgrid, dim3(256), 0, st, w.delta_part, N, w.delta);
    TSG_CHECK_LAUNCH();
    MmArgs mx = {};   // dX[c][i] = sum_j dOut[c][j] * exp(A[i][j] - lse[j])
    mx.A = (const bf16_t*)dout; mx.B = (const bf16_t*)A; mx.C = (bf16_t*)dX;
    mx.M = Cx; mx.N = K; mx.K = N; mx.sA = Cx * N; mx.sB = K * N; mx.sC = Cx * K; mx.lse = lse; mx.sL = N; mx.batch = B;
    if ((e = af_prepare(mx, w.Af, st))) return e;
    if ((e = launch_mm<false, false, 2, 0>(mx, st))) return e;
    MmArgs ma = {};   // dA[i][j] = P[i][j] * (sum_c X[c][i] * dOut[c][j] - delta[j])
    ma.A = (const bf16_t*)X; ma.B = (const bf16_t*)dout; ma.C = (bf16_t*)dA;
    ma.M = K; ma.N = N; ma.K = Cx; ma.sA = Cx * K; ma.sB = Cx * N; ma.sC = K * N;
    ma.lse = lse; ma.sL = N; ma.Araw = (const bf16_t*)A; ma.sR = K * N; ma.delta = w.delta; ma.sD = N; ma.batch = B;
    if ((e = launch_mm<true, true, 0, 1>(ma, st))) return e;
  }
  return 0;
}

}  // extern "C"
