// The classifier convolution of a segmentation head: nn.Conv2d(C_in, n_classes, kernel_size=1) with bias on a
// channels_last bf16 feature map — bisenet network.py:151-161 (`output = self.conv_1x1(fm)`, 256 / 64 -> 19), dfn and the
// 19-class heads of the other families.  The vendor library runs it as an implicit GEMM producing channels_last logits,
// which the criterion kernels (csrc/ohem.hip: per-class planes) then need copied to NCHW, and its backward is two more
// convolution kernels plus the copies back: 0.08 ms forward + 0.21 ms backward + 0.09 ms of layout copies and bias
// passes per BiSeNet step for 2.5 GFLOP (profiles/r04_eager_ops.txt).  The operation is a stream over x (134 MB for the
// 256-channel head at 16 x 128^2) with a 19 x C_in matrix that fits in registers:
//   forward   z[b, n, hw]  = bias[n] + sum_c W[n, c] x[b, hw, c]        MFMA (A = W held in registers, B = pixels straight
//                                                                      from global memory in fragment shape); D rows are
//                                                                      classes, so the PLANAR logits the criterion reads
//                                                                      come out as 64-byte row segments: no layout copy
//   dgrad     dx[b, hw, c] = sum_n dz[b, n, hw] W[n, c]                 MFMA with K = 32 padded classes: A = dz gathered from its
//                                                                      planes (2-byte loads, 64-byte segments), B = W in
//                                                                      registers; lane pairs trade a register so that the
//                                                                      stores are 4 bytes wide
//   wgrad     dW[n, c]     = sum_{b, hw} dz[b, n, hw] x[b, hw, c]       MFMA with K = pixels: A = 16-byte runs of the dz planes,
//                                                                      B = x gathered; per-block partials folded in fp64 in a
//                                                                      fixed order; dbias[n] = sum dz by a plane-sum kernel
// All three are HBM-bound (x / dx once).  n_classes <= 32, C_in in {32, 64, 128, 256}, H*W a multiple of 16.
#include "tsg_common.h"

namespace tsg {

typedef __attribute__((ext_vector_type(8))) __bf16 ch_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ch_f32x16;

constexpr int CH_MAXN = 32;          // classes (one 32-row MFMA tile)
constexpr int CH_MAXC = 256;

struct ChGeom { int64_t B, HW, P; int C, N; };

// ---------------------------------------------------------------- forward
// wave = 32 pixels per iteration; A fragments (the whole weight matrix) stay in registers: KS = C / 16 of them
template <int KS>
__global__ __launch_bounds__(256) void cls_fwd_k(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ bias, bf16_t* __restrict__ z, ChGeom g) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  // A fragment of k step ks: row (class) l31, k = 16 ks + 8 half .. + 7; classes >= N are zero rows
  ch_bf16x8 af[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    uint32_t pk[4] = {0u, 0u, 0u, 0u};
    if (l31 < g.N) {
      const float* wr = w + (int64_t)l31 * g.C + ks * 16 + half * 8;
      const float4 a = *reinterpret_cast<const float4*>(wr), b = *reinterpret_cast<const float4*>(wr + 4);
      pk[0] = pack2_bf16(a.x, a.y); pk[1] = pack2_bf16(a.z, a.w); pk[2] = pack2_bf16(b.x, b.y); pk[3] = pack2_bf16(b.z, b.w);
    }
    const uint4 v = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    af[ks] = __builtin_bit_cast(ch_bf16x8, v);
  }
  // bias of the 16 class rows this lane's accumulator registers hold: row = (r & 3) + 8 (r >> 2) + 4 half
  float bs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
    bs[r] = (bias && n < g.N) ? bias[n] : 0.f;
  }
  const int64_t ngroups = (g.P + 31) / 32;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  for (int64_t grp = wid; grp < ngroups; grp += nw) {
    const int64_t p = grp * 32 + l31;
    const bool ok = p < g.P;
    const bf16_t* xr = x + (ok ? p : g.P - 1) * g.C + half * 8;
    uint4 bv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bv[ks] = *reinterpret_cast<const uint4*>(xr + ks * 16);
    ch_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bs[r];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], __builtin_bit_cast(ch_bf16x8, bv[ks]), acc, 0, 0, 0);
    if (ok) {
      const int64_t b = p / g.HW, hw = p - b * g.HW;
      bf16_t* zb = z + b * g.N * g.HW + hw;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
        if (n < g.N) zb[(int64_t)n * g.HW] = f32_to_bf16(acc[r]);
      }
    }
  }
}

// ---------------------------------------------------------------- backward
// (A first version did both gradients on the VALU — a thread owning 8 channels x 19 classes in registers — and lost to
// the vendor kernels 2-4x: 160 values of prologue / epilogue per thread and 19 + 4 loads per 4 pixels, tools/bench_clshead.py.)
// Both are MFMA kernels with K = 32 (classes, zero padded) resp. K = pixels; the operand that is not k-contiguous in
// memory is gathered with 2-byte loads (coalesced across lanes: 64-byte row segments), which is cheap because every
// matrix here is skinny: 16 MFMAs per 32 pixels.
constexpr int CH_RUN = 8;            // H*W granularity of the backward kernels (one 16-byte run of a dz plane)

__device__ __forceinline__ float ch_px(uint2 v, int j) {
  const uint32_t word = j < 2 ? v.x : v.y;
  return (j & 1) ? __uint_as_float(word & 0xffff0000u) : __uint_as_float(word << 16);
}

// dx[px][c] = sum_n dz[n][px] W[n][c]:  D[32 px x 32 c] = A[32 px x 32 n] B[32 n x 32 c] per c tile
template <int CT>
__global__ __launch_bounds__(256) void cls_dgrad_k(const bf16_t* __restrict__ dz, const float* __restrict__ w,
                                                   bf16_t* __restrict__ dx, ChGeom g) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
  // B fragments: column c = 32 ct + l31, k = class 16 ks + 8 half + e (bf16-rounded master weight; classes >= N are zero)
  ch_bf16x8 bw[CT][2];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = 16 * ks + 8 * half + e;
        f[e] = n < g.N ? w[(int64_t)n * g.C + ct * 32 + l31] : 0.f;
      }
      const uint4 v = make_uint4(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]), pack2_bf16(f[6], f[7]));
      bw[ct][ks] = __builtin_bit_cast(ch_bf16x8, v);
    }
  const int64_t ngroups = (g.P + 31) / 32;
  const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  const bool even = (lane & 1) == 0;
  for (int64_t grp = wid; grp < ngroups; grp += nw) {
    // A fragments: row = pixel l31 of the group, k = class 16 ks + 8 half + e: eight 2-byte loads per k step
    const int64_t p = grp * 32 + l31;
    const bool ok = p < g.P;
    const int64_t pc = ok ? p : g.P - 1, b = pc / g.HW, hw = pc - b * g.HW;
    const bf16_t* zb = dz + b * g.N * g.HW + hw;
    ch_bf16x8 az[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint32_t h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int n = 16 * ks + 8 * half + e;
        h[e] = (ok && n < g.N) ? (uint32_t)zb[(int64_t)n * g.HW] : 0u;
      }
      const uint4 v = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
      az[ks] = __builtin_bit_cast(ch_bf16x8, v);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      ch_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az[0], bw[ct][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(az[1], bw[ct][1], acc, 0, 0, 0);
      // acc[r]: pixel row (r & 3) + 8 (r >> 2) + 4 half, channel l31.  Lane pairs trade one register so that every lane
      // stores two adjacent channels (4 bytes) of one row: the even lane row ra, the odd lane row rb
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int ra = 2 * q, rb = 2 * q + 1;
        const float got = __shfl_xor(even ? acc[rb] : acc[ra], 1, 64);
        const int r = even ? ra : rb;
        const int64_t pr = grp * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const uint32_t word = even ? pack2_bf16(acc[ra], got) : pack2_bf16(got, acc[rb]);
        if (pr < g.P) *reinterpret_cast<uint32_t*>(dx + pr * g.C + ct * 32 + (l31 & ~1)) = word;
      }
    }
  }
}

// dW[n][c] = sum_px dz[n][px] x[px][c]:  D[32 n x 32 c] += A[32 n x 16 px] B[16 px x 32 c]; part[block][n][c]
template <int CT>
__global__ __launch_bounds__(256) void cls_wgrad_k(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ x,
                                                   float* __restrict__ part, ChGeom g) {
  __shared__ float img[CT * 32 * 32];                                    // the block's sum, wave by wave (fixed order)
  const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31, wave = threadIdx.x >> 6;
  ch_f32x16 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
  const int64_t nsteps = g.P / 16;                                       // HW % 16 == 0 (host): a k step never leaves an image
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
  for (int64_t st = wid; st < nsteps; st += nw) {
    const int64_t p0 = st * 16 + 8 * half, b = p0 / g.HW, hw = p0 - b * g.HW;
    // A: row = class l31, k = the 8 pixels p0 .. p0 + 7 of its plane: one 16-byte load
    uint4 av = make_uint4(0u, 0u, 0u, 0u);
    if (l31 < g.N) av = *reinterpret_cast<const uint4*>(dz + (b * g.N + l31) * g.HW + hw);
    const bf16_t* xr = x + p0 * g.C + l31;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      uint32_t h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (uint32_t)xr[(int64_t)e * g.C + ct * 32];
      const uint4 bv = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
      acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ch_bf16x8, av), __builtin_bit_cast(ch_bf16x8, bv), acc[ct], 0, 0, 0);
    }
  }
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* q = img + (ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31;
          *q = (wv == 0 ? 0.f : *q) + acc[ct][r];
        }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < CT * 32 * 32; i += 256) {
    const int ct = i / 1024, n = (i / 32) % 32, c = i % 32;
    if (n < g.N) part[((int64_t)blockIdx.x * g.N + n) * g.C + ct * 32 + c] = img[i];
  }
}

// bias gradient: block (n, b) sums one plane of dz; part_b[b][n]
__global__ __launch_bounds__(256) void cls_dbias_k(const bf16_t* __restrict__ dz, float* __restrict__ part_b, ChGeom g) {
  __shared__ float smr[2 * 4];
  const int n = blockIdx.x;
  const int64_t b = blockIdx.y;
  const bf16_t* zp = dz + (b * g.N + n) * g.HW;
  float a = 0.f, dummy = 0.f;
  for (int64_t i = (int64_t)threadIdx.x * CH_RUN; i < g.HW; i += 256 * CH_RUN) {
    const uint4 v = *reinterpret_cast<const uint4*>(zp + i);
    const uint2 lo = make_uint2(v.x, v.y), hi = make_uint2(v.z, v.w);
#pragma unroll
    for (int j = 0; j < 4; ++j) a += ch_px(lo, j) + ch_px(hi, j);
  }
  block_sum2(a, dummy, smr);
  if (threadIdx.x == 0) part_b[b * g.N + n] = a;
}

// dw[i] = sum over the per-block partials (fixed order, fp64): 64 outputs per block, 4 thread groups each walk a quarter
// of the partials with 8 independent loads in flight (a single thread walking 256 partials was 80 us of pure latency)
__global__ __launch_bounds__(256) void cls_wgrad_fold(const float* __restrict__ part, const float* __restrict__ part_b, int nblk,
                                                      int nblk_b, int64_t nw, int N, float* __restrict__ dw,
                                                      float* __restrict__ db) {
  __shared__ double sm[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + col;
  const bool bias_blk = blockIdx.x == gridDim.x - 1;                      // the last block folds the bias partials instead
  const float* src = bias_blk ? part_b : part;
  const int64_t n_out = bias_blk ? N : nw, idx = bias_blk ? col : i;
  const int nb = bias_blk ? nblk_b : nblk;
  const int per = (nb + 3) / 4, b0 = grp * per, b1 = b0 + per < nb ? b0 + per : nb;
  double t = 0.0;
  if (idx < n_out) {
    for (int b = b0; b < b1; b += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = b + u < b1 ? src[(int64_t)(b + u) * n_out + idx] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) t += (double)v[u];
    }
  }
  sm[grp][col] = t;
  __syncthreads();
  if (grp == 0 && idx < n_out) {
    const double tot = (sm[0][col] + sm[1][col]) + (sm[2][col] + sm[3][col]);
    if (bias_blk) { if (db) db[idx] = (float)tot; }
    else dw[idx] = (float)tot;
  }
}

static int ch_geom(ChGeom* g, int64_t B, int64_t HW, int C, int N) {
  if (B <= 0 || HW <= 0 || C <= 0 || N <= 0 || N > CH_MAXN || C > CH_MAXC || C % 32 || C > 256 || (C & (C - 1)) || HW % 16) return TSG_E_SHAPE;
  if (B * HW > 0x3fffffffffLL / C) return TSG_E_SHAPE;
  g->B = B; g->HW = HW; g->P = B * HW; g->C = C; g->N = N;
  return 0;
}

constexpr int CH_WBLOCKS = 256;      // blocks (= partials) of the weight gradient: one per CU

}  // namespace tsg

using namespace tsg;

extern "C" {

int tsg_cls_head_supported(int dtype, int Cin, int n_classes, int64_t HW) {
  ChGeom g;
  return dtype == TSG_BF16 && ch_geom(&g, 1, HW, Cin, n_classes) == 0;
}

int tsg_cls_head_fwd(const void* x, const float* w, const float* bias, void* z, int64_t B, int64_t HW, int Cin, int n_classes,
                     void* stream) {
  if (!x || !w || !z) return TSG_E_NULL;
  ChGeom g;
  int e = ch_geom(&g, B, HW, Cin, n_classes);
  if (e) return e;
  if (!aligned16(x) || !aligned16(w)) return TSG_E_ALIGN;
  const int64_t ngroups = (g.P + 31) / 32;
  int64_t blocks = (ngroups + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
#define CH_F(KS) hipLaunchKernelGGL((cls_fwd_k<KS>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, w, bias, (bf16_t*)z, g)
  switch (Cin / 16) {
    case 1: CH_F(1); break; case 2: CH_F(2); break; case 4: CH_F(4); break; case 8: CH_F(8); break;
    case 16: CH_F(16); break;
    default: return TSG_E_SHAPE;
  }
#undef CH_F
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_cls_head_dgrad(const void* dz, const float* w, void* dx, int64_t B, int64_t HW, int Cin, int n_classes, void* stream) {
  if (!dz || !w || !dx) return TSG_E_NULL;
  ChGeom g;
  int e = ch_geom(&g, B, HW, Cin, n_classes);
  if (e) return e;
  if (!aligned16(dz) || !aligned16(dx)) return TSG_E_ALIGN;
  const int64_t ngroups = (g.P + 31) / 32;
  int64_t blocks = (ngroups + 3) / 4;
  if (blocks > 1024) blocks = 1024;
  hipStream_t st = (hipStream_t)stream;
#define CH_D(CT) hipLaunchKernelGGL((cls_dgrad_k<CT>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)dz, w, (bf16_t*)dx, g)
  switch (Cin / 32) {
    case 1: CH_D(1); break; case 2: CH_D(2); break; case 4: CH_D(4); break; case 8: CH_D(8); break;
    default: return TSG_E_SHAPE;
  }
#undef CH_D
  TSG_CHECK_LAUNCH();
  return 0;
}

size_t tsg_cls_head_wgrad_ws_bytes(int64_t B, int Cin, int n_classes) {
  if (B <= 0 || Cin <= 0 || n_classes <= 0) return 0;
  return ((size_t)CH_WBLOCKS * n_classes * Cin + (size_t)B * n_classes) * sizeof(float);
}

int tsg_cls_head_wgrad(const void* dz, const void* x, float* dw, float* dbias, int64_t B, int64_t HW, int Cin, int n_classes,
                       void* ws, size_t ws_bytes, void* stream) {
  if (!dz || !x || !dw || !ws) return TSG_E_NULL;
  ChGeom g;
  int e = ch_geom(&g, B, HW, Cin, n_classes);
  if (e) return e;
  if (B > 65535 || ws_bytes < tsg_cls_head_wgrad_ws_bytes(B, Cin, n_classes)) return B > 65535 ? TSG_E_SHAPE : TSG_E_WS;
  if (!aligned16(dz) || !aligned16(x) || !aligned16(ws)) return TSG_E_ALIGN;
  int64_t blocks = (g.P / 16 + 3) / 4;
  if (blocks > CH_WBLOCKS) blocks = CH_WBLOCKS;
  float* part = (float*)ws;
  float* part_b = part + (size_t)CH_WBLOCKS * n_classes * Cin;
  hipStream_t st = (hipStream_t)stream;
#define CH_W(CT) hipLaunchKernelGGL((cls_wgrad_k<CT>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)x, part, g)
  switch (Cin / 32) {
    case 1: CH_W(1); break; case 2: CH_W(2); break; case 4: CH_W(4); break; case 8: CH_W(8); break;
    default: return TSG_E_SHAPE;
  }
#undef CH_W
  TSG_CHECK_LAUNCH();
  if (dbias) {
    hipLaunchKernelGGL(cls_dbias_k, dim3((unsigned)n_classes, (unsigned)B), dim3(256), 0, st, (const bf16_t*)dz, part_b, g);
    TSG_CHECK_LAUNCH();
  }
  const int64_t nw = (int64_t)n_classes * Cin;
  hipLaunchKernelGGL(cls_wgrad_fold, dim3((unsigned)((nw + 63) / 64 + 1)), dim3(256), 0, st, part, part_b, (int)blocks, (int)B, nw,
                     n_classes, dw, dbias);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
