// Weight gradient of the 64 -> 64 channel 3x3 / stride 1 / padding 1 convolutions of ResNet-18's layer1
// (furnace/base_model/resnet.py:24-29 BasicBlock.conv1/conv2 with planes = 64; four per step at
// [16, 64, 256, 256]), channels_last bf16:
//
//     dw[oc][kh][kw][ci] = sum_{b, oh, ow} dy[b, oh, ow, oc] * x[b, oh + kh - 1, ow + kw - 1, ci]
//
// MIOpen runs this as a split-K implicit GEMM with a zero fill and a cast around it (198 + ~25 us per
// convolution on MI355X for 268 MB of operands).  Here: GEMM M = oc (64), N = (tap, ci) = 576, K = pixels.
// K is the pixel axis, so both operands are transposed while they are staged in LDS:
//   dy^T [oc][128 pixels of the tile]                       (two pixels per 32-bit write)
//   x^T  [ci][6 x 34 patch pixels], stored twice (element j at index j and at j + 1) so that the 8 consecutive
//        pixels of a fragment start on a 32-bit boundary for every tap column kw; the per-channel stride of
//        121 dwords spreads the 32 lanes of a ds_read_b32 over the 32 banks.
// A block's 4 waves own the whole [64 x 576] accumulator (9 tiles of 32x32 each: oc half x ci half x 9 taps)
// in MFMA registers across all its tiles; per-block partials are summed in fp64 in a fixed order.
#include "tsg_common.h"
#include <stdlib.h>

namespace tsg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// The LDS images are written with 16- and 32-bit stores and read back as 32- and 128-bit fragments: every
// access goes through may_alias types so that type-based alias analysis cannot reorder or drop them.
typedef uint32_t __attribute__((may_alias)) lds_u32;
typedef uint16_t __attribute__((may_alias)) lds_u16;
typedef bf16x8 __attribute__((may_alias)) lds_bf16x8;

constexpr int W3_C = 64;                    // channels in and out
constexpr int W3_N = 9 * W3_C;              // 576 = (tap, ci)
constexpr int W3_TH = 4, W3_TW = 32;        // output tile: 4 rows x 32 columns = 8 K-steps of 16 pixels
constexpr int W3_PR = W3_TH + 2;            // 6 patch rows
constexpr int W3_PP = (W3_TW + 2) / 2;      // 17 pixel pairs per patch row
constexpr int W3_RD = 20;                   // patch row stride in dwords (40 elements, 34 used)
constexpr int W3_CS = 121;                  // per-channel stride in dwords (6 x 20 = 120 used; odd => conflict-free)
constexpr int W3_COPY = W3_C * W3_CS;       // dwords per alignment copy
constexpr int W3_DS = 68;                   // dy^T row stride in dwords (128 pixels + 8 pad elements)
constexpr int W3_NCH = 8 * W3_PR * W3_PP;   // 816 (pixel pair, 8-channel group) chunks of a patch
constexpr int W3_NU = (W3_NCH + 255) / 256; // 4 per thread
constexpr int W3_NPART = 256;               // persistent blocks: the 144-register accumulator allows one block per CU

// `two` is the constant 2 passed at run time: the two 16-bit stores of alignment copy 1 are adjacent in memory, and
// with a compile-time distance the compiler merges them into one 32-bit store off its natural alignment.
struct W3Geom { int B, H, W, tiles_h, tiles_w, ntiles, two; };

__global__ __launch_bounds__(256) void conv3_wrw_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                   float* __restrict__ part, W3Geom g) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  lds_u32* dyT = reinterpret_cast<lds_u32*>(lds);       // [64][W3_DS]
  lds_u32* xT = dyT + W3_C * W3_DS;                     // [2 copies][64][W3_CS]
  lds_u16* xT16 = reinterpret_cast<lds_u16*>(xT);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, n = lane & 31;
  const int wm = wave >> 1, wh = wave & 1;              // oc half, ci half

  // ---- tile-independent staging descriptors
  const int spp = tid & 15, strow = (tid >> 4) & 1, spart = tid >> 5;      // dy: pixel pair, row (and +2), 8-oc group
  int xoff[W3_NU], xrc[W3_NU], xdst[W3_NU];             // x: element offset from &x[b, oh0, ow0, 0]; r | c << 8; LDS dword
#pragma unroll
  for (int u = 0; u < W3_NU; ++u) {
    const int q = tid + 256 * u;
    const int cpart = q / (W3_PR * W3_PP), rem = q % (W3_PR * W3_PP), r = rem / W3_PP, pc = rem % W3_PP;
    xrc[u] = q < W3_NCH ? (r | ((2 * pc) << 8)) : -1;
    xoff[u] = ((r - 1) * g.W + 2 * pc - 1) * W3_C + cpart * 8;
    xdst[u] = (cpart * 8) * W3_CS + r * W3_RD + pc;
  }
  // ---- fragment addressing (dwords)
  const lds_u32* afrag = dyT + (32 * wm + n) * W3_DS + 4 * half;          // + ks * 8
  const lds_u32* bfrag = xT + (32 * wh + n) * W3_CS + 4 * half;           // + copy / row / column offsets

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  uint4 rd[4];
  uint4 rx[W3_NU][2];
  auto fetch = [&](int tile) {
    const int ow0 = (tile % g.tiles_w) * W3_TW, oh0 = ((tile / g.tiles_w) % g.tiles_h) * W3_TH;
    const int b = tile / (g.tiles_w * g.tiles_h);
    const int64_t org = (((int64_t)b * g.H + oh0) * g.W + ow0) * W3_C;
    const bf16_t* dt = dy + org + ((int64_t)strow * g.W + 2 * spp) * W3_C + spart * 8;
#pragma unroll
    for (int rs = 0; rs < 2; ++rs)
#pragma unroll
      for (int px = 0; px < 2; ++px)
        rd[rs * 2 + px] = (oh0 + strow + 2 * rs < g.H && ow0 + 2 * spp + px < g.W)
                              ? *reinterpret_cast<const uint4*>(dt + ((int64_t)rs * 2 * g.W + px) * W3_C)
                              : make_uint4(0, 0, 0, 0);
    const bf16_t* xo = x + org;
#pragma unroll
    for (int u = 0; u < W3_NU; ++u) {
      const int ih = oh0 - 1 + (xrc[u] & 0xff), iw = ow0 - 1 + (xrc[u] >> 8);
      const bool rowok = xrc[u] >= 0 && ih >= 0 && ih < g.H;
      rx[u][0] = (rowok && iw >= 0 && iw < g.W) ? *reinterpret_cast<const uint4*>(xo + xoff[u]) : make_uint4(0, 0, 0, 0);
      rx[u][1] = (rowok && iw + 1 >= 0 && iw + 1 < g.W) ? *reinterpret_cast<const uint4*>(xo + xoff[u] + W3_C)
                                                         : make_uint4(0, 0, 0, 0);
    }
  };

  int tile = blockIdx.x;
  if (tile < g.ntiles) fetch(tile);
  for (; tile < g.ntiles; tile += gridDim.x) {
    __syncthreads();                                    // the previous tile's fragment reads are done
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      const uint32_t wa[4] = {rd[2 * rs].x, rd[2 * rs].y, rd[2 * rs].z, rd[2 * rs].w};
      const uint32_t wb[4] = {rd[2 * rs + 1].x, rd[2 * rs + 1].y, rd[2 * rs + 1].z, rd[2 * rs + 1].w};
      lds_u32* drow = dyT + (spart * 8) * W3_DS + 16 * (strow + 2 * rs) + spp;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        drow[e * W3_DS] = (e & 1) ? ((wa[e >> 1] >> 16) | (wb[e >> 1] & 0xffff0000u))
                                  : ((wa[e >> 1] & 0xffffu) | (wb[e >> 1] << 16));
    }
#pragma unroll
    for (int u = 0; u < W3_NU; ++u) {
      if (u < W3_NU - 1 || xrc[u] >= 0) {
        const uint32_t wa[4] = {rx[u][0].x, rx[u][0].y, rx[u][0].z, rx[u][0].w};      // patch column 2 pc
        const uint32_t wb[4] = {rx[u][1].x, rx[u][1].y, rx[u][1].z, rx[u][1].w};      // patch column 2 pc + 1
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t a = (e & 1) ? (wa[e >> 1] >> 16) : (wa[e >> 1] & 0xffffu);
          const uint32_t b2 = (e & 1) ? (wb[e >> 1] >> 16) : (wb[e >> 1] & 0xffffu);
          const int d = xdst[u] + e * W3_CS;
          xT[d] = a | (b2 << 16);                                                    // copy 0: element j at j
          xT16[2 * (W3_COPY + d) + 1] = (uint16_t)a;                                   // copy 1: element j at j + 1
          xT16[2 * (W3_COPY + d) + g.two] = (uint16_t)b2;
        }
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < g.ntiles) fetch(tile + gridDim.x);                   // in flight during the MFMAs
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                    // tile row ks >> 1, columns 16 (ks & 1) + 8 half ..
      const bf16x8 fa = *reinterpret_cast<const lds_bf16x8*>(afrag + ks * 8);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int sg = kw & 1;
          const lds_u32* q = bfrag + sg * W3_COPY + ((ks >> 1) + kh) * W3_RD + (ks & 1) * 8 + ((kw + sg) >> 1);
          union { uint32_t u[4]; bf16x8 v; } fb;
          fb.u[0] = q[0]; fb.u[1] = q[1]; fb.u[2] = q[2]; fb.u[3] = q[3];
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb.v, acc[kh * 3 + kw], 0, 0, 0);
        }
    }
  }
  // acc[t][r]: oc = 32 wm + (r & 3) + 8 (r >> 2) + 4 half, column = t * 64 + 32 wh + n
  float* out = part + (int64_t)blockIdx.x * W3_C * W3_N;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oc = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[oc * W3_N + t * W3_C + 32 * wh + n] = acc[t][r];
    }
}

// ---------------------------------------------------------------- variant 2: transposing LDS reads
// Same GEMM, but the tiles stay pixel-major in LDS exactly as they come from HBM (16-byte copies, no unpacking)
// and the K = pixel fragments are produced by gfx950's ds_read_b64_tr_b16.  Measured semantics
// (tools/probes/tr_probe.hip): in a 16-lane group lane i contributes the 4 bf16 at its own 8-byte-aligned address,
// S[i][0..3]; lane l receives S[4 j + (l >> 2)][l & 3], j = 0..3.  Pointing source lane 4 j + r at
// (pixel k0 + j, channels c0 + 4 r ..) therefore hands lane 4 r + c the channel c0 + 4 r + c at pixels k0 .. k0 + 3:
// two reads make one MFMA fragment, and a tap shift is a pixel offset (no alignment copies).
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef v4i16 __attribute__((address_space(3))) lds_v4i16;

constexpr int T3_RBE = 96;                          // bf16 elements per pixel row in LDS: 64 channels + 32 pad (192 B:
                                                    // two 16-lane groups of a transposing read hit 64 distinct banks)
constexpr int T3_PC = W3_TW + 2;                    // 34 patch columns
constexpr int T3_NPX = W3_PR * T3_PC;               // 204 patch pixels
constexpr int T3_DY = W3_TH * W3_TW * T3_RBE;       // elements of the dy tile
constexpr int T3_XU = (T3_NPX * 8 + 255) / 256;     // 7 x chunks (16 B) per thread
constexpr size_t T3_LDS = (size_t)(T3_DY + T3_NPX * T3_RBE) * sizeof(bf16_t);   // 63,744 B

// Normalise-on-load (AFF): x is the input of a BatchNorm + ReLU whose output the convolution consumed (csrc/conv64.hip):
// the x patch is transformed to relu(a x + b) — with tsg_bn_apply_fwd's arithmetic and rounding — while it is staged, so
// the weight gradient sees the activation the forward convolution saw without that activation ever being stored.
__device__ __forceinline__ uint4 w3_affine_relu(uint4 v, const float* __restrict__ ab, int part8) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
  const float4 a0 = *reinterpret_cast<const float4*>(ab + part8), a1 = *reinterpret_cast<const float4*>(ab + part8 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(ab + W3_C + part8), b1 = *reinterpret_cast<const float4*>(ab + W3_C + part8 + 4);
  const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = __uint_as_float(w[i] << 16), x1 = __uint_as_float(w[i] & 0xffff0000u);
    const float y0 = fmaf(x0, a[2 * i], b[2 * i]), y1 = fmaf(x1, a[2 * i + 1], b[2 * i + 1]);
    w[i] = pack2_bf16(y0 > 0.f ? y0 : 0.f, y1 > 0.f ? y1 : 0.f);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

template <bool AFF>
__global__ __launch_bounds__(256) void conv3_wrw_tr_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                      float* __restrict__ part, W3Geom g, const float* __restrict__ in_ab) {
  __shared__ __attribute__((aligned(16))) float abs_[AFF ? 2 * W3_C : 4];
  if (AFF && threadIdx.x < 2 * W3_C) abs_[threadIdx.x] = in_ab[threadIdx.x];   // visible after the tile loop's first barrier
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  bf16_t* dyL = reinterpret_cast<bf16_t*>(lds);       // [128 pixels][T3_RBE]
  bf16_t* xL = dyL + T3_DY;                           // [204 pixels][T3_RBE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wh = wave & 1;            // oc half, ci half
  const int half = lane >> 5, sub = (lane >> 4) & 1, i16 = lane & 15;

  // staging: chunk = (pixel, 8-channel part); thread -> part tid & 7, pixel (tid >> 3) + 32 u
  const int spart = tid & 7, spix = tid >> 3;
  int xr[T3_XU], xc[T3_XU];                           // patch row / column of x chunk u (row < 0: none)
#pragma unroll
  for (int u = 0; u < T3_XU; ++u) {
    const int pp = spix + 32 * u;
    xr[u] = pp < T3_NPX ? pp / T3_PC : -1;
    xc[u] = pp % T3_PC;
  }
  // fragment bases (elements): row offset of this lane's source pixel + its 4-channel group
  const int fbase = (8 * half + (i16 >> 2)) * T3_RBE + 16 * sub + 4 * (i16 & 3);
  const lds_v4i16* afr = (const lds_v4i16*)(dyL + fbase + 32 * wm);
  const lds_v4i16* bfr = (const lds_v4i16*)(xL + fbase + 32 * wh);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  uint4 rd[W3_TH];
  uint4 rx[T3_XU];
  auto fetch = [&](int tile) {
    const int ow0 = (tile % g.tiles_w) * W3_TW, oh0 = ((tile / g.tiles_w) % g.tiles_h) * W3_TH;
    const int b = tile / (g.tiles_w * g.tiles_h);
    const int64_t org = (((int64_t)b * g.H + oh0) * g.W + ow0) * W3_C;
    const bf16_t* dt = dy + org + (int64_t)spix * W3_C + spart * 8;
#pragma unroll
    for (int u = 0; u < W3_TH; ++u)
      rd[u] = (oh0 + u < g.H && ow0 + spix < g.W) ? *reinterpret_cast<const uint4*>(dt + (int64_t)u * g.W * W3_C)
                                                  : make_uint4(0, 0, 0, 0);
    const bf16_t* xo = x + org + spart * 8;
#pragma unroll
    for (int u = 0; u < T3_XU; ++u) {
      const int ih = oh0 - 1 + xr[u], iw = ow0 - 1 + xc[u];
      rx[u] = (xr[u] >= 0 && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
                  ? *reinterpret_cast<const uint4*>(xo + ((int64_t)(xr[u] - 1) * g.W + xc[u] - 1) * W3_C)
                  : make_uint4(0, 0, 0, 0);
    }
  };

  int tile = blockIdx.x;
  if (tile < g.ntiles) fetch(tile);
  for (; tile < g.ntiles; tile += gridDim.x) {
    __syncthreads();                                  // the previous tile's fragment reads are done
#pragma unroll
    for (int u = 0; u < W3_TH; ++u)
      *reinterpret_cast<uint4*>(dyL + (spix + 32 * u) * T3_RBE + spart * 8) = rd[u];
    const int cur_ow0 = (tile % g.tiles_w) * W3_TW, cur_oh0 = ((tile / g.tiles_w) % g.tiles_h) * W3_TH;
#pragma unroll
    for (int u = 0; u < T3_XU; ++u)
      if (u < T3_XU - 1 || xr[u] >= 0) {
        uint4 v = rx[u];
        if (AFF) {
          const int ih = cur_oh0 - 1 + xr[u], iw = cur_ow0 - 1 + xc[u];
          if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) v = w3_affine_relu(v, abs_, spart * 8);
        }
        *reinterpret_cast<uint4*>(xL + (spix + 32 * u) * T3_RBE + spart * 8) = v;
      }
    __syncthreads();
    if (tile + (int)gridDim.x < g.ntiles) fetch(tile + gridDim.x);       // in flight during the MFMAs
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                  // 16 pixels: tile row ks >> 1, columns 16 (ks & 1) + 8 half ..
      union { v4i16 q[2]; bf16x8 v; } fa;
      fa.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(afr + (ks * 16 + 0) * (T3_RBE / 4)));
      fa.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(afr + (ks * 16 + 4) * (T3_RBE / 4)));
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int px = ((ks >> 1) + kh) * T3_PC + (ks & 1) * 16 + kw;   // patch pixel of the fragment's first K
          union { v4i16 q[2]; bf16x8 v; } fb;
          fb.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(bfr + (px + 0) * (T3_RBE / 4)));
          fb.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(bfr + (px + 4) * (T3_RBE / 4)));
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.v, fb.v, acc[kh * 3 + kw], 0, 0, 0);
        }
    }
  }
  // acc[t][r]: oc = 32 wm + (r & 3) + 8 (r >> 2) + 4 half, column = t * 64 + 32 wh + (lane & 31)
  float* out = part + (int64_t)blockIdx.x * W3_C * W3_N;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oc = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[oc * W3_N + t * W3_C + 32 * wh + (lane & 31)] = acc[t][r];
    }
}

// ---------------------------------------------------------------- variant 3: any C_in, C_out that are multiples of 64
// The same kernel body over (oc tile, ci tile) PAIRS of 64 x 64 channels: a block owns one pair and every bpp-th pixel
// tile, reads the 64-channel slices of dy and x it needs (pixel strides C_out / C_in instead of 64) and writes one
// [64 x 576] partial.  128 -> 128 at 128^2 is 4 pairs, 512 -> 512 at 32^2 is 64 pairs: every operand slice is read by
// C_out / 64 (x) or C_in / 64 (dy) blocks -- 268 MB of (Infinity-Cache resident) traffic for each of the 77-GFLOP layers
// of ResNet-18, against ~85 us of MFMA time, so the kernel stays on the MFMA side of its roofline exactly like the
// 64 -> 64 case it was built for.  (furnace/base_model/resnet.py:24-29 every BasicBlock conv3x3 with stride 1; bisenet
// network.py:43-52 refines / arms, :140-156 head conv_3x3.)
struct W3GenGeom {
  int B, H, W, tiles_h, tiles_w, ntiles;   // H, W: OUTPUT size (= size of dy)
  int Hin, Win;                            // input size (= H, W for stride 1)
  int Cin, Cout, nci, npairs, bpp;         // channel counts, ci tiles, (oc tile, ci tile) pairs, blocks per pair
  int xs;                                  // XCDs that share one slot's pairs: 1 (bpp % 8 == 0) or 2 (bpp % 4 == 0)
  int prio;                                // s_setprio 1 around the MFMAs of a tile (TSG_MFMA_PRIO)
};

// stride S in {1, 2}: the x patch of a 4 x 32 output tile is (4 S + 3 - S) x (32 S + 3 - S) input pixels (6 x 34 / 9 x 65),
// still pixel-major in LDS; the K = output-pixel fragments of tap (kh, kw) start at patch pixel (S row + kh, S col + kw)
// and step S pixels per K index -- every lane of a transposing read supplies its own address, so a stride is free.
template <int S> struct W3S {
  static constexpr int PR = S * W3_TH + 3 - S, PC = S * W3_TW + 3 - S, NPX = PR * PC;
  static constexpr int XU = (NPX * 8 + 255) / 256;                                  // 16-byte x chunks per thread: 7 / 19
  static constexpr size_t LDS = (size_t)(T3_DY + NPX * T3_RBE) * sizeof(bf16_t);    // 63,744 B / 136,896 B
};

// PF = 1: one block per CU, the next tile's operands are prefetched into registers while the MFMAs run.
// PF = 0: no register prefetch (the 44 registers it costs go), two blocks per CU that cover for each other's loads,
// staging and barriers — with one block per CU the MFMA pipe idles ~2/3 of the time during those phases.
// PF = 2 (round 3): one block per CU, TWO tiles ahead by untracked loads (tsg_common.h: TSG_ASM_LD16 / vm_wait).  A tile
// is 72 MFMAs per wave = ~1 us and the measured tile time 2.1 us (128 -> 128 at 64^2: 32 tiles per block in 68 us, MFMA
// pipe busy 0.36-0.45), which looked like one tile of prefetch against a ~2 us load latency -- but with two tiles in flight
// the kernel is SLOWER (opt-in, see w3_pf2()).  The loads are unconditional (a position outside the image reads the tensor's first element
// and is zeroed when the set is written to LDS; a tile index beyond the block's last one re-reads the last one), so the
// number of loads in flight behind the set being consumed is the constant the s_waitcnt needs.
// BUF (round 3; default, TSG_CONV_WRW_BUF=0 selects the pointer loads): the fetch as raw buffer loads.  The ISA of the pointer version spends 293 instructions per tile and wave on the 11 loads
// (64-bit address arithmetic, one divergent branch per predicated load) before the tile's first MFMA can issue.  Here a
// thread's byte offsets are tile-independent 32-bit constants computed once, the tile contributes a scalar base (the
// buffer descriptor), and a position outside the image gets the offset 0x80000000, which the buffer unit answers with
// zeros (num_records 0x7fffffff): no branches, no 64-bit VALU, one basic block the scheduler places under the MFMAs
// (175 -> ~125 instructions, interleaved with the tile's MFMAs in the ISA).  Bit-equal to the pointer version on the
// bench's layers, ragged sizes, stride 2 and BN-on-load; 109 -> 90 us (256 -> 256 at 64^2), 138 -> 104 us (512 -> 512 at
// 32^2), 77 -> 61 us (stride 2, 128 -> 256 at 128^2), 88 -> 86 us (two blocks per CU): profiles/r03_conv3wrw_buffer_fetch.txt.
template <int S, int PF, bool AFF, bool BUF = false>
__global__ __launch_bounds__(256, PF ? 1 : 2) void conv3_wrw_gen_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                       float* __restrict__ part, W3GenGeom g, const float* __restrict__ in_ab) {
  typedef W3S<S> P;
  __shared__ __attribute__((aligned(16))) float abs_[AFF ? 2 * W3_C : 4];
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  bf16_t* dyL = reinterpret_cast<bf16_t*>(lds);       // [128 pixels][T3_RBE]
  bf16_t* xL = dyL + T3_DY;                           // [P::NPX pixels][T3_RBE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wh = wave & 1;            // oc half, ci half of the 64 x 64 pair
  const int half = lane >> 5, sub = (lane >> 4) & 1, i16 = lane & 15;
  // XCD-aware mapping (round 3).  Consecutive block ids go round-robin over the 8 XCDs (own L2 each).  With
  // pair = id % npairs the npairs blocks that stream the SAME pixel tiles (one per (oc tile, ci tile) pair: each x tile is
  // wanted by C_out / 64 of them, each dy tile by C_in / 64) landed on different XCDs, so every one of them fetched its
  // operands from HBM: 2.01x the algorithmic traffic over the step's 25 launches (profiles/traffic.json, round 2).  Now the
  // pairs of a slot are consecutive ids on ONE XCD: they run at the same time on the same tiles and the re-reads are
  // L2 hits.  bpp is a multiple of 8 (w3gen_geom).
  // xs == 2 (round 4; >= 64 pairs, i.e. 512 -> 512): a slot's pairs are split over TWO XCDs, so that 4 slots x 64 pairs
  // = 256 blocks fill the chip in ONE round with one resident block per CU (8 slots were 512 blocks = two rounds of 16
  // tiles each, and twice the partials); each XCD then streams every x slice and half of the dy slices of its slot.
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int pps = g.npairs / g.xs;                    // pairs per XCD of a slot
  const int pair = (g.xs == 2 ? (xcd & 1) * pps : 0) + jb % pps;
  const int slot = (jb / pps) * (8 / g.xs) + (g.xs == 2 ? xcd >> 1 : xcd);
  const int oc0 = (pair / g.nci) * W3_C, ci0 = (pair % g.nci) * W3_C;
  if (AFF && tid < 2 * W3_C) abs_[tid] = in_ab[(tid >> 6) * g.Cin + ci0 + (tid & 63)];   // a / b rows of this ci tile

  const int spart = tid & 7, spix = tid >> 3;
  int xr[P::XU], xc[P::XU];
#pragma unroll
  for (int u = 0; u < P::XU; ++u) {
    const int pp = spix + 32 * u;
    xr[u] = pp < P::NPX ? pp / P::PC : -1;
    xc[u] = pp % P::PC;
  }
  const int chan = 16 * sub + 4 * (i16 & 3);
  const lds_v4i16* afr = (const lds_v4i16*)(dyL + (8 * half + (i16 >> 2)) * T3_RBE + chan + 32 * wm);
  const lds_v4i16* bfr = (const lds_v4i16*)(xL + S * (8 * half + (i16 >> 2)) * T3_RBE + chan + 32 * wh);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  uint4 rd[W3_TH];
  uint4 rx[P::XU];
  // BUF: tile-independent byte offsets of this thread's chunks (relative to the tile's first dy pixel / patch pixel (0, 0))
  uint32_t dyo[BUF ? W3_TH : 1], xo32[BUF ? P::XU : 1];
  if (BUF) {
#pragma unroll
    for (int u = 0; u < W3_TH; ++u) dyo[BUF ? u : 0] = (uint32_t)((((int64_t)u * g.W + spix) * g.Cout + spart * 8) * 2);
#pragma unroll
    for (int u = 0; u < P::XU; ++u)
      xo32[BUF ? u : 0] = (uint32_t)((((int64_t)(xr[u] < 0 ? 0 : xr[u]) * g.Win + xc[u]) * g.Cin + spart * 8) * 2);
  }
  // BUF: the fetched tiles are slot, slot + bpp, ...: their (image, tile row, tile column) advance by constant steps with
  // carries instead of three integer divisions per tile (~60 scalar instructions of the fetch)
  int ftw = 0, fth = 0, fb = 0;
  const int dtw = g.bpp % g.tiles_w, dth = (g.bpp / g.tiles_w) % g.tiles_h, db = g.bpp / (g.tiles_w * g.tiles_h);
  if (BUF) { ftw = slot % g.tiles_w; fth = (slot / g.tiles_w) % g.tiles_h; fb = slot / (g.tiles_w * g.tiles_h); }
  auto fetch = [&](int tile) {
    const int ow0 = BUF ? ftw * W3_TW : (tile % g.tiles_w) * W3_TW;
    const int oh0 = BUF ? fth * W3_TH : ((tile / g.tiles_w) % g.tiles_h) * W3_TH;
    const int b = BUF ? fb : tile / (g.tiles_w * g.tiles_h);
    const int64_t pix = ((int64_t)b * g.H + oh0) * g.W + ow0;
    if (BUF) {
      constexpr uint32_t kOob = 0x80000000u;             // >= num_records: the buffer unit returns zeros
      const int ih0 = S * oh0 - 1, iw0 = S * ow0 - 1;
      const bf16_t* dbase = dy + pix * g.Cout + oc0;     // uniform: descriptor bases of this tile
      const bf16_t* xbase = x + (((int64_t)b * g.Hin + ih0) * g.Win + iw0) * g.Cin + ci0;
      const __amdgpu_buffer_rsrc_t rd_ = __builtin_amdgcn_make_buffer_rsrc((void*)dbase, 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rx_ = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, 0x7fffffff, 0x00020000);
      const bool live = tile < g.ntiles;                 // beyond the block's last tile every lane is out of bounds: no access
      const bool colok = live && ow0 + spix < g.W;
#pragma unroll
      for (int u = 0; u < W3_TH; ++u) {
        const uint32_t off = (colok && oh0 + u < g.H) ? dyo[BUF ? u : 0] : kOob;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rd_, (int)off, 0, 0);
        rd[u] = make_uint4(v.x, v.y, v.z, v.w);
      }
#pragma unroll
      for (int u = 0; u < P::XU; ++u) {
        // 0 <= ih0 + xr < Hin and 0 <= iw0 + xc < Win as two unsigned compares (xr = -1 marks a thread beyond the patch)
        const bool ok = live && (uint32_t)(ih0 + xr[u]) < (uint32_t)g.Hin && (uint32_t)(iw0 + xc[u]) < (uint32_t)g.Win && xr[u] >= 0;
        const uint32_t off = ok ? xo32[BUF ? u : 0] : kOob;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx_, (int)off, 0, 0);
        rx[u] = make_uint4(v.x, v.y, v.z, v.w);
      }
      // the next tile of this block
      ftw += dtw; if (ftw >= g.tiles_w) { ftw -= g.tiles_w; ++fth; }
      fth += dth; if (fth >= g.tiles_h) { fth -= g.tiles_h; ++fb; }
      fb += db;
      return;
    }
    const bf16_t* dt = dy + (pix + spix) * g.Cout + oc0 + spart * 8;
#pragma unroll
    for (int u = 0; u < W3_TH; ++u)
      rd[u] = (oh0 + u < g.H && ow0 + spix < g.W) ? *reinterpret_cast<const uint4*>(dt + (int64_t)u * g.W * g.Cout)
                                                  : make_uint4(0, 0, 0, 0);
    const int ih0 = S * oh0 - 1, iw0 = S * ow0 - 1;                       // input pixel of patch (0, 0)
    const bf16_t* xo = x + (((int64_t)b * g.Hin + ih0) * g.Win + iw0) * g.Cin + ci0 + spart * 8;
#pragma unroll
    for (int u = 0; u < P::XU; ++u) {
      const int ih = ih0 + xr[u], iw = iw0 + xc[u];
      rx[u] = (xr[u] >= 0 && ih >= 0 && ih < g.Hin && iw >= 0 && iw < g.Win)
                  ? *reinterpret_cast<const uint4*>(xo + ((int64_t)xr[u] * g.Win + xc[u]) * g.Cin)
                  : make_uint4(0, 0, 0, 0);
    }
  };
  // one tile's operands, registers -> LDS.  REZERO (PF = 2): the loads were unconditional, positions outside the image
  // become the exact zeros the predicated loads of PF 0 / 1 return
  auto stage = [&](const uint4* qd_, const uint4* qx_, int tile_, bool rezero) {
    const int cur_ow0 = (tile_ % g.tiles_w) * W3_TW, cur_oh0 = ((tile_ / g.tiles_w) % g.tiles_h) * W3_TH;
#pragma unroll
    for (int u = 0; u < W3_TH; ++u) {
      uint4 v = qd_[u];
      if (rezero && !(cur_oh0 + u < g.H && cur_ow0 + spix < g.W)) v = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(dyL + (spix + 32 * u) * T3_RBE + spart * 8) = v;
    }
#pragma unroll
    for (int u = 0; u < P::XU; ++u)
      if (u < P::XU - 1 || xr[u] >= 0) {
        uint4 v = qx_[u];
        const int ih = S * cur_oh0 - 1 + xr[u], iw = S * cur_ow0 - 1 + xc[u];
        const bool in = ih >= 0 && ih < g.Hin && iw >= 0 && iw < g.Win;
        if (AFF && in) v = w3_affine_relu(v, abs_, spart * 8);
        if (rezero && !in) v = make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(xL + (spix + 32 * u) * T3_RBE + spart * 8) = v;
      }
  };
  union Frag { v4i16 q[2]; bf16x8 v; };
  // the MFMAs of the tile in LDS.  dy fragments of the 8 k steps (16 output pixels each: tile row ks >> 1, columns
  // 16 (ks & 1) + 8 half ..) stay in registers; the x fragments are read once per PATCH row: the fragment tap kh of tile
  // row r needs (patch row S r + kh, columns S 16 c + kw ..) is the one tap kh - S of tile row r + 1 needs, so walking
  // patch rows instead of (row, kh) pairs reads 36 (stride 1) / 54 (stride 2) fragments per tile instead of 72 (neutral
  // in the step: this loop is not bound by LDS reads, profiles/r03_conv3wrw_shared_fragments.txt)
  auto mfma_tile = [&]() {
    if (g.prio) __builtin_amdgcn_s_setprio(1);
    Frag fa[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      fa[ks].q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(afr + (ks * 16 + 0) * (T3_RBE / 4)));
      fa[ks].q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(afr + (ks * 16 + 4) * (T3_RBE / 4)));
    }
#pragma unroll
    for (int pr = 0; pr < S * 3 + 3; ++pr)            // patch rows S r + kh, r < 4, kh < 3
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int px = pr * P::PC + S * c * 16 + kw;       // patch pixel of the fragment's first K
          Frag fb;
          fb.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(bfr + (px + 0) * (T3_RBE / 4)));
          fb.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(bfr + (px + 4 * S) * (T3_RBE / 4)));
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const int rr = pr - kh;
            if (rr >= 0 && rr % S == 0 && rr / S < 4)
              acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2 * (rr / S) + c].v, fb.v, acc[kh * 3 + kw], 0, 0, 0);
          }
        }
    if (g.prio) __builtin_amdgcn_s_setprio(0);
  };

  if (PF == 2) {
    constexpr int L = W3_TH + P::XU;                    // loads per set
    u32x4 qd[PF == 2 ? 2 : 1][W3_TH], qx[PF == 2 ? 2 : 1][PF == 2 ? P::XU : 1];
    // the block's last tile: what indices beyond it re-read
    const int last = slot < g.ntiles ? slot + (g.ntiles - 1 - slot) / g.bpp * g.bpp : 0;
#define TSG_W3_FETCH(SET, TILE)                                                                               \
    {                                                                                                         \
      const int t_ = (TILE) < g.ntiles ? (TILE) : last;                                                       \
      const int ow0_ = (t_ % g.tiles_w) * W3_TW, oh0_ = ((t_ / g.tiles_w) % g.tiles_h) * W3_TH;               \
      const int b_ = t_ / (g.tiles_w * g.tiles_h);                                                            \
      const int64_t pix_ = ((int64_t)b_ * g.H + oh0_) * g.W + ow0_;                                           \
      const bf16_t* dt_ = dy + (pix_ + spix) * g.Cout + oc0 + spart * 8;                                      \
      _Pragma("unroll") for (int u_ = 0; u_ < W3_TH; ++u_) {                                                  \
        const bf16_t* p_ = (oh0_ + u_ < g.H && ow0_ + spix < g.W) ? dt_ + (int64_t)u_ * g.W * g.Cout : dy;    \
        TSG_ASM_LD16(qd[SET][u_], p_, 0);                                                                     \
      }                                                                                                       \
      const int ih0_ = S * oh0_ - 1, iw0_ = S * ow0_ - 1;                                                     \
      const bf16_t* xo_ = x + (((int64_t)b_ * g.Hin + ih0_) * g.Win + iw0_) * g.Cin + ci0 + spart * 8;        \
      _Pragma("unroll") for (int u_ = 0; u_ < P::XU; ++u_) {                                                  \
        const int ih_ = ih0_ + xr[u_], iw_ = iw0_ + xc[u_];                                                   \
        const bf16_t* p_ = (xr[u_] >= 0 && ih_ >= 0 && ih_ < g.Hin && iw_ >= 0 && iw_ < g.Win)                \
                               ? xo_ + ((int64_t)xr[u_] * g.Win + xc[u_]) * g.Cin : x;                        \
        TSG_ASM_LD16(qx[SET][PF == 2 ? u_ : 0], p_, 0);                                                       \
      }                                                                                                       \
    }
    int tile = slot;
    if (tile < g.ntiles) {
      TSG_W3_FETCH(0, tile)
      TSG_W3_FETCH(1, tile + g.bpp)
      for (bool more = true; more;) {
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          __syncthreads();                               // the previous tile's fragment reads are done
          vm_wait<L>(qd[PF == 2 ? s_ : 0][0]);           // all but the other set's L loads have landed
#pragma unroll
          for (int u = 1; u < W3_TH; ++u) vm_tie(qd[PF == 2 ? s_ : 0][u]);
#pragma unroll
          for (int u = 0; u < P::XU; ++u) vm_tie(qx[PF == 2 ? s_ : 0][PF == 2 ? u : 0]);
          uint4 td[W3_TH], tx[P::XU];
#pragma unroll
          for (int u = 0; u < W3_TH; ++u) { const u32x4 v = qd[PF == 2 ? s_ : 0][u]; td[u] = make_uint4(v.x, v.y, v.z, v.w); }
#pragma unroll
          for (int u = 0; u < P::XU; ++u) { const u32x4 v = qx[PF == 2 ? s_ : 0][PF == 2 ? u : 0]; tx[u] = make_uint4(v.x, v.y, v.z, v.w); }
          stage(td, tx, tile, true);
          __syncthreads();
          TSG_W3_FETCH(PF == 2 ? s_ : 0, tile + 2 * g.bpp)   // into the set just written out; re-reads `last` at the tail
          mfma_tile();
          tile += g.bpp;
          if (tile >= g.ntiles) { more = false; break; }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)");                // the tail's re-reads still target qd / qx
    }
#undef TSG_W3_FETCH
  } else {
    int tile = slot;
    if (PF == 1 && tile < g.ntiles) fetch(tile);
    for (; tile < g.ntiles; tile += g.bpp) {
      if (PF == 0) fetch(tile);
      __syncthreads();
      stage(rd, rx, tile, false);
      __syncthreads();
      // BUF: issued unconditionally (beyond the last tile all of its lanes are out of bounds), so that the fetch and the
      // MFMAs are ONE basic block and the scheduler may issue the scalar / address work in the shadow of the MFMAs
      if (PF == 1 && (BUF || tile + g.bpp < g.ntiles)) fetch(tile + g.bpp);
      mfma_tile();
    }
  }
  // partial of this (pair, slot): [pair][slot][64 oc][9 taps][64 ci]
  float* out = part + ((int64_t)pair * g.bpp + slot) * W3_C * W3_N;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int oc = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
      out[oc * W3_N + t * W3_C + 32 * wh + (lane & 31)] = acc[t][r];
    }
}

// dw[oc0 + oc][tap][ci0 + 0..63] = sum over the pair's bpp partials (fixed order, fp64).  Block = one run of 64 ci.
__global__ __launch_bounds__(256) void conv3_wrw_gen_fold(const float* __restrict__ part, W3GenGeom g,
                                                          float* __restrict__ dw) {
  __shared__ double sm[16][64];
  const int run = blockIdx.x % (W3_C * 9), pair = blockIdx.x / (W3_C * 9);       // run = oc * 9 + tap inside the pair
  const int oc0 = (pair / g.nci) * W3_C, ci0 = (pair % g.nci) * W3_C;
  const int c4 = threadIdx.x & 15, gs = threadIdx.x >> 4;
  const float* src = part + (int64_t)pair * g.bpp * W3_C * W3_N + run * 64 + c4 * 4;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int s0 = gs; s0 < g.bpp; s0 += 16 * 8) {             // 8 predicated loads in flight, summed in order
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + 16 * u;
      v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(s < g.bpp ? s : gs) * W3_C * W3_N);
      if (s >= g.bpp) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
  }
  sm[gs][c4 * 4 + 0] = a0; sm[gs][c4 * 4 + 1] = a1; sm[gs][c4 * 4 + 2] = a2; sm[gs][c4 * 4 + 3] = a3;
  __syncthreads();
  if (threadIdx.x < 64) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sm[q][threadIdx.x];
    const int oc = run / 9, tap = run % 9;
    dw[((int64_t)(oc0 + oc) * 9 + tap) * g.Cin + ci0 + threadIdx.x] = (float)t;
  }
}

// Round 6: the same fold with the row lanes sized to the pair's slot count.  conv3_wrw_gen_fold gives every run of 64 entries
// 16 row lanes x 8 loads: right for 256 slots (layer1), but a 512 -> 512 layer has 4 slots per pair (64 pairs) — 124 of its
// 128 loads per 16 lanes were predicated re-reads of a neighbour's rows — and 128 -> 128 has 64.  Here T = 1 .. 16 row lanes
// (<= 8 rows each up to 128 slots) share one float4 of the output; lane t sums rows t, t + T, .. in row order (fp64), the T
// sums are folded in lane order: fixed order, bit-reproducible (the ORDER differs from conv3_wrw_gen_fold's, so the last
// bits of dw do too).  13.3 -> ~8 us per launch x 25 launches.
__global__ __launch_bounds__(256) void conv3_wrw_gen_fold2(const float* __restrict__ part, W3GenGeom g, int T,
                                                           float* __restrict__ dw) {
  __shared__ double sm[256][4];
  const int outs = 256 / T;                                  // float4 outputs per block
  const int t = threadIdx.x / outs, o = threadIdx.x - t * outs;
  constexpr int QP = W3_C * W3_N / 4;                        // 9216 float4 per pair
  const int64_t q_all = (int64_t)blockIdx.x * outs + o;
  const int pair = (int)(q_all / QP), q = (int)(q_all - (int64_t)pair * QP);
  const float* src = part + (int64_t)pair * g.bpp * W3_C * W3_N + (int64_t)q * 4;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int s0 = t; s0 < g.bpp; s0 += 8 * T) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int s = s0 + u * T;
      v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(s < g.bpp ? s : t) * W3_C * W3_N);
      if (s >= g.bpp) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
  }
  if (T > 1) {
    sm[threadIdx.x][0] = a0; sm[threadIdx.x][1] = a1; sm[threadIdx.x][2] = a2; sm[threadIdx.x][3] = a3;
    __syncthreads();
    if (t == 0) {
      for (int k = 1; k < T; ++k) {
        const double* r = sm[k * outs + o];
        a0 += r[0]; a1 += r[1]; a2 += r[2]; a3 += r[3];
      }
    }
  }
  if (t == 0) {
    const int e = q * 4, oc = e / W3_N, rem = e - oc * W3_N, tap = rem / W3_C, ci = rem - tap * W3_C;
    const int oc0 = (pair / g.nci) * W3_C, ci0 = (pair % g.nci) * W3_C;
    *reinterpret_cast<float4*>(dw + ((int64_t)(oc0 + oc) * 9 + tap) * g.Cin + ci0 + ci) =
        make_float4((float)a0, (float)a1, (float)a2, (float)a3);
  }
}

// dw[flat] = sum over the per-block partials (fixed order, fp64); 64 consecutive entries per block
__global__ __launch_bounds__(256) void conv3_wrw_fold(const float* __restrict__ part, int nparts, int64_t stride,
                                                      float* __restrict__ dw) {
  __shared__ double sm[16][64];
  const int c4 = threadIdx.x & 15, gs = threadIdx.x >> 4;
  const float* src = part + blockIdx.x * 64 + c4 * 4;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  for (int s0 = gs; s0 < nparts; s0 += 16 * 8) {            // 8 predicated loads in flight, summed in order
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int gidx = s0 + 16 * u;
      v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(gidx < nparts ? gidx : gs) * stride);
      if (gidx >= nparts) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { a0 += (double)v[u].x; a1 += (double)v[u].y; a2 += (double)v[u].z; a3 += (double)v[u].w; }
  }
  sm[gs][c4 * 4 + 0] = a0; sm[gs][c4 * 4 + 1] = a1; sm[gs][c4 * 4 + 2] = a2; sm[gs][c4 * 4 + 3] = a3;
  __syncthreads();
  if (threadIdx.x < 64) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sm[q][threadIdx.x];
    dw[blockIdx.x * 64 + threadIdx.x] = (float)t;
  }
}

constexpr size_t W3_LDS = (size_t)(W3_C * W3_DS + 2 * W3_COPY) * sizeof(uint32_t) + 16;   // 79,376 B

}  // namespace tsg

using namespace tsg;

extern "C" {

int tsg_conv3x3_wrw_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad, int dilation,
                              int groups) {
  return dtype == TSG_BF16 && Cin == W3_C && Cout == W3_C && kh == 3 && kw == 3 && stride == 1 && pad == 1 &&
         dilation == 1 && groups == 1;
}

size_t tsg_conv3x3_wrw_ws_bytes(void) { return (size_t)W3_NPART * W3_C * W3_N * sizeof(float); }

static int conv3_wrw_common(int variant, const void* x, const void* dy, float* dw, int64_t B, int64_t H, int64_t W,
                            void* ws, size_t ws_bytes, void* stream, const float* in_ab = nullptr) {
  if (!x || !dy || !dw || !ws) return TSG_E_NULL;
  if (in_ab && (variant != 1 || !aligned16(in_ab))) return in_ab && variant != 1 ? TSG_E_SHAPE : TSG_E_ALIGN;
  if (B <= 0 || H <= 0 || W <= 0) return TSG_E_SHAPE;
  const int64_t th = (H + W3_TH - 1) / W3_TH, tw = (W + W3_TW - 1) / W3_TW;
  if (B * th * tw > 0x7fffffffLL || H * W * W3_C > 0x7fffffffLL) return TSG_E_SHAPE;
  if (ws_bytes < tsg_conv3x3_wrw_ws_bytes()) return TSG_E_WS;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dw) || !aligned16(ws)) return TSG_E_ALIGN;
  W3Geom g;
  g.B = (int)B; g.H = (int)H; g.W = (int)W; g.tiles_h = (int)th; g.tiles_w = (int)tw; g.ntiles = (int)(B * th * tw); g.two = 2;
  hipStream_t st = (hipStream_t)stream;
  TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wrw_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)W3_LDS));
  const int grid = g.ntiles < W3_NPART ? g.ntiles : W3_NPART;
  if (variant == 1 && in_ab) {
    TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wrw_tr_k<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS));
    hipLaunchKernelGGL(conv3_wrw_tr_k<true>, dim3(grid), dim3(256), T3_LDS, st, (const bf16_t*)x, (const bf16_t*)dy,
                       (float*)ws, g, in_ab);
  } else if (variant == 1) {
    TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wrw_tr_k<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)T3_LDS));
    hipLaunchKernelGGL(conv3_wrw_tr_k<false>, dim3(grid), dim3(256), T3_LDS, st, (const bf16_t*)x, (const bf16_t*)dy,
                       (float*)ws, g, (const float*)nullptr);
  } else {
    hipLaunchKernelGGL(conv3_wrw_k, dim3(grid), dim3(256), W3_LDS, st, (const bf16_t*)x, (const bf16_t*)dy, (float*)ws,
                       g);
  }
  TSG_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv3_wrw_fold, dim3(W3_C * W3_N / 64), dim3(256), 0, st, (const float*)ws, grid,
                     (int64_t)W3_C * W3_N, dw);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_conv3x3_wrw(const void* x, const void* dy, float* dw, int64_t B, int64_t H, int64_t W, void* ws,
                    size_t ws_bytes, void* stream) {
  return conv3_wrw_common(0, x, dy, dw, B, H, W, ws, ws_bytes, stream);
}

int tsg_conv3x3_wrw_tr(const void* x, const void* dy, float* dw, int64_t B, int64_t H, int64_t W, void* ws,
                       size_t ws_bytes, void* stream) {
  return conv3_wrw_common(1, x, dy, dw, B, H, W, ws, ws_bytes, stream);
}

int tsg_conv3x3_wrw_tr_norm(const void* x, const float* in_ab, const void* dy, float* dw, int64_t B, int64_t H, int64_t W,
                            void* ws, size_t ws_bytes, void* stream) {
  if (!in_ab) return TSG_E_NULL;
  return conv3_wrw_common(1, x, dy, dw, B, H, W, ws, ws_bytes, stream, in_ab);
}

// TSG_CONV_WRW_OCC=1 (default since round 6) | 2: with 2, stride-1 layers with >= 1024 pixel tiles run the two-blocks-per-CU
// variant of the kernel (no register prefetch).  tools/bench_conv3wrw.py, us per layer ALONE, 1 -> 2: layer2 97 -> 88, layer3
// 107 -> 87, layer4 127 -> 104, head1 193 -> 156 (0.99 PF), head2 99 -> 89; the 64^2 / 32^2 maps (<= 512 tiles) are a few us
// better with one prefetching block per CU and keep it.  IN THE STEP the order is the other way round (round 6, interleaved
// pairs on two boxes, profiles/r06_wrw_one_block_per_cu_in_the_step.txt): one block per CU writes 256 partials instead of
// 512 (37.7 instead of 75 MB per launch, written and read again by the fold — and whoever runs next pays for a writer's
// write-backs) and leaves half of every CU's wave slots to the kernel of the other queue: one graph 12.40 ->
// 12.36 / 12.82 -> 12.72 ms, sixteen graphs 11.84 -> 11.80 / 12.16 -> 12.09 ms, eager with the side stream 12.26 -> 12.08 ms.
// TSG_CONV_WRW_PF2=1 (opt-in): the stride-1 layers on the two-tiles-ahead kernel, one block per CU.  Measured SLOWER
// (profiles/r03_conv3wrw_two_tiles_ahead.txt: 2.76 vs 2.37 ms per step, 1100 vs 1125 img/s on one box; layer3 at the same
// occupancy 123 vs 105 us): the tile time is not the load latency the 2.1 us per tile suggested
static bool w3_pf2() {
  static const bool v = [] { const char* e = getenv("TSG_CONV_WRW_PF2"); return e && e[0] == '1'; }();
  return v;
}
static bool w3_occ2(int64_t ntiles) {
  static const int v = [] { const char* e = getenv("TSG_CONV_WRW_OCC"); return e ? atoi(e) : 1; }();
  return !w3_pf2() && v == 2 && ntiles >= 1024;
}

static int w3gen_geom(W3GenGeom* g, int64_t B, int64_t Hin, int64_t Win, int Cin, int Cout, int stride) {
  if (B <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0 || Cin % W3_C || Cout % W3_C) return TSG_E_SHAPE;
  if (stride != 1 && stride != 2) return TSG_E_SHAPE;
  const int64_t H = (Hin - 1) / stride + 1, W = (Win - 1) / stride + 1;       // 3x3, padding 1
  const int64_t th = (H + W3_TH - 1) / W3_TH, tw = (W + W3_TW - 1) / W3_TW;
  if (B * th * tw > 0x7fffffffLL || B * Hin * Win * (int64_t)(Cin > Cout ? Cin : Cout) > 0x7fffffff00LL) return TSG_E_SHAPE;
  g->B = (int)B; g->H = (int)H; g->W = (int)W; g->Hin = (int)Hin; g->Win = (int)Win;
  g->tiles_h = (int)th; g->tiles_w = (int)tw; g->ntiles = (int)(B * th * tw);
  g->Cin = Cin; g->Cout = Cout; g->nci = Cin / W3_C; g->npairs = (Cout / W3_C) * g->nci;
  // one block per CU at a time (300 registers per lane): a second round of blocks would only add partials to fold
  // (measured: 256 blocks 826 img/s, 512 blocks 815, 128 blocks 789)
  static int target = 0;
  if (!target) { const char* e = getenv("TSG_CONV_WRW_BLOCKS"); target = e ? atoi(e) : 256; if (target < 1) target = 256; }
  int bpp = ((stride == 1 && w3_occ2(g->ntiles) ? 2 * target : target) + g->npairs - 1) / g->npairs;
  if (bpp > g->ntiles) bpp = g->ntiles;
  if (bpp < 1) bpp = 1;
  // slots per pair: a multiple of 8 (one XCD per slot residue), or of 4 with a slot's pairs on two XCDs when 8 slots
  // would be more blocks than CUs (TSG_CONV_WRW_XS2=0 restores the round-3 mapping)
  static const bool xs2 = [] { const char* e = getenv("TSG_CONV_WRW_XS2"); return !(e && e[0] == '0'); }();
  g->xs = (xs2 && bpp <= 4 && g->npairs % 2 == 0 && g->npairs * 8 > target) ? 2 : 1;
  g->bpp = g->xs == 2 ? (bpp + 3) / 4 * 4 : (bpp + 7) / 8 * 8;
  static const int prio = [] { const char* e = getenv("TSG_MFMA_PRIO"); return e ? atoi(e) : 0; }();
  g->prio = prio;
  return 0;
}

int tsg_conv3x3_wrw_gen_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad, int dilation,
                                  int groups) {
  return dtype == TSG_BF16 && Cin > 0 && Cout > 0 && Cin % W3_C == 0 && Cout % W3_C == 0 && kh == 3 && kw == 3 &&
         (stride == 1 || stride == 2) && pad == 1 && dilation == 1 && groups == 1;
}

size_t tsg_conv3x3_wrw_gen_ws_bytes(int64_t B, int64_t Hin, int64_t Win, int Cin, int Cout, int stride) {
  W3GenGeom g;
  if (w3gen_geom(&g, B, Hin, Win, Cin, Cout, stride)) return 0;
  return (size_t)g.npairs * g.bpp * W3_C * W3_N * sizeof(float);
}

static int conv3_wrw_gen_common(const void* x, const float* in_ab, const void* dy, float* dw, int64_t B, int64_t Hin,
                                int64_t Win, int Cin, int Cout, int stride, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !dy || !dw || !ws) return TSG_E_NULL;
  if (in_ab && !aligned16(in_ab)) return TSG_E_ALIGN;
  W3GenGeom g;
  int e = w3gen_geom(&g, B, Hin, Win, Cin, Cout, stride);
  if (e) return e;
  if (ws_bytes < tsg_conv3x3_wrw_gen_ws_bytes(B, Hin, Win, Cin, Cout, stride)) return TSG_E_WS;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dw) || !aligned16(ws)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  // raw buffer loads in the fetch (see the kernel's comment) unless TSG_CONV_WRW_BUF=0 or a thread's offsets inside a
  // tile could reach the 2 GB the descriptor covers (the base is per tile, so this takes a row of > 10^8 elements)
  static const bool buf_env = [] { const char* e = getenv("TSG_CONV_WRW_BUF"); return !(e && e[0] == '0'); }();
  const bool buf = buf_env && (int64_t)(3 * stride + 4) * Win * Cin * 2 < 0x40000000LL && (int64_t)5 * g.W * Cout * 2 < 0x40000000LL;
#define W3_GO2(SS, PFF, AF, BF)                                                                                   \
  do {                                                                                                            \
    TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_wrw_gen_k<SS, PFF, AF, BF>),                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3S<SS>::LDS));                  \
    hipLaunchKernelGGL((conv3_wrw_gen_k<SS, PFF, AF, BF>), dim3(g.npairs * g.bpp), dim3(256), W3S<SS>::LDS, st,   \
                       (const bf16_t*)x, (const bf16_t*)dy, (float*)ws, g, in_ab);                                \
  } while (0)
#define W3_GO(SS, PFF, AF) do { if (buf && PFF != 2) W3_GO2(SS, (PFF == 2 ? 1 : PFF), AF, true); else W3_GO2(SS, PFF, AF, false); } while (0)
  // stride 2 stays on PF = 1: two sets of its 23 loads do not fit the register file (the build spilled 72-84 VGPRs,
  // and a spilled register of an in-flight load is garbage)
  if (stride == 1 && w3_pf2()) { if (in_ab) W3_GO(1, 2, true); else W3_GO(1, 2, false); }
  else if (stride == 1 && w3_occ2(g.ntiles)) { if (in_ab) W3_GO(1, 0, true); else W3_GO(1, 0, false); }
  else if (stride == 1) { if (in_ab) W3_GO(1, 1, true); else W3_GO(1, 1, false); }
  else { if (in_ab) W3_GO(2, 1, true); else W3_GO(2, 1, false); }
#undef W3_GO
#undef W3_GO2
  TSG_CHECK_LAUNCH();
  // TSG_CONV_WRW_FOLD=2 (default) | 1: conv3_wrw_gen_fold2 (row lanes sized to the slot count) | the round-2 fold
  static const bool fold2 = [] { const char* e = getenv("TSG_CONV_WRW_FOLD"); return !(e && e[0] == '1'); }();
  if (fold2) {
    const int T = g.bpp <= 8 ? 1 : g.bpp <= 16 ? 2 : g.bpp <= 32 ? 4 : g.bpp <= 64 ? 8 : 16;
    const int64_t blocks = (int64_t)g.npairs * (W3_C * W3_N / 4) / (256 / T);
    hipLaunchKernelGGL(conv3_wrw_gen_fold2, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, g, T, dw);
  } else {
    hipLaunchKernelGGL(conv3_wrw_gen_fold, dim3(g.npairs * W3_C * 9), dim3(256), 0, st, (const float*)ws, g, dw);
  }
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_conv3x3_wrw_gen(const void* x, const void* dy, float* dw, int64_t B, int64_t Hin, int64_t Win, int Cin, int Cout,
                        int stride, void* ws, size_t ws_bytes, void* stream) {
  return conv3_wrw_gen_common(x, nullptr, dy, dw, B, Hin, Win, Cin, Cout, stride, ws, ws_bytes, stream);
}

int tsg_conv3x3_wrw_gen_norm(const void* x, const float* in_ab, const void* dy, float* dw, int64_t B, int64_t Hin,
                             int64_t Win, int Cin, int Cout, int stride, void* ws, size_t ws_bytes, void* stream) {
  if (!in_ab) return TSG_E_NULL;
  return conv3_wrw_gen_common(x, in_ab, dy, dw, B, Hin, Win, Cin, Cout, stride, ws, ws_bytes, stream);
}

}  // extern "C"

// ---------------------------------------------------------------- data gradient through a FORWARD convolution
// dx = conv(dy, w') with w'[ci][kh][kw][oc] = w[oc][2 - kh][2 - kw][ci] (stride 1, padding 1): the library's forward
// kernels for the symmetric 3x3 layers are 25-35 % faster than its backward-data kernels on gfx950
// (tools/probe_conv2.py), so the autograd function of these layers asks for a forward convolution with the rotated,
// transposed filter this kernel writes (bf16, channels_last) from the fp32 or bf16 master weight.
namespace tsg {
template <typename TI>
__global__ __launch_bounds__(256) void w3_rot180_t_k(const TI* __restrict__ w, bf16_t* __restrict__ out, int O, int I) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z, o0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int o = o0 + r, i = i0 + tx;
    tile[r][tx] = (o < O && i < I) ? ld1<TI>(w + ((int64_t)o * 9 + (8 - tap)) * I + i) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, o = o0 + tx;
    if (i < I && o < O) out[((int64_t)i * 9 + tap) * O + o] = f32_to_bf16(tile[tx][r]);
  }
}
}  // namespace tsg

extern "C" int tsg_conv3x3_weight_rot180_t(const void* w, int dtype, void* out, int O, int I, void* stream) {
  if (!w || !out) return TSG_E_NULL;
  if (O <= 0 || I <= 0) return TSG_E_SHAPE;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  dim3 grid((unsigned)((I + 31) / 32), (unsigned)((O + 31) / 32), 9);
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((tsg::w3_rot180_t_k<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)w, (tsg::bf16_t*)out, O, I);
  else
    hipLaunchKernelGGL((tsg::w3_rot180_t_k<tsg::bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, (const tsg::bf16_t*)w,
                       (tsg::bf16_t*)out, O, I);
  TSG_CHECK_LAUNCH();
  return 0;
}
