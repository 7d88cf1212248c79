// Stem convolution of the ResNet-18 context path and of BiSeNet's spatial path:
// Conv2d(3, 64, kernel 7, stride 2, padding 3, bias=False) on a [B,3,H,W] image
// (furnace/base_model/resnet.py:96-97 conv1; model/bisenet/*/network.py:116
// SpatialPath.conv_7x7).  The input needs no gradient, so the training step is
// forward + weight gradient.  Both are implicit GEMMs on the bf16 MFMA
// (32x32x16), memory-bound by the [B,OH,OW,64] activation (0.54 GB at B=16,
// 1024^2):
//
//   K axis of the GEMM: k = r * 8 + s, r = ic * 7 + kh (21 rows + 1 zero row),
//   s = kw + 1 (slot 0 is a zero tap) -> 176.  With that padding the 8 taps of a
//   K-fragment are 8 consecutive input columns starting at the even column
//   2*ow - 4, i.e. four aligned 32-bit LDS reads from the staged input patch.
//
//   forward : y[pixel][oc] = sum_k im2col[pixel][k] * w[oc][k]; weights live in
//             registers (A operand), pixels stream through LDS (B operand); the
//             tile is re-laid through LDS so every lane stores 16 B of NHWC.
//   wgrad   : dw[oc][k] = sum_pixel dy[pixel][oc] * im2col[pixel][k]; the GEMM K
//             axis is the pixel axis, so dy is transposed through LDS and the
//             patch is staged as even/odd column planes (two alignments each) to
//             keep the 8 consecutive pixels of a fragment contiguous.  Persistent
//             blocks accumulate in MFMA registers; per-block partials are summed
//             in a fixed order (fp64) => deterministic, no atomics.
//
// x: NCHW bf16.  y / dy: NHWC bf16 (channels_last).  w / dw: fp32 [64,3,7,7].
#include "tsg_common.h"

namespace tsg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int SC_OC = 64;
constexpr int SC_KP = 176;             // padded GEMM-K: 22 rows x 8 slots
constexpr int SC_KSTEPS = SC_KP / 16;  // 11
constexpr int SC_TH = 4, SC_TW = 32;   // output tile of a block: 4 rows (one per wave) x 32 columns
constexpr int SC_PR = 2 * SC_TH + 5;   // 13 input rows per channel
constexpr int SC_PD = SC_TW + 4;       // 36 dwords = 72 input columns (70 used), origin column 2*ow0 - 4
constexpr int SC_NPART = 768;          // persistent blocks of the weight-gradient kernel (3 per CU)

// ---------------------------------------------------------------- weights -> bf16 [64][176]
__global__ void stem_pack_w(const float* __restrict__ w, bf16_t* __restrict__ wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= SC_OC * SC_KP) return;
  const int oc = i / SC_KP, k = i % SC_KP, r = k >> 3, s = k & 7;
  float v = 0.f;
  if (r < 21 && s >= 1) v = w[oc * 147 + r * 7 + (s - 1)];
  wp[i] = f32_to_bf16(v);
}

struct StemGeom {
  int B, H, W, OH, OW, tiles_h, tiles_w, ntiles;
};

// ---------------------------------------------------------------- input patch of a tile
constexpr int SC_NPD = 3 * SC_PR * SC_PD;               // 1404 input dwords (bf16 pairs) per tile
constexpr int SC_NPF = (SC_NPD + 255) / 256;            // 6 per thread

struct TilePos { int b, oh0, ow0; };
__device__ __forceinline__ TilePos tile_pos(const StemGeom& g, int tile) {   // uniform: scalar ALU
  TilePos t;
  t.ow0 = (tile % g.tiles_w) * SC_TW;
  t.oh0 = ((tile / g.tiles_w) % g.tiles_h) * SC_TH;
  t.b = tile / (g.tiles_w * g.tiles_h);
  return t;
}

// The tile-independent half of the patch addressing, computed once per thread: pair `u` of this thread sits at
// input row 2*oh0 - 3 + rr[u], column 2*ow0 - 4 + cc[u] of channel ic, i.e. at element `off[u]` from
// &x[b, 0, 2*oh0, 2*ow0].  rr < 0 marks the slots past the 1404th pair.
struct PatchLane {
  int off[SC_NPF];
  int rc[SC_NPF];                                        // rr | cc << 8, or -1
};

__device__ __forceinline__ void patch_lane_init(const StemGeom& g, int tid, PatchLane& pl) {
#pragma unroll
  for (int u = 0; u < SC_NPF; ++u) {
    const int idx = tid + 256 * u;
    const int pr = idx / SC_PD, dc = idx % SC_PD, ic = pr / SC_PR, rr = pr % SC_PR;
    pl.rc[u] = idx < SC_NPD ? (rr | ((2 * dc) << 8)) : -1;
    pl.off[u] = (ic * g.H + rr - 3) * g.W + 2 * dc - 4;
  }
}

// iw is even and W is even: a pair is either fully inside the image or fully in the padding.
__device__ __forceinline__ void fetch_patch(const bf16_t* __restrict__ x, const StemGeom& g, const TilePos& tp,
                                            const PatchLane& pl, uint32_t (&rp)[SC_NPF]) {
  const bf16_t* org = x + ((int64_t)tp.b * 3 * g.H + 2 * tp.oh0) * g.W + 2 * tp.ow0;
  const int ih0 = 2 * tp.oh0 - 3, iw0 = 2 * tp.ow0 - 4;
  const bool interior = ih0 >= 0 && ih0 + SC_PR <= g.H && iw0 >= 0 && iw0 + 2 * SC_PD <= g.W;   // uniform
  if (interior) {
#pragma unroll
    for (int u = 0; u < SC_NPF; ++u) {
      rp[u] = 0u;
      if (u < SC_NPF - 1 || pl.rc[u] >= 0) rp[u] = *reinterpret_cast<const uint32_t*>(org + pl.off[u]);
    }
  } else {
#pragma unroll
    for (int u = 0; u < SC_NPF; ++u) {
      const int ih = ih0 + (pl.rc[u] & 0xff), iw = iw0 + (pl.rc[u] >> 8);
      rp[u] = 0u;
      if (pl.rc[u] >= 0 && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
        rp[u] = *reinterpret_cast<const uint32_t*>(org + pl.off[u]);
    }
  }
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: round to nearest even
  const f32x2_t f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}

// ---------------------------------------------------------------- forward
// 4 waves: wave = (row pair wr) * 2 + (oc half wm).  A wave keeps the weights of its 32 output channels in
// registers (11 K-fragments) and runs two pixel rows against them.
// STATS: the per-channel sum / square sum of the (bf16-rounded) outputs ride along — what the BatchNorm that follows
// every such stem would otherwise re-read the whole activation for (tsg_bn_stats).  partial[block][2][64], fp32.
// STORE = false (round 6): the statistics-only pass of the recomputing stem (below) — the same tiles, sums and partial
// rows, y never leaves the chip.
template <bool STATS, bool STORE = true>
__global__ __launch_bounds__(256) void stem_fwd_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                  bf16_t* __restrict__ y, StemGeom g, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) uint32_t patch[SC_NPD];                 // 5616 B
  __shared__ __attribute__((aligned(16))) bf16_t outs[SC_TH * SC_TW * 72];        // 18432 B: [pixel][64 oc + 8 pad]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int wm = wave & 1, wr = wave >> 1;

  bf16x8 fw[SC_KSTEPS];
#pragma unroll
  for (int t = 0; t < SC_KSTEPS; ++t)
    fw[t] = *reinterpret_cast<const bf16x8*>(wp + (wm * 32 + p) * SC_KP + t * 16 + half * 8);

  int rowoff[SC_KSTEPS];                               // patch row (in dwords) of this lane's K-fragment, tile row 0
#pragma unroll
  for (int t = 0; t < SC_KSTEPS; ++t) {
    int r = 2 * t + half;
    if (r >= 21) r = 0;                                // zero weights: any finite data will do
    rowoff[t] = ((r / 7) * SC_PR + r % 7) * SC_PD + p;
  }
  const int spl = tid >> 3, spart = tid & 7;           // store: tile column spl, 16-B part spart

  PatchLane pl;
  patch_lane_init(g, tid, pl);
  uint32_t rp[SC_NPF];
  float st1 = 0.f, st2 = 0.f;                          // STATS: channel tid & 63 over tile row tid >> 6
  int tile = blockIdx.x;
  TilePos tp = tile_pos(g, tile < g.ntiles ? tile : 0);
  if (tile < g.ntiles) fetch_patch(x, g, tp, pl, rp);
  for (; tile < g.ntiles; tile += gridDim.x) {
    __syncthreads();                                   // the previous tile's reads of patch/outs are done
#pragma unroll
    for (int u = 0; u < SC_NPF; ++u)
      if (tid + 256 * u < SC_NPD) patch[tid + 256 * u] = rp[u];
    __syncthreads();
    TilePos tn = tp;
    if (tile + (int)gridDim.x < g.ntiles) {            // in flight during the MFMAs below
      tn = tile_pos(g, tile + gridDim.x);
      fetch_patch(x, g, tn, pl, rp);
    }

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int t = 0; t < SC_KSTEPS; ++t) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t* q = patch + rowoff[t] + 2 * (2 * wr + i) * SC_PD;
        union { uint32_t u[4]; bf16x8 v; } fb;
        fb.u[0] = q[0]; fb.u[1] = q[1]; fb.u[2] = q[2]; fb.u[3] = q[3];
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[t], fb.v, acc[i], 0, 0, 0);
      }
    }

    // acc[i][r]: oc = 32 wm + (r & 3) + 8 (r >> 2) + 4 half, pixel = (row 2 wr + i, column p).  Re-lay as [pixel][oc].
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int oc0 = 32 * wm + 8 * gq + 4 * half;
        uint2 v;
        v.x = pack_bf16(acc[i][4 * gq + 0], acc[i][4 * gq + 1]);
        v.y = pack_bf16(acc[i][4 * gq + 2], acc[i][4 * gq + 3]);
        *reinterpret_cast<uint2*>(outs + ((2 * wr + i) * SC_TW + p) * 72 + oc0) = v;
      }
    __syncthreads();
    // thread -> (tile row qd, pixel spl, 16-B part spart): a wave stores 1 KB of consecutive NHWC bytes
    bf16_t* yt = y + (((int64_t)tp.b * g.OH + tp.oh0) * g.OW + tp.ow0) * SC_OC;
    const bool colok = tp.ow0 + spl < g.OW;
    if (STORE) {
#pragma unroll
      for (int qd = 0; qd < SC_TH; ++qd)
        if (tp.oh0 + qd < g.OH && colok)
          *reinterpret_cast<uint4*>(yt + ((int64_t)qd * g.OW + spl) * SC_OC + spart * 8) =
              *reinterpret_cast<const uint4*>(outs + (qd * SC_TW + spl) * 72 + spart * 8);
    }
    if (STATS) {
      const int c = tid & 63, qd = tid >> 6;
      if (tp.oh0 + qd < g.OH) {
        const int npx = g.OW - tp.ow0 < SC_TW ? g.OW - tp.ow0 : SC_TW;
        const bf16_t* col = outs + qd * SC_TW * 72 + c;
        if (npx == SC_TW) {
#pragma unroll 8
          for (int px = 0; px < SC_TW; ++px) { const float v = bf16_to_f32(col[px * 72]); st1 += v; st2 = fmaf(v, v, st2); }
        } else {
          for (int px = 0; px < npx; ++px) { const float v = bf16_to_f32(col[px * 72]); st1 += v; st2 = fmaf(v, v, st2); }
        }
      }
    }
    tp = tn;
  }
  if (STATS) {                                         // fold the four tile rows in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(outs);
    red[tid] = st1; red[256 + tid] = st2;
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, which = tid >> 6;
      const float* r = red + which * 256 + c;
      partial[((int64_t)blockIdx.x * 2 + which) * SC_OC + c] = (r[0] + r[64]) + (r[128] + r[192]);
    }
  }
}

// ---------------------------------------------------------------- weight gradient
constexpr int SC_DS = 136;                 // dy^T row: 128 pixels + 8 pad (272 B, 16-B multiple)
constexpr int SC_RS = 24;                  // plane row stride in dwords (bank pattern: rows land 24 apart)
constexpr int SC_RIC = 15;                 // plane rows per input channel (13 used); 15 = 7 mod 8 keeps r -> bank regular
constexpr int SC_PLANE = 1088;             // dwords per plane copy (45 rows x 24 = 1080, rounded to 17 x 64)

// parity q, alignment copy sg.  The bank offsets 0, 1, 4, 5 (with RS = 24 and RIC = 15) make the 32 lanes of a
// B-fragment ds_read_b32 hit 32 different banks (tools/emulate_stemconv.py checks it).
__device__ __forceinline__ int plane_base(int q, int sg) {
  return (q * 2 + sg) * SC_PLANE + sg + 4 * q;
}

// BNB: dy is not read but MADE — the BatchNorm (+ReLU) backward of the layer behind the stem,
//   dy = a dv + Bc (xc - mean) + C2,  dv = relu'(a xc + b) * da      (tsg_bn_bwd_apply with the recomputed mask, bf16-rounded),
// evaluated while the tile is staged from the gradient da w.r.t. the normalised activation and the stem's own output xc.
// The image needs no gradient, so this weight gradient is the ONLY consumer of dy: the 0.5 GB tensor is never written.
// __launch_bounds__(256, 3): SC_NPART = 3 x 256 persistent blocks must all be resident.  Left to itself the compiler gave
// the BNB form 180 VGPRs = two blocks per CU, so a third of the blocks ran as a second round (400 -> 360 us at 16 x 1024^2,
// round 6: profiles/r06_stem_wrw_bn_occupancy.txt); bounded it takes 160, spill-free.
template <bool BNB>
__global__ __launch_bounds__(256, 3) void stem_wrw_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                  float* __restrict__ part, StemGeom g, const bf16_t* __restrict__ xc,
                                                  const float* __restrict__ bp) {
  __shared__ __attribute__((aligned(16))) bf16_t dyT[SC_OC * SC_DS];       // 17408 B: [oc][pixel of the tile]
  __shared__ __attribute__((aligned(16))) uint32_t planes[4 * SC_PLANE];   // 17408 B
  __shared__ __attribute__((aligned(16))) float bps[BNB ? 5 * SC_OC : 4];  // the backward pack {a, b, mean, Bc, C2}
  if (BNB) for (int i = threadIdx.x; i < 5 * SC_OC; i += 256) bps[i] = bp[i];   // visible after the tile loop's first barrier
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, n = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  int bbase[3];                            // dword offset of this lane's tap (k index) for its 3 N-tiles
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int k = 32 * (3 * wn + j) + n;
    int r = k >> 3, s = k & 7;
    if (r >= 21) r = 20;                   // columns 168..191 of the GEMM are padding, never written out; alias a real lane
    const int ic = r / 7, kh = r % 7, q = s & 1, sh = s >> 1, sg = sh & 1;
    bbase[j] = plane_base(q, sg) + (ic * SC_RIC + kh) * SC_RS + ((sh + sg) >> 1);
  }
  bf16_t* pl16 = reinterpret_cast<bf16_t*>(planes);
  uint32_t* dyT32 = reinterpret_cast<uint32_t*>(dyT);

  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // dy staging: pixel pair spp of tile rows strow and strow + 2, group of 8 oc spart (two pixels -> one 32-bit LDS write)
  const int spp = tid & 15, strow = (tid >> 4) & 1, spart = tid >> 5;
  PatchLane pl;
  patch_lane_init(g, tid, pl);
  int pdst[SC_NPF];                                     // where pair u goes in plane (0,0), in bf16 elements
#pragma unroll
  for (int u = 0; u < SC_NPF; ++u) {
    const int idx = tid + 256 * u, pr = idx / SC_PD, dc = idx % SC_PD, ic = pr / SC_PR, rr = pr % SC_PR;
    pdst[u] = (ic * SC_RIC + rr) * SC_RS * 2 + dc;
  }
  uint4 rd[SC_TH];
  uint4 rc[BNB ? SC_TH : 1];                              // BNB: the stem output at the pixels of rd
  int okm = 0;                                            // BNB: which of the four are inside the image
  uint32_t rp[SC_NPF];
  auto fetch = [&](int tile) {
    const TilePos tp = tile_pos(g, tile);
    const int64_t off = (((int64_t)tp.b * g.OH + tp.oh0 + strow) * g.OW + tp.ow0 + 2 * spp) * SC_OC + spart * 8;
    okm = 0;
#pragma unroll
    for (int rs = 0; rs < 2; ++rs)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const bool ok = tp.oh0 + strow + 2 * rs < g.OH && tp.ow0 + 2 * spp + px < g.OW;
        const int64_t o = off + ((int64_t)rs * 2 * g.OW + px) * SC_OC;
        rd[rs * 2 + px] = ok ? *reinterpret_cast<const uint4*>(dy + o) : make_uint4(0, 0, 0, 0);
        if (BNB) {
          rc[rs * 2 + px] = ok ? *reinterpret_cast<const uint4*>(xc + o) : make_uint4(0, 0, 0, 0);
          okm |= (ok ? 1 : 0) << (rs * 2 + px);
        }
      }
    fetch_patch(x, g, tp, pl, rp);
  };
  int tile = blockIdx.x;
  if (tile < g.ntiles) fetch(tile);
  for (; tile < g.ntiles; tile += gridDim.x) {
    __syncthreads();                                      // previous tile's fragment reads are done
    if (BNB) {
#pragma unroll
      for (int q = 0; q < SC_TH; ++q) {
        uint32_t wd[4] = {rd[q].x, rd[q].y, rd[q].z, rd[q].w};
        const uint32_t wx[4] = {rc[q].x, rc[q].y, rc[q].z, rc[q].w};
        if (okm >> q & 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float o2[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int c = spart * 8 + 2 * i + h;
              const float dav = __uint_as_float(h ? (wd[i] & 0xffff0000u) : (wd[i] << 16));
              const float xv = __uint_as_float(h ? (wx[i] & 0xffff0000u) : (wx[i] << 16));
              const float a = bps[c], dv = fmaf(xv, a, bps[SC_OC + c]) > 0.f ? dav : 0.f;
              o2[h] = fmaf(a, dv, fmaf(bps[3 * SC_OC + c], xv - bps[2 * SC_OC + c], bps[4 * SC_OC + c]));
            }
            wd[i] = pack_bf16(o2[0], o2[1]);
          }
        }
        rd[q] = make_uint4(wd[0], wd[1], wd[2], wd[3]);
      }
    }
#pragma unroll
    for (int rs = 0; rs < 2; ++rs) {
      const uint32_t wa[4] = {rd[2 * rs].x, rd[2 * rs].y, rd[2 * rs].z, rd[2 * rs].w};                  // pixel 2 spp
      const uint32_t wb[4] = {rd[2 * rs + 1].x, rd[2 * rs + 1].y, rd[2 * rs + 1].z, rd[2 * rs + 1].w};  // pixel 2 spp + 1
      uint32_t* drow = dyT32 + (spart * 8) * (SC_DS / 2) + 16 * (strow + 2 * rs) + spp;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        drow[e * (SC_DS / 2)] = (e & 1) ? ((wa[e >> 1] >> 16) | (wb[e >> 1] & 0xffff0000u))
                                        : ((wa[e >> 1] & 0xffffu) | (wb[e >> 1] << 16));
    }
#pragma unroll
    for (int u = 0; u < SC_NPF; ++u) {
      if (u < SC_NPF - 1 || pl.rc[u] >= 0) {
        const bf16_t e0 = (bf16_t)(rp[u] & 0xffffu), e1 = (bf16_t)(rp[u] >> 16);
        bf16_t* d = pl16 + pdst[u];
        d[plane_base(0, 0) * 2] = e0;
        d[plane_base(0, 1) * 2 + 1] = e0;
        d[plane_base(1, 0) * 2] = e1;
        d[plane_base(1, 1) * 2 + 1] = e1;
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < g.ntiles) fetch(tile + gridDim.x);   // in flight during the MFMAs below
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {                      // 16 pixels per step: tile row ks >> 1, columns 16 (ks & 1) ..
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(dyT + (32 * wm + n) * SC_DS + ks * 16 + 8 * half);
      const int off = (2 * (ks >> 1)) * SC_RS + (ks & 1) * 8 + 4 * half;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const uint32_t* q = planes + bbase[j] + off;
        union { uint32_t u[4]; bf16x8 v; } fb;
        fb.u[0] = q[0]; fb.u[1] = q[1]; fb.u[2] = q[2]; fb.u[3] = q[3];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb.v, acc[j], 0, 0, 0);
      }
    }
  }
  // acc[j][r]: oc = 32 wm + (r & 3) + 8 (r >> 2) + 4 half, k = 32 (3 wn + j) + n
  float* out = part + (int64_t)blockIdx.x * SC_OC * SC_KP;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int k = 32 * (3 * wn + j) + n;
    if (k < SC_KP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[oc * SC_KP + k] = acc[j][r];
      }
    }
  }
}

// dw[oc][ic][kh][kw] = sum over the per-block partials, fixed order, fp64.  A block folds 64 consecutive
// entries of the [64][176] partial: 16 float4 columns x 16 interleaved slices of the partial index.
__global__ __launch_bounds__(256) void stem_wrw_fold(const float* __restrict__ part, int nparts,
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
      v[u] = *reinterpret_cast<const float4*>(src + (int64_t)(gidx < nparts ? gidx : gs) * SC_OC * SC_KP);
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
    const int flat = blockIdx.x * 64 + threadIdx.x, oc = flat / SC_KP, k = flat % SC_KP, r = k >> 3, sl = k & 7;
    if (r < 21 && sl >= 1) dw[oc * 147 + r * 7 + (sl - 1)] = (float)t;
  }
}


// =====================================================================================================================
// Round 6: the ResNet stem WITHOUT its 537 MB activation (furnace/base_model/resnet.py:96-100,131-133:
// maxpool(relu(bn1(conv1(img))))).  The 3 -> 64 7x7/2 convolution is 79 GFLOP = ~40 us of MFMA over a 100 MB image, its
// output 537 MB at 16 x 1024^2: storing y and reading it back (statistics consumer, BN + ReLU + pool, both backward
// passes, then the 537 MB gradient written and read again by the weight gradient) cost the step 0.87 ms.  Here y is
// RE-EVALUATED wherever it is needed and never written:
//   stem_fwd_k<true, false>      statistics of y (the partial rows tsg_stem_conv_fwd_stats writes)
//   stem_fwd_pool_k              conv -> BN -> ReLU -> maxpool(3, 2, 1): pooled map + one argmax byte per element
//   stem_pool_bwd_reduce_k       conv again; gradient of a stem pixel gathered from the <= 4 windows covering it, masked
//                                by the recomputed ReLU -> sum dy', sum dy' (y - mean)
//   stem_wrw_pool_k              conv again; dy = a dy' + Bc (y - mean) + C2 staged straight into the weight gradient's
//                                transposed LDS image; the patch that fed the forward MFMAs feeds the weight gradient's.
// Every recomputation rounds y to bf16 exactly as stem_fwd_k stores it, and the kernels are deterministic, so the four
// evaluations see identical values: results equal tsg_stem_conv_fwd_stats -> tsg_bn_relu_pool_* -> tsg_stem_conv_wrw
// (bit for bit in the pooled map, the argmax bytes and dw's operands; the BN sums differ in summation order only).
// =====================================================================================================================

// the 4 x 32-pixel tile of y of the staged patch -> outs[pixel][72] (bf16, the values stem_fwd_k stores)
__device__ __forceinline__ void stem_tile_to_lds(const uint32_t* __restrict__ patch, const bf16x8 (&fw)[SC_KSTEPS],
                                                 const int (&rowoff)[SC_KSTEPS], int wr, int wm, int half, int p,
                                                 bf16_t* __restrict__ outs) {
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int t = 0; t < SC_KSTEPS; ++t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t* q = patch + rowoff[t] + 2 * (2 * wr + i) * SC_PD;
      union { uint32_t u[4]; bf16x8 v; } fb;
      fb.u[0] = q[0]; fb.u[1] = q[1]; fb.u[2] = q[2]; fb.u[3] = q[3];
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[t], fb.v, acc[i], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int oc0 = 32 * wm + 8 * gq + 4 * half;
      uint2 v;
      v.x = pack_bf16(acc[i][4 * gq + 0], acc[i][4 * gq + 1]);
      v.y = pack_bf16(acc[i][4 * gq + 2], acc[i][4 * gq + 3]);
      *reinterpret_cast<uint2*>(outs + ((2 * wr + i) * SC_TW + p) * 72 + oc0) = v;
    }
}

// The four pool windows (K..K+1, M..M+1) of a 2 x 2 block of stem pixels, 8 channels: raw gradient words + argmax bytes.
// Unconditional loads from clamped addresses; a window outside the pooled map matches no position (bytes 0xff).
struct PoolWin4 {
  uint4 d[4];
  uint2 w[4];
};
__device__ __forceinline__ void pool_win4_load(PoolWin4& pw, const bf16_t* __restrict__ dpn, const uint8_t* __restrict__ idn,
                                               int K, int M, int PH, int PW) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int oy = K + (q >> 1), ox = M + (q & 1);
    const bool ok = oy < PH && ox < PW;
    const int64_t o = ((int64_t)(oy < PH ? oy : PH - 1) * PW + (ox < PW ? ox : PW - 1)) * SC_OC;
    pw.d[q] = *reinterpret_cast<const uint4*>(dpn + o);
    const uint2 u = *reinterpret_cast<const uint2*>(idn + o);
    pw.w[q] = ok ? u : make_uint2(0xffffffffu, 0xffffffffu);
  }
}
// gradients of the block's four pixels (q = 2 row + col), 8 channels — bnpool.hip block_grads, same additions in the same order
__device__ __forceinline__ void pool_win4_grads(const PoolWin4& pw, float (&gr)[4][8]) {
  float d[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t w4[4] = {pw.d[q].x, pw.d[q].y, pw.d[q].z, pw.d[q].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      d[q][2 * i] = __uint_as_float(w4[i] << 16);
      d[q][2 * i + 1] = __uint_as_float(w4[i] & 0xffff0000u);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int sh = 8 * (j & 3);
    const uint32_t b00 = ((j < 4 ? pw.w[0].x : pw.w[0].y) >> sh) & 0xffu, b01 = ((j < 4 ? pw.w[1].x : pw.w[1].y) >> sh) & 0xffu;
    const uint32_t b10 = ((j < 4 ? pw.w[2].x : pw.w[2].y) >> sh) & 0xffu, b11 = ((j < 4 ? pw.w[3].x : pw.w[3].y) >> sh) & 0xffu;
    gr[0][j] = b00 == 4u ? d[0][j] : 0.f;
    gr[1][j] = (b00 == 5u ? d[0][j] : 0.f) + (b01 == 3u ? d[1][j] : 0.f);
    gr[2][j] = (b00 == 7u ? d[0][j] : 0.f) + (b10 == 1u ? d[2][j] : 0.f);
    gr[3][j] = ((b00 == 8u ? d[0][j] : 0.f) + (b01 == 6u ? d[1][j] : 0.f)) +
               ((b10 == 2u ? d[2][j] : 0.f) + (b11 == 0u ? d[3][j] : 0.f));
  }
}

// ---------------------------------------------------------------- backward sums with y recomputed
// Consumer mapping of a tile: thread = (column pair m, row pair kk, 8-channel group spart) = one 2 x 2 block of stem pixels.
__global__ __launch_bounds__(256, 2) void stem_pool_bwd_reduce_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                                 const bf16_t* __restrict__ dpool,
                                                                 const uint8_t* __restrict__ idx, StemGeom g, int PH, int PW,
                                                                 const float* __restrict__ fp, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) uint32_t patch[SC_NPD];
  __shared__ __attribute__((aligned(16))) bf16_t outs[SC_TH * SC_TW * 72];        // 18432 B; at the end float [2][32][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int wm = wave & 1, wr = wave >> 1;
  bf16x8 fw[SC_KSTEPS];
#pragma unroll
  for (int t = 0; t < SC_KSTEPS; ++t)
    fw[t] = *reinterpret_cast<const bf16x8*>(wp + (wm * 32 + p) * SC_KP + t * 16 + half * 8);
  int rowoff[SC_KSTEPS];
#pragma unroll
  for (int t = 0; t < SC_KSTEPS; ++t) {
    int r = 2 * t + half;
    if (r >= 21) r = 0;
    rowoff[t] = ((r / 7) * SC_PR + r % 7) * SC_PD + p;
  }
  const int m = tid & 15, kk = (tid >> 4) & 1, spart = tid >> 5;
  float a[8], b[8], mu[8], a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = fp[spart * 8 + j]; b[j] = fp[SC_OC + spart * 8 + j]; mu[j] = fp[2 * SC_OC + spart * 8 + j];
    a1[j] = 0.f; a2[j] = 0.f;
  }
  PatchLane pl;
  patch_lane_init(g, tid, pl);
  uint32_t rp[SC_NPF];
  int tile = blockIdx.x;
  TilePos tp = tile_pos(g, tile < g.ntiles ? tile : 0);
  if (tile < g.ntiles) fetch_patch(x, g, tp, pl, rp);
  for (; tile < g.ntiles; tile += gridDim.x) {
    __syncthreads();                                   // the previous tile's reads of patch / outs are done
#pragma unroll
    for (int u = 0; u < SC_NPF; ++u)
      if (tid + 256 * u < SC_NPD) patch[tid + 256 * u] = rp[u];
    __syncthreads();
    TilePos tn = tp;
    if (tile + (int)gridDim.x < g.ntiles) {            // in flight during the MFMAs below
      tn = tile_pos(g, tile + gridDim.x);
      fetch_patch(x, g, tn, pl, rp);
    }
    PoolWin4 pw;
    {
      const int64_t pimg = (int64_t)tp.b * PH * PW * SC_OC + spart * 8;
      pool_win4_load(pw, dpool + pimg, idx + pimg, (tp.oh0 >> 1) + kk, (tp.ow0 >> 1) + m, PH, PW);
    }
    stem_tile_to_lds(patch, fw, rowoff, wr, wm, half, p, outs);
    __syncthreads();
    float gr[4][8];
    pool_win4_grads(pw, gr);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 2 * kk + (q >> 1), col = 2 * m + (q & 1);
      const bool ok = tp.oh0 + row < g.OH && tp.ow0 + col < g.OW;
      const uint4 yv = *reinterpret_cast<const uint4*>(outs + (row * SC_TW + col) * 72 + spart * 8);
      const uint32_t wy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = __uint_as_float((j & 1) ? (wy[j >> 1] & 0xffff0000u) : (wy[j >> 1] << 16));
        const float dv = (ok && fmaf(xv, a[j], b[j]) > 0.f) ? gr[q][j] : 0.f;
        a1[j] += dv;
        a2[j] = fmaf(dv, xv - mu[j], a2[j]);
      }
    }
    tp = tn;
  }
  // fold the 32 pixel-block lanes of every channel in a fixed order
  __syncthreads();
  float* red = reinterpret_cast<float*>(outs);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[(kk * 16 + m) * SC_OC + spart * 8 + j] = a1[j];
    red[2048 + (kk * 16 + m) * SC_OC + spart * 8 + j] = a2[j];
  }
  __syncthreads();
  if (tid < 128) {
    const int c = tid & 63, which = tid >> 6;
    const float* r = red + which * 2048 + c;
    float t = 0.f;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) t += r[q * SC_OC];
    partial[((int64_t)blockIdx.x * 2 + which) * SC_OC + c] = t;
  }
}

// ---------------------------------------------------------------- weight gradient with dy (and, RECOMP, y) made on the fly
// RECOMP = true : y of the tile is re-evaluated from the staged patch (nothing of the stem is in memory);
// RECOMP = false: y is READ (xc, the stored stem output) — the forward MFMAs, their 88 LDS reads and the re-lay of the tile
//                 are the larger half of the recomputing kernel's instructions, and the stem kernels are issue-bound.
template <bool RECOMP, int WAVES = RECOMP ? 2 : 3>
__global__ __launch_bounds__(256, WAVES) void stem_wrw_pool_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                          const bf16_t* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                          StemGeom g, int PH, int PW, const float* __restrict__ bp,
                                                          float* __restrict__ part, const bf16_t* __restrict__ xc) {
  __shared__ __attribute__((aligned(16))) uint32_t patch[RECOMP ? SC_NPD : 4];    // forward image of the input patch
  __shared__ __attribute__((aligned(16))) bf16_t outs[RECOMP ? SC_TH * SC_TW * 72 : 8];   // y tile [pixel][64 + 8]
  __shared__ __attribute__((aligned(16))) bf16_t dyT[SC_OC * SC_DS];              // dy^T [oc][pixel of the tile]
  __shared__ __attribute__((aligned(16))) uint32_t planes[4 * SC_PLANE];          // weight-gradient image of the patch
  __shared__ __attribute__((aligned(16))) float bps[5 * SC_OC];
  for (int i = threadIdx.x; i < 5 * SC_OC; i += 256) bps[i] = bp[i];              // visible after the tile loop's first barrier
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, n = lane & 31;
  const int fwm = wave & 1, fwr = wave >> 1;           // forward roles: oc half, row pair
  const int wm = wave >> 1, wn = wave & 1;             // weight-gradient roles: oc half, K-column group
  bf16x8 fw[SC_KSTEPS];
  int rowoff[SC_KSTEPS];
  if (RECOMP) {
#pragma unroll
    for (int t = 0; t < SC_KSTEPS; ++t)
      fw[t] = *reinterpret_cast<const bf16x8*>(wp + (fwm * 32 + n) * SC_KP + t * 16 + half * 8);
#pragma unroll
    for (int t = 0; t < SC_KSTEPS; ++t) {
      int r = 2 * t + half;
      if (r >= 21) r = 0;
      rowoff[t] = ((r / 7) * SC_PR + r % 7) * SC_PD + n;
    }
  }
  int bbase[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int k = 32 * (3 * wn + j) + n;
    int r = k >> 3, s = k & 7;
    if (r >= 21) r = 20;
    const int ic = r / 7, kh = r % 7, q = s & 1, sh = s >> 1, sg = sh & 1;
    bbase[j] = plane_base(q, sg) + (ic * SC_RIC + kh) * SC_RS + ((sh + sg) >> 1);
  }
  bf16_t* pl16 = reinterpret_cast<bf16_t*>(planes);
  uint32_t* dyT32 = reinterpret_cast<uint32_t*>(dyT);
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int m = tid & 15, kk = (tid >> 4) & 1, spart = tid >> 5;
  PatchLane pl;
  patch_lane_init(g, tid, pl);
  int pdst[SC_NPF];
#pragma unroll
  for (int u = 0; u < SC_NPF; ++u) {
    const int id = tid + 256 * u, pr = id / SC_PD, dc = id % SC_PD, ic = pr / SC_PR, rr = pr % SC_PR;
    pdst[u] = (ic * SC_RIC + rr) * SC_RS * 2 + dc;
  }
  uint32_t rp[SC_NPF];
  int tile = blockIdx.x;
  TilePos tp = tile_pos(g, tile < g.ntiles ? tile : 0);
  if (tile < g.ntiles) fetch_patch(x, g, tp, pl, rp);
  for (; tile < g.ntiles; tile += gridDim.x) {
    __syncthreads();                                   // the previous tile's fragment reads are done
#pragma unroll
    for (int u = 0; u < SC_NPF; ++u) {
      if (u < SC_NPF - 1 || pl.rc[u] >= 0) {
        if (RECOMP) patch[tid + 256 * u] = rp[u];
        const bf16_t e0 = (bf16_t)(rp[u] & 0xffffu), e1 = (bf16_t)(rp[u] >> 16);
        bf16_t* d = pl16 + pdst[u];
        d[plane_base(0, 0) * 2] = e0;
        d[plane_base(0, 1) * 2 + 1] = e0;
        d[plane_base(1, 0) * 2] = e1;
        d[plane_base(1, 1) * 2 + 1] = e1;
      }
    }
    __syncthreads();
    TilePos tn = tp;
    if (tile + (int)gridDim.x < g.ntiles) {
      tn = tile_pos(g, tile + gridDim.x);
      fetch_patch(x, g, tn, pl, rp);
    }
    PoolWin4 pw;
    {
      const int64_t pimg = (int64_t)tp.b * PH * PW * SC_OC + spart * 8;
      pool_win4_load(pw, dpool + pimg, idx + pimg, (tp.oh0 >> 1) + kk, (tp.ow0 >> 1) + m, PH, PW);
    }
    uint4 yv[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 2 * kk + (q >> 1), col = 2 * m + (q & 1);
      ok[q] = tp.oh0 + row < g.OH && tp.ow0 + col < g.OW;
      if (!RECOMP) {                                    // unconditional loads from clamped addresses (a pixel outside re-reads the tile's first)
        const int64_t o = (((int64_t)tp.b * g.OH + tp.oh0 + (ok[q] ? row : 0)) * g.OW + tp.ow0 + (ok[q] ? col : 0)) * SC_OC + spart * 8;
        yv[q] = *reinterpret_cast<const uint4*>(xc + o);
      }
    }
    if (RECOMP) {
      stem_tile_to_lds(patch, fw, rowoff, fwr, fwm, half, n, outs);
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q)
        yv[q] = *reinterpret_cast<const uint4*>(outs + ((2 * kk + (q >> 1)) * SC_TW + 2 * m + (q & 1)) * 72 + spart * 8);
    }
    {
      uint32_t dyw[4][4];                              // the four pixels' dy, 8 channels each, packed bf16
      // one channel PAIR at a time (word i of every 16-byte operand): the gradients of the four pixels from the four
      // windows (pool_win4_grads' additions in its order), then dy — 20 live values instead of 100
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t dq[4] = {i == 0 ? pw.d[0].x : i == 1 ? pw.d[0].y : i == 2 ? pw.d[0].z : pw.d[0].w,
                                i == 0 ? pw.d[1].x : i == 1 ? pw.d[1].y : i == 2 ? pw.d[1].z : pw.d[1].w,
                                i == 0 ? pw.d[2].x : i == 1 ? pw.d[2].y : i == 2 ? pw.d[2].z : pw.d[2].w,
                                i == 0 ? pw.d[3].x : i == 1 ? pw.d[3].y : i == 2 ? pw.d[3].z : pw.d[3].w};
        const uint32_t yq[4] = {i == 0 ? yv[0].x : i == 1 ? yv[0].y : i == 2 ? yv[0].z : yv[0].w,
                                i == 0 ? yv[1].x : i == 1 ? yv[1].y : i == 2 ? yv[1].z : yv[1].w,
                                i == 0 ? yv[2].x : i == 1 ? yv[2].y : i == 2 ? yv[2].z : yv[2].w,
                                i == 0 ? yv[3].x : i == 1 ? yv[3].y : i == 2 ? yv[3].z : yv[3].w};
        float o2[4][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = 2 * i + h, c = spart * 8 + j, sh = 8 * (j & 3);
          float d[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) d[q] = __uint_as_float(h ? (dq[q] & 0xffff0000u) : (dq[q] << 16));
          const uint32_t b00 = ((j < 4 ? pw.w[0].x : pw.w[0].y) >> sh) & 0xffu, b01 = ((j < 4 ? pw.w[1].x : pw.w[1].y) >> sh) & 0xffu;
          const uint32_t b10 = ((j < 4 ? pw.w[2].x : pw.w[2].y) >> sh) & 0xffu, b11 = ((j < 4 ? pw.w[3].x : pw.w[3].y) >> sh) & 0xffu;
          float gq[4];
          gq[0] = b00 == 4u ? d[0] : 0.f;
          gq[1] = (b00 == 5u ? d[0] : 0.f) + (b01 == 3u ? d[1] : 0.f);
          gq[2] = (b00 == 7u ? d[0] : 0.f) + (b10 == 1u ? d[2] : 0.f);
          gq[3] = ((b00 == 8u ? d[0] : 0.f) + (b01 == 6u ? d[1] : 0.f)) + ((b10 == 2u ? d[2] : 0.f) + (b11 == 0u ? d[3] : 0.f));
          const float av = bps[c], bv = bps[SC_OC + c], mv = bps[2 * SC_OC + c], bcv = bps[3 * SC_OC + c], c2v = bps[4 * SC_OC + c];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float xv = __uint_as_float(h ? (yq[q] & 0xffff0000u) : (yq[q] << 16));
            const float dv = fmaf(xv, av, bv) > 0.f ? gq[q] : 0.f;
            o2[q][h] = fmaf(av, dv, fmaf(bcv, xv - mv, c2v));
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) dyw[q][i] = ok[q] ? pack_bf16(o2[q][0], o2[q][1]) : 0u;
      }
#pragma unroll
      for (int rs = 0; rs < 2; ++rs) {                 // rows 2 kk + rs: pixel pair (2 m, 2 m + 1) -> one 32-bit LDS write per channel
        uint32_t* drow = dyT32 + (spart * 8) * (SC_DS / 2) + 16 * (2 * kk + rs) + m;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t wa = dyw[2 * rs][e >> 1], wb = dyw[2 * rs + 1][e >> 1];
          drow[e * (SC_DS / 2)] = (e & 1) ? ((wa >> 16) | (wb & 0xffff0000u)) : ((wa & 0xffffu) | (wb << 16));
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(dyT + (32 * wm + n) * SC_DS + ks * 16 + 8 * half);
      const int off = (2 * (ks >> 1)) * SC_RS + (ks & 1) * 8 + 4 * half;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const uint32_t* q = planes + bbase[j] + off;
        union { uint32_t u[4]; bf16x8 v; } fb;
        fb.u[0] = q[0]; fb.u[1] = q[1]; fb.u[2] = q[2]; fb.u[3] = q[3];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb.v, acc[j], 0, 0, 0);
      }
    }
    tp = tn;
  }
  float* out = part + (int64_t)blockIdx.x * SC_OC * SC_KP;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int k = 32 * (3 * wn + j) + n;
    if (k < SC_KP) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oc = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
        out[oc * SC_KP + k] = acc[j][r];
      }
    }
  }
}

// ---------------------------------------------------------------- forward: conv -> BN -> ReLU -> maxpool(3, 2, 1)
// Tile = 4 x 15 pooled pixels = the 9 x 31 stem pixels of their windows (rows 2 ph0 - 1 .., columns 2 pw0 - 1 ..; one
// 32-wide MFMA column tile): neighbouring tiles re-evaluate the shared window row / two columns instead of exchanging
// them (1.2 x the MFMA work of the plain convolution, which has the time: the pass is bound by its 200 MB of stores).
constexpr int FP_PRT = 4, FP_PCT = 15;                  // pooled rows / columns of a tile
constexpr int FP_YR = 2 * FP_PRT + 1;                   // 9 stem rows
constexpr int FP_PR = 2 * FP_YR + 5;                    // 23 input rows per channel
constexpr int FP_NPD = 3 * FP_PR * SC_PD;               // 2484 input dwords per tile
constexpr int FP_NPF = (FP_NPD + 255) / 256;            // 10 per thread

struct PoolGeom { int PH, PW, tph, tpw, ntiles; };

__global__ __launch_bounds__(256, 2) void stem_fwd_pool_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp,
                                                          bf16_t* __restrict__ ypool, uint8_t* __restrict__ idx,
                                                          StemGeom g, PoolGeom pg, const float* __restrict__ fp) {
  __shared__ __attribute__((aligned(16))) uint32_t patch[FP_NPD];                  // 9936 B
  __shared__ __attribute__((aligned(16))) bf16_t outs[FP_YR * SC_TW * 72];          // 41472 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int wm = wave & 1, wr = wave >> 1;
  bf16x8 fw[SC_KSTEPS];
#pragma unroll
  for (int t = 0; t < SC_KSTEPS; ++t)
    fw[t] = *reinterpret_cast<const bf16x8*>(wp + (wm * 32 + p) * SC_KP + t * 16 + half * 8);
  int rowoff[SC_KSTEPS];
#pragma unroll
  for (int t = 0; t < SC_KSTEPS; ++t) {
    int r = 2 * t + half;
    if (r >= 21) r = 0;
    rowoff[t] = ((r / 7) * FP_PR + r % 7) * SC_PD + p;
  }
  // patch addressing (the 23-row version of PatchLane): pair u at input row 2 yr0 - 3 + rr, column 2 yc0 - 4 + cc
  int poff[FP_NPF], prc[FP_NPF];
#pragma unroll
  for (int u = 0; u < FP_NPF; ++u) {
    const int id = tid + 256 * u;
    const int pr = id / SC_PD, dc = id % SC_PD, ic = pr / FP_PR, rr = pr % FP_PR;
    prc[u] = id < FP_NPD ? (rr | ((2 * dc) << 8)) : -1;
    poff[u] = (ic * g.H + rr - 3) * g.W + 2 * dc - 4;
  }
  uint32_t rp[FP_NPF];
  struct PT { int b, ph0, pw0; };
  auto tile_at = [&](int tile) {
    PT t;
    t.pw0 = (tile % pg.tpw) * FP_PCT;
    t.ph0 = ((tile / pg.tpw) % pg.tph) * FP_PRT;
    t.b = tile / (pg.tpw * pg.tph);
    return t;
  };
  auto fetch = [&](const PT& t) {
    const int yr0 = 2 * t.ph0 - 1, yc0 = 2 * t.pw0 - 1;
    const bf16_t* org = x + ((int64_t)t.b * 3 * g.H + 2 * yr0) * g.W + 2 * yc0;
    const int ih0 = 2 * yr0 - 3, iw0 = 2 * yc0 - 4;
#pragma unroll
    for (int u = 0; u < FP_NPF; ++u) {
      const int ih = ih0 + (prc[u] & 0xff), iw = iw0 + (prc[u] >> 8);
      const bool ok = prc[u] >= 0 && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
      const uint32_t v = *reinterpret_cast<const uint32_t*>(ok ? org + poff[u] : x);   // unconditional, clamped address
      rp[u] = ok ? v : 0u;
    }
  };
  const int spart = tid & 7;
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = fp[spart * 8 + j]; b[j] = fp[SC_OC + spart * 8 + j]; }
  int tile = blockIdx.x;
  PT tp = tile_at(tile < pg.ntiles ? tile : 0);
  if (tile < pg.ntiles) fetch(tp);
  for (; tile < pg.ntiles; tile += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < FP_NPF; ++u)
      if (tid + 256 * u < FP_NPD) patch[tid + 256 * u] = rp[u];
    __syncthreads();
    PT tn = tp;
    if (tile + (int)gridDim.x < pg.ntiles) {
      tn = tile_at(tile + gridDim.x);
      fetch(tn);
    }
    // stem rows i = 0..8 of the tile: row group wr = 0 takes 0..4, wr = 1 takes 5..8, two rows per MFMA chain pair
    const int i_beg = wr ? 5 : 0, i_end = wr ? 9 : 5;
    for (int i0 = i_beg; i0 < i_end; i0 += 2) {
      const bool two = i0 + 1 < i_end;                 // uniform per wave
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
      for (int t = 0; t < SC_KSTEPS; ++t) {
        const uint32_t* q0 = patch + rowoff[t] + 2 * i0 * SC_PD;
        union { uint32_t u[4]; bf16x8 v; } fb;
        fb.u[0] = q0[0]; fb.u[1] = q0[1]; fb.u[2] = q0[2]; fb.u[3] = q0[3];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[t], fb.v, acc[0], 0, 0, 0);
        if (two) {
          const uint32_t* q1 = q0 + 2 * SC_PD;
          fb.u[0] = q1[0]; fb.u[1] = q1[1]; fb.u[2] = q1[2]; fb.u[3] = q1[3];
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[t], fb.v, acc[1], 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i == 0 || two) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int oc0 = 32 * wm + 8 * gq + 4 * half;
            uint2 v;
            v.x = pack_bf16(acc[i][4 * gq + 0], acc[i][4 * gq + 1]);
            v.y = pack_bf16(acc[i][4 * gq + 2], acc[i][4 * gq + 3]);
            *reinterpret_cast<uint2*>(outs + ((i0 + i) * SC_TW + p) * 72 + oc0) = v;
          }
        }
      }
    }
    __syncthreads();
    // pool: item = (pooled row pr, pooled column pc, channel group); first maximum in scan order of the fp32 values
    // relu(a y + b), positions outside the stem map excluded — bn_relu_pool_fwd_k's rule, value for value
    const int yr0 = 2 * tp.ph0 - 1, yc0 = 2 * tp.pw0 - 1;
#pragma unroll 1
    for (int item = tid; item < FP_PRT * FP_PCT * 8; item += 256) {
      const int pq = item >> 3, pr = pq / FP_PCT, pc = pq - pr * FP_PCT;
      const int ph = tp.ph0 + pr, pwx = tp.pw0 + pc;
      if (ph >= pg.PH || pwx >= pg.PW) continue;
      float best[8];
      int am[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { best[j] = -1.f; am[j] = 0; }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        float rm[8];
        int rk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { rm[j] = -1.f; rk[j] = 0; }
        const int i = 2 * pr + ky, yr = yr0 + i;
        const bool row_ok = yr >= 0 && yr < g.OH;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int pp = 2 * pc + kx, yc = yc0 + pp;
          const bool ok = row_ok && yc >= 0 && yc < g.OW;
          const uint4 yv = *reinterpret_cast<const uint4*>(outs + (i * SC_TW + pp) * 72 + spart * 8);
          const uint32_t wy[4] = {yv.x, yv.y, yv.z, yv.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xv = __uint_as_float((j & 1) ? (wy[j >> 1] & 0xffff0000u) : (wy[j >> 1] << 16));
            float v = fmaxf(fmaf(xv, a[j], b[j]), 0.f);
            v = ok ? v : -1.f;
            if (v > rm[j]) { rm[j] = v; rk[j] = kx; }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (ky == 0) { best[j] = rm[j]; am[j] = rk[j]; }
          else if (rm[j] > best[j]) { best[j] = rm[j]; am[j] = 3 * ky + rk[j]; }
        }
      }
      uint32_t ov[4], w[2] = {0u, 0u};
#pragma unroll
      for (int j = 0; j < 4; ++j) ov[j] = pack_bf16(best[2 * j], best[2 * j + 1]);
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j >> 2] |= (uint32_t)am[j] << (8 * (j & 3));
      const int64_t o = (((int64_t)tp.b * pg.PH + ph) * pg.PW + pwx) * SC_OC + spart * 8;
      *reinterpret_cast<uint4*>(ypool + o) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
      *reinterpret_cast<uint2*>(idx + o) = make_uint2(w[0], w[1]);
    }
    tp = tn;
  }
}

static size_t sc_align(size_t v) { return (v + 255) / 256 * 256; }

static bool stem_geom(int64_t B, int64_t H, int64_t W, StemGeom* g) {
  if (B <= 0 || H <= 0 || W <= 0 || (W & 1)) return false;
  const int64_t OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int64_t th = (OH + SC_TH - 1) / SC_TH, tw = (OW + SC_TW - 1) / SC_TW;
  if (B * th * tw > 0x7fffffffLL || B * 3 * H * W > 0x7fffffffffffLL) return false;
  g->B = (int)B; g->H = (int)H; g->W = (int)W; g->OH = (int)OH; g->OW = (int)OW;
  g->tiles_h = (int)th; g->tiles_w = (int)tw; g->ntiles = (int)(B * th * tw);
  return true;
}

}  // namespace tsg

using namespace tsg;

extern "C" {

int tsg_stem_conv_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad, int dilation,
                            int groups, int64_t H, int64_t W) {
  return dtype == TSG_BF16 && Cin == 3 && Cout == SC_OC && kh == 7 && kw == 7 && stride == 2 && pad == 3 &&
         dilation == 1 && groups == 1 && H > 0 && W > 0 && (W & 1) == 0;
}

size_t tsg_stem_conv_ws_bytes(void) {
  return sc_align((size_t)SC_OC * SC_KP * sizeof(bf16_t)) + (size_t)SC_NPART * SC_OC * SC_KP * sizeof(float);
}

int tsg_stem_conv_fwd(const void* x, const float* w, void* y, int64_t B, int64_t H, int64_t W, void* ws,
                      size_t ws_bytes, void* stream) {
  if (!x || !w || !y || !ws) return TSG_E_NULL;
  StemGeom g;
  if (!stem_geom(B, H, W, &g)) return TSG_E_SHAPE;
  if (ws_bytes < sc_align((size_t)SC_OC * SC_KP * sizeof(bf16_t))) return TSG_E_WS;
  if (!aligned16(y) || !aligned16(ws) || (((uintptr_t)x) & 3u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* wp = (bf16_t*)ws;
  hipLaunchKernelGGL(stem_pack_w, dim3((SC_OC * SC_KP + 255) / 256), dim3(256), 0, st, w, wp);
  TSG_CHECK_LAUNCH();
  const int grid = g.ntiles < 768 ? g.ntiles : 768;             // 3 resident blocks per CU (164 VGPRs)
  hipLaunchKernelGGL(stem_fwd_k<false>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp, (bf16_t*)y, g,
                     (float*)nullptr);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_stem_conv_stats_partials(int64_t B, int64_t H, int64_t W) {
  StemGeom g;
  if (!stem_geom(B, H, W, &g)) return TSG_E_SHAPE;
  return g.ntiles < 768 ? g.ntiles : 768;
}

int tsg_stem_conv_fwd_stats(const void* x, const float* w, void* y, float* partial, int64_t B, int64_t H, int64_t W,
                            void* ws, size_t ws_bytes, void* stream) {
  if (!x || !w || !y || !ws || !partial) return TSG_E_NULL;
  StemGeom g;
  if (!stem_geom(B, H, W, &g)) return TSG_E_SHAPE;
  if (ws_bytes < sc_align((size_t)SC_OC * SC_KP * sizeof(bf16_t))) return TSG_E_WS;
  if (!aligned16(y) || !aligned16(ws) || (((uintptr_t)x) & 3u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* wp = (bf16_t*)ws;
  hipLaunchKernelGGL(stem_pack_w, dim3((SC_OC * SC_KP + 255) / 256), dim3(256), 0, st, w, wp);
  TSG_CHECK_LAUNCH();
  const int grid = g.ntiles < 768 ? g.ntiles : 768;
  hipLaunchKernelGGL(stem_fwd_k<true>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp, (bf16_t*)y, g,
                     partial);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_stem_conv_wrw(const void* x, const void* dy, float* dw, int64_t B, int64_t H, int64_t W, void* ws,
                      size_t ws_bytes, void* stream) {
  if (!x || !dy || !dw || !ws) return TSG_E_NULL;
  StemGeom g;
  if (!stem_geom(B, H, W, &g)) return TSG_E_SHAPE;
  if (ws_bytes < tsg_stem_conv_ws_bytes()) return TSG_E_WS;
  if (!aligned16(dy) || !aligned16(ws) || (((uintptr_t)x) & 3u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)((char*)ws + sc_align((size_t)SC_OC * SC_KP * sizeof(bf16_t)));
  const int grid = g.ntiles < SC_NPART ? g.ntiles : SC_NPART;
  hipLaunchKernelGGL(stem_wrw_k<false>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, part, g,
                     (const bf16_t*)nullptr, (const float*)nullptr);
  TSG_CHECK_LAUNCH();
  hipLaunchKernelGGL(stem_wrw_fold, dim3(SC_OC * SC_KP / 64), dim3(256), 0, st, (const float*)part, grid, dw);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_stem_conv_wrw_bn(const void* x, const void* da, const void* xc, const float* bp, float* dw, int64_t B, int64_t H,
                         int64_t W, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !da || !xc || !bp || !dw || !ws) return TSG_E_NULL;
  StemGeom g;
  if (!stem_geom(B, H, W, &g)) return TSG_E_SHAPE;
  if (ws_bytes < tsg_stem_conv_ws_bytes()) return TSG_E_WS;
  if (!aligned16(da) || !aligned16(xc) || !aligned16(ws) || (((uintptr_t)x) & 3u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)((char*)ws + sc_align((size_t)SC_OC * SC_KP * sizeof(bf16_t)));
  const int grid = g.ntiles < SC_NPART ? g.ntiles : SC_NPART;
  hipLaunchKernelGGL(stem_wrw_k<true>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)da, part, g,
                     (const bf16_t*)xc, bp);
  TSG_CHECK_LAUNCH();
  hipLaunchKernelGGL(stem_wrw_fold, dim3(SC_OC * SC_KP / 64), dim3(256), 0, st, (const float*)part, grid, dw);
  TSG_CHECK_LAUNCH();
  return 0;
}

// ---- the recomputing ResNet stem (round 6) ----------------------------------------------------------------------
static const int kStemPoolBlocks = 512;                // 2 resident blocks per CU (launch bounds of the kernels above)

static bool pool_geom(const StemGeom& g, PoolGeom* pg) {
  pg->PH = (g.OH - 1) / 2 + 1; pg->PW = (g.OW - 1) / 2 + 1;
  pg->tph = (pg->PH + FP_PRT - 1) / FP_PRT; pg->tpw = (pg->PW + FP_PCT - 1) / FP_PCT;
  const int64_t nt = (int64_t)g.B * pg->tph * pg->tpw;
  if (nt > 0x7fffffffLL) return false;
  pg->ntiles = (int)nt;
  return true;
}

int tsg_stem_conv_stats(const void* x, const float* w, float* partial, int64_t B, int64_t H, int64_t W, void* ws,
                        size_t ws_bytes, void* stream) {
  if (!x || !w || !ws || !partial) return TSG_E_NULL;
  StemGeom g;
  if (!stem_geom(B, H, W, &g)) return TSG_E_SHAPE;
  if (ws_bytes < sc_align((size_t)SC_OC * SC_KP * sizeof(bf16_t))) return TSG_E_WS;
  if (!aligned16(ws) || (((uintptr_t)x) & 3u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* wp = (bf16_t*)ws;
  hipLaunchKernelGGL(stem_pack_w, dim3((SC_OC * SC_KP + 255) / 256), dim3(256), 0, st, w, wp);
  TSG_CHECK_LAUNCH();
  const int grid = g.ntiles < 768 ? g.ntiles : 768;    // the partial rows of tsg_stem_conv_fwd_stats, bit for bit
  hipLaunchKernelGGL((stem_fwd_k<true, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp,
                     (bf16_t*)nullptr, g, partial);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_stem_conv_bn_relu_pool_fwd(const void* x, const float* w, const float* fp, void* ypool, void* argmax_u8, int64_t B,
                                   int64_t H, int64_t W, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !w || !fp || !ypool || !argmax_u8 || !ws) return TSG_E_NULL;
  StemGeom g;
  PoolGeom pg;
  if (!stem_geom(B, H, W, &g) || !pool_geom(g, &pg)) return TSG_E_SHAPE;
  if (ws_bytes < sc_align((size_t)SC_OC * SC_KP * sizeof(bf16_t))) return TSG_E_WS;
  if (!aligned16(ypool) || !aligned16(ws) || (((uintptr_t)x) & 3u) || (((uintptr_t)argmax_u8) & 7u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* wp = (bf16_t*)ws;
  hipLaunchKernelGGL(stem_pack_w, dim3((SC_OC * SC_KP + 255) / 256), dim3(256), 0, st, w, wp);
  TSG_CHECK_LAUNCH();
  const int grid = pg.ntiles < kStemPoolBlocks ? pg.ntiles : kStemPoolBlocks;
  hipLaunchKernelGGL(stem_fwd_pool_k, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp, (bf16_t*)ypool,
                     (uint8_t*)argmax_u8, g, pg, fp);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_stem_pool_bwd_num_partials(int64_t B, int64_t H, int64_t W) {
  StemGeom g;
  if (!stem_geom(B, H, W, &g)) return TSG_E_SHAPE;
  return g.ntiles < kStemPoolBlocks ? g.ntiles : kStemPoolBlocks;
}

int tsg_stem_conv_bn_relu_pool_bwd_reduce(const void* x, const float* w, const void* dpool, const void* argmax_u8,
                                          const float* fp, float* partial, int64_t B, int64_t H, int64_t W, void* ws,
                                          size_t ws_bytes, void* stream) {
  if (!x || !w || !dpool || !argmax_u8 || !fp || !partial || !ws) return TSG_E_NULL;
  StemGeom g;
  PoolGeom pg;
  if (!stem_geom(B, H, W, &g) || !pool_geom(g, &pg)) return TSG_E_SHAPE;
  if (ws_bytes < sc_align((size_t)SC_OC * SC_KP * sizeof(bf16_t))) return TSG_E_WS;
  if (!aligned16(dpool) || !aligned16(ws) || (((uintptr_t)x) & 3u) || (((uintptr_t)argmax_u8) & 7u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* wp = (bf16_t*)ws;
  hipLaunchKernelGGL(stem_pack_w, dim3((SC_OC * SC_KP + 255) / 256), dim3(256), 0, st, w, wp);
  TSG_CHECK_LAUNCH();
  const int grid = g.ntiles < kStemPoolBlocks ? g.ntiles : kStemPoolBlocks;
  hipLaunchKernelGGL(stem_pool_bwd_reduce_k, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp,
                     (const bf16_t*)dpool, (const uint8_t*)argmax_u8, g, pg.PH, pg.PW, fp, partial);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_stem_conv_wrw_bn_pool(const void* x, const float* w, const void* xc, const void* dpool, const void* argmax_u8,
                              const float* bp, float* dw, int64_t B, int64_t H, int64_t W, void* ws, size_t ws_bytes,
                              void* stream) {
  if (!x || !w || !dpool || !argmax_u8 || !bp || !dw || !ws) return TSG_E_NULL;
  if (xc && !aligned16(xc)) return TSG_E_ALIGN;
  StemGeom g;
  PoolGeom pg;
  if (!stem_geom(B, H, W, &g) || !pool_geom(g, &pg)) return TSG_E_SHAPE;
  if (ws_bytes < tsg_stem_conv_ws_bytes()) return TSG_E_WS;
  if (!aligned16(dpool) || !aligned16(ws) || (((uintptr_t)x) & 3u) || (((uintptr_t)argmax_u8) & 7u)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* wp = (bf16_t*)ws;
  float* part = (float*)((char*)ws + sc_align((size_t)SC_OC * SC_KP * sizeof(bf16_t)));
  hipLaunchKernelGGL(stem_pack_w, dim3((SC_OC * SC_KP + 255) / 256), dim3(256), 0, st, w, wp);
  TSG_CHECK_LAUNCH();
  int grid;
  if (xc) {                                            // y is read: 3 resident blocks per CU, the tile partition of tsg_stem_conv_wrw
    grid = g.ntiles < SC_NPART ? g.ntiles : SC_NPART;
    hipLaunchKernelGGL((stem_wrw_pool_k<false, 3>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp,
                       (const bf16_t*)dpool, (const uint8_t*)argmax_u8, g, pg.PH, pg.PW, bp, part, (const bf16_t*)xc);
  } else {
    grid = g.ntiles < kStemPoolBlocks ? g.ntiles : kStemPoolBlocks;
    hipLaunchKernelGGL((stem_wrw_pool_k<true, 2>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)wp,
                       (const bf16_t*)dpool, (const uint8_t*)argmax_u8, g, pg.PH, pg.PW, bp, part, (const bf16_t*)nullptr);
  }
  TSG_CHECK_LAUNCH();
  hipLaunchKernelGGL(stem_wrw_fold, dim3(SC_OC * SC_KP / 64), dim3(256), 0, st, (const float*)part, grid, dw);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
