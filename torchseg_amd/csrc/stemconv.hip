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
template <bool STATS>
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
#pragma unroll
    for (int qd = 0; qd < SC_TH; ++qd)
      if (tp.oh0 + qd < g.OH && colok)
        *reinterpret_cast<uint4*>(yt + ((int64_t)qd * g.OW + spl) * SC_OC + spart * 8) =
            *reinterpret_cast<const uint4*>(outs + (qd * SC_TW + spl) * 72 + spart * 8);
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

}  // extern "C"
