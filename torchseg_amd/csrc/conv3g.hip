// General 3x3 / stride 1 / padding 1 convolution, forward, on the bf16 MFMA, channels_last — every 3x3 layer of the
// networks with C_in a multiple of 16 and C_out a multiple of 64 that csrc/conv64.hip (64 -> 64) does not cover:
// ResNet-18 layer2-4 (furnace/base_model/resnet.py:24-29,36-53: 128 / 256 / 512 channels on 128^2 / 64^2 / 32^2 maps at
// BASELINE config 2), BiSeNet's attention-refinement 3x3, refines and heads (bisenet network.py:43-52,140-156).  The
// same kernel computes their DATA gradient: a stride-1 3x3 data gradient is the forward convolution of dy with the
// 180-degree-rotated, transposed filter, which tsg_conv3x3_gen_prep_filter(mode 1) writes.
//
// Round 2 left these layers on the vendor library (0.62-0.91 PF forward, 25-35 % slower backward, no way to attach the
// BatchNorm statistics of the output or the BatchNorm + ReLU of the input).  Shape of this kernel:
//   * implicit GEMM, D[oc][pixel] += W[oc][tap, ci] * X[tap, ci][pixel]: A operand = filter, B operand = pixels, 32x32x16
//     MFMAs, one K step = 16 input channels of one tap;
//   * a block (8 waves, 2 per SIMD) owns 128 (or 64) output channels x an 8 x 32 pixel tile and walks the input channels
//     in chunks of 16: per chunk the (8+2) x (32+2) x 16 input patch is staged ONCE and serves all nine taps from LDS
//     (48-byte pixel stride: the 32 lanes of a ds_read_b128 fragment fall on distinct bank groups), and the 9 x BN x 16
//     filter slab is a LINEAR copy: tsg_conv3x3_gen_prep_filter lays the filter out in MFMA fragment order
//     [oc tile][chunk][tap][32-channel block][lane][8], so an A fragment is 1 KB of consecutive LDS (conflict-free by
//     construction) — 166 staged bytes per MFMA against 512 for a 128 x 128 x 64 GEMM tile;
//   * both LDS images are double-buffered: the global loads of chunk c+1 are issued before the 36 MFMAs per wave of
//     chunk c and written behind them, one barrier per chunk;
//   * a wave computes 64 oc x 2 rows x 32 pixels (4 accumulators); the 12 pixel fragments of a chunk (4 patch rows x 3
//     column shifts) are read once and reused by the taps that share them: 30 LDS reads per 36 MFMAs;
//   * blocks are persistent over the pixel tiles of one oc tile, mapped XCD-aware (blocks that read the same pixels for
//     different oc tiles run on the same XCD at the same time, so the second read of x is an L2 hit);
//   * epilogue through LDS (the filter buffers) so that every lane stores 16 B of NHWC; STATS: per-channel sum / square
//     sum of the bf16-rounded outputs (the statistics pass of the SyncBatchNorm that follows these convolutions,
//     legacy/sync_bn/syncbn.py:86-98) accumulated over the block's tiles, partial[slot][2][C_out];
//   * AFF: the input is relu(a x + b) of the tensor that is read (BatchNorm + ReLU in front of the convolution,
//     resnet.py:36-46), applied while the patch is written to LDS with the values tsg_bn_apply_fwd would have stored.
// x, y: NHWC bf16.  wf: the prepared filter.  fp32 accumulation, one rounding to bf16 at the store.
#include "tsg_common.h"
#include <stdlib.h>
#include <type_traits>

namespace tsg {

typedef __attribute__((ext_vector_type(8))) __bf16 g3_bf16x8;
typedef __attribute__((ext_vector_type(16))) float g3_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int g3_u32x4;

constexpr int G3_TH = 8, G3_TW = 32;                     // output tile: 256 pixels
constexpr int G3_PH = G3_TH + 2, G3_PW = G3_TW + 2;      // input patch
constexpr int G3_KC = 16;                                // input channels per chunk
constexpr int G3_PS = 24;                                // LDS pixel stride in bf16: 32 B of data + 16 B of padding
constexpr int G3_NPX = G3_PH * G3_PW;                    // 340 patch pixels
constexpr int G3_PATCH = G3_NPX * G3_PS;                 // bf16 elements of one patch buffer (16,320 B)
constexpr int G3_NPV = G3_NPX * 2;                       // 16-byte vectors of a patch chunk: 680
constexpr int G3_MAX_AFF_C = 512;                        // normalise-on-load: a / b rows staged in LDS

struct G3Geom {
  int B, H, W, Cin, Cout;
  int tiles_h, tiles_w, ntiles;                          // pixel tiles (per oc tile)
  int nchunks, noct, nslots;                             // Cin / 16, Cout / BN, persistent blocks per oc tile
  int prio;                                              // s_setprio 1 around the 36 MFMAs of a chunk (TSG_MFMA_PRIO)
};

// BN output channels per block, NW waves: 8 waves = 2 (oc halves) x 4 (row pairs), one block per CU; 4 waves = 1 x 4 with
// BN = 64, two blocks per CU that run out of phase and cover each other's staging / barriers (the conv64 finding)
template <int BN, int NW> struct G3Cfg {
  static constexpr int THREADS = 64 * NW;
  static constexpr int WO = NW / 4;                      // waves along the output channels
  static constexpr int NOB = BN / 32 / WO;               // 32-channel blocks per wave
  static constexpr int FELEMS = 9 * BN * G3_KC;          // bf16 elements of one filter slab
  static constexpr int NFV = FELEMS / 8;                 // 16-byte vectors of a filter slab: 2304 / 1152
  static constexpr int NFU = (NFV + THREADS - 1) / THREADS;          // per thread: 5 / 3 / 5 (the last one partly)
  static constexpr int NPU = (G3_NPV + THREADS - 1) / THREADS;       // patch vectors per thread: 2 / 3
  static constexpr int OS = BN + 8;                      // epilogue staging: bf16 per pixel row
  static constexpr size_t LDS = (size_t)(2 * FELEMS + 2 * G3_PATCH) * 2 + 2 * G3_MAX_AFF_C * 4;
  static_assert(256 * OS <= 2 * FELEMS, "the output tile is staged in the two filter buffers");
};

// bf16(bf16 a + bf16 b) per element, fp32 add: what the eager `a + b` of two bf16 tensors computes
__device__ __forceinline__ uint4 g3_add_bf16x8(uint4 a, uint4 b) {
  const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    o[i] = pack2_bf16(__uint_as_float(x[i] << 16) + __uint_as_float(y[i] << 16),
                      __uint_as_float(x[i] & 0xffff0000u) + __uint_as_float(y[i] & 0xffff0000u));
  return make_uint4(o[0], o[1], o[2], o[3]);
}

__device__ __forceinline__ uint4 g3_affine_relu(uint4 v, const float* __restrict__ a, const float* __restrict__ b) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = __uint_as_float(w[i] << 16), x1 = __uint_as_float(w[i] & 0xffff0000u);
    const float y0 = fmaf(x0, a[2 * i], b[2 * i]), y1 = fmaf(x1, a[2 * i + 1], b[2 * i + 1]);
    w[i] = pack2_bf16(y0 > 0.f ? y0 : 0.f, y1 > 0.f ? y1 : 0.f);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// GLDS: the filter slab goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, destination =
// wave-uniform base + lane x 16 B — exactly the fragment-ordered slab, which is a linear copy), not through registers and
// ds_write_b128.  63 % of the bytes a block stages per chunk then bypass the register -> LDS write path (79 B / clk / CU,
// the path the PSA ablations found saturated) and 12-20 VGPRs are free.
template <int BN, int NW, bool AFF, bool STATS, bool GLDS>
__global__ __launch_bounds__(64 * NW, 2) void conv3g_fwd_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wf,
                                                             bf16_t* __restrict__ y, G3Geom g,
                                                             const float* __restrict__ in_ab,
                                                             float* __restrict__ partial,
                                                             const bf16_t* __restrict__ addend) {
  using Cfg = G3Cfg<BN, NW>;
  constexpr int NOB = Cfg::NOB, NFU = Cfg::NFU, NPU = Cfg::NPU, OS = Cfg::OS, OCB = BN / 32, G3_THREADS = Cfg::THREADS;
  extern __shared__ __attribute__((aligned(16))) unsigned char g3_smem[];
  bf16_t* fbuf = reinterpret_cast<bf16_t*>(g3_smem);                       // [2][FELEMS]
  bf16_t* pbuf = fbuf + 2 * Cfg::FELEMS;                                   // [2][G3_PATCH]
  float* abs_ = reinterpret_cast<float*>(pbuf + 2 * G3_PATCH);             // [2][Cin] (AFF)
  bf16_t* outs = fbuf;                                                     // epilogue: [256 pixels][OS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int wo = wave % Cfg::WO, wp = wave / Cfg::WO;    // oc part of the tile, row pair of the tile
  // XCD-aware persistent mapping: the blocks of one slot (same pixel tiles, all oc tiles) sit on one XCD
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int oct = jb % g.noct, slot = (jb / g.noct) * 8 + xcd;

  if (AFF) {
    for (int i = tid; i < 2 * g.Cin; i += G3_THREADS) abs_[i] = in_ab[i];
  }

  // staging descriptors
  int prc[NPU];                                          // patch vector u: pr | pc << 8 | part << 16, or -1
#pragma unroll
  for (int u = 0; u < NPU; ++u) {
    const int v = tid + G3_THREADS * u, pp = v >> 1;
    prc[u] = v < G3_NPV ? ((pp / G3_PW) | ((pp % G3_PW) << 8) | ((v & 1) << 16)) : -1;
  }
  uint4 rf[GLDS ? 1 : NFU], rp[NPU];
  constexpr int NFW = (Cfg::NFV / 64 + NW - 1) / NW;       // GLDS: 1 KB pieces of a slab per wave (5 / 3 / 5)
  const bf16_t* wslab = wf + (int64_t)oct * g.nchunks * Cfg::FELEMS;

  // STATS: the thread's 8 channels (16-byte part tid % (BN / 8)) over the pixels it stores: tid / (BN / 8) + k * 4096 / BN
  float st1[8], st2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { st1[e] = 0.f; st2[e] = 0.f; }

  for (int tile = slot; tile < g.ntiles; tile += g.nslots) {
    const int ow0 = (tile % g.tiles_w) * G3_TW, oh0 = ((tile / g.tiles_w) % g.tiles_h) * G3_TH;
    const int bimg = tile / (g.tiles_w * g.tiles_h);
    const bf16_t* ximg = x + (int64_t)bimg * g.H * g.W * g.Cin;

    auto fetch = [&](int chunk, int buf) {
      const bf16_t* ws = wslab + (int64_t)chunk * Cfg::FELEMS;
      if (GLDS) {
        unsigned char* fb = reinterpret_cast<unsigned char*>(fbuf + buf * Cfg::FELEMS);
#pragma unroll
        for (int u = 0; u < NFW; ++u) {
          const int q = wave + NW * u;                           // wave-uniform piece index
          if (u < NFW - 1 || q < Cfg::NFV / 64)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ws + ((int64_t)q * 64 + lane) * 8),
                                             (__attribute__((address_space(3))) void*)(fb + q * 1024), 16, 0, 0);
        }
      } else {
#pragma unroll
        for (int u = 0; u < NFU; ++u) {
          const int v = tid + G3_THREADS * u;
          rf[u] = make_uint4(0u, 0u, 0u, 0u);
          if (u < NFU - 1 || v < Cfg::NFV) rf[u] = *reinterpret_cast<const uint4*>(ws + (int64_t)v * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < NPU; ++u) {
        const int ih = oh0 - 1 + (prc[u] & 0xff), iw = ow0 - 1 + ((prc[u] >> 8) & 0xff);
        rp[u] = make_uint4(0u, 0u, 0u, 0u);
        if (prc[u] >= 0 && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
          rp[u] = *reinterpret_cast<const uint4*>(ximg + ((int64_t)ih * g.W + iw) * g.Cin + chunk * G3_KC +
                                                  ((prc[u] >> 16) & 1) * 8);
      }
    };
    auto stage = [&](int chunk, int buf) {
      if (!GLDS) {
        bf16_t* fb = fbuf + buf * Cfg::FELEMS;
#pragma unroll
        for (int u = 0; u < NFU; ++u) {
          const int v = tid + G3_THREADS * u;
          if (u < NFU - 1 || v < Cfg::NFV) *reinterpret_cast<uint4*>(fb + v * 8) = rf[u];
        }
      }
      bf16_t* pbw = pbuf + buf * G3_PATCH;
#pragma unroll
      for (int u = 0; u < NPU; ++u)
        if (prc[u] >= 0) {
          uint4 v = rp[u];
          const int part = (prc[u] >> 16) & 1;
          if (AFF) {
            const int ih = oh0 - 1 + (prc[u] & 0xff), iw = ow0 - 1 + ((prc[u] >> 8) & 0xff);
            if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) {     // padding pixels stay exactly zero
              const int c0 = chunk * G3_KC + part * 8;
              v = g3_affine_relu(v, abs_ + c0, abs_ + g.Cin + c0);
            }
          }
          const int pp = (prc[u] & 0xff) * G3_PW + ((prc[u] >> 8) & 0xff);
          *reinterpret_cast<uint4*>(pbw + pp * G3_PS + part * 8) = v;
        }
    };

    g3_f32x16 acc[NOB][2];
#pragma unroll
    for (int j = 0; j < NOB; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    if (GLDS) __syncthreads();                           // the DMA writes LDS at once: the previous tile's epilogue must be done
    fetch(0, 0);
    if (!GLDS) __syncthreads();                          // the previous tile's epilogue is done with the buffers; abs_ visible
    stage(0, 0);
    __syncthreads();

    for (int c = 0; c < g.nchunks; ++c) {
      const int buf = c & 1;
      if (c + 1 < g.nchunks) fetch(c + 1, buf ^ 1);      // in flight during the MFMAs of this chunk
      const bf16_t* pb = pbuf + buf * G3_PATCH + ((2 * wp) * G3_PW + p) * G3_PS + half * 8;
      const bf16_t* fa = fbuf + buf * Cfg::FELEMS + ((wo * NOB) * 64 + lane) * 8;
      g3_bf16x8 bq[4][3];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          bq[pr][kw] = *reinterpret_cast<const g3_bf16x8*>(pb + (pr * G3_PW + kw) * G3_PS);
      // the filter fragments of tap t + 1 are read while the MFMAs of tap t run (two register sets; with one, every tap
      // began with an exposed LDS latency: ds_read, s_waitcnt lgkmcnt(0), 4 MFMAs in the ISA)
      g3_bf16x8 af[2][NOB];
#pragma unroll
      for (int j = 0; j < NOB; ++j) af[0][j] = *reinterpret_cast<const g3_bf16x8*>(fa + (j * 64) * 8);
      // (the scheduler otherwise sinks every read to just before its first use, to save registers this kernel has to spare)
      __builtin_amdgcn_sched_barrier(0);
      // two independent blocks share a CU: the one inside its MFMA cluster outranks the one that is staging (guide T5:
      // pays only where waves are at DIFFERENT phases, which is exactly the two-blocks-per-CU arrangement)
      if (g.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int kh = t / 3, kw = t % 3;
        if (t + 1 < 9) {
#pragma unroll
          for (int j = 0; j < NOB; ++j)
            af[(t + 1) & 1][j] = *reinterpret_cast<const g3_bf16x8*>(fa + (((t + 1) * OCB + j) * 64) * 8);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < NOB; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t & 1][j], bq[i + kh][kw], acc[j][i], 0, 0, 0);
      }
      if (g.prio) __builtin_amdgcn_s_setprio(0);
      if (c + 1 < g.nchunks) stage(c + 1, buf ^ 1);
      __syncthreads();
    }

    // ---- epilogue: acc[j][i][r] is oc = (wo NOB + j) 32 + (r & 3) + 8 (r >> 2) + 4 half, pixel (row 2 wp + i, column p)
#pragma unroll
    for (int j = 0; j < NOB; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          uint2 v;
          v.x = pack2_bf16(acc[j][i][4 * gq + 0], acc[j][i][4 * gq + 1]);
          v.y = pack2_bf16(acc[j][i][4 * gq + 2], acc[j][i][4 * gq + 3]);
          *reinterpret_cast<uint2*>(outs + ((2 * wp + i) * G3_TW + p) * OS + (wo * NOB + j) * 32 + 8 * gq + 4 * half) = v;
        }
    __syncthreads();
    constexpr int VPP = BN / 8;                          // 16-byte vectors per pixel
    const int64_t img_off = (int64_t)bimg * g.H * g.W * g.Cout + oct * BN;
    bf16_t* yimg = y + img_off;
#pragma unroll
    for (int k = 0; k < 256 * VPP / G3_THREADS; ++k) {
      const int v = tid + G3_THREADS * k, px = v / VPP, part = v % VPP;
      const int oh = oh0 + (px >> 5), ow = ow0 + (px & 31);
      if (oh < g.H && ow < g.W) {
        uint4 o = *reinterpret_cast<const uint4*>(outs + px * OS + part * 8);
        const int64_t off = ((int64_t)oh * g.W + ow) * g.Cout + part * 8;
        // addend: y = bf16(bf16(conv) + addend) — the gradient that reaches the same tensor through a skip connection
        // (resnet.py:48-52: out += residual), summed here instead of by a separate pass over three tensors
        if (addend) o = g3_add_bf16x8(o, *reinterpret_cast<const uint4*>(addend + img_off + off));
        *reinterpret_cast<uint4*>(yimg + off) = o;
        if (STATS) {                                     // the values just stored: no second pass over the tile
          const uint32_t w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(w[e] << 16), hi = __uint_as_float(w[e] & 0xffff0000u);
            st1[2 * e] += lo; st2[2 * e] = fmaf(lo, lo, st2[2 * e]);
            st1[2 * e + 1] += hi; st2[2 * e + 1] = fmaf(hi, hi, st2[2 * e + 1]);
          }
        }
      }
    }
    // the next tile's first __syncthreads() (after its fetch) orders these reads before the buffers are rewritten
  }

  if (STATS) {                                           // fold the pixel groups in a fixed order
    constexpr int VPP = BN / 8, NG = G3_THREADS / VPP;    // 32 / 64 pixel groups
    __syncthreads();
    float* red = reinterpret_cast<float*>(fbuf);         // [NG][2][BN] floats: <= 32 KB, the filter buffers are free now
    const int part = tid % VPP, q = tid / VPP;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(q * 2 + 0) * BN + part * 8 + e] = st1[e];
      red[(q * 2 + 1) * BN + part * 8 + e] = st2[e];
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int c = tid % BN, which = tid / BN;
      float sacc = 0.f;
      for (int qq = 0; qq < NG; ++qq) sacc += red[(qq * 2 + which) * BN + c];
      partial[((int64_t)slot * 2 + which) * g.Cout + oct * BN + c] = sacc;
    }
  }
}

// ---- second structure for the large maps (>= 512 block tiles of 64 oc x 16 x 32 pixels): conv3h_fwd_k ----------------
// Counters of the kernel above on the 128-channel / 128^2 layers (profiles/r03_conv3g_sq_counters.txt): MFMA pipe busy 0.52,
// the waves wait for instruction issue 43 % of their cycles.  The ISA of its K loop shows why: per chunk of 36 MFMAs a wave
// runs ~75 address / bounds / branch instructions for its three patch loads and five DMA pieces, then waits for 16 fragment
// reads with nothing to issue, then 3 ds_write_b128 behind exec branches, then the barrier: roughly 700 cycles of overhead
// against 1,152 cycles of MFMA, and two waves per SIMD do not interleave well enough to hide that.  This kernel halves the
// overhead per MFMA and removes the per-chunk VALU work:
//   * a wave owns 64 oc x 4 rows x 32 pixels = 8 accumulators: 72 MFMAs (2,304 cycles) per chunk and barrier, 36 fragment
//     reads instead of 60 for the same MFMAs (the six patch rows of one column shift serve three kernel rows);
//   * BOTH operands arrive by LDS-DMA (buffer_load_dwordx4 ... lds): the filter slab as before, the (16+2) x (32+2) x 16
//     patch as 20 pieces of 1 KB whose per-lane source offsets are computed ONCE per tile; out-of-image pixels carry an
//     offset beyond the buffer's num_records, for which the load returns zeros (the padding), so a chunk's staging is ten
//     DMA instructions per wave with an SGPR chunk offset: no registers, no ds_write, no branches;
//   * the patch is stored unpadded (32 B per pixel) so that 2 x (filter + patch) = 76 KB and two blocks still share a CU;
//     the B-fragment reads are then 2-way bank conflicted (8 LDS cycles instead of 4), which the LDS pipe has to spare
//     (it is ~35 % busy with them);
//   * the DMA pieces of chunk c+1 are issued between the first MFMA groups of chunk c, the fragment reads of MFMA group
//     s+1 before group s: after the eight reads that open a chunk every instruction sits in an MFMA shadow;
//   * STATS: the channel sums of a tile are folded across lanes (same 16-byte part) and waves through a 2 KB corner of
//     LDS into ONE register per thread instead of sixteen.
// Same prepared filter (BN = 64 layout), same partial[slot][2][C_out] contract and slot count as conv3g_fwd_k.
constexpr int H3_TH = 16, H3_PH = H3_TH + 2;             // output rows per tile, patch rows
constexpr int H3_NPX = H3_PH * G3_PW;                    // 612 patch pixels
constexpr int H3_PPIECES = 20;                           // 1 KB DMA pieces of a patch chunk (612 x 32 B = 19,584 B)
constexpr int H3_PBYTES = H3_PPIECES * 1024;
constexpr int H3_FBYTES = 9 * 64 * G3_KC * 2;            // 18,432
constexpr int H3_FPIECES = H3_FBYTES / 1024;             // 18
constexpr int H3_OS = 72;                                // epilogue staging: bf16 per pixel row
constexpr int H3_RED = 2 * H3_FBYTES + 2 * H3_PBYTES;               // float [4 waves][2][64] behind the staging buffers
constexpr size_t H3_LDS = H3_RED + 2048;                           // 79,872 B: two blocks per CU (<= 81,920 each)
static_assert(512 * H3_OS * 2 <= H3_RED, "the output tile is staged in the filter and patch buffers");

template <bool STATS>
__global__ __launch_bounds__(256, 2) void conv3h_fwd_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wf,
                                                        bf16_t* __restrict__ y, G3Geom g, float* __restrict__ partial,
                                                        const bf16_t* __restrict__ addend) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char h3_smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int oct = jb % g.noct, slot = (jb / g.noct) * 8 + xcd;
  bf16_t* outs = reinterpret_cast<bf16_t*>(h3_smem);
  float* red = reinterpret_cast<float*>(h3_smem + H3_RED);

  // the block's filter slabs [chunk][tap][ocb][lane][8] as one buffer; piece q of chunk c at byte (c 18 + q) 1024
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(wf + (int64_t)oct * g.nchunks * (H3_FBYTES / 2)), 0, g.nchunks * H3_FBYTES, 0x00020000);
  const unsigned char* fa = h3_smem + lane * 16;                                           // A fragment (tap, ocb) at + (tap 2 + ocb) 1024
  const unsigned char* pb = h3_smem + 2 * H3_FBYTES + ((4 * wave) * G3_PW + p) * 32 + half * 16;   // B (row, shift) at + (row 34 + shift) 32

  float stacc = 0.f;                                     // STATS: thread (c = tid & 63, which = tid >> 6 & 1) of the first 128
  bool red_full = false;

  for (int tile = slot; tile < g.ntiles; tile += g.nslots) {
    const int ow0 = (tile % g.tiles_w) * G3_TW, oh0 = ((tile / g.tiles_w) % g.tiles_h) * H3_TH;
    const int bimg = tile / (g.tiles_w * g.tiles_h);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (int64_t)bimg * g.H * g.W * g.Cin), 0, g.H * g.W * g.Cin * 2, 0x00020000);
    int voff[5];                                          // patch piece wave + 4 u: this lane's 16 bytes
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int v = (wave + 4 * u) * 64 + lane, q = v >> 1;
      const int ih = oh0 - 1 + q / G3_PW, iw = ow0 - 1 + q % G3_PW;
      const bool ok = q < H3_NPX && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
      voff[u] = ok ? (ih * g.W + iw) * g.Cin * 2 + (v & 1) * 16 : (int)0x80000000;
    }
    auto dma_patch = [&](int u, int chunk, int buf) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(h3_smem + 2 * H3_FBYTES + buf * H3_PBYTES + (wave + 4 * u) * 1024),
                                               16, voff[u], chunk * (G3_KC * 2), 0, 0);
    };
    auto dma_filter = [&](int u, int chunk, int buf) {
      const int q = wave + 4 * u;
      if (u < 4 || q < H3_FPIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(h3_smem + buf * H3_FBYTES + q * 1024), 16, lane * 16,
                                                 (chunk * H3_FPIECES + q) * 1024, 0, 0);
    };

    g3_f32x16 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

    __syncthreads();                                     // the previous tile's epilogue is done with LDS
    if (STATS && red_full && tid < 128) {                // fold the previous tile's four wave sums (order fixed)
      const int c = tid & 63, which = tid >> 6;
      stacc += (red[(0 * 2 + which) * 64 + c] + red[(1 * 2 + which) * 64 + c]) +
               (red[(2 * 2 + which) * 64 + c] + red[(3 * 2 + which) * 64 + c]);
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) dma_patch(u, 0, 0);
#pragma unroll
    for (int u = 0; u < 5; ++u) dma_filter(u, 0, 0);
    __syncthreads();                                     // (waits for the DMA: vmcnt(0) in front of the barrier)

    auto chunk_body = [&](auto BUFC, int c) {
      constexpr int buf = decltype(BUFC)::value;
      const bool more = c + 1 < g.nchunks;
      const unsigned char* fab = fa + buf * H3_FBYTES;
      const unsigned char* pbb = pb + buf * H3_PBYTES;
      g3_bf16x8 bq[2][6], af[2][2];
#pragma unroll
      for (int r = 0; r < 6; ++r) bq[0][r] = *reinterpret_cast<const g3_bf16x8*>(pbb + (r * G3_PW) * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j) af[0][j] = *reinterpret_cast<const g3_bf16x8*>(fab + j * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 9; ++s) {                      // MFMA group s: column shift kw = s / 3, kernel row kh = s % 3
        const int kw = s / 3, kh = s % 3;
        if (s + 1 < 9) {
          const int t1 = ((s + 1) % 3) * 3 + (s + 1) / 3;
#pragma unroll
          for (int j = 0; j < 2; ++j) af[(s + 1) & 1][j] = *reinterpret_cast<const g3_bf16x8*>(fab + (t1 * 2 + j) * 1024);
        }
        if (kw + 1 < 3) {
#pragma unroll
          for (int e = 0; e < 2; ++e)
            bq[(kw + 1) & 1][2 * kh + e] = *reinterpret_cast<const g3_bf16x8*>(pbb + ((2 * kh + e) * G3_PW + kw + 1) * 32);
        }
        if (more) {                                      // chunk c + 1: the patch (HBM / L2) first, the filter (L2) behind it
          if (s == 0) { dma_patch(0, c + 1, buf ^ 1); dma_patch(1, c + 1, buf ^ 1); }
          if (s == 1) { dma_patch(2, c + 1, buf ^ 1); dma_patch(3, c + 1, buf ^ 1); }
          if (s == 2) { dma_patch(4, c + 1, buf ^ 1); dma_filter(0, c + 1, buf ^ 1); }
          if (s == 3) { dma_filter(1, c + 1, buf ^ 1); dma_filter(2, c + 1, buf ^ 1); }
          if (s == 4) { dma_filter(3, c + 1, buf ^ 1); dma_filter(4, c + 1, buf ^ 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][j], bq[kw & 1][i + kh], acc[j][i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    };
    for (int c = 0; c < g.nchunks; c += 2) {             // nchunks is even (host check): buffer indices are immediates
      chunk_body(std::integral_constant<int, 0>{}, c);
      chunk_body(std::integral_constant<int, 1>{}, c + 1);
    }

    // ---- epilogue: acc[j][i][r] is oc = j 32 + (r & 3) + 8 (r >> 2) + 4 half, pixel (row 4 wave + i, column p)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // STATS: pixels of a partial tile that lie outside the image are staged as zeros (they are not stored, and the
      // statistics below read the staged tile)
      const bool inside = !STATS || (oh0 + 4 * wave + i < g.H && ow0 + p < g.W);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          uint2 v;
          v.x = pack2_bf16(acc[j][i][4 * gq + 0], acc[j][i][4 * gq + 1]);
          v.y = pack2_bf16(acc[j][i][4 * gq + 2], acc[j][i][4 * gq + 3]);
          if (!inside) v = make_uint2(0u, 0u);
          *reinterpret_cast<uint2*>(outs + ((4 * wave + i) * G3_TW + p) * H3_OS + j * 32 + 8 * gq + 4 * half) = v;
        }
    }
    const int64_t img_off = (int64_t)bimg * g.H * g.W * g.Cout + oct * 64;
    bf16_t* yimg = y + img_off;
    const int part = tid & 7;
    // the addend's vectors are requested one group of four rows ahead of the stores that use them (the first group before
    // the barrier): their latency overlaps the staging instead of sitting in front of every store
    const bool has_add = !STATS && addend != nullptr;
    const int ow = ow0 + (tid >> 3);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(addend + (int64_t)bimg * g.H * g.W * g.Cout), 0, g.H * g.W * g.Cout * 2, 0x00020000);
    const int av0 = ow < g.W ? (oh0 * g.W + ow) * g.Cout * 2 + oct * 128 + part * 16 : (int)0x80000000;
    const int arow = g.W * g.Cout * 2;
    g3_u32x4 adc[4], adn[4];
    __builtin_amdgcn_sched_barrier(0);                   // (after the staging stores: the accumulators' registers are free)
    if (has_add) {
#pragma unroll
      for (int e = 0; e < 4; ++e) adc[e] = __builtin_amdgcn_raw_buffer_load_b128(ra, av0, e * arow, 0);
    }
    __syncthreads();
    {
      bf16_t* yp = yimg + ((int64_t)oh0 * g.W + ow) * g.Cout + part * 8;       // row k of the tile: + k W C_out
      const bf16_t* op = outs + (tid >> 3) * H3_OS + part * 8;
      const int rows = ow < g.W ? min(16, g.H - oh0) : 0;
#pragma unroll 1
      for (int k4 = 0; k4 < 16; k4 += 4) {
        if (has_add && k4 < 12) {
#pragma unroll
          for (int e = 0; e < 4; ++e) adn[e] = __builtin_amdgcn_raw_buffer_load_b128(ra, av0, (k4 + 4 + e) * arow, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (k4 + e < rows) {
            uint4 o = *reinterpret_cast<const uint4*>(op + 32 * (k4 + e) * H3_OS);
            if (has_add) o = g3_add_bf16x8(o, make_uint4(adc[e][0], adc[e][1], adc[e][2], adc[e][3]));
            *reinterpret_cast<uint4*>(yp) = o;
          }
          yp += (int64_t)g.W * g.Cout;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) adc[e] = adn[e];
      }
    }
    if (STATS) {
      // Channel sums of the staged (bf16-rounded) tile ON THE MATRIX CORES: the wave's 128 pixels are the K dimension.  A
      // transposing read (the weight-gradient kernel's recipe, csrc/conv3wrw.hip) hands lane (c, half) the channel
      // cb 32 + c at 8 pixels; that register is both the A fragment Y^T[c][pixel] and the B fragment Y[pixel][c], so
      //   ones x Y  -> every row of D1 holds sum_px y[px][c]            (row 0: lanes 0-31, register 0)
      //   Y^T x Y   -> the diagonal of D2 holds sum_px y[px][c]^2       (exact products of bf16, fp32 accumulation)
      // 32 MFMAs + 32 LDS reads per wave and tile instead of unpack / add / fma per element and 48 cross-lane shuffles, and
      // no statistics registers live across the K loop; time-neutral at the bench shapes (+3 ... +9 us per launch for the
      // statistics either way, profiles/r04_conv3h.txt).
      typedef short h3_v4i16 __attribute__((ext_vector_type(4)));
      typedef h3_v4i16 __attribute__((address_space(3))) h3_lds_v4i16;
      const int sub = (lane >> 4) & 1, i16 = lane & 15;
      const bf16_t* fr = outs + (128 * wave + 8 * half + (i16 >> 2)) * H3_OS + 16 * sub + 4 * (i16 & 3);
      g3_f32x16 d1[2], d2[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { d1[cb][r] = 0.f; d2[cb][r] = 0.f; }
      union { uint32_t u[4]; g3_bf16x8 v; } ones;
#pragma unroll
      for (int e = 0; e < 4; ++e) ones.u[e] = 0x3f803f80u;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          union { h3_v4i16 q[2]; g3_bf16x8 v; } f;
          f.q[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h3_lds_v4i16*)(fr + (ks * 16 + 0) * H3_OS + cb * 32));
          f.q[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((h3_lds_v4i16*)(fr + (ks * 16 + 4) * H3_OS + cb * 32));
          d1[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones.v, f.v, d1[cb], 0, 0, 0);
          d2[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.v, f.v, d2[cb], 0, 0, 0);
        }
      // D row (r & 3) + 8 (r >> 2) + 4 half, column lane & 31: channel c's diagonal element sits in the lane with
      // half == (c >> 2 & 1), register (c & 3) + 4 (c >> 3)
      const int c = lane & 31, rc = (c & 3) + 4 * (c >> 3);
      const bool mine = half == ((c >> 2) & 1);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        float sq = d2[cb][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) sq = rc == r ? d2[cb][r] : sq;
        if (half == 0) red[(wave * 2 + 0) * 64 + cb * 32 + c] = d1[cb][0];
        if (mine) red[(wave * 2 + 1) * 64 + cb * 32 + c] = sq;
      }
      red_full = true;
    }
  }

  if (STATS) {
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, which = tid >> 6;
      if (red_full)
        stacc += (red[(0 * 2 + which) * 64 + c] + red[(1 * 2 + which) * 64 + c]) +
                 (red[(2 * 2 + which) * 64 + c] + red[(3 * 2 + which) * 64 + c]);
      partial[((int64_t)slot * 2 + which) * g.Cout + oct * 64 + c] = stacc;
    }
  }
}

// ---- data gradient of the STRIDE-2 3x3 / padding 1 convolutions: conv3s2d_k ---------------------------------------------
// ResNet's first block of layer2-4 (furnace/base_model/resnet.py:24-29 conv3x3(inplanes, planes, stride), :36-53) with 64 ->
// 128, 128 -> 256, 256 -> 512 channels: the vendor library's backward-data kernels run them at 0.13-0.2 PF (141 / 92 / 95 us
// at the bench shape, profiles/r04_eager_ops.txt) and autograd then adds the shortcut branch's gradient in a separate pass
// over three tensors (60 / 33 / 17 us).  dx by OUTPUT PARITY: with ih = 2 a + pa, iw = 2 b + pb,
//   dx[2a  ][2b  ] = dy[a][b] w11
//   dx[2a  ][2b+1] = dy[a][b+1] w10 + dy[a][b] w12
//   dx[2a+1][2b  ] = dy[a+1][b] w01 + dy[a][b] w21
//   dx[2a+1][2b+1] = dy[a+1][b+1] w00 + dy[a+1][b] w02 + dy[a][b+1] w20 + dy[a][b] w22
// (w_khkw = w[co][ci][kh][kw], summed over co): four small stride-1 correlations of the SAME dy patch, nine MFMA taps per
// dy pixel in all, no zero-stuffed operand.  A block owns 8 x 32 dy pixels (-> 16 x 64 of dx) x 32 dx channels; a wave 2
// dy rows = 4 parities x 2 rows = 8 accumulators; K = the convolution's output channels in chunks of 32 (two MFMA K
// steps per barrier: 36 MFMAs per wave).  Staging as in conv3h_fwd_k: everything by LDS-DMA, the (8+1) x (32+1) dy patch
// as two 16-channel planes of 32-byte pixels, out-of-image pixels by out-of-range buffer offsets; the filter is the
// mode-1 prepared filter at tile width 32 ([ci tile][chunk][tap'][lane][8] with tap' = 8 - (3 kh + kw)).  `addend`
// (the gradient that reaches x through the shortcut branch, resnet.py:48-52) joins in the epilogue.
constexpr int D2_TH = 8, D2_PR = D2_TH + 1, D2_PC = G3_TW + 1;      // dy tile rows, patch rows / columns
constexpr int D2_NPX = D2_PR * D2_PC;                    // 297 patch pixels
constexpr int D2_PVEC = D2_NPX * 2;                      // 16-byte vectors of one 16-channel plane: 594
constexpr int D2_PPIECES = (2 * D2_PVEC + 63) / 64;      // 19 pieces of 1 KB
constexpr int D2_PBYTES = D2_PPIECES * 1024;             // 19,456
constexpr int D2_FBYTES = 2 * 9 * 32 * G3_KC * 2;        // 18,432: two 16-channel slabs of a 32-wide tile
constexpr int D2_FPIECES = D2_FBYTES / 1024;             // 18
constexpr int D2_OS = 36;                                // epilogue staging: bf16 per dx pixel (32 + 4)
constexpr size_t D2_LDS = 2 * (D2_FBYTES + D2_PBYTES);   // 75,776 B: two blocks per CU
static_assert(16 * 64 * D2_OS * 2 <= D2_LDS, "the dx tile is staged in the filter and patch buffers");

// sub (round 6): the addend is COMPACT — [B, H, W, Cout] on dy's grid, the gradient of x[:, :, ::2, ::2] (the residual block's
// 1x1 / stride-2 shortcut convolution, resnet.py:139-146, run as a stride-1 convolution of the sub-sampled map): it is added at
// the even pixels of dx only, and the zero-filled full-size tensor the vendor library's stride-2 data gradient used to write
// (and this epilogue to read back) never exists.
__global__ __launch_bounds__(256, 2) void conv3s2d_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wf,
                                                      bf16_t* __restrict__ dx, G3Geom g, const bf16_t* __restrict__ addend,
                                                      int sub) {
  // g: B; H, W = dy's size; Cin = the convolution's OUTPUT channels (the K of this product); Cout = dx channels;
  //    tiles over dy; nchunks = Cin / 32; noct = Cout / 32; prio carries dx's height | width << 16
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char d2_smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
  const int oct = jb % g.noct, slot = (jb / g.noct) * 8 + xcd;
  const int XH = g.prio & 0xffff, XW = (g.prio >> 16) & 0xffff;
  bf16_t* outs = reinterpret_cast<bf16_t*>(d2_smem);

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(wf + (int64_t)oct * g.nchunks * (D2_FBYTES / 2)), 0, g.nchunks * D2_FBYTES, 0x00020000);
  const unsigned char* fa = d2_smem + lane * 16;                    // A (k step, tap') at + (ks 9 + tap') 1024
  const unsigned char* pb = d2_smem + D2_FBYTES + ((2 * wave) * D2_PC + p) * 32 + half * 16;   // B (ks, row, shift) at + ks 594 16 + (row 33 + shift) 32

  for (int tile = slot; tile < g.ntiles; tile += g.nslots) {
    const int b0 = (tile % g.tiles_w) * G3_TW, a0 = ((tile / g.tiles_w) % g.tiles_h) * D2_TH;
    const int bimg = tile / (g.tiles_w * g.tiles_h);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dy + (int64_t)bimg * g.H * g.W * g.Cin), 0, g.H * g.W * g.Cin * 2, 0x00020000);
    int voff[5];                                          // patch piece wave + 4 u: this lane's 16 bytes
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int v = (wave + 4 * u) * 64 + lane;
      const int ks = v >= D2_PVEC ? 1 : 0, vv = v - ks * D2_PVEC, q = vv >> 1;
      const int a = a0 + q / D2_PC, b = b0 + q % D2_PC;
      const bool ok = v < 2 * D2_PVEC && a < g.H && b < g.W;
      voff[u] = ok ? (a * g.W + b) * g.Cin * 2 + ks * 32 + (vv & 1) * 16 : (int)0x80000000;
    }
    auto dma = [&](int u, int chunk, int buf) {          // patch piece u and filter piece u of this wave
      const int q = wave + 4 * u;
      if (u < 4 || q < D2_PPIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_ptr_t)(d2_smem + buf * (D2_FBYTES + D2_PBYTES) + D2_FBYTES + q * 1024),
                                                 16, voff[u], chunk * 64, 0, 0);
      if (u < 4 || q < D2_FPIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(d2_smem + buf * (D2_FBYTES + D2_PBYTES) + q * 1024), 16,
                                                 lane * 16, (chunk * D2_FPIECES + q) * 1024, 0, 0);
    };

    g3_f32x16 acc[4][2];                                  // [2 pa + pb][dy row of the wave]
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[c][r][e] = 0.f;

    __syncthreads();                                     // the previous tile's epilogue is done with LDS
#pragma unroll
    for (int u = 0; u < 5; ++u) dma(u, 0, 0);
    __syncthreads();

    for (int c = 0; c < g.nchunks; ++c) {
      const int buf = c & 1;
      const bool more = c + 1 < g.nchunks;
      const unsigned char* fab = fa + buf * (D2_FBYTES + D2_PBYTES);
      const unsigned char* pbb = pb + buf * (D2_FBYTES + D2_PBYTES);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        g3_bf16x8 bq[3][2], af[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int dc = 0; dc < 2; ++dc)
            bq[r][dc] = *reinterpret_cast<const g3_bf16x8*>(pbb + ks * (D2_PVEC * 16) + (r * D2_PC + dc) * 32);
#pragma unroll
        for (int t = 0; t < 9; ++t) af[t] = *reinterpret_cast<const g3_bf16x8*>(fab + (ks * 9 + (8 - t)) * 1024);
        if (more) {
          if (ks == 0) { dma(0, c + 1, buf ^ 1); dma(1, c + 1, buf ^ 1); dma(2, c + 1, buf ^ 1); }
          else { dma(3, c + 1, buf ^ 1); dma(4, c + 1, buf ^ 1); }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {                     // tap (kh, kw) -> parity class and patch offset
          const int kh = t / 3, kw = t % 3;
          const int pa = kh == 1 ? 0 : 1, dr = kh == 0 ? 1 : 0;
          const int pbp = kw == 1 ? 0 : 1, dc = kw == 0 ? 1 : 0;
#pragma unroll
          for (int r = 0; r < 2; ++r)
            acc[2 * pa + pbp][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], bq[r + dr][dc], acc[2 * pa + pbp][r], 0, 0, 0);
        }
      }
      __syncthreads();
    }

    // ---- epilogue: acc[2 pa + pb][r][e]: channel (e & 3) + 8 (e >> 2) + 4 half of dx pixel
    //      (row 2 (2 wave + r) + pa, column 2 p + pb) of the 16 x 64 tile
#pragma unroll
    for (int cls = 0; cls < 4; ++cls)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int row = 2 * (2 * wave + r) + (cls >> 1), col = 2 * p + (cls & 1);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          uint2 v;
          v.x = pack2_bf16(acc[cls][r][4 * gq + 0], acc[cls][r][4 * gq + 1]);
          v.y = pack2_bf16(acc[cls][r][4 * gq + 2], acc[cls][r][4 * gq + 3]);
          *reinterpret_cast<uint2*>(outs + (row * 64 + col) * D2_OS + 8 * gq + 4 * half) = v;
        }
      }
    const int64_t img_off = (int64_t)bimg * XH * XW * g.Cout + oct * 32;
    const int part = tid & 3;
    // the addend's vectors are requested one group of four rows ahead of the stores that use them, the first group before
    // the barrier: their latency (HBM: the tensor was written by another kernel) overlaps the staging instead of sitting in
    // front of every store (64 -> 128 @ 256^2 with addend: 129 -> 110 us, profiles/r04_s2_dgrad.txt)
    const bool has_add = addend != nullptr;
    const int iw = 2 * b0 + (tid >> 2);                   // row k, column tid >> 2 of the 16 x 64 tile
    const __amdgpu_buffer_rsrc_t ra = sub
        ? __builtin_amdgcn_make_buffer_rsrc((void*)(addend + (int64_t)bimg * g.H * g.W * g.Cout), 0, g.H * g.W * g.Cout * 2, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc((void*)(addend + (int64_t)bimg * XH * XW * g.Cout), 0, XH * XW * g.Cout * 2, 0x00020000);
    // full-size addend: row k of the tile at av0 + k arow.  Compact addend: only even rows and even columns of dx have one,
    // row k at (k >> 1) arow; everything else gets the out-of-range offset the buffer unit answers with zeros (o + 0 = o)
    const int av0 = iw >= XW ? (int)0x80000000
                  : !sub ? ((2 * a0) * XW + iw) * g.Cout * 2 + oct * 64 + part * 16
                  : (iw & 1) ? (int)0x80000000 : (a0 * g.W + (iw >> 1)) * g.Cout * 2 + oct * 64 + part * 16;
    const int arow = (sub ? g.W : XW) * g.Cout * 2;
    auto aload = [&](int k) {                              // k = k4 + e, k4 a multiple of 4: parity of k = parity of e
      return sub ? __builtin_amdgcn_raw_buffer_load_b128(ra, (k & 1) ? (int)0x80000000 : av0, (k >> 1) * arow, 0)
                 : __builtin_amdgcn_raw_buffer_load_b128(ra, av0, k * arow, 0);
    };
    g3_u32x4 adc[4], adn[4];
    __builtin_amdgcn_sched_barrier(0);                   // (after the staging stores: the accumulators' registers are free)
    if (has_add) {
#pragma unroll
      for (int e = 0; e < 4; ++e) adc[e] = aload(e);
    }
    __syncthreads();
    {
      bf16_t* xp = dx + img_off + ((int64_t)(2 * a0) * XW + iw) * g.Cout + part * 8;
      const bf16_t* op = outs + (tid >> 2) * D2_OS + part * 8;
      const int rows = iw < XW ? min(16, XH - 2 * a0) : 0;
#pragma unroll 1
      for (int k4 = 0; k4 < 16; k4 += 4) {
        if (has_add && k4 < 12) {
#pragma unroll
          for (int e = 0; e < 4; ++e) adn[e] = aload(k4 + 4 + e);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (k4 + e < rows) {
            // (D2_OS = 36 elements: rows are 8-byte aligned only)
            const uint2 lo = *reinterpret_cast<const uint2*>(op + 64 * (k4 + e) * D2_OS);
            const uint2 hi = *reinterpret_cast<const uint2*>(op + 64 * (k4 + e) * D2_OS + 4);
            uint4 o = make_uint4(lo.x, lo.y, hi.x, hi.y);
            if (has_add) o = g3_add_bf16x8(o, make_uint4(adc[e][0], adc[e][1], adc[e][2], adc[e][3]));
            *reinterpret_cast<uint4*>(xp) = o;
          }
          xp += (int64_t)XW * g.Cout;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) adc[e] = adn[e];
      }
    }
  }
}

// ---- filter preparation: fp32 / bf16 master weight [O][3][3][I] (channels_last filter) -> bf16 in fragment order
//   out[oc tile][chunk][tap][ocb][lane][e] = W'[oc = tile BN + ocb 32 + (lane & 31)][tap][ci = chunk 16 + (lane >> 5) 8 + e]
// mode 0: W' = w (forward: C_out' = O, C_in' = I).
// mode 1: W'[oc'][tap][ci'] = w[ci'][8 - tap][oc'] (data gradient: C_out' = I, C_in' = O).
template <typename TI>
__global__ __launch_bounds__(256) void g3_prep_filter_k(const TI* __restrict__ w, bf16_t* __restrict__ out, int O, int I,
                                                        int BN, int mode, int64_t nvec) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (v >= nvec) return;
  const int Co = mode ? I : O, Ci = mode ? O : I;        // of the convolution that will run
  const int nch = Ci / G3_KC, ocb_n = BN / 32;
  int64_t r = v;
  const int ln = (int)(r % 64); r /= 64;
  const int ocb = (int)(r % ocb_n); r /= ocb_n;
  const int tap = (int)(r % 9); r /= 9;
  const int chunk = (int)(r % nch); r /= nch;
  const int tile = (int)r;
  const int oc = tile * BN + ocb * 32 + (ln & 31), ci0 = chunk * G3_KC + (ln >> 5) * 8;
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = ci0 + e;
    float val = 0.f;
    if (oc < Co) val = mode ? ld1<TI>(w + ((int64_t)ci * 9 + (8 - tap)) * I + oc) : ld1<TI>(w + ((int64_t)oc * 9 + tap) * I + ci);
    f[e] = val;
  }
  uint4 o;
  o.x = pack2_bf16(f[0], f[1]); o.y = pack2_bf16(f[2], f[3]); o.z = pack2_bf16(f[4], f[5]); o.w = pack2_bf16(f[6], f[7]);
  *reinterpret_cast<uint4*>(out + v * 8) = o;
}

// Output channels per block.  Default 64 with 4-wave blocks, two per CU: measured over BiSeNet's ten layer shapes
// (tools/bench_conv3g.py, profiles/r03_conv3g_configs.log) 1266 us per step of forwards against 1278 us for one 8-wave
// 128-channel block per CU and 1418 us for 8-wave 64-channel blocks — two independent blocks per CU run out of phase and
// cover each other's staging and barriers, which is worth more than reading x once instead of twice (the second read is
// an L2 hit: XCD-aware mapping).  TSG_CONV3G_BN=128 selects the wide tile where C_out allows it (tests cover both).
// The prepared filter is laid out for the width it was made for.
static int g3_bn(int64_t B, int64_t H, int64_t W, int Cout) {
  (void)B; (void)H; (void)W;
  if (Cout % 128) return 64;
  if (const char* e = getenv("TSG_CONV3G_BN")) {
    const int v = atoi(e);
    if (v == 64 || v == 128) return v;
  }
  return 64;
}

// TSG_CONV3G_GLDS=1|0 (default 1): filter slab by LDS-DMA.  profiles/r03_conv3g_glds.log: per step of forwards 1258 vs
// 1320 us, data gradients 1211 vs 1220 us, the bench 1125.3 vs 1123.3 img/s in two interleaved pairs — small, repeatable
static bool g3_glds() {
  static const int v = [] { const char* e = getenv("TSG_CONV3G_GLDS"); return e ? atoi(e) : 1; }();
  return v != 0;
}

// waves per block: 4 (two blocks per CU) for 64-wide tiles unless TSG_CONV3G_NW=8
static int g3_nw(int BN) {
  static const int forced = [] { const char* e = getenv("TSG_CONV3G_NW"); return e ? atoi(e) : 0; }();
  if (BN == 128) return 8;
  return forced == 8 ? 8 : 4;
}

static int g3_geom(G3Geom* g, int64_t B, int64_t H, int64_t W, int Cin, int Cout, int BN) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % G3_KC || Cout % 64) return TSG_E_SHAPE;
  if ((BN != 64 && BN != 128) || Cout % BN) return TSG_E_SHAPE;
  const int64_t th = (H + G3_TH - 1) / G3_TH, tw = (W + G3_TW - 1) / G3_TW;
  if (B * th * tw > 0x7fffffffLL || H * W * (int64_t)(Cin > Cout ? Cin : Cout) > 0x7fffffffLL) return TSG_E_SHAPE;
  g->B = (int)B; g->H = (int)H; g->W = (int)W; g->Cin = Cin; g->Cout = Cout;
  g->tiles_h = (int)th; g->tiles_w = (int)tw; g->ntiles = (int)(B * th * tw);
  g->nchunks = Cin / G3_KC; g->noct = Cout / BN;
  // one 8-wave block (~110 KB of LDS) or two 4-wave blocks (~74 KB each) per CU: ~256 / ~512 blocks in all, a multiple
  // of 8 slots per oc tile (XCD mapping)
  static int target = 0;
  if (!target) { const char* e = getenv("TSG_CONV3G_BLOCKS"); target = e ? atoi(e) : 256; if (target < 8) target = 256; }
  int64_t ns = (g3_nw(BN) == 4 ? 2 * target : target) / g->noct;
  if (ns > g->ntiles) ns = g->ntiles;
  ns = (ns + 7) / 8 * 8;
  if (ns < 8) ns = 8;
  g->nslots = (int)ns;
  static const int prio = [] { const char* e = getenv("TSG_MFMA_PRIO"); return e ? atoi(e) : 0; }();
  g->prio = prio;
  return 0;
}

// conv3h_fwd_k (16-row tiles, 8 accumulators per wave) takes the problem when its tiles fill the two-blocks-per-CU grid:
// >= 512 block tiles (BiSeNet-R18 at the bench shape: every 128^2 layer, the 64^2 layers with >= 256 output channels).
// Not with normalise-on-load (the patch bypasses the registers) and not with 128-wide filter layouts.  The slot count
// equals conv3g_fwd_k's for the same problem, so the statistics partial has the same rows whichever kernel runs.
// TSG_CONV3G_V2=0 keeps every problem on conv3g_fwd_k, =2 sends every problem it can run to conv3h_fwd_k (tests).
static bool g3_v2_geom(G3Geom* g, int BN) {
  const char* e = getenv("TSG_CONV3G_V2");               // 0: never, 1 (default): where it fills the grid, 2: wherever it can run
  const int on = e ? atoi(e) : 1;
  if (!on || BN != 64 || g3_nw(64) != 4 || (g->nchunks & 1)) return false;
  const int64_t th = (g->H + H3_TH - 1) / H3_TH;
  const int64_t nt = (int64_t)g->B * th * g->tiles_w;
  // byte offsets into x, the addend and the filter are 32-bit buffer offsets
  if ((int64_t)g->H * g->W * g->Cin * 2 > 0x7fffffffLL || (int64_t)g->H * g->W * g->Cout * 2 > 0x7fffffffLL ||
      (int64_t)g->nchunks * H3_FBYTES > 0x7fffffffLL)
    return false;
  if (on != 2 && (nt < g->nslots || nt * g->noct < 512)) return false;
  g->tiles_h = (int)th;
  g->ntiles = (int)nt;
  return true;
}

}  // namespace tsg

using namespace tsg;

extern "C" {

int tsg_conv3x3_gen_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad, int dilation, int groups) {
  return dtype == TSG_BF16 && Cin > 0 && Cout > 0 && Cin % G3_KC == 0 && Cout % 64 == 0 && kh == 3 && kw == 3 &&
         stride == 1 && pad == 1 && dilation == 1 && groups == 1;
}

/* bf16 elements of the prepared filter of a convolution with C_out output and C_in input channels */
int64_t tsg_conv3x3_gen_filter_elems(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % G3_KC || Cout % 64) return TSG_E_SHAPE;
  return (int64_t)9 * Cin * Cout;
}

/* output channels per block (64 or 128) tsg_conv3x3_gen_fwd should use for this problem: pass it to the filter
 * preparation and to the forward call */
int tsg_conv3x3_gen_tile(int64_t B, int64_t H, int64_t W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % G3_KC || Cout % 64) return TSG_E_SHAPE;
  return g3_bn(B, H, W, Cout);
}

int tsg_conv3x3_gen_prep_filter(const void* w, int dtype, void* out, int O, int I, int mode, int BN, void* stream) {
  if (!w || !out) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (mode != 0 && mode != 1) return TSG_E_SHAPE;
  const int Co = mode ? I : O, Ci = mode ? O : I;
  if (O <= 0 || I <= 0 || Ci % G3_KC || (BN != 32 && BN != 64 && BN != 128) || Co % BN) return TSG_E_SHAPE;   // 32: conv3s2d_k
  if (!aligned16(out)) return TSG_E_ALIGN;
  const int64_t nvec = (int64_t)9 * Ci * Co / 8;
  const unsigned grid = (unsigned)((nvec + 255) / 256);
  if (dtype == TSG_F32)
    hipLaunchKernelGGL((g3_prep_filter_k<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)w,
                       (bf16_t*)out, O, I, BN, mode, nvec);
  else
    hipLaunchKernelGGL((g3_prep_filter_k<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w,
                       (bf16_t*)out, O, I, BN, mode, nvec);
  TSG_CHECK_LAUNCH();
  return 0;
}

/* data gradient of a 3x3 / stride 2 / padding 1 convolution C_in -> C_out: dy [B,OH,OW,C_out] -> dx [B,H,W,C_in], OH =
 * (H - 1) / 2 + 1; wf = tsg_conv3x3_gen_prep_filter(w, mode 1, BN 32) */
int tsg_conv3x3_s2_dgrad_supported(int dtype, int Cin, int Cout) {
  return dtype == TSG_BF16 && Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % 32 == 0;
}

static int s2_dgrad_common(const void* dy, const void* wf, void* dx, const void* addend, int sub, int64_t B, int64_t H, int64_t W,
                           int Cin, int Cout, void* stream) {
  if (!dy || !wf || !dx) return TSG_E_NULL;
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32 || H > 0xffff || W > 0xffff) return TSG_E_SHAPE;
  if (!aligned16(dy) || !aligned16(wf) || !aligned16(dx) || (addend && !aligned16(addend))) return TSG_E_ALIGN;
  const int64_t OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int64_t th = (OH + D2_TH - 1) / D2_TH, tw = (OW + G3_TW - 1) / G3_TW;
  if (B * th * tw > 0x7fffffffLL || OH * OW * (int64_t)Cout * 2 > 0x7fffffffLL || H * W * (int64_t)Cin * 2 > 0x7fffffffLL ||
      (int64_t)9 * Cin * Cout * 2 > 0x7fffffffLL)
    return TSG_E_SHAPE;                                  // 32-bit buffer offsets (bytes) into dy, the addend and the filter
  G3Geom g;
  g.B = (int)B; g.H = (int)OH; g.W = (int)OW; g.Cin = Cout; g.Cout = Cin;      // the product's K = the convolution's C_out
  g.tiles_h = (int)th; g.tiles_w = (int)tw; g.ntiles = (int)(B * th * tw);
  g.nchunks = Cout / 32; g.noct = Cin / 32;
  int64_t ns = 512 / g.noct;
  if (ns > g.ntiles) ns = g.ntiles;
  ns = (ns + 7) / 8 * 8;
  if (ns < 8) ns = 8;
  g.nslots = (int)ns;
  g.prio = (int)H | ((int)W << 16);
  TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3s2d_k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)D2_LDS));
  hipLaunchKernelGGL(conv3s2d_k, dim3(g.nslots * g.noct), dim3(256), D2_LDS, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)wf, (bf16_t*)dx, g, (const bf16_t*)addend, sub);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_conv3x3_s2_dgrad(const void* dy, const void* wf, void* dx, const void* addend, int64_t B, int64_t H, int64_t W,
                         int Cin, int Cout, void* stream) {
  return s2_dgrad_common(dy, wf, dx, addend, 0, B, H, W, Cin, Cout, stream);
}

/* the same with a COMPACT addend [B, OH, OW, Cin] (the gradient of x[:, :, ::2, ::2]): added at the even pixels of dx */
int tsg_conv3x3_s2_dgrad_subadd(const void* dy, const void* wf, void* dx, const void* addend_sub, int64_t B, int64_t H,
                                int64_t W, int Cin, int Cout, void* stream) {
  if (!addend_sub) return TSG_E_NULL;
  return s2_dgrad_common(dy, wf, dx, addend_sub, 1, B, H, W, Cin, Cout, stream);
}

/* which kernel tsg_conv3x3_gen_fwd runs for this problem: 0 = conv3g_fwd_k (8-row tiles), 1 = conv3h_fwd_k (16-row tiles,
 * all staging by LDS-DMA; never with in_ab).  Negative = error code. */
int tsg_conv3x3_gen_variant(int64_t B, int64_t H, int64_t W, int Cin, int Cout, int BN, int with_in_ab) {
  G3Geom g;
  int e = g3_geom(&g, B, H, W, Cin, Cout, BN);
  if (e) return e;
  return (!with_in_ab && g3_v2_geom(&g, BN)) ? 1 : 0;
}

/* rows of the statistics partial the forward writes when `partial` is given: partial[rows][2][Cout] */
int tsg_conv3x3_gen_stats_partials(int64_t B, int64_t H, int64_t W, int Cin, int Cout, int BN) {
  G3Geom g;
  int e = g3_geom(&g, B, H, W, Cin, Cout, BN);
  return e ? e : g.nslots;
}

int tsg_conv3x3_gen_fwd(const void* x, const void* wf, void* y, float* partial, const float* in_ab, const void* addend,
                        int64_t B, int64_t H, int64_t W, int Cin, int Cout, int BN, void* stream) {
  if (!x || !wf || !y) return TSG_E_NULL;
  if (addend && (partial || !aligned16(addend))) return partial ? TSG_E_SHAPE : TSG_E_ALIGN;   // statistics are of the convolution
  G3Geom g;
  int e = g3_geom(&g, B, H, W, Cin, Cout, BN);
  if (e) return e;
  if (in_ab && Cin > G3_MAX_AFF_C) return TSG_E_SHAPE;
  if (!aligned16(x) || !aligned16(wf) || !aligned16(y) || (in_ab && !aligned16(in_ab))) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int grid = g.nslots * g.noct;
#define G3_GO(BNN, NWW, AF, STT, GL)                                                                                \
  do {                                                                                                              \
    constexpr size_t lds_bytes = G3Cfg<BNN, NWW>::LDS;                                                              \
    TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3g_fwd_k<BNN, NWW, AF, STT, GL>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));                       \
    hipLaunchKernelGGL((conv3g_fwd_k<BNN, NWW, AF, STT, GL>), dim3(grid), dim3(64 * NWW), lds_bytes, st,            \
                       (const bf16_t*)x, (const bf16_t*)wf, (bf16_t*)y, g, in_ab, partial, (const bf16_t*)addend);  \
  } while (0)
#define G3_PICK2(BNN, NWW, GL)                                                                                      \
  do {                                                                                                              \
    if (in_ab) { if (partial) G3_GO(BNN, NWW, true, true, GL); else G3_GO(BNN, NWW, true, false, GL); }             \
    else { if (partial) G3_GO(BNN, NWW, false, true, GL); else G3_GO(BNN, NWW, false, false, GL); }                 \
  } while (0)
#define G3_PICK(BNN, NWW)                                                                                           \
  do { if (g3_glds()) G3_PICK2(BNN, NWW, true); else G3_PICK2(BNN, NWW, false); } while (0)
  if (!in_ab && g3_v2_geom(&g, BN)) {
    if (partial) {
      TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3h_fwd_k<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)H3_LDS));
      hipLaunchKernelGGL((conv3h_fwd_k<true>), dim3(grid), dim3(256), H3_LDS, st, (const bf16_t*)x, (const bf16_t*)wf,
                         (bf16_t*)y, g, partial, (const bf16_t*)addend);
    } else {
      TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3h_fwd_k<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)H3_LDS));
      hipLaunchKernelGGL((conv3h_fwd_k<false>), dim3(grid), dim3(256), H3_LDS, st, (const bf16_t*)x, (const bf16_t*)wf,
                         (bf16_t*)y, g, partial, (const bf16_t*)addend);
    }
  } else if (BN == 128) G3_PICK(128, 8);
  else if (g3_nw(64) == 8) G3_PICK(64, 8);
  else G3_PICK(64, 4);
#undef G3_PICK
#undef G3_PICK2
#undef G3_GO
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
