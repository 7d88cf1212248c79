// Forward of the 64 -> 64 channel 3x3 / stride 1 / padding 1 convolutions on the bf16 MFMA, channels_last:
// ResNet-18 layer1's four conv3x3(64, 64) (furnace/base_model/resnet.py:24-29,36-53) on the largest activations of the
// context path ([16, 64, 256, 256] at BASELINE config 2: 134 MB in, 134 MB out, 77 GFLOP each).  The same kernel
// computes their DATA gradient (a stride-1 3x3 data gradient is the forward convolution of dy with the 180-degree-rotated,
// transposed filter, tsg_conv3x3_weight_rot180_t), so it runs eight times per step.  The vendor library's kernels reach
// 0.45 PF here (172 us, tools/probe_conv2.py): with only 64 output channels a tile of the implicit GEMM has too little
// reuse for its generic LDS pipeline.
//
// Shape of the kernel (the same split as csrc/stemconv.hip's forward):
//   * the FILTER lives in registers: a wave owns 32 output channels and keeps all 9 x 4 K-fragments of them (144 VGPRs)
//     for the whole launch — the A operand never touches LDS again;
//   * pixels stream through LDS: a block computes 4 rows x 32 columns of output from a (4+2) x (32+2) x 64 input patch
//     (pixel stride 144 B so that the 32 lanes of a ds_read_b128 B-fragment spread over the banks); the patch of the
//     next tile is fetched into registers while the MFMAs of the current one run;
//   * 72 MFMAs (32x32x16) per wave and tile, one 16-byte LDS read each; the tile is re-laid through LDS so that every
//     lane stores 16 B of NHWC;
//   * STATS: per-channel sum / square sum of the bf16-rounded outputs in the epilogue (the statistics pass of the
//     BatchNorm that follows every one of these convolutions), partial[block][2][64].
// x, y: NHWC bf16.  w: bf16 [oc][kh][kw][ci] (the channels_last filter layout).
#include "tsg_common.h"
#include <stdlib.h>

namespace tsg {

typedef __attribute__((ext_vector_type(8))) __bf16 c6_bf16x8;
typedef __attribute__((ext_vector_type(16))) float c6_f32x16;

constexpr int C6_C = 64;
constexpr int C6_TH = 4, C6_TW = 32;                     // output tile
constexpr int C6_PH = C6_TH + 2, C6_PW = C6_TW + 2;      // input patch, pixels
constexpr int C6_PS = 72;                                // LDS pixel stride in bf16 (144 B)
constexpr int C6_NV = C6_PH * C6_PW * 8;                 // 16-byte vectors of a patch: 1632
constexpr int C6_NF = (C6_NV + 255) / 256;               // 7 per thread (the last one partly)
constexpr int C6_AHEAD = 12;                             // B fragments in flight ahead of the MFMA that consumes them

struct C6Geom { int B, H, W, tiles_h, tiles_w, ntiles; };

struct C6Tile { int b, oh0, ow0; };
__device__ __forceinline__ C6Tile c6_tile(const C6Geom& g, int tile) {
  C6Tile t;
  t.ow0 = (tile % g.tiles_w) * C6_TW;
  t.oh0 = ((tile / g.tiles_w) % g.tiles_h) * C6_TH;
  t.b = tile / (g.tiles_w * g.tiles_h);
  return t;
}

// vector `u` of this thread is 16-byte part tid & 7 of patch pixel (pr, pc) = input pixel (oh0 - 1 + pr, ow0 - 1 + pc);
// rc[u] = pr | pc << 8, or -1 past the end of the patch (one register per vector: the kernel is register-bound)
struct C6Lane { int rc[C6_NF]; };

__device__ __forceinline__ void c6_lane_init(int tid, C6Lane& l) {
#pragma unroll
  for (int u = 0; u < C6_NF; ++u) {
    const int v = tid + 256 * u, px = v >> 3;
    l.rc[u] = v < C6_NV ? ((px / C6_PW) | ((px % C6_PW) << 8)) : -1;
  }
}

__device__ __forceinline__ void c6_fetch(const bf16_t* __restrict__ x, const C6Geom& g, const C6Tile& t,
                                         const C6Lane& l, int part8, uint4 (&rp)[C6_NF]) {
  const bf16_t* xb = x + (int64_t)t.b * g.H * g.W * C6_C + part8;
#pragma unroll
  for (int u = 0; u < C6_NF; ++u) {
    const int ih = t.oh0 - 1 + (l.rc[u] & 0xff), iw = t.ow0 - 1 + (l.rc[u] >> 8);
    rp[u] = make_uint4(0u, 0u, 0u, 0u);
    if (l.rc[u] >= 0 && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
      rp[u] = *reinterpret_cast<const uint4*>(xb + ((int64_t)ih * g.W + iw) * C6_C);
  }
}

// Normalise-on-load (AFF): the input of the convolution is relu(a x + b) of the tensor that is read — the BatchNorm + ReLU
// that precedes these convolutions in the spatial path and inside a BasicBlock (seg_oprs.py:39-46, resnet.py:36-46) —
// applied while the patch is written to LDS, with the values tsg_bn_apply_fwd would have stored (same fma, same
// rounding to bf16), so the normalised activation is never written or re-read.  Padding pixels stay exactly zero.
__device__ __forceinline__ uint4 c6_affine_relu(uint4 v, const float* __restrict__ ab, int part8) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
  const float4 a0 = *reinterpret_cast<const float4*>(ab + part8), a1 = *reinterpret_cast<const float4*>(ab + part8 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(ab + C6_C + part8), b1 = *reinterpret_cast<const float4*>(ab + C6_C + part8 + 4);
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

// bf16(bf16 a + bf16 b) per element, fp32 add: what the eager `a + b` of two bf16 tensors computes
__device__ __forceinline__ uint4 c6_add_bf16x8(uint4 a, uint4 b) {
  const uint32_t x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    o[i] = pack2_bf16(__uint_as_float(x[i] << 16) + __uint_as_float(y[i] << 16),
                      __uint_as_float(x[i] & 0xffff0000u) + __uint_as_float(y[i] & 0xffff0000u));
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// ---- backward sums of the BatchNorm in FRONT of the convolution, in the epilogue of its data gradient (round 6) ----------
// The data gradient computed here is dy' of a BatchNorm -> ReLU pair (seg_oprs.py:39-46, resnet.py:36-46): SyncBN's backward
// starts with sum dy' m and sum dy' m (x - mean) per channel, m = (a x + b > 0) (syncbn_kernel.cu:160-174 with the ReLU mask
// folded in; csrc/bn.hip bn_reduce_* <MODE 1, MASK 2>), a pass that re-reads the tensor this kernel has just written plus
// x.  Here the thread that stores 16 bytes of the gradient reads the 16 bytes of x beside them and accumulates its eight
// channels' two sums from the bf16-ROUNDED values it stores — the values the separate pass would read — so only the
// summation order differs from that pass (fp32 per thread over its pixels, then a fixed-order fold; fp64 across blocks in
// tsg_bn_bwd_coeffs as before).  abm: LDS copy of the pack's rows a, b, mean ([3][64]).
struct C6Bsum { float s1[8], s2[8]; };
__device__ __forceinline__ void c6_bsum_zero(C6Bsum& a) {
#pragma unroll
  for (int e = 0; e < 8; ++e) { a.s1[e] = 0.f; a.s2[e] = 0.f; }
}
__device__ __forceinline__ void c6_bsum_acc(C6Bsum& acc, uint4 o, uint4 xv, const float* __restrict__ abm, int part8) {
  const uint32_t ow[4] = {o.x, o.y, o.z, o.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
  const float4 a0 = *reinterpret_cast<const float4*>(abm + part8), a1 = *reinterpret_cast<const float4*>(abm + part8 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(abm + C6_C + part8), b1 = *reinterpret_cast<const float4*>(abm + C6_C + part8 + 4);
  const float4 m0 = *reinterpret_cast<const float4*>(abm + 2 * C6_C + part8), m1 = *reinterpret_cast<const float4*>(abm + 2 * C6_C + part8 + 4);
  const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  const float mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float d0 = __uint_as_float(ow[i] << 16), d1 = __uint_as_float(ow[i] & 0xffff0000u);
    const float x0 = __uint_as_float(xw[i] << 16), x1 = __uint_as_float(xw[i] & 0xffff0000u);
    const float g0 = fmaf(x0, a[2 * i], b[2 * i]) > 0.f ? d0 : 0.f;            // bn_reduce_nhwc<., ., 1, 2>: the same mask
    const float g1 = fmaf(x1, a[2 * i + 1], b[2 * i + 1]) > 0.f ? d1 : 0.f;
    acc.s1[2 * i] += g0;
    acc.s2[2 * i] = fmaf(g0, x0 - mu[2 * i], acc.s2[2 * i]);
    acc.s1[2 * i + 1] += g1;
    acc.s2[2 * i + 1] = fmaf(g1, x1 - mu[2 * i + 1], acc.s2[2 * i + 1]);
  }
}
// 4 waves: wave = (row pair wr) * 2 + (oc half wm)
template <bool STATS, int OCC, bool AFF>
__global__ __launch_bounds__(256, OCC) void conv64_fwd_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                    bf16_t* __restrict__ y, C6Geom g, float* __restrict__ partial,
                                                    const float* __restrict__ in_ab,
                                                    const bf16_t* __restrict__ addend) {
  __shared__ __attribute__((aligned(16))) bf16_t patch[C6_PH * C6_PW * C6_PS];     // 29376 B
  __shared__ __attribute__((aligned(16))) bf16_t outs[C6_TH * C6_TW * C6_PS];      // 18432 B: [pixel][64 oc + 8 pad]
  __shared__ __attribute__((aligned(16))) float abs_[AFF ? 2 * C6_C : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int wm = wave & 1, wr = wave >> 1;
  if (AFF && tid < 2 * C6_C) abs_[tid] = in_ab[tid];      // visible after the first barrier of the tile loop

  c6_bf16x8 fw[9][4];                                    // filter fragments: oc = 32 wm + p, ci = 16 kc + 8 half ..
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
      fw[t][kc] = *reinterpret_cast<const c6_bf16x8*>(w + ((wm * 32 + p) * 9 + t) * C6_C + kc * 16 + half * 8);

  const int spl = tid >> 3, spart = tid & 7;             // store: tile column spl, 16-B part spart
  C6Lane ln;
  c6_lane_init(tid, ln);
  uint4 rp[C6_NF];
  float st1 = 0.f, st2 = 0.f;                            // STATS: channel tid & 63 over tile row tid >> 6
  int tile = blockIdx.x;
  C6Tile tp = c6_tile(g, tile < g.ntiles ? tile : 0);
  const int part8 = (tid & 7) * 8;
  // OCC == 1: one block per CU, the next tile's patch is prefetched into registers during the MFMAs.
  // OCC == 2: no register prefetch (its 28 registers are what does not fit twice), two blocks per CU cover for each
  // other's loads, staging and epilogue.
  if (OCC == 1 && tile < g.ntiles) c6_fetch(x, g, tp, ln, part8, rp);
  for (; tile < g.ntiles; tile += gridDim.x) {
    if (OCC != 1) c6_fetch(x, g, tp, ln, part8, rp);
    __syncthreads();                                     // the previous tile's reads of patch / outs are done
#pragma unroll
    for (int u = 0; u < C6_NF; ++u)
      if (u < C6_NF - 1 || ln.rc[u] >= 0) {
        uint4 v = rp[u];
        if (AFF) {
          const int ih = tp.oh0 - 1 + (ln.rc[u] & 0xff), iw = tp.ow0 - 1 + (ln.rc[u] >> 8);
          if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) v = c6_affine_relu(v, abs_, part8);
        }
        *reinterpret_cast<uint4*>(patch + ((tid + 256 * u) >> 3) * C6_PS + part8) = v;
      }
    __syncthreads();
    C6Tile tn = tp;
    if (tile + (int)gridDim.x < g.ntiles) {              // in flight during the MFMAs below
      tn = c6_tile(g, tile + gridDim.x);
      if (OCC == 1) c6_fetch(x, g, tn, ln, part8, rp);
    }

    c6_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const bf16_t* pb = patch + ((2 * wr) * C6_PW + p) * C6_PS + half * 8;
    // 72 (tap, channel chunk, row) steps, software-pipelined by hand: the B fragment of step s + C6_AHEAD is requested
    // before the MFMA of step s issues, and the scheduler is told not to move anything across a step.  With one wave per
    // SIMD nothing else hides the ~130-cycle LDS latency; left to itself the compiler serialised load -> wait -> MFMA over
    // the last third of the loop.
    auto frag = [&](int s) {
      const int t = s >> 3, kc = (s >> 1) & 3, i = s & 1, kh = t / 3, kw = t % 3;
      return *reinterpret_cast<const c6_bf16x8*>(pb + ((i + kh) * C6_PW + kw) * C6_PS + kc * 16);
    };
    constexpr int AHEAD = OCC == 1 ? C6_AHEAD : 4;       // two waves per SIMD hide most of the latency themselves
    c6_bf16x8 ring[AHEAD];
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) ring[s] = frag(s);
#pragma unroll
    for (int s = 0; s < 72; ++s) {
      const c6_bf16x8 fb = ring[s % AHEAD];
      if (s + AHEAD < 72) ring[s % AHEAD] = frag(s + AHEAD);
      acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[s >> 3][(s >> 1) & 3], fb, acc[s & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // acc[i][r]: oc = 32 wm + (r & 3) + 8 (r >> 2) + 4 half, pixel = (row 2 wr + i, column p).  Re-lay as [pixel][oc].
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int oc0 = 32 * wm + 8 * gq + 4 * half;
        uint2 v;
        v.x = pack2_bf16(acc[i][4 * gq + 0], acc[i][4 * gq + 1]);
        v.y = pack2_bf16(acc[i][4 * gq + 2], acc[i][4 * gq + 3]);
        *reinterpret_cast<uint2*>(outs + ((2 * wr + i) * C6_TW + p) * C6_PS + oc0) = v;
      }
    const int64_t tile_off = (((int64_t)tp.b * g.H + tp.oh0) * g.W + tp.ow0) * C6_C;
    bf16_t* yt = y + tile_off;
    const bool colok = tp.ow0 + spl < g.W;
    // addend: y = bf16(bf16(conv) + addend): the skip connection's gradient joins the data gradient here.  Its four vectors
    // are requested BEFORE the barrier, so their latency overlaps the staging instead of sitting in front of every store
    // (16 x 64 x 256^2: 126-149 -> 120-143 us, profiles/r04_conv3h.txt; csrc/conv3g.hip's epilogues do the same)
    uint4 ad[C6_TH];
    __builtin_amdgcn_sched_barrier(0);                   // (after the staging stores: the accumulators' registers are free)
    if (addend) {
#pragma unroll
      for (int qd = 0; qd < C6_TH; ++qd) {
        ad[qd] = make_uint4(0u, 0u, 0u, 0u);
        if (tp.oh0 + qd < g.H && colok)
          ad[qd] = *reinterpret_cast<const uint4*>(addend + tile_off + ((int64_t)qd * g.W + spl) * C6_C + spart * 8);
      }
    }
    __syncthreads();
#pragma unroll
    for (int qd = 0; qd < C6_TH; ++qd)
      if (tp.oh0 + qd < g.H && colok) {
        uint4 o = *reinterpret_cast<const uint4*>(outs + (qd * C6_TW + spl) * C6_PS + spart * 8);
        const int64_t off = ((int64_t)qd * g.W + spl) * C6_C + spart * 8;
        if (addend) o = c6_add_bf16x8(o, ad[qd]);
        *reinterpret_cast<uint4*>(yt + off) = o;
      }
    if (STATS) {
      const int c = tid & 63, qd = tid >> 6;
      if (tp.oh0 + qd < g.H) {
        const int npx = g.W - tp.ow0 < C6_TW ? g.W - tp.ow0 : C6_TW;
        const bf16_t* col = outs + qd * C6_TW * C6_PS + c;
        if (npx == C6_TW) {
#pragma unroll 8
          for (int px = 0; px < C6_TW; ++px) { const float v = bf16_to_f32(col[px * C6_PS]); st1 += v; st2 = fmaf(v, v, st2); }
        } else {
          for (int px = 0; px < npx; ++px) { const float v = bf16_to_f32(col[px * C6_PS]); st1 += v; st2 = fmaf(v, v, st2); }
        }
      }
    }
    tp = tn;
  }
  if (STATS) {                                           // fold the four tile rows in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(outs);
    red[tid] = st1; red[256 + tid] = st2;
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, which = tid >> 6;
      const float* r = red + which * 256 + c;
      partial[((int64_t)blockIdx.x * 2 + which) * C6_C + c] = (r[0] + r[64]) + (r[128] + r[192]);
    }
  }
}

// =====================================================================================================================
// Round 5: the same kernel with the patch staged by LDS-DMA and double-buffered (the launches WITHOUT normalise-on-load:
// six of the step's eight).  conv64_fwd_k<., 2, false> runs load -> wait -> LDS write -> barrier -> 72 MFMAs -> epilogue
// per tile with nothing of its own in flight during the MFMAs (the 28 prefetch registers do not fit twice per CU): the
// matrix pipe was busy 0.33 of the time, and the second block of the CU only half covers a tile's 1-2 us of load latency
// with its own 1 us of MFMAs.  Here the NEXT tile's patch is requested by buffer_load_dwordx4 ... lds into the other of
// two patch buffers right after the barrier that opens a tile — no staging registers, no ds_write, no VALU — and lands
// while the MFMAs and the epilogue of the current tile run.
//   * DMA writes 1 KB per wave instruction, lane L at byte 16 L: pixels are 128 B apart (no padding possible).  Unpadded,
//     the 16 lanes of a ds_read_b128 phase would all hit the same four banks; so the eight 16-byte parts of a pixel are
//     stored XOR-swizzled by the pixel's patch COLUMN: part j of column c sits at slot j ^ ((c >> 1) & 7).  Sixteen
//     consecutive columns then cover every slot twice, once on an even and once on an odd pixel, i.e. on the two 128-byte
//     halves of the bank period: conflict-free.  The swizzle costs the DMA nothing (each lane picks WHICH global 16 bytes it
//     fetches) and the reader one v_xad_u32 per fragment;
//   * padding pixels = buffer offsets beyond num_records (the load returns zeros), as in conv3h_fwd_k;
//   * LDS: 2 x 26 KB patch buffers + the 18 KB output staging tile = 70 KB: two blocks per CU.
constexpr int C6D_PIECES = 26;                           // 204 pixels x 8 parts = 1632 vectors = 25.5 wave pieces of 64
constexpr int C6D_PBYTES = C6D_PIECES * 1024;
constexpr int C6D_OUTS = 2 * C6D_PBYTES;
constexpr size_t C6D_LDS = (size_t)C6D_OUTS + (size_t)C6_TH * C6_TW * C6_PS * 2;     // 71,680 B

// BSUM (round 6): the launch is the DATA gradient of a convolution behind BatchNorm -> ReLU (BasicBlock's bn1 -> relu -> conv2,
// resnet.py:36-46); its epilogue also emits that BatchNorm's backward sums (c6_bsum_* above).  The kernel has no register
// left (252 of 256 at two blocks per CU), so a thread's 16 sums live only inside one tile's epilogue: they are folded over the
// eight lanes of the wave that store the same 16-byte part (lanes l, l + 8, ..: three ds_bpermute steps, fixed order) and
// accumulated into a per-wave LDS row; the four rows are folded at the end.  partial: [grid][2][64].
template <bool STATS, bool BSUM>
__global__ __launch_bounds__(256, 2) void conv64_dma_fwd_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, C6Geom g, float* __restrict__ partial,
                                                          const bf16_t* __restrict__ addend, const bf16_t* __restrict__ bx,
                                                          const float* __restrict__ bfp) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char c6d_smem[];
  __shared__ __attribute__((aligned(16))) float abm[BSUM ? 3 * C6_C : 4];
  __shared__ __attribute__((aligned(16))) float wred[BSUM ? 4 * 2 * C6_C : 4];
  bf16_t* outs = reinterpret_cast<bf16_t*>(c6d_smem + C6D_OUTS);
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, p = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wr = wave >> 1;
  if (BSUM) {                                            // visible after the first barrier of the tile loop
    if (tid < 3 * C6_C) abm[tid] = bfp[tid];
    wred[tid] = 0.f; wred[256 + tid] = 0.f;
  }

  c6_bf16x8 fw[9][4];                                    // filter fragments: oc = 32 wm + p, ci = 16 kc + 8 half ..
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
      fw[t][kc] = *reinterpret_cast<const c6_bf16x8*>(w + ((wm * 32 + p) * 9 + t) * C6_C + kc * 16 + half * 8);
  const int spl = tid >> 3, spart = tid & 7;             // store: tile column spl, 16-B part spart

  // DMA: piece q = wave + 4 u; this lane's vector v = 64 q + lane = slot (v & 7) of patch pixel v >> 3.  Everything about
  // a vector is recomputed per tile from v (a dozen VALU instructions per piece against ~2300 cycles of MFMAs): seven
  // pieces' worth of per-lane constants were the registers that spilled, and a scratch reload in front of a DMA is an
  // s_waitcnt vmcnt(0), i.e. a wait for the PREVIOUS piece's DMA
  auto dma = [&](const C6Tile& t, int buf) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (int64_t)t.b * g.H * g.W * C6_C), 0, g.H * g.W * C6_C * 2, 0x00020000);
    int ln = lane;
    asm volatile("" : "+v"(ln));                         // opaque per call: keeps the compiler from hoisting the per-lane
    //                                                      geometry out of the tile loop (= 20 more live registers = spills)
#pragma unroll
    for (int u = 0; u < 7; ++u)
      if (wave + 4 * u < C6D_PIECES) {                   // wave-uniform
        const int v = (wave + 4 * u) * 64 + ln, px = v >> 3;
        const int pr = px / C6_PW, pc = px - pr * C6_PW;
        const int ih = t.oh0 - 1 + pr, iw = t.ow0 - 1 + pc;
        const bool ok = v < C6_NV && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
        const int voff = ok ? (ih * g.W + iw) * (C6_C * 2) + (((v & 7) ^ ((pc >> 1) & 7)) << 4) : (int)0x80000000;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(c6d_smem + buf * C6D_PBYTES + (wave + 4 * u) * 1024), 16, voff,
                                                 0, 0, 0);
      }
  };
  // B fragment of step s: patch pixel (2 wr + i + kh, p + kw), 16-byte part 2 kc + half, at slot part ^ swizzle(column)
  int swz16[3], pcol[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    swz16[kw] = (((p + kw) >> 1) & 7) << 4;
    pcol[kw] = ((2 * wr) * C6_PW + p + kw) * 128;
  }

  float st1 = 0.f, st2 = 0.f;                            // STATS: channel tid & 63 over tile row tid >> 6
  int tile = blockIdx.x, buf = 0;
  C6Tile tp = c6_tile(g, tile < g.ntiles ? tile : 0);
  if (tile < g.ntiles) dma(tp, 0);
  for (; tile < g.ntiles; tile += gridDim.x, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the tile's patch have landed ...
    __syncthreads();                                     // ... and everybody's; the previous tile is done with outs and buf ^ 1
    C6Tile tn = tp;
    if (tile + (int)gridDim.x < g.ntiles) {              // in flight during the MFMAs and the epilogue below
      tn = c6_tile(g, tile + gridDim.x);
      dma(tn, buf ^ 1);
    }

    c6_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned char* pbuf = c6d_smem + buf * C6D_PBYTES;
    auto frag = [&](int s) {
      const int t = s >> 3, kc = (s >> 1) & 3, i = s & 1, kh = t / 3, kw = t % 3;
      const int off = (swz16[kw] ^ (((kc * 2) << 4) | (half << 4))) + pcol[kw];
      return *reinterpret_cast<const c6_bf16x8*>(pbuf + off + (i + kh) * (C6_PW * 128));
    };
    constexpr int AHEAD = 4;
    c6_bf16x8 ring[AHEAD];
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) ring[s] = frag(s);
#pragma unroll
    for (int s = 0; s < 72; ++s) {
      const c6_bf16x8 fb = ring[s % AHEAD];
      if (s + AHEAD < 72) ring[s % AHEAD] = frag(s + AHEAD);
      acc[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[s >> 3][(s >> 1) & 3], fb, acc[s & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // acc[i][r]: oc = 32 wm + (r & 3) + 8 (r >> 2) + 4 half, pixel = (row 2 wr + i, column p).  Re-lay as [pixel][oc].
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int oc0 = 32 * wm + 8 * gq + 4 * half;
        uint2 v;
        v.x = pack2_bf16(acc[i][4 * gq + 0], acc[i][4 * gq + 1]);
        v.y = pack2_bf16(acc[i][4 * gq + 2], acc[i][4 * gq + 3]);
        *reinterpret_cast<uint2*>(outs + ((2 * wr + i) * C6_TW + p) * C6_PS + oc0) = v;
      }
    const int64_t tile_off = (((int64_t)tp.b * g.H + tp.oh0) * g.W + tp.ow0) * C6_C;
    bf16_t* yt = y + tile_off;
    const bool colok = tp.ow0 + spl < g.W;
    uint4 ad[C6_TH];
    __builtin_amdgcn_sched_barrier(0);
    if (addend) {
#pragma unroll
      for (int qd = 0; qd < C6_TH; ++qd) {
        ad[qd] = make_uint4(0u, 0u, 0u, 0u);
        if (tp.oh0 + qd < g.H && colok)
          ad[qd] = *reinterpret_cast<const uint4*>(addend + tile_off + ((int64_t)qd * g.W + spl) * C6_C + spart * 8);
      }
    }
    if (BSUM) {                                          // x beside the four vectors this thread stores: clamped addresses,
#pragma unroll                                           // unconditional loads (a load in a branch is waited for in that branch)
      for (int qd = 0; qd < C6_TH; ++qd) {
        const bool ok = tp.oh0 + qd < g.H && colok;
        ad[qd] = *reinterpret_cast<const uint4*>(bx + (ok ? tile_off + ((int64_t)qd * g.W + spl) * C6_C + spart * 8 : 0));
      }
    }
    __syncthreads();
    C6Bsum bs;
    if (BSUM) c6_bsum_zero(bs);
#pragma unroll
    for (int qd = 0; qd < C6_TH; ++qd)
      if (tp.oh0 + qd < g.H && colok) {
        uint4 o = *reinterpret_cast<const uint4*>(outs + (qd * C6_TW + spl) * C6_PS + spart * 8);
        const int64_t off = ((int64_t)qd * g.W + spl) * C6_C + spart * 8;
        if (addend && !BSUM) o = c6_add_bf16x8(o, ad[qd]);
        *reinterpret_cast<uint4*>(yt + off) = o;
        if (BSUM) c6_bsum_acc(bs, o, ad[qd], abm, spart * 8);
      }
    if (BSUM) {
      float* mine = wred + wave * 2 * C6_C + spart * 8;  // lanes 0-7 of the wave own [2][8 parts x 8 channels]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s1 = bs.s1[e], s2 = bs.s2[e];
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
        if (lane < 8) { mine[e] += s1; mine[C6_C + e] += s2; }
      }
    }
    if (STATS) {
      const int c = tid & 63, qd = tid >> 6;
      if (tp.oh0 + qd < g.H) {
        const int npx = g.W - tp.ow0 < C6_TW ? g.W - tp.ow0 : C6_TW;
        const bf16_t* col = outs + qd * C6_TW * C6_PS + c;
        if (npx == C6_TW) {
#pragma unroll 8
          for (int px = 0; px < C6_TW; ++px) { const float v = bf16_to_f32(col[px * C6_PS]); st1 += v; st2 = fmaf(v, v, st2); }
        } else {
          for (int px = 0; px < npx; ++px) { const float v = bf16_to_f32(col[px * C6_PS]); st1 += v; st2 = fmaf(v, v, st2); }
        }
      }
    }
    tp = tn;
  }
  if (STATS) {                                           // fold the four tile rows in a fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(outs);
    red[tid] = st1; red[256 + tid] = st2;
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, which = tid >> 6;
      const float* r = red + which * 256 + c;
      partial[((int64_t)blockIdx.x * 2 + which) * C6_C + c] = (r[0] + r[64]) + (r[128] + r[192]);
    }
  }
  if (BSUM) {                                            // fold the four waves' rows in a fixed order
    __syncthreads();
    if (tid < 128) {
      const float* r = wred + tid;                       // tid = which * 64 + c
      partial[(int64_t)blockIdx.x * 2 * C6_C + tid] = (r[0] + r[128]) + (r[256] + r[384]);
    }
  }
}

// =====================================================================================================================
// Stride 2 (BiSeNet's SpatialPath.conv_3x3_1 / conv_3x3_2, network.py:117-118: 64 -> 64, 3x3 / 2 / 1, on the 512^2 and
// 256^2 maps).  Both directions move 5 bytes of activation per 2 flops/byte of MFMA work: memory-bound, so the aim is
// to stream the big tensor exactly once at the HBM rate, which the vendor kernels miss by 1.7x (forward) and 3x (data
// gradient, tools/probe_conv2.py).
//   forward  : tile = 2 output rows x 32 columns from a 5 x 65 input patch; the patch is stored as an EVEN-column and an
//              ODD-column plane per row, so that the 32 lanes of a B fragment (output columns q .. q+31, input columns
//              2q + kw) read 32 consecutive pixels of one plane (144-byte stride: conflict-free).
//   backward : by output parity.  dx[2m+a][2q+b] only sees the taps with kh = 1 - a (mod 2), kw = 1 - b (mod 2):
//                (0,0) 1 tap, (0,1) 2, (1,0) 2, (1,1) 4 — 9 tap-products per 2 x 2 block of dx instead of 36, no zero
//              insertion, no atomics, every dx element written once.  Tile = 2 dy rows x 32 dy columns (3 x 33 patch)
//              -> 4 x 64 pixels of dx, the four parities computed one after the other into one staging tile.
//              The A operand is the transposed filter: tsg_conv3x3_weight_rot180_t's output, indexed back by tap.
// =====================================================================================================================
constexpr int S2_TH = 2, S2_TW = 32;
constexpr int S2_PH = 2 * S2_TH + 1, S2_PW = 2 * S2_TW + 1;      // 5 x 65 input pixels
constexpr int S2_NE = S2_TW + 1;                                 // even-plane pixels per row (33), odd plane: 32
constexpr int S2_NV = S2_PH * S2_PW * 8;                         // 2600 vectors
constexpr int S2_NF = (S2_NV + 255) / 256;                       // 11

struct S2Geom { int B, H, W, OH, OW, tiles_h, tiles_w, ntiles; };
__device__ __forceinline__ C6Tile s2_tile(const S2Geom& g, int tile, int th, int tw) {
  C6Tile t;
  t.ow0 = (tile % g.tiles_w) * tw;
  t.oh0 = ((tile / g.tiles_w) % g.tiles_h) * th;
  t.b = tile / (g.tiles_w * g.tiles_h);
  return t;
}

template <int NF, int NV, int PW>
struct S2Lane {
  int rc[NF];
  __device__ __forceinline__ void init(int tid) {
#pragma unroll
    for (int u = 0; u < NF; ++u) {
      const int v = tid + 256 * u, px = v >> 3;
      rc[u] = v < NV ? ((px / PW) | ((px % PW) << 8)) : -1;
    }
  }
};

template <bool STATS, int OCC, bool AFF>
__global__ __launch_bounds__(256, OCC) void conv64_fwd_s2_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, S2Geom g, float* __restrict__ partial,
                                                          const float* __restrict__ in_ab) {
  __shared__ __attribute__((aligned(16))) bf16_t patch[S2_PH * S2_PW * C6_PS];     // 46800 B
  __shared__ __attribute__((aligned(16))) bf16_t outs[S2_TH * S2_TW * C6_PS];      // 9216 B
  __shared__ __attribute__((aligned(16))) float abs_[AFF ? 2 * C6_C : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int wm = wave & 1, wr = wave >> 1;
  if (AFF && tid < 2 * C6_C) abs_[tid] = in_ab[tid];
  c6_bf16x8 fw[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
      fw[t][kc] = *reinterpret_cast<const c6_bf16x8*>(w + ((wm * 32 + p) * 9 + t) * C6_C + kc * 16 + half * 8);
  const int part8 = (tid & 7) * 8;
  S2Lane<S2_NF, S2_NV, S2_PW> ln;
  ln.init(tid);
  uint4 rp[S2_NF];
  auto fetch = [&](const C6Tile& t) {
    const bf16_t* xb = x + (int64_t)t.b * g.H * g.W * C6_C + part8;
#pragma unroll
    for (int u = 0; u < S2_NF; ++u) {
      const int ih = 2 * t.oh0 - 1 + (ln.rc[u] & 0xff), iw = 2 * t.ow0 - 1 + (ln.rc[u] >> 8);
      rp[u] = make_uint4(0u, 0u, 0u, 0u);
      if (ln.rc[u] >= 0 && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W)
        rp[u] = *reinterpret_cast<const uint4*>(xb + ((int64_t)ih * g.W + iw) * C6_C);
    }
  };
  float st1 = 0.f, st2 = 0.f;
  int tile = blockIdx.x;
  C6Tile tp = s2_tile(g, tile < g.ntiles ? tile : 0, S2_TH, S2_TW);
  if (OCC == 1 && tile < g.ntiles) fetch(tp);
  for (; tile < g.ntiles; tile += gridDim.x) {
    if (AFF) {
      // normalise-on-load variant: the patch is loaded, transformed and written four vectors at a time (all eleven in
      // registers next to the transform's temporaries would spill)
      __syncthreads();
      const bf16_t* xb = x + (int64_t)tp.b * g.H * g.W * C6_C + part8;
#pragma unroll
      for (int u0 = 0; u0 < S2_NF; u0 += 4) {
        uint4 v[4];
        bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int u = u0 + k < S2_NF ? u0 + k : S2_NF - 1;
          const int ih = 2 * tp.oh0 - 1 + (ln.rc[u] & 0xff), iw = 2 * tp.ow0 - 1 + (ln.rc[u] >> 8);
          ok[k] = u0 + k < S2_NF && ln.rc[u] >= 0 && ih >= 0 && ih < g.H && iw >= 0 && iw < g.W;
          v[k] = make_uint4(0u, 0u, 0u, 0u);
          if (ok[k]) v[k] = *reinterpret_cast<const uint4*>(xb + ((int64_t)ih * g.W + iw) * C6_C);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int u = u0 + k;
          if (u < S2_NF && (u < S2_NF - 1 || ln.rc[u] >= 0)) {
            const int pr = ln.rc[u] & 0xff, pc = ln.rc[u] >> 8;
            const int pix = pr * S2_PW + ((pc & 1) ? S2_NE + (pc >> 1) : (pc >> 1));
            if (ok[k]) v[k] = c6_affine_relu(v[k], abs_, part8);
            *reinterpret_cast<uint4*>(patch + pix * C6_PS + part8) = v[k];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    } else {
    if (OCC != 1) fetch(tp);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < S2_NF; ++u) {
      if (u < S2_NF - 1 || ln.rc[u] >= 0) {
        const int pr = ln.rc[u] & 0xff, pc = ln.rc[u] >> 8;
        const int pix = pr * S2_PW + ((pc & 1) ? S2_NE + (pc >> 1) : (pc >> 1));     // even plane, then odd plane
        *reinterpret_cast<uint4*>(patch + pix * C6_PS + part8) = rp[u];
      }
    }
    __syncthreads();
    }
    C6Tile tn = tp;
    if (tile + (int)gridDim.x < g.ntiles) {
      tn = s2_tile(g, tile + gridDim.x, S2_TH, S2_TW);
      if (OCC == 1) fetch(tn);
    }
    c6_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int kh = t / 3, kw = t % 3;
      const bf16_t* pb = patch + ((2 * wr + kh) * S2_PW + (kw == 1 ? S2_NE + p : p + (kw >> 1))) * C6_PS + half * 8;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const c6_bf16x8 fb = *reinterpret_cast<const c6_bf16x8*>(pb + kc * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[t][kc], fb, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int oc0 = 32 * wm + 8 * gq + 4 * half;
      uint2 v;
      v.x = pack2_bf16(acc[4 * gq + 0], acc[4 * gq + 1]);
      v.y = pack2_bf16(acc[4 * gq + 2], acc[4 * gq + 3]);
      *reinterpret_cast<uint2*>(outs + (wr * S2_TW + p) * C6_PS + oc0) = v;
    }
    __syncthreads();
    bf16_t* yt = y + (((int64_t)tp.b * g.OH + tp.oh0) * g.OW + tp.ow0) * C6_C;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int v = tid + 256 * k, qd = v >> 8, spl = (v >> 3) & 31, spart = v & 7;
      if (tp.oh0 + qd < g.OH && tp.ow0 + spl < g.OW)
        *reinterpret_cast<uint4*>(yt + ((int64_t)qd * g.OW + spl) * C6_C + spart * 8) =
            *reinterpret_cast<const uint4*>(outs + (qd * S2_TW + spl) * C6_PS + spart * 8);
    }
    if (STATS) {
      const int c = tid & 63, grp = tid >> 6;              // channel c over 16 of the tile's 64 pixels
      const int qd = grp >> 1, px0 = (grp & 1) * 16;
      if (tp.oh0 + qd < g.OH) {
        const bf16_t* col = outs + (qd * S2_TW + px0) * C6_PS + c;
#pragma unroll 8
        for (int px = 0; px < 16; ++px)
          if (tp.ow0 + px0 + px < g.OW) { const float v = bf16_to_f32(col[px * C6_PS]); st1 += v; st2 = fmaf(v, v, st2); }
      }
    }
    tp = tn;
  }
  if (STATS) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(patch);
    red[tid] = st1; red[256 + tid] = st2;
    __syncthreads();
    if (tid < 128) {
      const int c = tid & 63, which = tid >> 6;
      const float* r = red + which * 256 + c;
      partial[((int64_t)blockIdx.x * 2 + which) * C6_C + c] = (r[0] + r[64]) + (r[128] + r[192]);
    }
  }
}

// ---- data gradient, stride 2.  g.H / g.W: dx (= the forward input) size; g.OH / g.OW: dy size; tiles over dy.
constexpr int D2_TH = 2, D2_TW = 32;
constexpr int D2_PH = D2_TH + 1, D2_PW = D2_TW + 1;              // 3 x 33 dy pixels
constexpr int D2_NV = D2_PH * D2_PW * 8;                         // 792
constexpr int D2_NF = (D2_NV + 255) / 256;                       // 4
constexpr int D2_OW = 2 * D2_TW;                                 // 64 dx columns per tile

template <int OCC, bool BSUM>
__global__ __launch_bounds__(256, OCC) void conv64_dgrad_s2_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wt,
                                                            bf16_t* __restrict__ dx, S2Geom g,
                                                            const bf16_t* __restrict__ bx, const float* __restrict__ bfp,
                                                            float* __restrict__ bpart) {
  __shared__ __attribute__((aligned(16))) bf16_t patch[D2_PH * D2_PW * C6_PS];             // 14256 B
  __shared__ __attribute__((aligned(16))) bf16_t outs[2 * D2_TH * D2_OW * C6_PS];          // 36864 B: 4 x 64 dx pixels
  __shared__ __attribute__((aligned(16))) float abm[BSUM ? 3 * C6_C : 4];
  // BSUM: a thread's 16 sums live inside one tile's epilogue only (144 filter registers + two blocks per CU leave no room to
  // carry them across the MFMA phase): folded over the eight lanes that store the same 16-byte part and accumulated into a
  // per-wave LDS row, as in conv64_dma_fwd_k<., true>
  __shared__ __attribute__((aligned(16))) float wred[BSUM ? 4 * 2 * C6_C : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int wm = wave & 1, wr = wave >> 1;
  if (BSUM) {                                             // visible after the first barrier of the tile loop
    if (tid < 3 * C6_C) abm[tid] = bfp[tid];
    wred[tid] = 0.f; wred[256 + tid] = 0.f;
  }
  // transposed filter: wt[ci][kh'][kw'][co] = w[co][2 - kh'][2 - kw'][ci]; fragment of forward tap (kh, kw): rows ci, K = co
  c6_bf16x8 fw[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
      fw[t][kc] = *reinterpret_cast<const c6_bf16x8*>(wt + ((wm * 32 + p) * 9 + (8 - t)) * C6_C + kc * 16 + half * 8);
  const int part8 = (tid & 7) * 8;
  S2Lane<D2_NF, D2_NV, D2_PW> ln;
  ln.init(tid);
  uint4 rp[D2_NF];
  auto fetch = [&](const C6Tile& t) {
    const bf16_t* db = dy + (int64_t)t.b * g.OH * g.OW * C6_C + part8;
#pragma unroll
    for (int u = 0; u < D2_NF; ++u) {
      const int oh = t.oh0 + (ln.rc[u] & 0xff), ow = t.ow0 + (ln.rc[u] >> 8);
      rp[u] = make_uint4(0u, 0u, 0u, 0u);
      if (ln.rc[u] >= 0 && oh < g.OH && ow < g.OW)
        rp[u] = *reinterpret_cast<const uint4*>(db + ((int64_t)oh * g.OW + ow) * C6_C);
    }
  };
  int tile = blockIdx.x;
  C6Tile tp = s2_tile(g, tile < g.ntiles ? tile : 0, D2_TH, D2_TW);
  if (OCC == 1 && tile < g.ntiles) fetch(tp);
  for (; tile < g.ntiles; tile += gridDim.x) {
    if (OCC != 1) fetch(tp);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < D2_NF; ++u)
      if (u < D2_NF - 1 || ln.rc[u] >= 0)
        *reinterpret_cast<uint4*>(patch + ((ln.rc[u] & 0xff) * D2_PW + (ln.rc[u] >> 8)) * C6_PS + part8) = rp[u];
    __syncthreads();
    C6Tile tn = tp;
    if (tile + (int)gridDim.x < g.ntiles) {
      tn = s2_tile(g, tile + gridDim.x, D2_TH, D2_TW);
      if (OCC == 1) fetch(tn);
    }
    // B fragment of dy pixel (row wr + dr, column p + dc)
    const bf16_t* pb = patch + (wr * D2_PW + p) * C6_PS + half * 8;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int a = ph >> 1, b = ph & 1;
      c6_f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ih = 0; ih <= a; ++ih) {
        const int kh = a ? (ih ? 0 : 2) : 1, dr = a ? ih : 0;          // a = 1: kh = 2 reads dy[m], kh = 0 reads dy[m+1]
#pragma unroll
        for (int iw = 0; iw <= b; ++iw) {
          const int kw = b ? (iw ? 0 : 2) : 1, dc = b ? iw : 0;
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) {
            const c6_bf16x8 fb = *reinterpret_cast<const c6_bf16x8*>(pb + (dr * D2_PW + dc) * C6_PS + kc * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[kh * 3 + kw][kc], fb, acc, 0, 0, 0);
          }
        }
      }
      // acc[r]: ci = 32 wm + (r & 3) + 8 (r >> 2) + 4 half, dx pixel (row 2 wr + a, column 2 p + b) of the tile
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int c0 = 32 * wm + 8 * gq + 4 * half;
        uint2 v;
        v.x = pack2_bf16(acc[4 * gq + 0], acc[4 * gq + 1]);
        v.y = pack2_bf16(acc[4 * gq + 2], acc[4 * gq + 3]);
        *reinterpret_cast<uint2*>(outs + ((2 * wr + a) * D2_OW + 2 * p + b) * C6_PS + c0) = v;
      }
    }
    const int64_t tile_off = (((int64_t)tp.b * g.H + 2 * tp.oh0) * g.W + 2 * tp.ow0) * C6_C;
    // BSUM: the eight vectors of x beside the eight this thread stores, requested before the barrier (their latency overlaps
    // the staging; the accumulators' registers are free by now)
    uint4 xq[BSUM ? 4 : 1];
    auto xload = [&](int k) {
      const int v = tid + 256 * k, qd = v >> 9, spl = (v >> 3) & 63, spart = v & 7;
      const bool ok = 2 * tp.oh0 + qd < g.H && 2 * tp.ow0 + spl < g.W;
      const int64_t off = ok ? tile_off + ((int64_t)qd * g.W + spl) * C6_C + spart * 8 : 0;       // clamped: unconditional load
      return *reinterpret_cast<const uint4*>(bx + off);
    };
    if (BSUM) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < 4; ++k) xq[BSUM ? k : 0] = xload(k);
    }
    __syncthreads();
    bf16_t* xt = dx + tile_off;
    C6Bsum bs;
    if (BSUM) c6_bsum_zero(bs);
    if (!BSUM) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int v = tid + 256 * k, qd = v >> 9, spl = (v >> 3) & 63, spart = v & 7;
        if (2 * tp.oh0 + qd < g.H && 2 * tp.ow0 + spl < g.W)
          *reinterpret_cast<uint4*>(xt + ((int64_t)qd * g.W + spl) * C6_C + spart * 8) =
              *reinterpret_cast<const uint4*>(outs + (qd * D2_OW + spl) * C6_PS + spart * 8);
      }
    } else {
      // two rounds of four vectors (tile rows 0-1, then 2-3); a consumed x register is refilled with the vector of the
      // second round at once, so that only four are ever live next to the 144 registers of the filter
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int v = tid + 256 * k, qd = v >> 9, spl = (v >> 3) & 63, spart = v & 7;
        if (2 * tp.oh0 + qd < g.H && 2 * tp.ow0 + spl < g.W) {
          const uint4 o = *reinterpret_cast<const uint4*>(outs + (qd * D2_OW + spl) * C6_PS + spart * 8);
          *reinterpret_cast<uint4*>(xt + ((int64_t)qd * g.W + spl) * C6_C + spart * 8) = o;
          c6_bsum_acc(bs, o, xq[k & 3], abm, spart * 8);
        }
        if (k < 4) {
          xq[k] = xload(4 + k);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (BSUM) {
      float* mine = wred + wave * 2 * C6_C + (tid & 7) * 8;  // lanes 0-7 of the wave own [2][8 parts x 8 channels]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float s1 = bs.s1[e], s2 = bs.s2[e];
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
        if (lane < 8) { mine[e] += s1; mine[C6_C + e] += s2; }
      }
    }
    tp = tn;
  }
  if (BSUM) {                                              // fold the four waves' rows in a fixed order
    __syncthreads();
    if (tid < 128) {
      const float* r = wred + tid;                         // tid = which * 64 + c
      bpart[(int64_t)blockIdx.x * 2 * C6_C + tid] = (r[0] + r[128]) + (r[256] + r[384]);
    }
  }
}

static bool s2_geom(int64_t B, int64_t H, int64_t W, int th, int tw, S2Geom* g) {
  if (B <= 0 || H <= 0 || W <= 0) return false;
  const int64_t OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int64_t nh = (OH + th - 1) / th, nw = (OW + tw - 1) / tw;
  if (B * nh * nw > 0x7fffffffLL || H * W * C6_C > 0x7fffffffLL) return false;
  g->B = (int)B; g->H = (int)H; g->W = (int)W; g->OH = (int)OH; g->OW = (int)OW;
  g->tiles_h = (int)nh; g->tiles_w = (int)nw; g->ntiles = (int)(B * nh * nw);
  return true;
}

// TSG_C64_OCC=2 (default): two blocks per CU WITHOUT the register prefetch of the next patch — the 28-44 registers it
// costs are what keeps two waves per SIMD from fitting next to the 144 filter registers, and a second block hides a
// block's loads, staging, barriers and epilogue better than the prefetch hid its loads (tools/bench_conv64.py, us:
// stride 1 at 256^2 105 -> 84 (0.92 PF), stride 2 at 512^2 forward 173 -> 144, data gradient 173 -> 155).
// 1: one prefetching block per CU.  Either way no spills: with __launch_bounds__(256, 2) AND the prefetch the compiler
// spilled ~35 registers to scratch and the stride-1 kernel took 162 us.
static int c6_occ() {
  static const int occ = [] { const char* e = getenv("TSG_C64_OCC"); return e ? atoi(e) : 2; }();
  return occ == 1 ? 1 : 2;
}

static bool c6_geom(int64_t B, int64_t H, int64_t W, C6Geom* g) {
  if (B <= 0 || H <= 0 || W <= 0) return false;
  const int64_t th = (H + C6_TH - 1) / C6_TH, tw = (W + C6_TW - 1) / C6_TW;
  if (B * th * tw > 0x7fffffffLL || H * W * C6_C > 0x7fffffffLL) return false;
  g->B = (int)B; g->H = (int)H; g->W = (int)W; g->tiles_h = (int)th; g->tiles_w = (int)tw; g->ntiles = (int)(B * th * tw);
  return true;
}

}  // namespace tsg

using namespace tsg;

extern "C" {

int tsg_conv3x3_c64_supported(int dtype, int Cin, int Cout, int kh, int kw, int stride, int pad, int dilation,
                              int groups) {
  return dtype == TSG_BF16 && Cin == C6_C && Cout == C6_C && kh == 3 && kw == 3 && (stride == 1 || stride == 2) &&
         pad == 1 && dilation == 1 && groups == 1;
}

int tsg_conv3x3_c64_stats_partials(int64_t B, int64_t H, int64_t W) {
  C6Geom g;
  if (!c6_geom(B, H, W, &g)) return TSG_E_SHAPE;
  return g.ntiles < 256 * c6_occ() ? g.ntiles : 256 * c6_occ();
}

int tsg_conv3x3_c64_fwd(const void* x, const void* w, void* y, float* partial, const float* in_ab, const void* addend,
                        int64_t B, int64_t H, int64_t W, void* stream) {
  if (!x || !w || !y) return TSG_E_NULL;
  if (in_ab && !aligned16(in_ab)) return TSG_E_ALIGN;
  if (addend && (partial || !aligned16(addend))) return partial ? TSG_E_SHAPE : TSG_E_ALIGN;
  C6Geom g;
  if (!c6_geom(B, H, W, &g)) return TSG_E_SHAPE;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int occ = c6_occ();                              // also fixes the number of statistics partials
  const int grid = g.ntiles < 256 * occ ? g.ntiles : 256 * occ;
#define C6_GO(ST, OC, AF) hipLaunchKernelGGL((conv64_fwd_k<ST, OC, AF>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, \
                                             (const bf16_t*)w, (bf16_t*)y, g, partial, in_ab, (const bf16_t*)addend)
  // TSG_CONV64_DMA=1|0 (default 1): the launches without normalise-on-load on conv64_dma_fwd_k (patch by LDS-DMA, double
  // buffered); needs two blocks per CU (occ 2) and 32-bit byte offsets into one image
  static const bool use_dma = [] { const char* e = getenv("TSG_CONV64_DMA"); return !(e && e[0] == '0'); }();
  if (!in_ab && use_dma && occ == 2 && (int64_t)g.H * g.W * C6_C * 2 < 0x7fffffffLL) {
    if (partial) {
      TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv64_dma_fwd_k<true, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C6D_LDS));
      hipLaunchKernelGGL((conv64_dma_fwd_k<true, false>), dim3(grid), dim3(256), C6D_LDS, st, (const bf16_t*)x, (const bf16_t*)w,
                         (bf16_t*)y, g, partial, (const bf16_t*)addend, (const bf16_t*)nullptr, (const float*)nullptr);
    } else {
      TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv64_dma_fwd_k<false, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C6D_LDS));
      hipLaunchKernelGGL((conv64_dma_fwd_k<false, false>), dim3(grid), dim3(256), C6D_LDS, st, (const bf16_t*)x, (const bf16_t*)w,
                         (bf16_t*)y, g, partial, (const bf16_t*)addend, (const bf16_t*)nullptr, (const float*)nullptr);
    }
    TSG_CHECK_LAUNCH();
    return 0;
  }
  if (in_ab) { if (partial) C6_GO(true, 2, true); else C6_GO(false, 2, true); }
  else if (partial) { if (occ == 1) C6_GO(true, 1, false); else C6_GO(true, 2, false); }
  else { if (occ == 1) C6_GO(false, 1, false); else C6_GO(false, 2, false); }
#undef C6_GO
  TSG_CHECK_LAUNCH();
  return 0;
}

// Data gradient of a 64 -> 64 / stride-1 convolution (dy, wt = tsg_conv3x3_weight_rot180_t(w)) that ALSO emits the backward
// sums of the BatchNorm -> ReLU in front of the convolution (bn_x: that BatchNorm's input, bn_fp: its forward pack
// [3][64] = a, b, mean): partial [tsg_conv3x3_c64_dgrad_bnsums_partials][2][64] = {sum dy' m, sum dy' m (x - mean)},
// what tsg_bn_bwd_reduce(relu = 1, y = NULL) computes from the stored gradient in a pass of its own.
int tsg_conv3x3_c64_dgrad_bnsums_partials(int64_t B, int64_t H, int64_t W) {
  C6Geom g;
  if (!c6_geom(B, H, W, &g)) return TSG_E_SHAPE;
  if (c6_occ() != 2 || (int64_t)g.H * g.W * C6_C * 2 >= 0x7fffffffLL) return 0;       // the LDS-DMA kernel only
  return g.ntiles < 512 ? g.ntiles : 512;
}

int tsg_conv3x3_c64_dgrad_bnsums(const void* dy, const void* wt, void* dx, const void* bn_x, const float* bn_fp,
                                 float* partial, int64_t B, int64_t H, int64_t W, void* stream) {
  if (!dy || !wt || !dx || !bn_x || !bn_fp || !partial) return TSG_E_NULL;
  C6Geom g;
  if (!c6_geom(B, H, W, &g)) return TSG_E_SHAPE;
  if (c6_occ() != 2 || (int64_t)g.H * g.W * C6_C * 2 >= 0x7fffffffLL) return TSG_E_SHAPE;
  if (!aligned16(dy) || !aligned16(wt) || !aligned16(dx) || !aligned16(bn_x) || !aligned16(bn_fp)) return TSG_E_ALIGN;
  const int grid = g.ntiles < 512 ? g.ntiles : 512;
  TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv64_dma_fwd_k<false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C6D_LDS));
  hipLaunchKernelGGL((conv64_dma_fwd_k<false, true>), dim3(grid), dim3(256), C6D_LDS, (hipStream_t)stream, (const bf16_t*)dy,
                     (const bf16_t*)wt, (bf16_t*)dx, g, partial, (const bf16_t*)nullptr, (const bf16_t*)bn_x, bn_fp);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_conv3x3_c64_s2_stats_partials(int64_t B, int64_t H, int64_t W) {
  S2Geom g;
  if (!s2_geom(B, H, W, S2_TH, S2_TW, &g)) return TSG_E_SHAPE;
  return g.ntiles < 256 * c6_occ() ? g.ntiles : 256 * c6_occ();
}

int tsg_conv3x3_c64_s2_fwd(const void* x, const void* w, void* y, float* partial, const float* in_ab, int64_t B,
                           int64_t H, int64_t W, void* stream) {
  if (!x || !w || !y) return TSG_E_NULL;
  if (in_ab && !aligned16(in_ab)) return TSG_E_ALIGN;
  S2Geom g;
  if (!s2_geom(B, H, W, S2_TH, S2_TW, &g)) return TSG_E_SHAPE;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int occ = c6_occ();
  const int grid = g.ntiles < 256 * occ ? g.ntiles : 256 * occ;
#define C6_GO(ST, OC, AF) hipLaunchKernelGGL((conv64_fwd_s2_k<ST, OC, AF>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, \
                                             (const bf16_t*)w, (bf16_t*)y, g, partial, in_ab)
  if (in_ab) { if (partial) C6_GO(true, 2, true); else C6_GO(false, 2, true); }
  else if (partial) { if (occ == 1) C6_GO(true, 1, false); else C6_GO(true, 2, false); }
  else { if (occ == 1) C6_GO(false, 1, false); else C6_GO(false, 2, false); }
#undef C6_GO
  TSG_CHECK_LAUNCH();
  return 0;
}

static int c6_s2_dgrad_common(const void* dy, const void* wt, void* dx, const void* bn_x, const float* bn_fp, float* partial,
                              int64_t B, int64_t H, int64_t W, void* stream) {
  if (!dy || !wt || !dx) return TSG_E_NULL;
  S2Geom g;
  if (!s2_geom(B, H, W, D2_TH, D2_TW, &g)) return TSG_E_SHAPE;
  if (!aligned16(dy) || !aligned16(wt) || !aligned16(dx)) return TSG_E_ALIGN;
  if (partial && (!aligned16(bn_x) || !aligned16(bn_fp))) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int occ = c6_occ();
  const int grid = g.ntiles < 256 * occ ? g.ntiles : 256 * occ;
#define C6_GO(OC, BS) hipLaunchKernelGGL((conv64_dgrad_s2_k<OC, BS>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dy,  \
                                         (const bf16_t*)wt, (bf16_t*)dx, g, (const bf16_t*)bn_x, bn_fp, partial)
  if (partial) { if (occ == 1) C6_GO(1, true); else C6_GO(2, true); }
  else { if (occ == 1) C6_GO(1, false); else C6_GO(2, false); }
#undef C6_GO
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_conv3x3_c64_s2_dgrad(const void* dy, const void* wt, void* dx, int64_t B, int64_t H, int64_t W, void* stream) {
  return c6_s2_dgrad_common(dy, wt, dx, nullptr, nullptr, nullptr, B, H, W, stream);
}

int tsg_conv3x3_c64_s2_dgrad_partials(int64_t B, int64_t H, int64_t W) {
  S2Geom g;
  if (!s2_geom(B, H, W, D2_TH, D2_TW, &g)) return TSG_E_SHAPE;
  return g.ntiles < 256 * c6_occ() ? g.ntiles : 256 * c6_occ();
}

int tsg_conv3x3_c64_s2_dgrad_bnsums(const void* dy, const void* wt, void* dx, const void* bn_x, const float* bn_fp,
                                    float* partial, int64_t B, int64_t H, int64_t W, void* stream) {
  if (!bn_x || !bn_fp || !partial) return TSG_E_NULL;
  return c6_s2_dgrad_common(dy, wt, dx, bn_x, bn_fp, partial, B, H, W, stream);
}

}  // extern "C"
