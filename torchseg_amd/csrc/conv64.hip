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

namespace tsg {

typedef __attribute__((ext_vector_type(8))) __bf16 c6_bf16x8;
typedef __attribute__((ext_vector_type(16))) float c6_f32x16;

constexpr int C6_C = 64;
constexpr int C6_TH = 4, C6_TW = 32;                     // output tile
constexpr int C6_PH = C6_TH + 2, C6_PW = C6_TW + 2;      // input patch, pixels
constexpr int C6_PS = 72;                                // LDS pixel stride in bf16 (144 B)
constexpr int C6_NV = C6_PH * C6_PW * 8;                 // 16-byte vectors of a patch: 1632
constexpr int C6_NF = (C6_NV + 255) / 256;               // 7 per thread (the last one partly)
constexpr int C6_BLOCKS = 512;                           // persistent blocks: 2 per CU (~230 VGPRs)

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

// 4 waves: wave = (row pair wr) * 2 + (oc half wm)
template <bool STATS>
__global__ __launch_bounds__(256, 2) void conv64_fwd_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                    bf16_t* __restrict__ y, C6Geom g, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) bf16_t patch[C6_PH * C6_PW * C6_PS];     // 29376 B
  __shared__ __attribute__((aligned(16))) bf16_t outs[C6_TH * C6_TW * C6_PS];      // 18432 B: [pixel][64 oc + 8 pad]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, p = lane & 31;
  const int wm = wave & 1, wr = wave >> 1;

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
  if (tile < g.ntiles) c6_fetch(x, g, tp, ln, part8, rp);
  for (; tile < g.ntiles; tile += gridDim.x) {
    __syncthreads();                                     // the previous tile's reads of patch / outs are done
#pragma unroll
    for (int u = 0; u < C6_NF; ++u)
      if (u < C6_NF - 1 || ln.rc[u] >= 0)
        *reinterpret_cast<uint4*>(patch + ((tid + 256 * u) >> 3) * C6_PS + part8) = rp[u];
    __syncthreads();
    C6Tile tn = tp;
    if (tile + (int)gridDim.x < g.ntiles) {              // in flight during the MFMAs below
      tn = c6_tile(g, tile + gridDim.x);
      c6_fetch(x, g, tn, ln, part8, rp);
    }

    c6_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const bf16_t* pb = patch + ((2 * wr) * C6_PW + p) * C6_PS + half * 8;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int kh = t / 3, kw = t % 3;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const c6_bf16x8 fb = *reinterpret_cast<const c6_bf16x8*>(pb + ((i + kh) * C6_PW + kw) * C6_PS + kc * 16);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fw[t][kc], fb, acc[i], 0, 0, 0);
        }
      }
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
    __syncthreads();
    bf16_t* yt = y + (((int64_t)tp.b * g.H + tp.oh0) * g.W + tp.ow0) * C6_C;
    const bool colok = tp.ow0 + spl < g.W;
#pragma unroll
    for (int qd = 0; qd < C6_TH; ++qd)
      if (tp.oh0 + qd < g.H && colok)
        *reinterpret_cast<uint4*>(yt + ((int64_t)qd * g.W + spl) * C6_C + spart * 8) =
            *reinterpret_cast<const uint4*>(outs + (qd * C6_TW + spl) * C6_PS + spart * 8);
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
  return dtype == TSG_BF16 && Cin == C6_C && Cout == C6_C && kh == 3 && kw == 3 && stride == 1 && pad == 1 &&
         dilation == 1 && groups == 1;
}

int tsg_conv3x3_c64_stats_partials(int64_t B, int64_t H, int64_t W) {
  C6Geom g;
  if (!c6_geom(B, H, W, &g)) return TSG_E_SHAPE;
  return g.ntiles < C6_BLOCKS ? g.ntiles : C6_BLOCKS;
}

int tsg_conv3x3_c64_fwd(const void* x, const void* w, void* y, float* partial, int64_t B, int64_t H, int64_t W,
                        void* stream) {
  if (!x || !w || !y) return TSG_E_NULL;
  C6Geom g;
  if (!c6_geom(B, H, W, &g)) return TSG_E_SHAPE;
  if (!aligned16(x) || !aligned16(w) || !aligned16(y)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int grid = g.ntiles < C6_BLOCKS ? g.ntiles : C6_BLOCKS;
  if (partial)
    hipLaunchKernelGGL(conv64_fwd_k<true>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, g,
                       partial);
  else
    hipLaunchKernelGGL(conv64_fwd_k<false>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y,
                       g, (float*)nullptr);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
