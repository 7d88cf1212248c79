// 1x1 convolutions on globally pooled feature maps, [B, C_in, 1, 1] -> [B, C_out, 1, 1]: the channel-attention branches of
// BiSeNet / DFN (furnace/seg_opr/seg_oprs.py:199-205 AttentionRefinement, :222-231 FeatureFusion, :113 SELayer's pool) and
// BiSeNet's global context (bisenet network.py:34-39).  Five such layers run per BiSeNet-R18 step.  Each is a
// [B x C_in] x [C_in x C_out] product with B = 16: the vendor library spends 11-18 us per forward and 31-44 us per
// backward on them (naive kernels, zero fills, casts), plus an autocast copy of the fp32 weight per use
// (profiles/r04_eager_ops.txt: 0.27 ms + 0.05 ms of a 13.6 ms step for 2 MFLOP).  Here: one wave per 32 output columns,
// 32x32x16 bf16 MFMAs with the batch as M (rows >= B are zero), the fp32 master weight rounded to bf16 in registers
// (what the autocast copy would hold), fp32 accumulation, one launch forward and one launch for both gradients.
//   forward   y[b][o]   = sum_ci x[b][ci] w[o][ci]                 M = b, N = o,  K = ci
//   data      dx[b][ci] = sum_o  dy[b][o] w[o][ci]                 M = b, N = ci, K = o
//   weight    dw[o][ci] = sum_b  dy[b][o] x[b][ci]   (fp32 out)    M = o, N = ci, K = b
#include "tsg_common.h"

namespace tsg {

typedef __attribute__((ext_vector_type(8))) __bf16 vc_bf16x8;
typedef __attribute__((ext_vector_type(16))) float vc_f32x16;

union VcFrag { uint32_t u[4]; uint4 q; vc_bf16x8 v; };

__device__ __forceinline__ VcFrag vc_zero() { VcFrag f; f.q = make_uint4(0u, 0u, 0u, 0u); return f; }

// 8 consecutive fp32 -> bf16 (round to nearest even, as tensor.to(bfloat16))
__device__ __forceinline__ VcFrag vc_round8(const float* __restrict__ p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  VcFrag f;
  f.u[0] = pack2_bf16(a.x, a.y); f.u[1] = pack2_bf16(a.z, a.w);
  f.u[2] = pack2_bf16(b.x, b.y); f.u[3] = pack2_bf16(b.z, b.w);
  return f;
}

__global__ __launch_bounds__(64) void vec1x1_fwd_k(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                   bf16_t* __restrict__ y, int B, int Cin, int Cout) {
  const int lane = threadIdx.x, n = lane & 31, half = lane >> 5;
  const int o = blockIdx.x * 32 + n;
  vc_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
  for (int k0 = 0; k0 < Cin; k0 += 16) {
    VcFrag a = vc_zero(), b = vc_zero();
    if (n < B) a.q = *reinterpret_cast<const uint4*>(x + (int64_t)n * Cin + k0 + 8 * half);
    if (o < Cout) b = vc_round8(w + (int64_t)o * Cin + k0 + 8 * half);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
  }
  if (o < Cout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bb = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (bb < B) y[(int64_t)bb * Cout + o] = (bf16_t)(pack2_bf16(acc[r], 0.f) & 0xffffu);
    }
  }
}

// blocks [0, ntw): 32 x 32 tiles of dw; blocks [ntw, ntw + ceil(Cin / 32)): 32 columns of dx
__global__ __launch_bounds__(64) void vec1x1_bwd_k(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                   const float* __restrict__ w, bf16_t* __restrict__ dx,
                                                   float* __restrict__ dw, int B, int Cin, int Cout, int ntw) {
  const int lane = threadIdx.x, n = lane & 31, half = lane >> 5;
  const int cit = (Cin + 31) / 32;
  vc_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if ((int)blockIdx.x < ntw) {
    const int ot = blockIdx.x / cit, ct = blockIdx.x % cit;
    const int o = ot * 32 + n, ci = ct * 32 + n;                   // this lane's A row (o) and B column (ci)
    for (int b0 = 0; b0 < B; b0 += 16) {
      uint32_t ae[8], be[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int bb = b0 + 8 * half + e;
        ae[e] = (bb < B && o < Cout) ? dy[(int64_t)bb * Cout + o] : 0u;
        be[e] = (bb < B && ci < Cin) ? x[(int64_t)bb * Cin + ci] : 0u;
      }
      VcFrag a, b;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a.u[e] = ae[2 * e] | (ae[2 * e + 1] << 16); b.u[e] = be[2 * e] | (be[2 * e + 1] << 16); }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
    }
    if (ci < Cin) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int oo = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (oo < Cout) dw[(int64_t)oo * Cin + ci] = acc[r];
      }
    }
    return;
  }
  const int ci = ((int)blockIdx.x - ntw) * 32 + n;
#pragma unroll 2
  for (int k0 = 0; k0 < Cout; k0 += 16) {
    VcFrag a = vc_zero(), b;
    if (n < B) a.q = *reinterpret_cast<const uint4*>(dy + (int64_t)n * Cout + k0 + 8 * half);
    float we[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) we[e] = ci < Cin ? w[(int64_t)(k0 + 8 * half + e) * Cin + ci] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) b.u[e] = pack2_bf16(we[2 * e], we[2 * e + 1]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
  }
  if (ci < Cin) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bb = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (bb < B) dx[(int64_t)bb * Cin + ci] = (bf16_t)(pack2_bf16(acc[r], 0.f) & 0xffffu);
    }
  }
}

static bool vc_ok(int B, int Cin, int Cout) {
  return B > 0 && B <= 32 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 16 == 0 && Cin <= 4096 && Cout <= 4096;
}

}  // namespace tsg

using namespace tsg;

extern "C" {

int tsg_conv1x1_vec_supported(int B, int Cin, int Cout) { return vc_ok(B, Cin, Cout) ? 1 : 0; }

int tsg_conv1x1_vec_fwd(const void* x, const float* w, void* y, int B, int Cin, int Cout, void* stream) {
  if (!x || !w || !y) return TSG_E_NULL;
  if (!vc_ok(B, Cin, Cout)) return TSG_E_SHAPE;
  if (!aligned16(x) || !aligned16(w)) return TSG_E_ALIGN;
  hipLaunchKernelGGL(vec1x1_fwd_k, dim3((Cout + 31) / 32), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)x, w,
                     (bf16_t*)y, B, Cin, Cout);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_conv1x1_vec_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, int B, int Cin, int Cout,
                        void* stream) {
  if (!dy || !x || !w || !dw) return TSG_E_NULL;
  if (!vc_ok(B, Cin, Cout)) return TSG_E_SHAPE;
  if (!aligned16(dy) || !aligned16(w)) return TSG_E_ALIGN;
  const int ntw = ((Cout + 31) / 32) * ((Cin + 31) / 32);
  const int grid = ntw + (dx ? (Cin + 31) / 32 : 0);
  hipLaunchKernelGGL(vec1x1_bwd_k, dim3(grid), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)x, w,
                     (bf16_t*)dx, dw, B, Cin, Cout, ntw);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
