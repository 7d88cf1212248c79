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

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the WHOLE pooled layer in one launch per direction — convolution, BatchNorm over the batch (training statistics
// or running statistics) and the activation behind it (ReLU of ConvBnRelu, the nn.Sigmoid of the attention branches:
// seg_oprs.py:199-205, :222-231; bisenet network.py:34-39).  Under graph replay a launch costs >= 4.8 us whatever it does
// (DESIGN.md 4.3), and a pooled [B, C, 1, 1] layer ran vec1x1 + bn_reduce + bn_finalize + bn_fwd (+ sigmoid) forward and
// (sigmoid_backward +) bn_reduce + bn_bwd_coeffs + bn_bwd + vec1x1 backward: 9-10 launches for 16 x C numbers.  A
// BatchNorm over [B, C, 1, 1] needs nothing but the channel's own B values, which one wave holds after its MFMAs.
// Arithmetic = the unfused kernels' (bn.hip bn_finalize_k / bn_bwd_coeffs_k / bn_fwd / bn_bwd: E[x^2] - mean^2 in fp64,
// a = gamma invstd, y = fma(x, a, b), dx = fma(a, dy', fma(Bc, x - mean, C2)), every tensor rounded to bf16 where the
// unfused path stores one) except that the B-term sums are formed in fp64 directly instead of fp32 partial rows.
//   bnmode 0: no BatchNorm   1: batch statistics (+ running statistics update)   2: running statistics
//   act    0: none           1: ReLU                                             2: sigmoid
struct VcBnArgs {
  const float* gamma; const float* beta;    // may be NULL (1 / 0)
  float* rmean; float* rvar; long long* nbt;  // running statistics (bnmode 1: updated when non-NULL; bnmode 2: read)
  float* stats;                             // out, bnmode 1 / 2: [4][Cout] = {a, b, mean, invstd}
  float eps, momentum;
};

__device__ __forceinline__ float vc_bf16r(float v) { return __uint_as_float(pack2_bf16(v, 0.f) << 16); }
__device__ __forceinline__ float vc_sigmoid(float v) { return 1.f / (1.f + expf(-v)); }

template <int BNMODE, int ACT>
__global__ __launch_bounds__(64) void vec1x1_bnact_fwd_k(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                         bf16_t* __restrict__ out, bf16_t* __restrict__ yc, int B, int Cin,
                                                         int Cout, VcBnArgs bn) {
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
  if (BNMODE == 1 && blockIdx.x == 0 && lane == 0 && bn.nbt) *bn.nbt += 1;
  const int oc = o < Cout ? o : Cout - 1;                        // lanes beyond Cout compute on channel Cout - 1 and store nothing
  float v[16];                                                   // the convolution's output as the unfused path stores it
  double p1 = 0.0, p2 = 0.0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int bb = (r & 3) + 8 * (r >> 2) + 4 * half;
    v[r] = vc_bf16r(acc[r]);
    if (BNMODE == 1 && bb < B) { p1 += (double)v[r]; p2 += (double)v[r] * (double)v[r]; }
  }
  float a = 1.f, b = 0.f;
  if (BNMODE == 1) {
    const double t1 = p1 + __shfl_xor(p1, 32, 64), t2 = p2 + __shfl_xor(p2, 32, 64);   // the channel's other 16 batch rows
    const double count = (double)B;
    const double m = t1 / count;
    double sumvar = t2 - t1 * m;
    if (sumvar < 0.0) sumvar = 0.0;
    const float mf = (float)m, isf = (float)(1.0 / sqrt(sumvar / count + (double)bn.eps));
    a = (bn.gamma ? bn.gamma[oc] : 1.f) * isf;
    b = fmaf(-mf, a, bn.beta ? bn.beta[oc] : 0.f);
    if (half == 0 && o < Cout) {
      bn.stats[o] = a; bn.stats[Cout + o] = b; bn.stats[2 * Cout + o] = mf; bn.stats[3 * Cout + o] = isf;
      if (bn.rmean) bn.rmean[o] = (float)((1.0 - (double)bn.momentum) * (double)bn.rmean[o] + (double)bn.momentum * m);
      if (bn.rvar) bn.rvar[o] = (float)((1.0 - (double)bn.momentum) * (double)bn.rvar[o] + (double)bn.momentum * (sumvar / (count - 1.0)));
    }
  } else if (BNMODE == 2) {
    const float mf = bn.rmean[oc], isf = (float)(1.0 / sqrt((double)bn.rvar[oc] + (double)bn.eps));
    a = (bn.gamma ? bn.gamma[oc] : 1.f) * isf;
    b = fmaf(-mf, a, bn.beta ? bn.beta[oc] : 0.f);
    if (half == 0 && o < Cout) { bn.stats[o] = a; bn.stats[Cout + o] = b; bn.stats[2 * Cout + o] = mf; bn.stats[3 * Cout + o] = isf; }
  }
  if (o < Cout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bb = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (bb < B) {
        float t = v[r];
        if (BNMODE != 0) {
          yc[(int64_t)bb * Cout + o] = (bf16_t)(pack2_bf16(t, 0.f) & 0xffffu);
          t = fmaf(t, a, b);
          if (ACT == 1) t = t > 0.f ? t : 0.f;                    // bn_fwd's fused ReLU: one rounding after it
          if (ACT == 2) t = vc_bf16r(t);                          // BatchNorm output as stored, THEN nn.Sigmoid on it
        } else if (ACT == 1) {
          t = t > 0.f ? t : 0.f;
        }
        if (ACT == 2) t = vc_sigmoid(t);
        out[(int64_t)bb * Cout + o] = (bf16_t)(pack2_bf16(t, 0.f) & 0xffffu);
      }
    }
  }
}

// The gradient w.r.t. the CONVOLUTION's output of channel c for all B rows -> dst[b] (bf16 words), from the gradient w.r.t. the
// layer's output: activation backward (ATen's sigmoid_backward / threshold_backward arithmetic and rounding), then the BatchNorm
// backward of bn.hip (sums over the batch in fp64, coefficients of bn_bwd_coeffs_k, dx of bn_bwd).  Returns the channel's
// (dbeta, dgamma) sums.  `out` = the layer's output, `yc` = the stored convolution output (BNMODE != 0).
// Rows are read 16 at a time, ALL loads of a chunk before any arithmetic (clamped rows, the validity applied to the values): one
// memory round trip per chunk instead of one per row (the first build took 21-29 us per launch, most of it 16 x 3 dependent loads).
template <int BNMODE, int ACT>
__device__ __forceinline__ void vc_layer_grad(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                              const bf16_t* __restrict__ yc, const float* __restrict__ stats, int B, int Cout,
                                              int c, uint16_t* __restrict__ dst, int dstride, float& dbeta, float& dgamma) {
  float a = 1.f, bsh = 0.f, mu = 0.f, isf = 1.f;
  if (BNMODE != 0) { a = stats[c]; bsh = stats[Cout + c]; mu = stats[2 * Cout + c]; isf = stats[3 * Cout + c]; }
  double t1 = 0.0, t2 = 0.0;
  float dz[32], xv[32];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    if (ch == 1 && B <= 16) {                                     // uniform
#pragma unroll
      for (int e = 0; e < 16; ++e) { dz[16 + e] = 0.f; xv[16 + e] = 0.f; }
      break;
    }
    uint16_t rd[16], ro[16], ry[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int b = 16 * ch + e, bc_ = b < B ? b : B - 1;
      rd[e] = dout[(int64_t)bc_ * Cout + c];
      ro[e] = out[(int64_t)bc_ * Cout + c];
      ry[e] = BNMODE != 0 ? yc[(int64_t)bc_ * Cout + c] : (uint16_t)0;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int b = 16 * ch + e;
      float g = bf16_to_f32(rd[e]);
      const float ov = bf16_to_f32(ro[e]);
      float xx = 0.f;
      if (ACT == 2) g = vc_bf16r((g * (1.f - ov)) * ov);          // sigmoid_backward: a * (1 - y) * y in float, one rounding
      if (BNMODE != 0) {
        xx = bf16_to_f32(ry[e]);
        if (ACT == 1) g = fmaf(xx, a, bsh) > 0.f ? g : 0.f;       // bn_bwd's recomputed ReLU mask
      } else if (ACT == 1) {
        g = ov > 0.f ? g : 0.f;                                   // threshold_backward on the stored output
      }
      if (b >= B) { g = 0.f; xx = mu; }
      if (BNMODE != 0) { t1 += (double)g; t2 += (double)(g * (xx - mu)); }
      dz[b] = g; xv[b] = xx;
    }
  }
  float bc = 0.f, c2 = 0.f;
  dbeta = 0.f; dgamma = 0.f;
  if (BNMODE != 0) {
    const double is = (double)isf;
    dbeta = (float)t1;
    dgamma = (float)(t2 * is);
    if (BNMODE == 1) {
      const double count = (double)B;
      const double k0 = t1 / count, k1 = t2 * is / count;
      bc = (float)(-(double)a * k1 * is);
      c2 = (float)(-(double)a * k0);
    }
  }
#pragma unroll
  for (int b = 0; b < 32; ++b) {
    if (b < B) {
      const float d = BNMODE != 0 ? fmaf(a, dz[b], fmaf(bc, xv[b] - mu, c2)) : dz[b];
      dst[b * dstride] = (uint16_t)(pack2_bf16(d, 0.f) & 0xffffu);
    }
  }
}

// blocks [0, ntw): 32 x 32 tiles of dw (the ci-tile-0 block of an o tile also writes dgamma / dbeta); blocks [ntw, ..): 32
// columns of dx.  Every block first forms the convolution-output gradient of the channels it needs in LDS ([B][Cs] bf16).
template <int BNMODE, int ACT>
__global__ __launch_bounds__(256) void vec1x1_bnact_bwd_k(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ out,
                                                         const bf16_t* __restrict__ yc, const float* __restrict__ stats,
                                                         const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                         bf16_t* __restrict__ dx, float* __restrict__ dw,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int Cin,
                                                         int Cout, int ntw) {
  extern __shared__ __attribute__((aligned(16))) uint16_t dyl[];   // [32][Cs], rows >= B zero
  const int lane = threadIdx.x & 63, n = lane & 31, half = lane >> 5;
  const int cit = (Cin + 31) / 32;
  const bool wtile = (int)blockIdx.x < ntw;
  const int ot = wtile ? blockIdx.x / cit : 0, ct = wtile ? blockIdx.x % cit : 0;
  const int c_lo = wtile ? ot * 32 : 0, Cs = wtile ? 32 : Cout;   // channels this block needs
  // phase 1, all four waves: a channel per thread
  for (int i = threadIdx.x; i < 32 * Cs; i += 256) dyl[i] = 0;
  __syncthreads();
  for (int cc = threadIdx.x; cc < Cs; cc += 256) {
    const int c = c_lo + cc;
    if (c < Cout) {
      float db, dg;
      vc_layer_grad<BNMODE, ACT>(dout, out, yc, stats, B, Cout, c, dyl + cc, Cs, db, dg);
      if (BNMODE != 0 && wtile && ct == 0) {
        if (dgamma) dgamma[c] = dg;
        if (dbeta) dbeta[c] = db;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;                                    // phase 2, one wave: the MFMAs
  vc_f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (wtile) {
    const int ci = ct * 32 + n;                                     // this lane's A row is channel c_lo + n, its B column ci
    for (int b0 = 0; b0 < B; b0 += 16) {
      uint32_t ae[8], be[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int bb = b0 + 8 * half + e;
        ae[e] = dyl[bb * Cs + n];
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
    VcFrag a, b;
    a.q = *reinterpret_cast<const uint4*>(dyl + n * Cs + k0 + 8 * half);
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

int tsg_conv1x1_vec_bnact_fwd(const void* x, const float* w, void* out, void* yc, float* stats, const float* gamma,
                              const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                              float eps, float momentum, int bnmode, int act, int B, int Cin, int Cout, void* stream) {
  if (!x || !w || !out) return TSG_E_NULL;
  if (!vc_ok(B, Cin, Cout) || bnmode < 0 || bnmode > 2 || act < 0 || act > 2) return TSG_E_SHAPE;
  if (bnmode != 0 && (!yc || !stats)) return TSG_E_NULL;
  if (bnmode == 2 && (!running_mean || !running_var)) return TSG_E_NULL;
  if (bnmode == 1 && B < 2) return TSG_E_SHAPE;
  if (!aligned16(x) || !aligned16(w)) return TSG_E_ALIGN;
  VcBnArgs bn = {gamma, beta, running_mean, running_var, num_batches_tracked, stats, eps, momentum};
  const dim3 grid((Cout + 31) / 32), blk(64);
  hipStream_t st = (hipStream_t)stream;
#define L_(M, A) hipLaunchKernelGGL((vec1x1_bnact_fwd_k<M, A>), grid, blk, 0, st, (const bf16_t*)x, w, (bf16_t*)out, (bf16_t*)yc, B, Cin, Cout, bn)
#define LA_(M) do { if (act == 0) L_(M, 0); else if (act == 1) L_(M, 1); else L_(M, 2); } while (0)
  if (bnmode == 0) LA_(0); else if (bnmode == 1) LA_(1); else LA_(2);
#undef LA_
#undef L_
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_conv1x1_vec_bnact_bwd(const void* dout, const void* out, const void* yc, const float* stats, const void* x,
                              const float* w, void* dx, float* dw, float* dgamma, float* dbeta, int bnmode, int act, int B,
                              int Cin, int Cout, void* stream) {
  if (!dout || !out || !x || !w || !dw) return TSG_E_NULL;
  if (!vc_ok(B, Cin, Cout) || bnmode < 0 || bnmode > 2 || act < 0 || act > 2) return TSG_E_SHAPE;
  if (bnmode != 0 && (!yc || !stats)) return TSG_E_NULL;
  if (!aligned16(w)) return TSG_E_ALIGN;
  const int ntw = ((Cout + 31) / 32) * ((Cin + 31) / 32);
  const int grid = ntw + (dx ? (Cin + 31) / 32 : 0);
  const size_t sh = (size_t)32 * (size_t)(Cout > 32 ? Cout : 32) * sizeof(uint16_t);
  if (sh > 160 * 1024) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
#define L_(M, A) do { \
    if (sh > 64 * 1024) TSG_HIP(hipFuncSetAttribute((const void*)vec1x1_bnact_bwd_k<M, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
    hipLaunchKernelGGL((vec1x1_bnact_bwd_k<M, A>), dim3(grid), dim3(256), sh, st, (const bf16_t*)dout, (const bf16_t*)out, \
                       (const bf16_t*)yc, stats, (const bf16_t*)x, w, (bf16_t*)dx, dw, dgamma, dbeta, B, Cin, Cout, ntw); } while (0)
#define LA_(M) do { if (act == 0) L_(M, 0); else if (act == 1) L_(M, 1); else L_(M, 2); } while (0)
  if (bnmode == 0) LA_(0); else if (bnmode == 1) LA_(1); else LA_(2);
#undef LA_
#undef L_
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
