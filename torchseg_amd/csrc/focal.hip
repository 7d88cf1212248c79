// Sigmoid focal loss of DFN's border branch for gfx950.
//
// Restates SigmoidFocalLoss.forward (furnace/seg_opr/loss_opr.py:23-45)
// exactly, including its use of the SIGMOID where the logit was intended
// (the "TODO" at :32): with p = sigmoid(x), m = [t != ignore], t' = m*t,
//   max_val = clamp(-p, min=0)                       (= 0 since p > 0)
//   pos = (1-p)^gamma * (p - p*t')
//   neg = p^gamma * (max_val + log(exp(-max_val) + exp(-p - max_val)))
//   loss = mean over ALL pixels of -(alpha*pos + (1-alpha)*neg) * m
// The reference spends ~15 element-wise temporaries; here: one read of
// (pred, target) forward, one read + one write backward.  HBM-bound.
#include "tsg_common.h"
#include <math.h>

namespace tsg {
constexpr int kT = 256;

template <int LT> __device__ __forceinline__ float lab_f(const void* p, int64_t i);
template <> __device__ __forceinline__ float lab_f<TSG_I64>(const void* p, int64_t i) { return (float)((const int64_t*)p)[i]; }
template <> __device__ __forceinline__ float lab_f<TSG_U8>(const void* p, int64_t i) { return (float)((const uint8_t*)p)[i]; }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <typename T, int LT>
__global__ __launch_bounds__(kT) void focal_fwd_k(const T* __restrict__ pred, const void* __restrict__ tgt,
                                                  int64_t P, float ignore, float gamma, float alpha,
                                                  float* __restrict__ part) {
  __shared__ float sm[2 * (kT / 64)];
  float acc = 0.f, dummy = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < P; i += (int64_t)gridDim.x * kT) {
    const float t = lab_f<LT>(tgt, i);
    const float m = t != ignore ? 1.f : 0.f;
    const float tt = m * t;
    const float p = sigmoidf_(ld1<T>(pred + i));
    float mv = -p; mv = mv < 0.f ? 0.f : mv;
    const float pos = powf(1.f - p, gamma) * (p - p * tt);
    const float neg = powf(p, gamma) * (mv + logf(expf(-mv) + expf(-p - mv)));
    acc += -(alpha * pos + (1.f - alpha) * neg) * m;
  }
  block_sum2(acc, dummy, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ __launch_bounds__(kT) void focal_finish_k(const float* __restrict__ part, int grid, double P,
                                                     float* __restrict__ loss) {
  __shared__ double dsm[kT / 64];
  double a = 0;
  for (int i = threadIdx.x; i < grid; i += kT) a += part[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) dsm[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = 0;
    for (int i = 0; i < kT / 64; ++i) a += dsm[i];
    loss[0] = (float)(a / P);
  }
}

template <typename T, int LT>
__global__ __launch_bounds__(kT) void focal_bwd_k(const T* __restrict__ pred, const void* __restrict__ tgt,
                                                  int64_t P, float ignore, float gamma, float alpha,
                                                  const float* __restrict__ gscale, T* __restrict__ dpred) {
  const float g = gscale[0] / (float)P;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < P; i += (int64_t)gridDim.x * kT) {
    const float t = lab_f<LT>(tgt, i);
    float d = 0.f;
    if (t != ignore) {
      const float p = sigmoidf_(ld1<T>(pred + i));
      const float q = 1.f - p;
      // d pos / dp with pos = (1-p)^g * p * (1-t)
      const float dpos = (1.f - t) * (powf(q, gamma) - gamma * p * powf(q, gamma - 1.f));
      // d neg / dp with neg = p^g * log(1 + e^-p)
      const float L = logf(1.f + expf(-p));
      const float dneg = gamma * powf(p, gamma - 1.f) * L - powf(p, gamma) / (1.f + expf(p));
      d = -g * (alpha * dpos + (1.f - alpha) * dneg) * p * q;
    }
    st1<T>(dpred + i, d);
  }
}

static int fgrid(int64_t P) {
  int64_t g = (P + kT - 1) / kT;
  if (g > 2048) g = 2048;
  return (int)(g < 1 ? 1 : g);
}
}  // namespace tsg

using namespace tsg;

extern "C" {

size_t tsg_focal_ws_bytes(int64_t P) { return (size_t)fgrid(P) * sizeof(float); }

int tsg_focal_fwd(const void* pred, int dtype, const void* target, int ltype, int64_t P,
                  int64_t ignore_label, float gamma, float alpha, float* loss, void* ws,
                  size_t ws_bytes, void* stream) {
  if (!pred || !target || !loss || !ws) return TSG_E_NULL;
  if (P <= 0) return TSG_E_SHAPE;
  if (ws_bytes < tsg_focal_ws_bytes(P)) return TSG_E_WS;
  hipStream_t st = (hipStream_t)stream;
  const int grid = fgrid(P);
  const float ig = (float)ignore_label;
#define GO(T, LT) hipLaunchKernelGGL((focal_fwd_k<T, LT>), dim3(grid), dim3(kT), 0, st, (const T*)pred, \
                                     target, P, ig, gamma, alpha, (float*)ws)
  if (dtype == TSG_F32) { if (ltype == TSG_I64) GO(float, TSG_I64); else if (ltype == TSG_U8) GO(float, TSG_U8); else return TSG_E_DTYPE; }
  else if (dtype == TSG_BF16) { if (ltype == TSG_I64) GO(bf16_t, TSG_I64); else if (ltype == TSG_U8) GO(bf16_t, TSG_U8); else return TSG_E_DTYPE; }
  else return TSG_E_DTYPE;
#undef GO
  TSG_CHECK_LAUNCH();
  hipLaunchKernelGGL(focal_finish_k, dim3(1), dim3(kT), 0, st, (const float*)ws, grid, (double)P, loss);
  TSG_CHECK_LAUNCH();
  return 0;
}

int tsg_focal_bwd(const void* pred, int dtype, const void* target, int ltype, int64_t P,
                  int64_t ignore_label, float gamma, float alpha, const float* gscale, void* dpred,
                  void* stream) {
  if (!pred || !target || !gscale || !dpred) return TSG_E_NULL;
  if (P <= 0) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int grid = fgrid(P) * 4;
  const float ig = (float)ignore_label;
#define GO(T, LT) hipLaunchKernelGGL((focal_bwd_k<T, LT>), dim3(grid), dim3(kT), 0, st, (const T*)pred, \
                                     target, P, ig, gamma, alpha, gscale, (T*)dpred)
  if (dtype == TSG_F32) { if (ltype == TSG_I64) GO(float, TSG_I64); else if (ltype == TSG_U8) GO(float, TSG_U8); else return TSG_E_DTYPE; }
  else if (dtype == TSG_BF16) { if (ltype == TSG_I64) GO(bf16_t, TSG_I64); else if (ltype == TSG_U8) GO(bf16_t, TSG_U8); else return TSG_E_DTYPE; }
  else return TSG_E_DTYPE;
#undef GO
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
