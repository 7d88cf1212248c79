// Evaluation metric on the GPU: the confusion matrix of
// furnace/seg_opr/metric.py:9-19 (`hist_info`) with the class arg-max of the
// evaluator (furnace/engine/evaluator.py: `pred = score.argmax(...)`) fused in,
// so the full-resolution score map is read once and never written back.
//
//   k        = (gt >= 0) & (gt < n_cl)
//   hist     = bincount(n_cl * gt[k] + pred[k], minlength = n_cl^2)
//   labeled  = sum(k);  correct = sum(pred[k] == gt[k])
//
// Integer work: per-block LDS histogram (integer atomics commute => the result
// does not depend on scheduling), flushed with 64-bit global atomics.  HBM-bound:
// C * elem_size + label bytes per pixel for the fused form.
#include "tsg_common.h"

namespace tsg {

template <typename G> __device__ __forceinline__ long load_label(const G* p, int64_t i) { return (long)p[i]; }

struct ConfAcc {
  uint32_t* hist;       // LDS, n_cl * n_cl
  unsigned long long labeled, correct, invalid;
  int n_cl;
  __device__ __forceinline__ void add(long g, long p) {
    if (g < 0 || g >= n_cl) return;
    if (p < 0 || p >= n_cl) { ++invalid; return; }      // numpy's bincount/reshape would raise here
    ++labeled;
    correct += (p == g);
    atomicAdd(&hist[g * n_cl + p], 1u);
  }
};

__device__ __forceinline__ void conf_flush(ConfAcc& a, unsigned long long* out, unsigned long long* cnt_sm) {
  __syncthreads();
  const int nn = a.n_cl * a.n_cl;
  for (int i = threadIdx.x; i < nn; i += blockDim.x) {
    const uint32_t v = a.hist[i];
    if (v) atomicAdd(&out[i], (unsigned long long)v);
  }
  unsigned long long l = a.labeled, c = a.correct, b = a.invalid;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    l += __shfl_xor(l, o, 64); c += __shfl_xor(c, o, 64); b += __shfl_xor(b, o, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { cnt_sm[3 * w] = l; cnt_sm[3 * w + 1] = c; cnt_sm[3 * w + 2] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    l = c = b = 0;
    for (int i = 0; i < nw; ++i) { l += cnt_sm[3 * i]; c += cnt_sm[3 * i + 1]; b += cnt_sm[3 * i + 2]; }
    if (l) atomicAdd(&out[nn], l);
    if (c) atomicAdd(&out[nn + 1], c);
    if (b) atomicAdd(&out[nn + 2], b);
  }
}

__device__ __forceinline__ ConfAcc conf_init(uint32_t* hist, int n_cl) {
  for (int i = threadIdx.x; i < n_cl * n_cl; i += blockDim.x) hist[i] = 0u;
  __syncthreads();
  ConfAcc a; a.hist = hist; a.labeled = a.correct = a.invalid = 0; a.n_cl = n_cl;
  return a;
}

template <typename PT, typename GT>
__global__ __launch_bounds__(256) void confusion_map_k(const PT* __restrict__ pred, const GT* __restrict__ gt,
                                                       int64_t P, int n_cl, unsigned long long* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
  __shared__ unsigned long long cnt_sm[12];
  ConfAcc a = conf_init(hist, n_cl);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256)
    a.add(load_label(gt, i), load_label(pred, i));
  conf_flush(a, out, cnt_sm);
}

// first maximum in channel order; a NaN wins over any number and the first NaN is kept (numpy / torch argmax)
__device__ __forceinline__ void argmax_step(float v, int c, float& best, int& idx) {
  if (v > best || (v != v && best == best)) { best = v; idx = c; }
}

template <typename T, typename GT>
__global__ __launch_bounds__(256) void confusion_logits_k(const T* __restrict__ z, const GT* __restrict__ gt,
                                                          int64_t B, int C, int64_t HW, int n_cl,
                                                          unsigned long long* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t hist[];
  __shared__ unsigned long long cnt_sm[12];
  constexpr int V = Vec<T>::N;
  ConfAcc a = conf_init(hist, n_cl);
  // 16-byte loads need every channel plane aligned: HW a multiple of the vector width, else all pixels go scalar
  const int64_t nv = (HW % V == 0) ? HW / V : 0, per_b = nv;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B * per_b; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / per_b, hw = (i % per_b) * V;
    const T* zp = z + b * C * HW + hw;
    Vec<T> v;
    v.load(zp);
    float best[V]; int idx[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { best[j] = v.v[j]; idx[j] = 0; }
    for (int c = 1; c < C; ++c) {
      v.load(zp + (int64_t)c * HW);
#pragma unroll
      for (int j = 0; j < V; ++j) argmax_step(v.v[j], c, best[j], idx[j]);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) a.add(load_label(gt, b * HW + hw + j), idx[j]);
  }
  const int64_t tail = HW - nv * V;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B * tail; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / tail, hw = nv * V + i % tail;
    const T* zp = z + b * C * HW + hw;
    float best = ld1<T>(zp); int idx = 0;
    for (int c = 1; c < C; ++c) argmax_step(ld1<T>(zp + (int64_t)c * HW), c, best, idx);
    a.add(load_label(gt, b * HW + hw), idx);
  }
  conf_flush(a, out, cnt_sm);
}

}  // namespace tsg

using namespace tsg;

namespace {
template <typename F> int with_lds(F kernel_ptr, size_t lds) {
  if (lds > 48 * 1024)
    TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel_ptr), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
  return 0;
}
int grid_for(int64_t items) {
  int64_t g = (items + 255) / 256;
  if (g > 2048) g = 2048;
  return (int)(g < 1 ? 1 : g);
}
}  // namespace

extern "C" {

#define TSG_CONF_CASE(KERN, ...)                                                         \
  do {                                                                                   \
    int e = with_lds(&KERN, lds);                                                        \
    if (e) return e;                                                                     \
    hipLaunchKernelGGL(KERN, dim3(grid), dim3(256), lds, st, __VA_ARGS__);               \
    TSG_CHECK_LAUNCH();                                                                  \
    return 0;                                                                            \
  } while (0)

int tsg_confusion_map(const void* pred, int pred_dtype, const void* gt, int gt_dtype, int64_t P, int n_cl,
                      int64_t* out, void* stream) {
  if (!pred || !gt || !out) return TSG_E_NULL;
  if (P < 0 || n_cl <= 0 || n_cl > 192) return TSG_E_SHAPE;
  if ((pred_dtype != TSG_I64 && pred_dtype != TSG_U8) || (gt_dtype != TSG_I64 && gt_dtype != TSG_U8)) return TSG_E_DTYPE;
  if (P == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)n_cl * n_cl * sizeof(uint32_t);
  const int grid = grid_for(P);
  unsigned long long* o = (unsigned long long*)out;
  if (pred_dtype == TSG_I64 && gt_dtype == TSG_I64)
    TSG_CONF_CASE((confusion_map_k<int64_t, int64_t>), (const int64_t*)pred, (const int64_t*)gt, P, n_cl, o);
  if (pred_dtype == TSG_I64 && gt_dtype == TSG_U8)
    TSG_CONF_CASE((confusion_map_k<int64_t, uint8_t>), (const int64_t*)pred, (const uint8_t*)gt, P, n_cl, o);
  if (pred_dtype == TSG_U8 && gt_dtype == TSG_I64)
    TSG_CONF_CASE((confusion_map_k<uint8_t, int64_t>), (const uint8_t*)pred, (const int64_t*)gt, P, n_cl, o);
  TSG_CONF_CASE((confusion_map_k<uint8_t, uint8_t>), (const uint8_t*)pred, (const uint8_t*)gt, P, n_cl, o);
}

int tsg_confusion_logits(const void* logits, int dtype, const void* gt, int gt_dtype, int64_t B, int C, int64_t HW,
                         int n_cl, int64_t* out, void* stream) {
  if (!logits || !gt || !out) return TSG_E_NULL;
  if (B < 0 || C <= 0 || HW < 0 || n_cl <= 0 || n_cl > 192) return TSG_E_SHAPE;
  if ((dtype != TSG_F32 && dtype != TSG_BF16) || (gt_dtype != TSG_I64 && gt_dtype != TSG_U8)) return TSG_E_DTYPE;
  if (B == 0 || HW == 0) return 0;
  const int V = dtype == TSG_F32 ? 4 : 8;
  if (HW % V == 0 && !aligned16(logits)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)n_cl * n_cl * sizeof(uint32_t);
  const int grid = grid_for(HW % V == 0 ? B * (HW / V) : B * HW);
  unsigned long long* o = (unsigned long long*)out;
  if (dtype == TSG_F32 && gt_dtype == TSG_I64)
    TSG_CONF_CASE((confusion_logits_k<float, int64_t>), (const float*)logits, (const int64_t*)gt, B, C, HW, n_cl, o);
  if (dtype == TSG_F32 && gt_dtype == TSG_U8)
    TSG_CONF_CASE((confusion_logits_k<float, uint8_t>), (const float*)logits, (const uint8_t*)gt, B, C, HW, n_cl, o);
  if (dtype == TSG_BF16 && gt_dtype == TSG_I64)
    TSG_CONF_CASE((confusion_logits_k<bf16_t, int64_t>), (const bf16_t*)logits, (const int64_t*)gt, B, C, HW, n_cl, o);
  TSG_CONF_CASE((confusion_logits_k<bf16_t, uint8_t>), (const bf16_t*)logits, (const uint8_t*)gt, B, C, HW, n_cl, o);
}

}  // extern "C"
