// Fused SGD step over a flat fp32 bucket (torch.optim.SGD as configured at
// model/bisenet/cityscapes.bisenet.R18/train.py:86-89: momentum 0.9, weight
// decay per group, dampening 0, no nesterov):
//   g = grad*grad_scale + wd*p ; buf = first ? g : momentum*buf + g ; p -= lr*buf
// One read of (p, grad, buf), one write of (p, buf): 20 B/element, HBM-bound.
#include "tsg_common.h"

namespace tsg {
__global__ __launch_bounds__(256) void sgd_k(float* __restrict__ p, const float* __restrict__ g,
                                             float* __restrict__ buf, int64_t n, float lr, float mom,
                                             float wd, float gs, int first, int vec) {
  const int64_t n4 = vec ? n / 4 : 0;     // 16-byte path only when all three pointers are aligned
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(buf)[i];
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = gg[j] * gs + wd * pp[j];
      bb[j] = first ? d : mom * bb[j] + d;
      pp[j] -= lr * bb[j];
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(buf)[i] = make_float4(bb[0], bb[1], bb[2], bb[3]);
  }
  // tail
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = g[i] * gs + wd * p[i];
    const float b = first ? d : mom * buf[i] + d;
    buf[i] = b;
    p[i] -= lr * b;
  }
}
}  // namespace tsg

namespace tsg {
// same update with the learning rate read from device memory, so a captured hipGraph
// keeps following the schedule (train.py:133-139 rewrites lr every iteration)
__global__ __launch_bounds__(256) void sgd_dev_k(float* __restrict__ p, const float* __restrict__ g,
                                                 float* __restrict__ buf, int64_t n,
                                                 const float* __restrict__ lr_dev, float lr_mult, float mom,
                                                 float wd, float gs, int vec) {
  const float lr = lr_dev[0] * lr_mult;
  const int64_t n4 = vec ? n / 4 : 0;     // gradients that are views into a DDP bucket may be unaligned
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 bv = reinterpret_cast<float4*>(buf)[i];
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = gg[j] * gs + wd * pp[j];
      bb[j] = mom * bb[j] + d;
      pp[j] -= lr * bb[j];
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    reinterpret_cast<float4*>(buf)[i] = make_float4(bb[0], bb[1], bb[2], bb[3]);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = g[i] * gs + wd * p[i];
    const float b = mom * buf[i] + d;
    buf[i] = b;
    p[i] -= lr * b;
  }
}
}  // namespace tsg

extern "C" int tsg_sgd_step_dev(float* param, const float* grad, float* momentum_buf, int64_t n,
                                const float* lr_dev, float lr_mult, float momentum, float weight_decay,
                                float grad_scale, void* stream) {
  if (!param || !grad || !momentum_buf || !lr_dev) return TSG_E_NULL;
  if (n <= 0) return TSG_E_SHAPE;
  const int vec = tsg::aligned16(param) && tsg::aligned16(grad) && tsg::aligned16(momentum_buf);
  int64_t g = ((vec ? n / 4 : n) + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(tsg::sgd_dev_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, param, grad,
                     momentum_buf, n, lr_dev, lr_mult, momentum, weight_decay, grad_scale, vec);
  TSG_CHECK_LAUNCH();
  return 0;
}

extern "C" int tsg_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr,
                            float momentum, float weight_decay, float grad_scale, int first_step,
                            void* stream) {
  if (!param || !grad || !momentum_buf) return TSG_E_NULL;
  if (n <= 0) return TSG_E_SHAPE;
  const int vec = tsg::aligned16(param) && tsg::aligned16(grad) && tsg::aligned16(momentum_buf);
  int64_t g = ((vec ? n / 4 : n) + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(tsg::sgd_k, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, param, grad,
                     momentum_buf, n, lr, momentum, weight_decay, grad_scale, first_step, vec);
  TSG_CHECK_LAUNCH();
  return 0;
}


// ---------------------------------------------------------------- multi-tensor step
// One launch for up to TSG_SGD_MAX_SEGS parameter tensors: the pointer table travels by value in the kernel
// argument block (copied at launch, so the host can rebuild it every step without a staging buffer), the
// block -> (tensor, chunk) map is static per model and lives in device memory.
namespace tsg {
constexpr int kSgdChunk = 4096;            // elements per block: 256 threads x 4 float4
struct SgdSegs {
  float* p[TSG_SGD_MAX_SEGS];
  const float* g[TSG_SGD_MAX_SEGS];
  float* b[TSG_SGD_MAX_SEGS];
  int n[TSG_SGD_MAX_SEGS];
  unsigned char grp[TSG_SGD_MAX_SEGS];
  float mom[TSG_SGD_MAX_GROUPS];
  float wd[TSG_SGD_MAX_GROUPS];
};
static_assert(sizeof(SgdSegs) <= 4000, "kernel argument block is limited to 4 KiB (the three other arguments take 24 B)");

__global__ __launch_bounds__(256) void sgd_multi_k(const SgdSegs s, const int2* __restrict__ map,
                                                   const float* __restrict__ lr_dev, float gs) {
  const int2 m = map[blockIdx.x];
  const int seg = m.x;
  float* __restrict__ p = s.p[seg];
  const float* __restrict__ g = s.g[seg];
  float* __restrict__ buf = s.b[seg];
  const int n = s.n[seg], grp = s.grp[seg];
  const float lr = lr_dev[grp], mom = s.mom[grp], wd = s.wd[grp];
  const int base = m.y * kSgdChunk;
  const int end = base + kSgdChunk < n ? base + kSgdChunk : n;
  const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)buf)) & 15u) == 0;
  if (vec) {
    const int end4 = base + ((end - base) & ~3);
    for (int i = base + 4 * threadIdx.x; i < end4; i += 1024) {
      float4 pv = *reinterpret_cast<float4*>(p + i);
      const float4 gv = *reinterpret_cast<const float4*>(g + i);
      float4 bv = *reinterpret_cast<float4*>(buf + i);
      float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = gg[j] * gs + wd * pp[j];
        bb[j] = mom * bb[j] + d;
        pp[j] -= lr * bb[j];
      }
      *reinterpret_cast<float4*>(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *reinterpret_cast<float4*>(buf + i) = make_float4(bb[0], bb[1], bb[2], bb[3]);
    }
    for (int i = end4 + threadIdx.x; i < end; i += 256) {
      const float d = g[i] * gs + wd * p[i];
      const float b = mom * buf[i] + d;
      buf[i] = b;
      p[i] -= lr * b;
    }
  } else {
    for (int i = base + threadIdx.x; i < end; i += 256) {
      const float d = g[i] * gs + wd * p[i];
      const float b = mom * buf[i] + d;
      buf[i] = b;
      p[i] -= lr * b;
    }
  }
}
}  // namespace tsg

extern "C" int64_t tsg_sgd_multi_blockmap(const int64_t* numel, int nseg, int* map_host, int64_t cap_blocks) {
  if (!numel || nseg <= 0 || nseg > TSG_SGD_MAX_SEGS) return TSG_E_SHAPE;
  int64_t nb = 0;
  for (int i = 0; i < nseg; ++i) {
    if (numel[i] <= 0 || numel[i] > 0x7fffffffLL) return TSG_E_SHAPE;
    const int64_t c = (numel[i] + tsg::kSgdChunk - 1) / tsg::kSgdChunk;
    if (map_host) {
      if (nb + c > cap_blocks) return TSG_E_WS;
      for (int64_t j = 0; j < c; ++j) { map_host[2 * (nb + j)] = i; map_host[2 * (nb + j) + 1] = (int)j; }
    }
    nb += c;
  }
  return nb;
}

extern "C" int tsg_sgd_multi_step_dev(const uint64_t* params, const uint64_t* grads, const uint64_t* bufs,
                                      const int64_t* numel, const int* group, int nseg, const float* lr_dev,
                                      const float* momentum, const float* weight_decay, int ngroups,
                                      const int* blockmap_dev, int64_t nblocks, float grad_scale, void* stream) {
  if (!params || !grads || !bufs || !numel || !group || !lr_dev || !momentum || !weight_decay || !blockmap_dev)
    return TSG_E_NULL;
  if (nseg <= 0 || nseg > TSG_SGD_MAX_SEGS || ngroups <= 0 || ngroups > TSG_SGD_MAX_GROUPS) return TSG_E_SHAPE;
  if (nblocks != tsg_sgd_multi_blockmap(numel, nseg, nullptr, 0)) return TSG_E_SHAPE;
  tsg::SgdSegs s = {};
  for (int i = 0; i < nseg; ++i) {
    if (!params[i] || !grads[i] || !bufs[i]) return TSG_E_NULL;
    if (group[i] < 0 || group[i] >= ngroups) return TSG_E_SHAPE;
    s.p[i] = (float*)(uintptr_t)params[i];
    s.g[i] = (const float*)(uintptr_t)grads[i];
    s.b[i] = (float*)(uintptr_t)bufs[i];
    s.n[i] = (int)numel[i];
    s.grp[i] = (unsigned char)group[i];
  }
  for (int i = 0; i < ngroups; ++i) { s.mom[i] = momentum[i]; s.wd[i] = weight_decay[i]; }
  hipLaunchKernelGGL(tsg::sgd_multi_k, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, s,
                     reinterpret_cast<const int2*>(blockmap_dev), lr_dev, grad_scale);
  TSG_CHECK_LAUNCH();
  return 0;
}


// ---------------------------------------------------------------- multi-tensor copy (gradient -> DDP bucket)
namespace tsg {
struct CopySegs {
  const float* s[TSG_SGD_MAX_SEGS];
  float* d[TSG_SGD_MAX_SEGS];
  int n[TSG_SGD_MAX_SEGS];
};

__global__ __launch_bounds__(256) void multi_copy_k(const CopySegs c, const int2* __restrict__ map, float scale) {
  const int2 m = map[blockIdx.x];
  const float* __restrict__ src = c.s[m.x];
  float* __restrict__ dst = c.d[m.x];
  const int n = c.n[m.x];
  const int base = m.y * kSgdChunk;
  const int end = base + kSgdChunk < n ? base + kSgdChunk : n;
  const bool vec = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15u) == 0;
  if (vec) {
    const int end4 = base + ((end - base) & ~3);
    for (int i = base + 4 * threadIdx.x; i < end4; i += 1024) {
      float4 v = *reinterpret_cast<const float4*>(src + i);
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
      *reinterpret_cast<float4*>(dst + i) = v;
    }
    for (int i = end4 + threadIdx.x; i < end; i += 256) dst[i] = src[i] * scale;
  } else {
    for (int i = base + threadIdx.x; i < end; i += 256) dst[i] = src[i] * scale;
  }
}
}  // namespace tsg

extern "C" int tsg_multi_copy_f32(const uint64_t* src, const uint64_t* dst, const int64_t* numel, int nseg,
                                  const int* blockmap_dev, int64_t nblocks, float scale, void* stream) {
  if (!src || !dst || !numel || !blockmap_dev) return TSG_E_NULL;
  if (nseg <= 0 || nseg > TSG_SGD_MAX_SEGS) return TSG_E_SHAPE;
  if (nblocks != tsg_sgd_multi_blockmap(numel, nseg, nullptr, 0)) return TSG_E_SHAPE;
  tsg::CopySegs c = {};
  for (int i = 0; i < nseg; ++i) {
    if (!src[i] || !dst[i]) return TSG_E_NULL;
    c.s[i] = (const float*)(uintptr_t)src[i];
    c.d[i] = (float*)(uintptr_t)dst[i];
    c.n[i] = (int)numel[i];
  }
  hipLaunchKernelGGL(tsg::multi_copy_k, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, c,
                     reinterpret_cast<const int2*>(blockmap_dev), scale);
  TSG_CHECK_LAUNCH();
  return 0;
}


// ---------------------------------------------------------------- bf16 shadows of the fp32 master filters
// The convolutions compute in bf16 (autocast), so every step each fp32 filter used to be cast to bf16 again (one tiny
// launch per convolution) and, for the layers whose data gradient runs as a forward convolution, rotated / transposed
// by another one.  This kernel refreshes all of them in ONE launch, right after the optimizer step: table[e] (device
// memory, static per model) = {w, wb, wrt, n, O, I}; wb[i] = bf16(w[i]) in the filter's own memory order; for 3x3
// filters stored [O][kh][kw][I] (channels_last) also wrt[ci][2 - kh][2 - kw][o] = bf16(w[o][kh][kw][ci]) when wrt != 0.
namespace tsg {
struct ShadowEntry {
  const float* w;
  bf16_t* wb;
  bf16_t* wrt;
  bf16_t* wf0;        // round 5: the filter in MFMA fragment order for tsg_conv3x3_gen_fwd (g3_prep_filter_k mode 0), or 0
  bf16_t* wf1;        // ... and for the data gradient (mode 1: rot180 + transpose; tile width bn1 = 32 for tsg_conv3x3_s2_dgrad)
  int n, O, I, bn0, bn1, pad;
};
static_assert(sizeof(ShadowEntry) == 64, "layout shared with torchseg_amd/shadow.py");

// element offset of W'[oc][tap][ci] in the fragment-order image of csrc/conv3g.hip (g3_prep_filter_k):
//   out[oc tile][chunk][tap][ocb][lane][e],  oc = tile BN + ocb 32 + (lane & 31),  ci = chunk 16 + (lane >> 5) 8 + e
__device__ __forceinline__ int64_t g3_frag_offset(int oc, int tap, int ci, int Ci, int BN) {
  const int nch = Ci >> 4, ocb_n = BN >> 5;
  const int tile = oc / BN, rem = oc - tile * BN, ocb = rem >> 5, ln = ((ci >> 3) & 1) * 32 + (rem & 31);
  return ((((int64_t)(tile * nch + (ci >> 4)) * 9 + tap) * ocb_n + ocb) * 64 + ln) * 8 + (ci & 7);
}

// map[block] = {entry, chunk | kind << 28}.  kind 0: 4096 SOURCE elements of the entry -> wb, wrt, wf0 (and wf1, when the host
// asks for it there: TSG_SHADOW_WF1_PASS=0).  kind 1 (round 6): 4096 DESTINATION elements of wf1.  Walking the source, the data-
// gradient image was written 2 bytes at a time, 16 bytes apart (consecutive c_in are consecutive LANES of a fragment, 8
// consecutive c_out its 16 bytes): every store a partial line — the launch took 98 us for 52 MB of parameters.  Walking the
// destination a thread builds one 16-byte vector from 8 loads w[o + j][tap][ci], each of which is coalesced ACROSS the lanes
// (32 consecutive ci), and a wave stores 1 KB contiguous.
constexpr int kShadowKindShift = 28;

__global__ __launch_bounds__(256) void weight_shadow_k(const ShadowEntry* __restrict__ table, const int2* __restrict__ map) {
  const int2 m = map[blockIdx.x];
  const ShadowEntry e = table[m.x];
  const int kind = m.y >> kShadowKindShift, chunk = m.y & ((1 << kShadowKindShift) - 1);
  const int base = chunk * kSgdChunk;
  const int end = base + kSgdChunk < e.n ? base + kSgdChunk : e.n;
  const int tapI = 9 * e.I;
  if (kind == 1) {                                     // e.n = 9 O I elements of wf1, 8 per thread and round
    const int nch = e.O >> 4, ocb_n = e.bn1 >> 5;
    for (int d = base + threadIdx.x * 8; d < end; d += 256 * 8) {
      const int ln = (d >> 3) & 63;
      int rest = d >> 9;
      const int ocb = rest % ocb_n; rest /= ocb_n;
      const int tp = rest % 9; rest /= 9;
      const int c16 = rest % nch, tile = rest / nch;
      const int ci = tile * e.bn1 + ocb * 32 + (ln & 31);              // W'[ci][tp][o] = w[o][8 - tp][ci]
      const int o0 = c16 * 16 + (ln >> 5) * 8;
      const float* src = e.w + ((int64_t)o0 * 9 + (8 - tp)) * e.I + ci;
      uint32_t pk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        pk[j] = (uint32_t)f32_to_bf16(src[(int64_t)(2 * j) * tapI]) | ((uint32_t)f32_to_bf16(src[(int64_t)(2 * j + 1) * tapI]) << 16);
      *reinterpret_cast<uint4*>(e.wf1 + d) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
    }
    return;
  }
  if (kind == 2) {                                     // the FORWARD image by destination: W'[oc][tp][ci0 .. ci0 + 7] = w[oc][tp][ci0 ..]
    const int nch = e.I >> 4, ocb_n = e.bn0 >> 5;
    for (int d = base + threadIdx.x * 8; d < end; d += 256 * 8) {
      const int ln = (d >> 3) & 63;
      int rest = d >> 9;
      const int ocb = rest % ocb_n; rest /= ocb_n;
      const int tp = rest % 9; rest /= 9;
      const int c16 = rest % nch, tile = rest / nch;
      const int oc = tile * e.bn0 + ocb * 32 + (ln & 31), ci0 = c16 * 16 + (ln >> 5) * 8;
      const float4* src = reinterpret_cast<const float4*>(e.w + ((int64_t)oc * 9 + tp) * e.I + ci0);
      const float4 a = src[0], b = src[1];
      *reinterpret_cast<uint4*>(e.wf0 + d) = make_uint4(
          (uint32_t)f32_to_bf16(a.x) | ((uint32_t)f32_to_bf16(a.y) << 16), (uint32_t)f32_to_bf16(a.z) | ((uint32_t)f32_to_bf16(a.w) << 16),
          (uint32_t)f32_to_bf16(b.x) | ((uint32_t)f32_to_bf16(b.y) << 16), (uint32_t)f32_to_bf16(b.z) | ((uint32_t)f32_to_bf16(b.w) << 16));
    }
    return;
  }
  const bool wf1_here = e.wf1 && (e.pad & 1) == 0;     // pad bit 0: wf1 has blocks of kind 1, bit 1: wf0 has blocks of kind 2
  const bool wf0_here = e.wf0 && (e.pad & 2) == 0;
  const bool geo = e.wrt || wf0_here || wf1_here;
  for (int i = base + threadIdx.x; i < end; i += 256) {
    const bf16_t v = f32_to_bf16(e.w[i]);
    if (e.wb) e.wb[i] = v;
    if (geo) {                                         // w is [O][3][3][I] (a channels_last 3x3 filter)
      const int o = i / tapI, r = i - o * tapI, tap = r / e.I, ci = r - tap * e.I;
      if (e.wrt) e.wrt[(ci * 9 + (8 - tap)) * e.O + o] = v;
      if (wf0_here) e.wf0[g3_frag_offset(o, tap, ci, e.I, e.bn0)] = v;          // W' = w: C_out' = O, C_in' = I
      if (wf1_here) e.wf1[g3_frag_offset(ci, 8 - tap, o, e.O, e.bn1)] = v;      // W'[ci][8 - tap][o] = w[o][tap][ci]
    }
  }
}
}  // namespace tsg

extern "C" size_t tsg_weight_shadow_entry_bytes(void) { return sizeof(tsg::ShadowEntry); }

extern "C" int tsg_weight_shadow_refresh(const void* table_dev, const int* blockmap_dev, int64_t nblocks, void* stream) {
  if (!table_dev || !blockmap_dev) return TSG_E_NULL;
  if (nblocks <= 0 || nblocks > 0x7fffffffLL) return TSG_E_SHAPE;
  hipLaunchKernelGGL(tsg::weight_shadow_k, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                     (const tsg::ShadowEntry*)table_dev, reinterpret_cast<const int2*>(blockmap_dev));
  TSG_CHECK_LAUNCH();
  return 0;
}
