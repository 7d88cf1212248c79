// OHEM 2-D cross entropy for gfx950.
//
// Restates ProbOhemCrossEntropy2d.forward (furnace/seg_opr/loss_opr.py:68-98):
//   valid = t != ignore ; prob = softmax(pred, 1) ; mask_prob = prob[t] (1 where
//   invalid) ; thr = max(thresh, k-th smallest mask_prob) with k = min(P,
//   min_kept) ; kept = mask_prob <= thr ; loss = CE(pred, t | valid & kept).
// The reference materialises the softmax, transposes it, runs a full
// torch.sort over P = B*H*W floats and a second log-softmax inside CE
// (>= 6 passes over the logits + an O(P log P) sort).  Here:
//   pass A  one read of the logits: online softmax per pixel -> nll = lse - x_t
//           and lse (8 B/pixel), block partial sums for the `thr == thresh`
//           outcome, and a coarse histogram of the IEEE bit patterns of
//           p_t = exp(-nll) above thresh (p in [0,1] => bit patterns are
//           monotone, so a radix select returns exactly sort(p)[k-1]);
//   decide  (1 block) picks the branch; when the k-th value lies above thresh,
//           1-2 refinement passes over nll (4 B/pixel) pin it bit-exactly and
//           pass C re-sums nll over p <= thr;
//   bwd     one read of logits + one write of dlogits, skipping reads for
//           pixels that were not kept.
// Everything stays on the device (no host sync); all cross-block reductions are
// integer atomics or fixed-order partials, so results are deterministic.
//
// HBM-bound: algorithmic bytes per pixel = C*s (fwd) + 2*C*s (bwd) + ~33 B side
// arrays (labels i64 x2, nll w+r x2, lse w+r).
#include "tsg_common.h"
#include "tsg_resample.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>

namespace tsg {

constexpr int kT = 256;

struct SelState {
  int32_t branch;       // 0: thr = thresh; 1: thr = k-th value (> thresh); 2: no OHEM
  int32_t pad0;
  int64_t num_valid;
  int64_t n_kept;
  int64_t krem;         // remaining 1-based rank inside the current prefix
  int64_t lo_d;         // lower bound of the current prefix in d-space
  uint32_t thr_bits;    // final threshold (float bits)
  int32_t n_bad;        // labels outside [0, C) other than ignore_label (the reference device-asserts on them)
  double sum;           // sum of w*nll over kept
  double wsum;          // sum of w over kept (denominator when weighted)
};

struct BlkPart {
  float sum_le, sum_valid, wsum_le, wsum_valid;
  int32_t cnt_le_all, cnt_le_valid, cnt_valid, cnt_bad;  // cnt_bad: labels that are neither ignore_label nor in [0, C)
};

// ---- workspace carving ------------------------------------------------------
struct OhemWs {
  SelState* st;
  uint32_t* hist[3];
  BlkPart* part;      // [grid]
  float* csum;        // [grid] pass C
  float* cwsum;       // [grid]
  int32_t* ccnt;      // [grid]
  size_t zero_bytes;  // leading bytes that must be zeroed per call (state + hists)
  size_t total;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static OhemWs carve(void* base, const tsg_ohem_plan& pl) {
  OhemWs w;
  char* p = (char*)base;
  size_t off = 0;
  w.st = (SelState*)(p + off); off += align_up(sizeof(SelState), 256);
  for (int l = 0; l < 3; ++l) {
    w.hist[l] = (uint32_t*)(p + off);
    off += align_up((size_t)(l < pl.levels ? pl.bins[l] : 0) * sizeof(uint32_t), 256);
  }
  w.zero_bytes = off;
  w.part = (BlkPart*)(p + off); off += align_up((size_t)pl.grid * sizeof(BlkPart), 256);
  w.csum = (float*)(p + off); off += align_up((size_t)pl.grid * sizeof(float), 256);
  w.cwsum = (float*)(p + off); off += align_up((size_t)pl.grid * sizeof(float), 256);
  w.ccnt = (int32_t*)(p + off); off += align_up((size_t)pl.grid * sizeof(int32_t), 256);
  w.total = off;
  return w;
}

// d-space: d = bits(p) - tb - 1 for p > thresh (tb = bits(thresh), or -1 when
// thresh < 0 so that every p >= 0 maps to d >= 0).
static void level_plan(int64_t range, int* levels, int* shift, int* bins) {
  int nb = 0;
  while (((int64_t)1 << nb) < range) ++nb;
  if (range <= 0) { *levels = 0; shift[0] = shift[1] = shift[2] = 0; bins[0] = bins[1] = bins[2] = 0; return; }
  shift[0] = nb > 11 ? nb - 11 : 0;
  bins[0] = (int)((range + ((int64_t)1 << shift[0]) - 1) >> shift[0]);
  if (shift[0] == 0) { *levels = 1; shift[1] = shift[2] = 0; bins[1] = bins[2] = 0; return; }
  shift[1] = shift[0] > 12 ? shift[0] - 12 : 0;
  bins[1] = 1 << (shift[0] - shift[1]);
  if (shift[1] == 0) { *levels = 2; shift[2] = 0; bins[2] = 0; return; }
  shift[2] = 0;
  bins[2] = 1 << shift[1];
  *levels = 3;
}

__device__ __forceinline__ float prob_of_nll(float nll) { return expf(-nll); }

// A label takes part in the loss iff it is not ignore_label AND names a class.  The reference indexes
// prob[target] / weight[target] with it (loss_opr.py:82-83,98), which device-asserts on anything else; here such a
// pixel is dropped like an ignored one and counted (sel[5]), so no kernel ever indexes weight[] / the class loop
// with an unchecked value.
__device__ __forceinline__ bool label_is_class(int64_t lab, int64_t ignore_label, int C) {
  return lab != ignore_label && (uint64_t)lab < (uint64_t)C;
}

template <int LT> struct Lab;
template <> struct Lab<TSG_I64> {
  typedef int64_t type;
  static __device__ __forceinline__ int64_t get(const void* p, int64_t i) { return ((const int64_t*)p)[i]; }
};
template <> struct Lab<TSG_U8> {
  typedef uint8_t type;
  static __device__ __forceinline__ int64_t get(const void* p, int64_t i) { return (int64_t)((const uint8_t*)p)[i]; }
};

template <typename T, int V> struct PixVec;
template <> struct PixVec<float, 4> : Vec<float> {};
template <> struct PixVec<bf16_t, 8> : Vec<bf16_t> {};
template <typename T> struct PixVec<T, 1> {
  float v[1];
  __device__ __forceinline__ void load(const T* p) { v[0] = ld1<T>(p); }
  __device__ __forceinline__ void store(T* p) const { st1<T>(p, v[0]); }
};

// per-thread running sums of pass A (both outcomes of the threshold decision are
// prepared, see ohem_decide0) and their deterministic block fold
struct PassAcc {
  float sum_le = 0.f, sum_valid = 0.f, wsum_le = 0.f, wsum_valid = 0.f;
  int cnt_le_all = 0, cnt_le_valid = 0, cnt_valid = 0, cnt_bad = 0;
};

__device__ __forceinline__ void account(PassAcc& a, bool valid, float nl, float w, float thresh,
                                        int64_t tb, int shift0, int bins0, uint32_t* lh) {
  const float pr = valid ? prob_of_nll(nl) : 1.f;  // masked_fill_(~valid, 1), loss_opr.py:81
  if (valid) { a.cnt_valid++; a.sum_valid += w * nl; a.wsum_valid += w; }
  if (pr <= thresh) {
    a.cnt_le_all++;
    if (valid) { a.cnt_le_valid++; a.sum_le += w * nl; a.wsum_le += w; }
  } else if (bins0 > 0) {
    const int64_t d = (int64_t)__float_as_uint(pr) - tb - 1;
    int bin = (int)(d >> shift0);
    bin = bin < 0 ? 0 : (bin >= bins0 ? bins0 - 1 : bin);   // NaN logits must not index outside the histogram
    atomicAdd(&lh[bin], 1u);
  }
}

// flush the LDS histogram and write this block's partial (fixed order); contains barriers
__device__ __forceinline__ void fold_block(PassAcc& a, const uint32_t* lh, int bins0,
                                           uint32_t* __restrict__ hist0, BlkPart* __restrict__ part) {
  __shared__ float fsm[4 * (kT / 64)];
  __shared__ int ism[4 * (kT / 64)];
  const int tid = threadIdx.x;
  __syncthreads();
  for (int i = tid; i < bins0; i += kT) {
    const uint32_t h = lh[i];
    if (h) atomicAdd(&hist0[i], h);
  }
  a.sum_le = wave_sum(a.sum_le); a.sum_valid = wave_sum(a.sum_valid);
  a.wsum_le = wave_sum(a.wsum_le); a.wsum_valid = wave_sum(a.wsum_valid);
  a.cnt_le_all = wave_sum(a.cnt_le_all); a.cnt_le_valid = wave_sum(a.cnt_le_valid); a.cnt_valid = wave_sum(a.cnt_valid);
  a.cnt_bad = wave_sum(a.cnt_bad);
  const int lane = tid & 63, wv = tid >> 6;
  if (lane == 0) {
    fsm[wv * 4 + 0] = a.sum_le; fsm[wv * 4 + 1] = a.sum_valid; fsm[wv * 4 + 2] = a.wsum_le; fsm[wv * 4 + 3] = a.wsum_valid;
    ism[wv * 4 + 0] = a.cnt_le_all; ism[wv * 4 + 1] = a.cnt_le_valid; ism[wv * 4 + 2] = a.cnt_valid; ism[wv * 4 + 3] = a.cnt_bad;
  }
  __syncthreads();
  if (tid == 0) {
    BlkPart bp = {0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0};
    for (int i = 0; i < kT / 64; ++i) {
      bp.sum_le += fsm[i * 4 + 0]; bp.sum_valid += fsm[i * 4 + 1];
      bp.wsum_le += fsm[i * 4 + 2]; bp.wsum_valid += fsm[i * 4 + 3];
      bp.cnt_le_all += ism[i * 4 + 0]; bp.cnt_le_valid += ism[i * 4 + 1]; bp.cnt_valid += ism[i * 4 + 2]; bp.cnt_bad += ism[i * 4 + 3];
    }
    part[blockIdx.x] = bp;
  }
}

// =============================================================================
// pass A
// =============================================================================
template <typename T, int V, int LT>
__global__ __launch_bounds__(kT) void ohem_pass_a(
    const T* __restrict__ logits, const void* __restrict__ labels, int64_t P, int C,
    int64_t HW, int64_t ignore_label, float thresh, int64_t tb, int shift0, int bins0,
    const float* __restrict__ weight, float* __restrict__ nll_out,
    float* __restrict__ lse_out, uint32_t* __restrict__ hist0, BlkPart* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lh[];  // bins0 + reduction scratch
  const int tid = threadIdx.x;
  for (int i = tid; i < bins0; i += kT) lh[i] = 0;
  __syncthreads();

  PassAcc acc;
  const int64_t nvec = P / V;  // V divides HW (checked on the host) hence P
  for (int64_t v = (int64_t)blockIdx.x * kT + tid; v < nvec; v += (int64_t)gridDim.x * kT) {
    const int64_t p0 = v * V;
    const int64_t b = p0 / HW;
    const int64_t q = p0 - b * HW;
    const T* base = logits + (b * C) * HW + q;
    int t[V];
    bool valid[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int64_t lab = Lab<LT>::get(labels, p0 + j);
      valid[j] = label_is_class(lab, ignore_label, C);
      if (!valid[j] && lab != ignore_label) acc.cnt_bad++;
      t[j] = valid[j] ? (int)lab : 0;  // loss_opr.py:72
    }
    float m[V], s[V], xt[V];
    {
      PixVec<T, V> px;
      px.load(base);
#pragma unroll
      for (int j = 0; j < V; ++j) { m[j] = px.v[j]; s[j] = 1.f; xt[j] = px.v[j]; }
    }
#pragma unroll 4
    for (int c = 1; c < C; ++c) {
      PixVec<T, V> px;
      px.load(base + (int64_t)c * HW);
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const float x = px.v[j];
        if (t[j] == c) xt[j] = x;
        const float mn = fmaxf(m[j], x);
        s[j] = s[j] * __expf(m[j] - mn) + __expf(x - mn);
        m[j] = mn;
      }
    }
    float nl[V], ls[V];
#pragma unroll
    for (int j = 0; j < V; ++j) {
      ls[j] = m[j] + logf(s[j]);
      float n_ = ls[j] - xt[j];
      n_ = n_ < 0.f ? 0.f : n_;
      nl[j] = valid[j] ? n_ : 0.f;
    }
    // side arrays (vector stores when V is the native width)
    if (V == 4) {
      *reinterpret_cast<float4*>(nll_out + p0) = make_float4(nl[0], nl[1 % V], nl[2 % V], nl[3 % V]);
      *reinterpret_cast<float4*>(lse_out + p0) = make_float4(ls[0], ls[1 % V], ls[2 % V], ls[3 % V]);
    } else if (V == 8) {
      *reinterpret_cast<float4*>(nll_out + p0) = make_float4(nl[0], nl[1 % V], nl[2 % V], nl[3 % V]);
      *reinterpret_cast<float4*>(nll_out + p0 + 4) = make_float4(nl[4 % V], nl[5 % V], nl[6 % V], nl[7 % V]);
      *reinterpret_cast<float4*>(lse_out + p0) = make_float4(ls[0], ls[1 % V], ls[2 % V], ls[3 % V]);
      *reinterpret_cast<float4*>(lse_out + p0 + 4) = make_float4(ls[4 % V], ls[5 % V], ls[6 % V], ls[7 % V]);
    } else {
      nll_out[p0] = nl[0];
      lse_out[p0] = ls[0];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float w = (weight && valid[j]) ? weight[t[j]] : 1.f;
      account(acc, valid[j], nl[j], w, thresh, tb, shift0, bins0, lh);
    }
  }
  fold_block(acc, lh, bins0, hist0, part);
}

// =============================================================================
// decide: single block.  Reduces the block partials (fp64, fixed order), picks
// the branch (loss_opr.py:78-90) and, on the k-th-value branch, walks hist0.
// =============================================================================
__device__ void scan_hist(const uint32_t* hist, int bins, int shift, SelState* st) {
  // thread 0 only: find the bin holding rank krem
  int64_t r = st->krem, cum = 0;
  int b = 0;
  for (; b < bins; ++b) {
    const int64_t h = hist[b];
    if (cum + h >= r) break;
    cum += h;
  }
  if (b >= bins) b = bins - 1;  // cannot happen when counts are consistent
  st->krem = r - cum;
  st->lo_d += (int64_t)b << shift;
}

__global__ __launch_bounds__(kT) void ohem_decide0(
    const BlkPart* __restrict__ part, int grid, const uint32_t* __restrict__ hist0, int bins0,
    int shift0, int levels, int64_t P, int64_t min_kept, float thresh, int64_t tb,
    int weighted, SelState* __restrict__ st) {
  __shared__ double dsm[4][kT / 64];
  __shared__ long long lsm[3][kT / 64];
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  long long c0 = 0, c1 = 0, c2 = 0;
  int nbad = 0;
  // fixed assignment of partials to threads, fixed combine order => deterministic
  for (int i = threadIdx.x; i < grid; i += kT) {
    const BlkPart bp = part[i];
    a0 += bp.sum_le; a1 += bp.sum_valid; a2 += bp.wsum_le; a3 += bp.wsum_valid;
    c0 += bp.cnt_le_all; c1 += bp.cnt_le_valid; c2 += bp.cnt_valid;
    nbad += bp.cnt_bad;
  }
  if (threadIdx.x == 0) st->n_bad = 0;
  __syncthreads();
  if (nbad) atomicAdd(&st->n_bad, nbad);   // integer: order-independent
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
  c0 = (long long)wave_sum((double)c0); c1 = (long long)wave_sum((double)c1); c2 = (long long)wave_sum((double)c2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    dsm[0][wv] = a0; dsm[1][wv] = a1; dsm[2][wv] = a2; dsm[3][wv] = a3;
    lsm[0][wv] = c0; lsm[1][wv] = c1; lsm[2][wv] = c2;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  a0 = a1 = a2 = a3 = 0; c0 = c1 = c2 = 0;
  for (int i = 0; i < kT / 64; ++i) {
    a0 += dsm[0][i]; a1 += dsm[1][i]; a2 += dsm[2][i]; a3 += dsm[3][i];
    c0 += lsm[0][i]; c1 += lsm[1][i]; c2 += lsm[2][i];
  }
  const int64_t num_valid = c2, cnt_le_all = c0, cnt_le_valid = c1;
  st->num_valid = num_valid;
  st->lo_d = 0;
  if (min_kept > num_valid || num_valid == 0 || min_kept <= 0) {
    // loss_opr.py:78-80 (only logs) / :80 num_valid == 0 / :85 min_kept == 0: plain CE
    st->branch = 2;
    st->n_kept = num_valid;
    st->sum = a1;
    st->wsum = weighted ? a3 : (double)num_valid;
    st->thr_bits = 0x7f800000u;  // +inf: everything valid is kept
    return;
  }
  const int64_t k = P < min_kept ? P : min_kept;  // loss_opr.py:87
  if (k <= cnt_le_all || levels == 0) {
    st->branch = 0;  // k-th smallest <= thresh  =>  threshold stays thresh (loss_opr.py:84,88)
    st->n_kept = cnt_le_valid;
    st->sum = a0;
    st->wsum = weighted ? a2 : (double)cnt_le_valid;
    st->thr_bits = __float_as_uint(thresh);
    return;
  }
  st->branch = 1;
  st->krem = k - cnt_le_all;
  scan_hist(hist0, bins0, shift0, st);
  if (levels == 1) st->thr_bits = (uint32_t)(st->lo_d + tb + 1);
}

// refinement level l >= 1: histogram of the sub-bin index of every element
// whose d lies inside the current prefix [lo_d, lo_d + 2^shift_prev).
template <int XF>
__global__ __launch_bounds__(kT) void sel_refine(
    const float* __restrict__ v, int64_t n, float thresh, int64_t tb, int shift_prev,
    int shift, int bins, uint32_t* __restrict__ hist, const SelState* __restrict__ st) {
  if (st->branch != 1) return;
  extern __shared__ __attribute__((aligned(16))) uint32_t lh[];
  for (int i = threadIdx.x; i < bins; i += kT) lh[i] = 0;
  __syncthreads();
  const int64_t lo = st->lo_d, hi = lo + ((int64_t)1 << shift_prev);
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < n; i += (int64_t)gridDim.x * kT) {
    const float pr = XF ? prob_of_nll(v[i]) : v[i];
    if (pr > thresh) {
      const int64_t d = (int64_t)__float_as_uint(pr) - tb - 1;
      if (d >= lo && d < hi) {
        int bin = (int)((d - lo) >> shift);
        if (bin < bins) atomicAdd(&lh[bin], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < bins; i += kT) {
    const uint32_t h = lh[i];
    if (h) atomicAdd(&hist[i], h);
  }
}

__global__ void sel_decide(const uint32_t* __restrict__ hist, int bins, int shift, int last,
                           int64_t tb, SelState* __restrict__ st) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || st->branch != 1) return;
  scan_hist(hist, bins, shift, st);
  if (last) st->thr_bits = (uint32_t)(st->lo_d + tb + 1);
}

// pass C: re-sum nll over valid & p <= thr (k-th value branch only)
template <int LT>
__global__ __launch_bounds__(kT) void ohem_pass_c(
    const float* __restrict__ nll, const void* __restrict__ labels, int64_t P, int C,
    int64_t ignore_label, const float* __restrict__ weight, const SelState* __restrict__ st,
    float* __restrict__ csum, float* __restrict__ cwsum, int32_t* __restrict__ ccnt) {
  if (st->branch != 1) return;
  __shared__ float sm[2 * (kT / 64)];
  __shared__ int ism[kT / 64];
  const float thr = __uint_as_float(st->thr_bits);
  float s = 0.f, ws = 0.f;
  int cnt = 0;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < P; i += (int64_t)gridDim.x * kT) {
    const int64_t lab = Lab<LT>::get(labels, i);
    if (label_is_class(lab, ignore_label, C)) {
      const float nl = nll[i];
      if (prob_of_nll(nl) <= thr) {
        const float w = weight ? weight[lab] : 1.f;
        s += w * nl; ws += w; cnt++;
      }
    }
  }
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) ism[threadIdx.x >> 6] = cnt;
  block_sum2(s, ws, sm);  // contains a __syncthreads()
  if (threadIdx.x == 0) {
    int c = 0;
    for (int i = 0; i < kT / 64; ++i) c += ism[i];
    csum[blockIdx.x] = s; cwsum[blockIdx.x] = ws; ccnt[blockIdx.x] = c;
  }
}

__global__ __launch_bounds__(kT) void ohem_finish(
    const float* __restrict__ csum, const float* __restrict__ cwsum,
    const int32_t* __restrict__ ccnt, int grid, int weighted, SelState* __restrict__ st,
    float* __restrict__ loss, int32_t* __restrict__ sel) {
  __shared__ double dsm[3][kT / 64];
  if (st->branch == 1) {
    double a = 0, b = 0, c = 0;
    for (int i = threadIdx.x; i < grid; i += kT) { a += csum[i]; b += cwsum[i]; c += ccnt[i]; }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { dsm[0][threadIdx.x >> 6] = a; dsm[1][threadIdx.x >> 6] = b; dsm[2][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
      a = b = c = 0;
      for (int i = 0; i < kT / 64; ++i) { a += dsm[0][i]; b += dsm[1][i]; c += dsm[2][i]; }
      st->sum = a;
      st->n_kept = (int64_t)c;
      st->wsum = weighted ? b : c;
    }
  }
  if (threadIdx.x == 0) {
    loss[0] = (float)(st->sum / st->wsum);  // 0/0 -> NaN like CrossEntropyLoss over no pixels
    sel[0] = (int32_t)st->thr_bits;
    sel[1] = (int32_t)(st->n_kept > 0x7fffffffLL ? 0x7fffffff : st->n_kept);
    sel[2] = (int32_t)(st->num_valid > 0x7fffffffLL ? 0x7fffffff : st->num_valid);
    sel[3] = st->branch;
    sel[5] = st->n_bad;
    // denominator as float for the backward pass
    reinterpret_cast<float*>(sel)[4] = (float)st->wsum;
  }
}

// =============================================================================
// backward
// =============================================================================
template <typename T, int V, int LT>
__global__ __launch_bounds__(kT) void ohem_bwd_k(
    const T* __restrict__ logits, const void* __restrict__ labels, int64_t P, int C, int64_t HW,
    int64_t ignore_label, const float* __restrict__ weight, const float* __restrict__ nll,
    const float* __restrict__ lse, const int32_t* __restrict__ sel,
    const float* __restrict__ gscale, T* __restrict__ dlogits) {
  const float thr = __uint_as_float((uint32_t)sel[0]);
  const int branch = sel[3];
  const float denom = reinterpret_cast<const float*>(sel)[4];
  const float g = gscale[0] / denom;
  const int64_t nvec = P / V;
  for (int64_t v = (int64_t)blockIdx.x * kT + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * kT) {
    const int64_t p0 = v * V;
    const int64_t b = p0 / HW;
    const int64_t q = p0 - b * HW;
    const int64_t boff = (b * C) * HW + q;
    float coef[V], ls[V];
    int t[V];
    bool any = false;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const int64_t lab = Lab<LT>::get(labels, p0 + j);
      const bool valid = label_is_class(lab, ignore_label, C);
      t[j] = valid ? (int)lab : -1;
      bool kept = valid;
      if (valid && branch != 2) kept = prob_of_nll(nll[p0 + j]) <= thr;
      const float w = (weight && valid) ? weight[lab] : 1.f;
      coef[j] = kept ? g * w : 0.f;
      ls[j] = lse[p0 + j];
      any |= kept;
    }
    if (any) {
#pragma unroll 4
      for (int c = 0; c < C; ++c) {
        PixVec<T, V> px;
        px.load(logits + boff + (int64_t)c * HW);
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const float sm = __expf(px.v[j] - ls[j]);
          px.v[j] = coef[j] * (sm - (t[j] == c ? 1.f : 0.f));
        }
        px.store(dlogits + boff + (int64_t)c * HW);
      }
    } else {
      PixVec<T, V> z;
#pragma unroll
      for (int j = 0; j < V; ++j) z.v[j] = 0.f;
      for (int c = 0; c < C; ++c) z.store(dlogits + boff + (int64_t)c * HW);
    }
  }
}

// =============================================================================
// Fused head: bilinear (align_corners=True) upsample of the low-resolution logits
// evaluated INSIDE the OHEM kernels, so the full-resolution logits (59.8 M
// elements per image for BiSeNet's three heads) are never written or re-read
// (SURVEY.md §8f-1).  z is [B, C, IH, IW]; the virtual logits are [B, C, OH, OW].
//
// Both directions are "column walkers": a thread owns ONE output column ox and
// walks down a band of output rows.  For its column it keeps, per class, the two
// horizontally interpolated source rows H0[c] / H1[c] (rows y0 / y1) in registers;
// they change only when y0 advances (every ~scale output rows), so per pixel and
// class the logit costs 2 FMAs:  v_c = (1-ly)*H0[c] + ly*H1[c]   — the same
// operations, in the same order, as the 4-tap formula of aten::upsample_bilinear2d.
// z itself is tiny (L2-resident) and read with wave-broadcast loads.
//   forward : per pixel max / sum-exp over the C register values (1 exp per class),
//             then the shared pass-A accounting.  HBM: label read + nll/lse write.
//   backward: g_c = coef*(softmax_c - [c==t]) is folded straight into the vertical
//             transposed taps (accA/accB per class, flushed when a source row is
//             complete) -> V[B,C,IH,OW] fp32; a small second kernel applies the
//             horizontal transposed taps.  No atomics, fixed order => deterministic.
// =============================================================================
constexpr int kFwdBand = 32;                 // output rows per forward band
constexpr int kBwdCH = 8;                    // source rows owned by a backward block

template <typename T, int CMAX>
__device__ __forceinline__ void load_hrow(const T* __restrict__ zb, int C, int64_t plane, int IW, int y,
                                          int x0, int x1, float lx, float (&H)[CMAX]) {
  const float hx = 1.f - lx;
  const T* r = zb + (int64_t)y * IW;
#pragma unroll
  for (int c = 0; c < CMAX; ++c)
    if (c < C) H[c] = hx * ld1<T>(r + c * plane + x0) + lx * ld1<T>(r + c * plane + x1);
}

template <typename T, int LT, int CMAX>
__global__ __launch_bounds__(kT) void ohem_up_pass_a(
    const T* __restrict__ z, const void* __restrict__ labels, int64_t B, int C, int IH, int IW,
    int OH, int OW, float sy, float sx, int64_t ignore_label, float thresh, int64_t tb, int shift0,
    int bins0, const float* __restrict__ weight, float* __restrict__ nll_out,
    float* __restrict__ lse_out, uint32_t* __restrict__ hist0, BlkPart* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lh[];
  __shared__ uint8_t lab_s[kFwdBand * kT];          // labels of the current band (255 = ignored), C <= 32
  const int tid = threadIdx.x;
  for (int i = tid; i < bins0; i += kT) lh[i] = 0;
  __syncthreads();
  PassAcc acc;
  const int xblocks = (OW + kT - 1) / kT, bands = (OH + kFwdBand - 1) / kFwdBand;
  const int64_t total = B * bands * (int64_t)xblocks;
  const int64_t plane = (int64_t)IH * IW;
  for (int64_t tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int xb = (int)(tile % xblocks);
    const int band = (int)((tile / xblocks) % bands);
    const int64_t b = tile / ((int64_t)xblocks * bands);
    const int ox = xb * kT + tid;
    if (ox >= OW) continue;
    int x0, x1; float lx;
    src_index(sx, ox, IW, x0, x1, lx);
    const T* zb = z + b * C * plane;
    float H0[CMAX], H1[CMAX];
    int cy0 = -1, cy1 = -1;
    const int oy_end = (band + 1) * kFwdBand < OH ? (band + 1) * kFwdBand : OH;
    // side data of the whole band first (kFwdBand independent loads in flight per thread);
    // each thread only ever reads back its own column, so no barrier is needed.
#pragma unroll
    for (int u = 0; u < kFwdBand; ++u) {
      const int oy = band * kFwdBand + u;
      int64_t lab = ignore_label;
      if (oy < oy_end) lab = Lab<LT>::get(labels, (b * OH + oy) * (int64_t)OW + ox);
      const bool is_cls = label_is_class(lab, ignore_label, C);
      if (!is_cls && lab != ignore_label) acc.cnt_bad++;
      lab_s[u * kT + tid] = is_cls ? (uint8_t)lab : (uint8_t)255;
    }
    for (int oy = band * kFwdBand; oy < oy_end; ++oy) {
      int y0, y1; float ly;
      src_index(sy, oy, IH, y0, y1, ly);
      if (y0 != cy0) {
        if (y0 == cy1) {
#pragma unroll
          for (int c = 0; c < CMAX; ++c) H0[c] = H1[c];
        } else {
          load_hrow<T, CMAX>(zb, C, plane, IW, y0, x0, x1, lx, H0);
        }
        cy0 = y0; cy1 = -1;
      }
      if (y1 != cy1) {
        if (y1 == cy0) {
#pragma unroll
          for (int c = 0; c < CMAX; ++c) H1[c] = H0[c];
        } else {
          load_hrow<T, CMAX>(zb, C, plane, IW, y1, x0, x1, lx, H1);
        }
        cy1 = y1;
      }
      const int64_t gp = (b * OH + oy) * (int64_t)OW + ox;
      const int lab = lab_s[(oy - band * kFwdBand) * kT + tid];
      const bool valid = lab != 255;
      const int t = valid ? lab : 0;
      const float hy = 1.f - ly;
      float v[CMAX];
      float m = -INFINITY, xt = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c) {
        if (c < C) {
          v[c] = hy * H0[c] + ly * H1[c];
          m = fmaxf(m, v[c]);
          if (c == t) xt = v[c];
        }
      }
      float ssum = 0.f;
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) ssum += __expf(v[c] - m);
      const float ls = m + logf(ssum);
      float nl = ls - xt;
      nl = nl < 0.f ? 0.f : nl;
      nl = valid ? nl : 0.f;
      nll_out[gp] = nl;
      lse_out[gp] = ls;
      const float w = (weight && valid) ? weight[t] : 1.f;
      account(acc, valid, nl, w, thresh, tb, shift0, bins0, lh);
    }
  }
  fold_block(acc, lh, bins0, hist0, part);
}

// backward, vertical part.  grid = (x-blocks, source-row bands, B); V[b][c][iy][ox] fp32.
template <typename T, int LT, int CMAX>
__global__ __launch_bounds__(kT) void ohem_up_bwd_v(
    const T* __restrict__ z, const void* __restrict__ labels, int C, int IH, int IW, int OH, int OW,
    float sy, float sx, int64_t ignore_label, const float* __restrict__ weight,
    const float* __restrict__ nll, const float* __restrict__ lse, const int32_t* __restrict__ sel,
    const float* __restrict__ gscale, float* __restrict__ V) {
  int ox = blockIdx.x * kT + threadIdx.x;
  const bool live = ox < OW;
  if (!live) return;
  const int r0 = blockIdx.y * kBwdCH;
  const int r1 = (r0 + kBwdCH < IH) ? r0 + kBwdCH : IH;
  const int64_t b = blockIdx.z;
  const float thr = __uint_as_float((uint32_t)sel[0]);
  const int branch = sel[3];
  const float g = gscale[0] / reinterpret_cast<const float*>(sel)[4];
  const int64_t plane = (int64_t)IH * IW;
  const T* zb = z + b * C * plane;
  float* Vb = V + b * C * (int64_t)IH * OW;
  int x0, x1; float lx;
  src_index(sx, ox, IW, x0, x1, lx);
  // output rows whose y0 lies in [r0-1, r1-1]
  int oy_lo = 0, oy_hi = OH - 1;
  if (sy > 0.f) {
    const float inv = 1.f / sy;
    int l = (int)ceilf((float)(r0 - 1) * inv) - 1;
    int h = (int)floorf((float)r1 * inv) + 1;
    oy_lo = l < 0 ? 0 : l;
    oy_hi = h > OH - 1 ? OH - 1 : h;
  }
  float H0[CMAX], H1[CMAX], accA[CMAX], accB[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; ++c) { accA[c] = 0.f; accB[c] = 0.f; H0[c] = 0.f; H1[c] = 0.f; }
  int cy0 = -1, cy1 = -1;
  int cur = -2;                                      // source row accA belongs to
  auto flush = [&](int row, const float (&a)[CMAX]) {
    if (row >= r0 && row < r1) {
#pragma unroll
      for (int c = 0; c < CMAX; ++c)
        if (c < C) Vb[((int64_t)c * IH + row) * OW + ox] = a[c];
    }
  };
  constexpr int G = 16;                                // rows whose side data are fetched together
  __shared__ float coef_s[G * kT];                     // g * w_t for kept pixels, 0 otherwise
  __shared__ float lse_s[G * kT];
  __shared__ uint8_t lab_s[G * kT];
  const int tid = threadIdx.x;
  for (int oyg = oy_lo; oyg <= oy_hi; oyg += G) {
    // independent loads first; a thread only reads back its own column => no barrier
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const bool in = oyg + u <= oy_hi;
      const int64_t gp = (b * OH + (in ? oyg + u : oy_hi)) * (int64_t)OW + ox;
      const int64_t lab = in ? Lab<LT>::get(labels, gp) : ignore_label;
      const float nl = nll[gp];
      const float ls = lse[gp];
      const bool valid = label_is_class(lab, ignore_label, C);
      bool kept = valid;
      if (valid && branch != 2) kept = prob_of_nll(nl) <= thr;
      coef_s[u * kT + tid] = kept ? g * (weight ? weight[lab] : 1.f) : 0.f;
      lse_s[u * kT + tid] = ls;
      lab_s[u * kT + tid] = valid ? (uint8_t)lab : (uint8_t)255;
    }
    const int oyg_end = (oyg + G - 1 < oy_hi) ? oyg + G - 1 : oy_hi;
    for (int oy = oyg; oy <= oyg_end; ++oy) {
      int y0, y1; float ly;
      src_index(sy, oy, IH, y0, y1, ly);
      if (y0 < r0 - 1 || y0 > r1 - 1) continue;
      if (y0 != cur) {
        if (cur >= 0) {
          flush(cur, accA);
          if (y0 == cur + 1) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c) { accA[c] = accB[c]; accB[c] = 0.f; }
          } else {
            flush(cur + 1, accB);
#pragma unroll
            for (int c = 0; c < CMAX; ++c) { accA[c] = 0.f; accB[c] = 0.f; }
          }
        }
        cur = y0;
      }
      if (y0 != cy0) {
        if (y0 == cy1) {
#pragma unroll
          for (int c = 0; c < CMAX; ++c) H0[c] = H1[c];
        } else {
          load_hrow<T, CMAX>(zb, C, plane, IW, y0, x0, x1, lx, H0);
        }
        cy0 = y0; cy1 = -1;
      }
      if (y1 != cy1) {
        if (y1 == cy0) {
#pragma unroll
          for (int c = 0; c < CMAX; ++c) H1[c] = H0[c];
        } else {
          load_hrow<T, CMAX>(zb, C, plane, IW, y1, x0, x1, lx, H1);
        }
        cy1 = y1;
      }
      const float coef = coef_s[(oy - oyg) * kT + tid];
      if (coef != 0.f) {
        const float ls = lse_s[(oy - oyg) * kT + tid];
        const int t = lab_s[(oy - oyg) * kT + tid];
        const float hy = 1.f - ly;
        const bool same = (y1 == y0);
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
          if (c < C) {
            const float v = hy * H0[c] + ly * H1[c];
            const float gc = coef * (__expf(v - ls) - (c == t ? 1.f : 0.f));
            if (same) accA[c] += hy * gc + ly * gc;
            else { accA[c] += hy * gc; accB[c] += ly * gc; }
          }
        }
      }
    }
  }
  if (cur >= 0) { flush(cur, accA); flush(cur + 1, accB); }
}

// backward, horizontal part: dz[b,c,iy,ix] = sum_ox wx(ox, ix) * V[b,c,iy,ox]
template <typename T, int MAXF>
__global__ __launch_bounds__(kT) void ohem_up_bwd_h(const float* __restrict__ V, T* __restrict__ dz,
                                                    int64_t rows, int IW, int OW, float sx) {
  const int64_t total = rows * IW;
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < total; i += (int64_t)gridDim.x * kT) {
    const int ix = (int)(i % IW);
    const int64_t row = i / IW;
    int xlo, xhi;
    footprint(sx, ix, OW, xlo, xhi);
    const float* v = V + row * OW + xlo;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < MAXF; ++j)
      if (xlo + j <= xhi) acc += tap_weight(sx, xlo + j, IW, ix) * v[j];
    st1<T>(dz + i, acc);
  }
}

static bool up_fused_ok(int C, int IH, int IW, int OH, int OW) {
  if (C > 32 || C < 1) return false;
  if (OH < 2 * IH || OW < 2 * IW) return false;             // only genuine up-sampling is fused
  const float sx = ac_scale(IW, OW), sy = ac_scale(IH, OH);
  if (sx <= 0.f || sy <= 0.f) return false;
  return (int)floorf(2.f / sx) + 4 <= 37;                    // horizontal footprint the gather unrolls
}

static int pixel_grid(int64_t nvec) {
  int64_t g = (nvec + kT - 1) / kT;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

static int64_t thresh_tb(float thresh) {
  if (thresh < 0.f) return -1;
  union { float f; uint32_t u; } cv;
  cv.f = thresh;
  return (int64_t)cv.u;
}

}  // namespace tsg

using namespace tsg;

static int ohem_select_tail(const tsg_ohem_plan& pl, OhemWs& w, const void* labels, int ltype,
                            int64_t ignore_label, float thresh, int64_t min_kept, const float* weight,
                            float* nll, float* loss, int32_t* sel, int64_t tb, hipStream_t st) {
  const int bins0 = pl.levels > 0 ? pl.bins[0] : 0;
  hipLaunchKernelGGL(ohem_decide0, dim3(1), dim3(kT), 0, st, w.part, pl.grid, w.hist[0], bins0,
                     pl.shift[0], pl.levels, pl.P, min_kept, thresh, tb, weight ? 1 : 0, w.st);
  TSG_CHECK_LAUNCH();
  for (int l = 1; l < pl.levels; ++l) {
    hipLaunchKernelGGL((sel_refine<1>), dim3(pl.grid), dim3(kT), (size_t)pl.bins[l] * 4, st, nll, pl.P,
                       thresh, tb, pl.shift[l - 1], pl.shift[l], pl.bins[l], w.hist[l], w.st);
    TSG_CHECK_LAUNCH();
    hipLaunchKernelGGL(sel_decide, dim3(1), dim3(64), 0, st, w.hist[l], pl.bins[l], pl.shift[l],
                       l == pl.levels - 1 ? 1 : 0, tb, w.st);
    TSG_CHECK_LAUNCH();
  }
  if (ltype == TSG_I64)
    hipLaunchKernelGGL((ohem_pass_c<TSG_I64>), dim3(pl.grid), dim3(kT), 0, st, nll, labels, pl.P, pl.C,
                       ignore_label, weight, w.st, w.csum, w.cwsum, w.ccnt);
  else
    hipLaunchKernelGGL((ohem_pass_c<TSG_U8>), dim3(pl.grid), dim3(kT), 0, st, nll, labels, pl.P, pl.C,
                       ignore_label, weight, w.st, w.csum, w.cwsum, w.ccnt);
  TSG_CHECK_LAUNCH();
  hipLaunchKernelGGL(ohem_finish, dim3(1), dim3(kT), 0, st, w.csum, w.cwsum, w.ccnt, pl.grid,
                     weight ? 1 : 0, w.st, loss, sel);
  TSG_CHECK_LAUNCH();
  return 0;
}

// mask_prob of loss_opr.py:81-83 as the selection kernels see it: the target-class probability recomputed from
// nll with the SAME device expression (prob_of_nll), 1 for pixels that take no part.
template <int LT>
__global__ __launch_bounds__(kT) void target_prob_k(const float* __restrict__ nll, const void* __restrict__ labels,
                                                    int64_t P, int C, int64_t ignore_label, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kT + threadIdx.x; i < P; i += (int64_t)gridDim.x * kT)
    out[i] = label_is_class(Lab<LT>::get(labels, i), ignore_label, C) ? prob_of_nll(nll[i]) : 1.f;
}

extern "C" {

int tsg_ohem_make_plan(int64_t B, int C, int64_t HW, float thresh, tsg_ohem_plan* plan) {
  if (!plan) return TSG_E_NULL;
  if (B <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  if (!(thresh == thresh)) return TSG_E_SHAPE;
  plan->P = B * HW;
  plan->C = C;
  plan->grid = pixel_grid(plan->P / 4);
  const int64_t tb = thresh_tb(thresh);
  const int64_t range = (int64_t)0x3f800000 - tb;  // number of distinct p in (thresh, 1]
  level_plan(range, &plan->levels, plan->shift, plan->bins);
  plan->thresh_bits = thresh < 0.f ? 0u : (uint32_t)tb;
  OhemWs w = carve(nullptr, *plan);
  plan->ws_bytes = w.total;
  return 0;
}

int tsg_ohem_fwd(const void* logits, int dtype, const void* labels, int ltype, int64_t B, int C,
                 int64_t HW, int64_t ignore_label, float thresh, int64_t min_kept,
                 const float* weight, float* nll, float* lse, float* loss, int32_t* sel,
                 void* ws, size_t ws_bytes, void* stream) {
  if (!logits || !labels || !nll || !lse || !loss || !sel || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (ltype != TSG_I64 && ltype != TSG_U8) return TSG_E_DTYPE;
  tsg_ohem_plan pl;
  int e = tsg_ohem_make_plan(B, C, HW, thresh, &pl);
  if (e) return e;
  if (ws_bytes < pl.ws_bytes) return TSG_E_WS;
  if (!aligned16(ws)) return TSG_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  OhemWs w = carve(ws, pl);
  TSG_HIP(hipMemsetAsync(ws, 0, w.zero_bytes, st));
  const int64_t tb = thresh_tb(thresh);
  const int native = dtype == TSG_BF16 ? 8 : 4;
  const bool vec = (HW % native == 0) && aligned16(logits) && aligned16(nll) && aligned16(lse);
  const int bins0 = pl.levels > 0 ? pl.bins[0] : 0;
  static const size_t pa_throttle = [] { const char* e = getenv("TSG_OHEM_FWD_LDS"); return e ? (size_t)atol(e) : (size_t)0; }();
  size_t sh = (size_t)(bins0 > 0 ? bins0 : 1) * sizeof(uint32_t);
  if (sh < pa_throttle) sh = pa_throttle;
#define PA(T, VV, LTT)                                                                         \
  hipLaunchKernelGGL((ohem_pass_a<T, VV, LTT>), dim3(pl.grid), dim3(kT), sh, st, (const T*)logits, \
                     labels, pl.P, C, HW, ignore_label, thresh, tb, pl.shift[0], bins0, weight, \
                     nll, lse, w.hist[0], w.part)
  if (dtype == TSG_F32) {
    if (vec) { if (ltype == TSG_I64) PA(float, 4, TSG_I64); else PA(float, 4, TSG_U8); }
    else     { if (ltype == TSG_I64) PA(float, 1, TSG_I64); else PA(float, 1, TSG_U8); }
  } else {
    if (vec) { if (ltype == TSG_I64) PA(bf16_t, 8, TSG_I64); else PA(bf16_t, 8, TSG_U8); }
    else     { if (ltype == TSG_I64) PA(bf16_t, 1, TSG_I64); else PA(bf16_t, 1, TSG_U8); }
  }
#undef PA
  TSG_CHECK_LAUNCH();
  return ohem_select_tail(pl, w, labels, ltype, ignore_label, thresh, min_kept, weight, nll, loss, sel, tb, st);
}

int tsg_ohem_bwd(const void* logits, int dtype, const void* labels, int ltype, int64_t B, int C,
                 int64_t HW, int64_t ignore_label, const float* weight, const float* nll,
                 const float* lse, const int32_t* sel, const float* gscale, void* dlogits,
                 void* ws, void* stream) {
  (void)ws;
  if (!logits || !labels || !nll || !lse || !sel || !gscale || !dlogits) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (ltype != TSG_I64 && ltype != TSG_U8) return TSG_E_DTYPE;
  if (B <= 0 || C <= 0 || HW <= 0) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t P = B * HW;
  const int native = dtype == TSG_BF16 ? 8 : 4;
  const bool vec = (HW % native == 0) && aligned16(logits) && aligned16(dlogits);
  const int V = vec ? native : 1;
  const int grid = pixel_grid(P / V);
  // occupancy throttle (tuning knob): unused dynamic LDS caps the resident waves per CU
  // (measured on MI355X, 16x19x1024^2 bf16: 0 B -> 407 us, 40 KB (4 blocks/CU) -> 354 us, 64 KB -> 476 us: with 38
  //  concurrent 2-MB-strided class planes per block, fewer resident blocks thrash DRAM pages / L2 less)
  static const size_t throttle = [] { const char* e = getenv("TSG_OHEM_BWD_LDS"); return e ? (size_t)atol(e) : (size_t)40000; }();
#define PB(T, VV, LTT)                                                                          \
  hipLaunchKernelGGL((ohem_bwd_k<T, VV, LTT>), dim3(grid), dim3(kT), throttle, st, (const T*)logits, labels, \
                     P, C, HW, ignore_label, weight, nll, lse, sel, gscale, (T*)dlogits)
  if (dtype == TSG_F32) {
    if (vec) { if (ltype == TSG_I64) PB(float, 4, TSG_I64); else PB(float, 4, TSG_U8); }
    else     { if (ltype == TSG_I64) PB(float, 1, TSG_I64); else PB(float, 1, TSG_U8); }
  } else {
    if (vec) { if (ltype == TSG_I64) PB(bf16_t, 8, TSG_I64); else PB(bf16_t, 8, TSG_U8); }
    else     { if (ltype == TSG_I64) PB(bf16_t, 1, TSG_I64); else PB(bf16_t, 1, TSG_U8); }
  }
#undef PB
  TSG_CHECK_LAUNCH();
  return 0;
}

// ---- fused upsample + OHEM ------------------------------------------------------
int tsg_ohem_up_supported(int C, int IH, int IW, int OH, int OW, float thresh) {
  (void)thresh;
  return up_fused_ok(C, IH, IW, OH, OW) ? 1 : 0;
}

size_t tsg_ohem_up_bwd_ws_bytes(int64_t B, int C, int IH, int OW) {
  if (B <= 0 || C <= 0 || IH <= 0 || OW <= 0) return 0;
  return (size_t)B * C * IH * OW * sizeof(float);
}

int tsg_ohem_up_fwd(const void* z, int dtype, const void* labels, int ltype, int64_t B, int C, int IH,
                    int IW, int OH, int OW, int64_t ignore_label, float thresh, int64_t min_kept,
                    const float* weight, float* nll, float* lse, float* loss, int32_t* sel, void* ws,
                    size_t ws_bytes, void* stream) {
  if (!z || !labels || !nll || !lse || !loss || !sel || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (ltype != TSG_I64 && ltype != TSG_U8) return TSG_E_DTYPE;
  if (IH <= 0 || IW <= 0 || !up_fused_ok(C, IH, IW, OH, OW)) return TSG_E_SHAPE;
  tsg_ohem_plan pl;
  int e = tsg_ohem_make_plan(B, C, (int64_t)OH * OW, thresh, &pl);
  if (e) return e;
  if (ws_bytes < pl.ws_bytes) return TSG_E_WS;
  if (!aligned16(ws)) return TSG_E_ALIGN;
  const int bins0 = pl.levels > 0 ? pl.bins[0] : 0;
  hipStream_t st = (hipStream_t)stream;
  OhemWs w = carve(ws, pl);
  TSG_HIP(hipMemsetAsync(ws, 0, w.zero_bytes, st));
  const int64_t tb = thresh_tb(thresh);
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  const size_t sh = (size_t)(bins0 > 0 ? bins0 : 1) * sizeof(uint32_t);
#define PA(T, LTT, CM)                                                                                \
  hipLaunchKernelGGL((ohem_up_pass_a<T, LTT, CM>), dim3(pl.grid), dim3(kT), sh, st, (const T*)z, labels, B, C, \
                     IH, IW, OH, OW, sy, sx, ignore_label, thresh, tb, pl.shift[0], bins0, weight, nll, \
                     lse, w.hist[0], w.part)
#define PC(T, LTT) do { if (C <= 20) PA(T, LTT, 20); else PA(T, LTT, 32); } while (0)
  if (dtype == TSG_F32) { if (ltype == TSG_I64) PC(float, TSG_I64); else PC(float, TSG_U8); }
  else { if (ltype == TSG_I64) PC(bf16_t, TSG_I64); else PC(bf16_t, TSG_U8); }
#undef PC
#undef PA
  TSG_CHECK_LAUNCH();
  return ohem_select_tail(pl, w, labels, ltype, ignore_label, thresh, min_kept, weight, nll, loss, sel, tb, st);
}

int tsg_ohem_up_bwd(const void* z, int dtype, const void* labels, int ltype, int64_t B, int C, int IH,
                    int IW, int OH, int OW, int64_t ignore_label, const float* weight, const float* nll,
                    const float* lse, const int32_t* sel, const float* gscale, void* dz, void* ws,
                    size_t ws_bytes, void* stream) {
  if (!z || !labels || !nll || !lse || !sel || !gscale || !dz || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (ltype != TSG_I64 && ltype != TSG_U8) return TSG_E_DTYPE;
  if (B <= 0 || !up_fused_ok(C, IH, IW, OH, OW)) return TSG_E_SHAPE;
  if (ws_bytes < tsg_ohem_up_bwd_ws_bytes(B, C, IH, OW)) return TSG_E_WS;
  hipStream_t st = (hipStream_t)stream;
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  float* V = (float*)ws;
  dim3 grid((unsigned)((OW + kT - 1) / kT), (unsigned)((IH + kBwdCH - 1) / kBwdCH), (unsigned)B);
#define PB(T, LTT, CM)                                                                                \
  hipLaunchKernelGGL((ohem_up_bwd_v<T, LTT, CM>), grid, dim3(kT), 0, st, (const T*)z, labels, C, IH, IW, OH, OW, \
                     sy, sx, ignore_label, weight, nll, lse, sel, gscale, V)
#define PC(T, LTT) do { if (C <= 20) PB(T, LTT, 20); else PB(T, LTT, 32); } while (0)
  if (dtype == TSG_F32) { if (ltype == TSG_I64) PC(float, TSG_I64); else PC(float, TSG_U8); }
  else { if (ltype == TSG_I64) PC(bf16_t, TSG_I64); else PC(bf16_t, TSG_U8); }
#undef PC
#undef PB
  TSG_CHECK_LAUNCH();
  const int64_t rows = B * C * IH;
  const int need = (int)floorf(2.f / sx) + 4;
  int64_t g2 = (rows * IW + kT - 1) / kT;
  if (g2 > 8192) g2 = 8192;
#define PH(T, F) hipLaunchKernelGGL((ohem_up_bwd_h<T, F>), dim3((unsigned)g2), dim3(kT), 0, st, V, (T*)dz, rows, IW, OW, sx)
  if (dtype == TSG_F32) { if (need <= 9) PH(float, 9); else if (need <= 21) PH(float, 21); else PH(float, 37); }
  else { if (need <= 9) PH(bf16_t, 9); else if (need <= 21) PH(bf16_t, 21); else PH(bf16_t, 37); }
#undef PH
  TSG_CHECK_LAUNCH();
  return 0;
}

// ---- standalone exact k-th order statistic -----------------------------------
static void kth_plan(tsg_ohem_plan* pl, int64_t n) {
  pl->P = n;
  pl->C = 1;
  pl->grid = pixel_grid(n);
  level_plan((int64_t)1 << 31, &pl->levels, pl->shift, pl->bins);  // all non-negative floats
  OhemWs w = carve(nullptr, *pl);
  pl->ws_bytes = w.total;
}

size_t tsg_kth_ws_bytes(int64_t n) {
  tsg_ohem_plan pl;
  kth_plan(&pl, n);
  return pl.ws_bytes;
}

__global__ void kth_init(SelState* st, int64_t k) {
  st->branch = 1; st->krem = k; st->lo_d = 0;
}
__global__ void kth_out(const SelState* st, float* out) { out[0] = __uint_as_float(st->thr_bits); }

int tsg_ohem_target_prob(const float* nll, const void* labels, int ltype, int64_t P, int C, int64_t ignore_label,
                         float* prob, void* stream) {
  if (!nll || !labels || !prob) return TSG_E_NULL;
  if (P <= 0 || C <= 0) return TSG_E_SHAPE;
  if (ltype != TSG_I64 && ltype != TSG_U8) return TSG_E_DTYPE;
  const int grid = (int)((P + kT - 1) / kT < 4096 ? (P + kT - 1) / kT : 4096);
  if (ltype == TSG_I64)
    hipLaunchKernelGGL((target_prob_k<TSG_I64>), dim3(grid), dim3(kT), 0, (hipStream_t)stream, nll, labels, P, C, ignore_label, prob);
  else
    hipLaunchKernelGGL((target_prob_k<TSG_U8>), dim3(grid), dim3(kT), 0, (hipStream_t)stream, nll, labels, P, C, ignore_label, prob);
  TSG_CHECK_LAUNCH();
  return 0;
}

// level 0 of the standalone select is a refine over the whole range
int tsg_kth_value(const float* v, int64_t n, int64_t k, float* out, void* ws, size_t ws_bytes,
                  void* stream) {
  if (!v || !out || !ws) return TSG_E_NULL;
  if (n <= 0 || k < 1 || k > n) return TSG_E_SHAPE;
  tsg_ohem_plan pl;
  kth_plan(&pl, n);
  if (ws_bytes < pl.ws_bytes) return TSG_E_WS;
  hipStream_t st = (hipStream_t)stream;
  OhemWs w = carve(ws, pl);
  TSG_HIP(hipMemsetAsync(ws, 0, w.zero_bytes, st));
  hipLaunchKernelGGL(kth_init, dim3(1), dim3(1), 0, st, w.st, k);
  const int64_t tb = -1;
  int prev_shift = 31;
  for (int l = 0; l < pl.levels; ++l) {
    hipLaunchKernelGGL((sel_refine<0>), dim3(pl.grid), dim3(kT), (size_t)pl.bins[l] * 4, st, v, n,
                       -1.0f, tb, prev_shift, pl.shift[l], pl.bins[l], w.hist[l], w.st);
    TSG_CHECK_LAUNCH();
    hipLaunchKernelGGL(sel_decide, dim3(1), dim3(64), 0, st, w.hist[l], pl.bins[l], pl.shift[l],
                       l == pl.levels - 1 ? 1 : 0, tb, w.st);
    TSG_CHECK_LAUNCH();
    prev_shift = pl.shift[l];
  }
  hipLaunchKernelGGL(kth_out, dim3(1), dim3(1), 0, st, w.st, out);
  TSG_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
