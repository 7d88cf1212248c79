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
  int32_t pad1;
  double sum;           // sum of w*nll over kept
  double wsum;          // sum of w over kept (denominator when weighted)
};

struct BlkPart {
  float sum_le, sum_valid, wsum_le, wsum_valid;
  int32_t cnt_le_all, cnt_le_valid, cnt_valid, pad;
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
  int cnt_le_all = 0, cnt_le_valid = 0, cnt_valid = 0;
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
  __shared__ int ism[3 * (kT / 64)];
  const int tid = threadIdx.x;
  __syncthreads();
  for (int i = tid; i < bins0; i += kT) {
    const uint32_t h = lh[i];
    if (h) atomicAdd(&hist0[i], h);
  }
  a.sum_le = wave_sum(a.sum_le); a.sum_valid = wave_sum(a.sum_valid);
  a.wsum_le = wave_sum(a.wsum_le); a.wsum_valid = wave_sum(a.wsum_valid);
  a.cnt_le_all = wave_sum(a.cnt_le_all); a.cnt_le_valid = wave_sum(a.cnt_le_valid); a.cnt_valid = wave_sum(a.cnt_valid);
  const int lane = tid & 63, wv = tid >> 6;
  if (lane == 0) {
    fsm[wv * 4 + 0] = a.sum_le; fsm[wv * 4 + 1] = a.sum_valid; fsm[wv * 4 + 2] = a.wsum_le; fsm[wv * 4 + 3] = a.wsum_valid;
    ism[wv * 3 + 0] = a.cnt_le_all; ism[wv * 3 + 1] = a.cnt_le_valid; ism[wv * 3 + 2] = a.cnt_valid;
  }
  __syncthreads();
  if (tid == 0) {
    BlkPart bp = {0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0};
    for (int i = 0; i < kT / 64; ++i) {
      bp.sum_le += fsm[i * 4 + 0]; bp.sum_valid += fsm[i * 4 + 1];
      bp.wsum_le += fsm[i * 4 + 2]; bp.wsum_valid += fsm[i * 4 + 3];
      bp.cnt_le_all += ism[i * 3 + 0]; bp.cnt_le_valid += ism[i * 3 + 1]; bp.cnt_valid += ism[i * 3 + 2];
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
      valid[j] = lab != ignore_label;
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
  // fixed assignment of partials to threads, fixed combine order => deterministic
  for (int i = threadIdx.x; i < grid; i += kT) {
    const BlkPart bp = part[i];
    a0 += bp.sum_le; a1 += bp.sum_valid; a2 += bp.wsum_le; a3 += bp.wsum_valid;
    c0 += bp.cnt_le_all; c1 += bp.cnt_le_valid; c2 += bp.cnt_valid;
  }
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
    const float* __restrict__ nll, const void* __restrict__ labels, int64_t P,
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
    if (lab != ignore_label) {
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
      const bool valid = lab != ignore_label;
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
// =============================================================================
constexpr int kFTX = 64, kFTY = 16;          // forward tile: 64 x 16 full-resolution pixels

// forward: per block-tile, stage the touched low-res window of every class in LDS
// ([C][RH][RW] floats), then every thread interpolates its 4 pixels class by class
// (4 LDS reads + 3 lerps) feeding the same online softmax / accounting as pass A.
template <typename T, int LT>
__global__ __launch_bounds__(kT) void ohem_up_pass_a(
    const T* __restrict__ z, const void* __restrict__ labels, int64_t B, int C, int IH, int IW,
    int OH, int OW, float sy, float sx, int RH, int RW, int64_t ignore_label, float thresh,
    int64_t tb, int shift0, int bins0, const float* __restrict__ weight,
    float* __restrict__ nll_out, float* __restrict__ lse_out, uint32_t* __restrict__ hist0,
    BlkPart* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lh[];   // [bins0] then zt
  float* zt = reinterpret_cast<float*>(lh + ((bins0 + 3) & ~3));
  const int tid = threadIdx.x;
  for (int i = tid; i < bins0; i += kT) lh[i] = 0;
  PassAcc acc;
  const int tiles_x = (OW + kFTX - 1) / kFTX, tiles_y = (OH + kFTY - 1) / kFTY;
  const int64_t total = B * tiles_y * (int64_t)tiles_x;
  const int prow = tid / (kFTX / 4), pxg = tid % (kFTX / 4);     // 16 rows x 16 groups of 4 pixels
  const int plane = RH * RW;
  for (int64_t tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int tx = (int)(tile % tiles_x);
    const int ty = (int)((tile / tiles_x) % tiles_y);
    const int64_t b = tile / ((int64_t)tiles_x * tiles_y);
    const int oy_first = ty * kFTY, ox_first = tx * kFTX;
    int Y0, X0, dum; float fd;
    src_index(sy, oy_first, IH, Y0, dum, fd);
    src_index(sx, ox_first, IW, X0, dum, fd);
    __syncthreads();                                             // previous tile done with zt / lh init visible
    for (int i = tid; i < C * plane; i += kT) {
      const int c = i / plane, r = i - c * plane;
      const int ry = r / RW, rx = r - ry * RW;
      int yy = Y0 + ry; yy = yy > IH - 1 ? IH - 1 : yy;
      int xx = X0 + rx; xx = xx > IW - 1 ? IW - 1 : xx;
      zt[i] = ld1<T>(z + ((b * C + c) * IH + yy) * (int64_t)IW + xx);
    }
    __syncthreads();
    const int oy = oy_first + prow;
    const int ox0 = ox_first + pxg * 4;
    if (oy < OH && ox0 < OW) {
      int y0, y1; float ly;
      src_index(sy, oy, IH, y0, y1, ly);
      const float hy = 1.f - ly;
      const int r0 = (y0 - Y0) * RW, r1 = (y1 - Y0) * RW;
      int o00[4], o01[4], t[4];
      float lx[4];
      bool inb[4], valid[4];
      const int64_t p0 = (b * OH + oy) * (int64_t)OW + ox0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        inb[j] = ox0 + j < OW;
        int x0, x1;
        src_index(sx, inb[j] ? ox0 + j : OW - 1, IW, x0, x1, lx[j]);
        o00[j] = x0 - X0; o01[j] = x1 - X0;
        const int64_t lab = inb[j] ? Lab<LT>::get(labels, p0 + j) : ignore_label;
        valid[j] = lab != ignore_label;
        t[j] = valid[j] ? (int)lab : 0;
      }
      float m[4], sacc[4], xt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { m[j] = -INFINITY; sacc[j] = 0.f; xt[j] = 0.f; }
      for (int c = 0; c < C; ++c) {
        const float* pz = zt + c * plane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float hx = 1.f - lx[j];
          const float top = hx * pz[r0 + o00[j]] + lx[j] * pz[r0 + o01[j]];
          const float bot = hx * pz[r1 + o00[j]] + lx[j] * pz[r1 + o01[j]];
          const float x = hy * top + ly * bot;
          if (t[j] == c) xt[j] = x;
          const float mn = fmaxf(m[j], x);
          sacc[j] = sacc[j] * __expf(m[j] - mn) + __expf(x - mn);
          m[j] = mn;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (inb[j]) {
          const float ls = m[j] + logf(sacc[j]);
          float nl = ls - xt[j];
          nl = nl < 0.f ? 0.f : nl;
          nl = valid[j] ? nl : 0.f;
          nll_out[p0 + j] = nl;
          lse_out[p0 + j] = ls;
          const float w = (weight && valid[j]) ? weight[t[j]] : 1.f;
          account(acc, valid[j], nl, w, thresh, tb, shift0, bins0, lh);
        }
      }
    }
  }
  fold_block(acc, lh, bins0, hist0, part);
}

// backward: a block owns CH x CW low-res cells of one image and produces dz for all
// C classes of them.  Classes are processed CHUNK at a time:
//   phase 1  every thread evaluates g_c = coef * (softmax_c - [c == t]) for its (<= KP)
//            full-res pixels of the block's footprint region (interpolating the logit
//            from the LDS copy of z) -> gbuf[CHUNK][region]
//   phase 2a horizontal taps:  hsum[c][row][ix] = sum_ox wx(ox, ix) * g[c][row][ox]
//   phase 2b vertical taps:    dz[c][iy][ix]    = sum_row wy(row, iy) * hsum[c][row][ix]
// Tap weights come from tap_weight() (bit-identical to the forward) via small LDS
// tables; no atomics, fixed summation order => deterministic.
constexpr int kKP = 12;        // region pixels per thread (region <= 3072)
constexpr int kChunk = 4;
constexpr int kMaxFoot = 40;   // taps per source index along one axis (scale factor <= 16)

struct UpBwdGeom { int CH, CW, RHf, RWf, FY, FX; };

template <typename T, int LT>
__global__ __launch_bounds__(kT) void ohem_up_bwd_k(
    const T* __restrict__ z, const void* __restrict__ labels, int C, int IH, int IW, int OH, int OW,
    float sy, float sx, UpBwdGeom gm, int64_t ignore_label, const float* __restrict__ weight,
    const float* __restrict__ nll, const float* __restrict__ lse, const int32_t* __restrict__ sel,
    const float* __restrict__ gscale, T* __restrict__ dz) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int CH = gm.CH, CW = gm.CW, RHf = gm.RHf, RWf = gm.RWf;
  const int ZH = CH + 2, ZW = CW + 2, zplane = ZH * ZW;
  float* zt = sm;                                   // [C][ZH][ZW]
  float* gbuf = zt + C * zplane;                    // [kChunk][RHf*RWf]
  float* hsum = gbuf + kChunk * RHf * RWf;          // [kChunk][RHf][CW]
  float* wyt = hsum + kChunk * RHf * CW;            // [CH][FY]
  float* wxt = wyt + CH * gm.FY;                    // [CW][FX]
  int* lo_y = reinterpret_cast<int*>(wxt + CW * gm.FX);   // [CH]
  int* lo_x = lo_y + CH;                            // [CW]
  const int tid = threadIdx.x;
  const int tiles_x = (IW + CW - 1) / CW, tiles_y = (IH + CH - 1) / CH;
  const int txi = blockIdx.x % tiles_x, tyi = (blockIdx.x / tiles_x) % tiles_y;
  const int64_t b = blockIdx.x / (tiles_x * tiles_y);
  const int r0 = tyi * CH, r1 = (r0 + CH < IH) ? r0 + CH : IH;
  const int c0 = txi * CW, c1 = (c0 + CW < IW) ? c0 + CW : IW;
  // footprint region of the owned cells in full-res coordinates
  int oy_lo, oy_hi, ox_lo, ox_hi, d0, d1;
  footprint(sy, r0, OH, oy_lo, d1); footprint(sy, r1 - 1, OH, d0, oy_hi);
  footprint(sx, c0, OW, ox_lo, d1); footprint(sx, c1 - 1, OW, d0, ox_hi);
  const int nrows = oy_hi - oy_lo + 1, ncols = ox_hi - ox_lo + 1, NP = nrows * ncols;
  // stage z window [r0-1, r1] x [c0-1, c1] (clamped) and the tap-weight tables
  for (int i = tid; i < C * zplane; i += kT) {
    const int c = i / zplane, r = i - c * zplane;
    int yy = r0 - 1 + r / ZW, xx = c0 - 1 + r % ZW;
    yy = yy < 0 ? 0 : (yy > IH - 1 ? IH - 1 : yy);
    xx = xx < 0 ? 0 : (xx > IW - 1 ? IW - 1 : xx);
    zt[i] = ld1<T>(z + ((b * C + c) * IH + yy) * (int64_t)IW + xx);
  }
  for (int i = tid; i < CH * gm.FY; i += kT) {
    const int iyl = i / gm.FY, j = i - iyl * gm.FY;
    float w = 0.f;
    if (r0 + iyl < r1) {
      int lo, hi;
      footprint(sy, r0 + iyl, OH, lo, hi);
      if (j == 0) lo_y[iyl] = lo;
      if (lo + j <= hi) w = tap_weight(sy, lo + j, IH, r0 + iyl);
    } else if (j == 0) lo_y[iyl] = 0;
    wyt[i] = w;
  }
  for (int i = tid; i < CW * gm.FX; i += kT) {
    const int ixl = i / gm.FX, j = i - ixl * gm.FX;
    float w = 0.f;
    if (c0 + ixl < c1) {
      int lo, hi;
      footprint(sx, c0 + ixl, OW, lo, hi);
      if (j == 0) lo_x[ixl] = lo;
      if (lo + j <= hi) w = tap_weight(sx, lo + j, IW, c0 + ixl);
    } else if (j == 0) lo_x[ixl] = 0;
    wxt[i] = w;
  }
  // per-thread pixel state, kept in registers across the class chunks
  const float thr = __uint_as_float((uint32_t)sel[0]);
  const int branch = sel[3];
  const float g = gscale[0] / reinterpret_cast<const float*>(sel)[4];
  float coef[kKP], lsv[kKP];
  int tt[kKP];
#pragma unroll
  for (int k = 0; k < kKP; ++k) {
    const int p = tid + k * kT;
    coef[k] = 0.f; lsv[k] = 0.f; tt[k] = -1;
    if (p < NP) {
      const int oy = oy_lo + p / ncols, ox = ox_lo + p % ncols;
      int y0, y1, x0, x1; float ly, lx;
      src_index(sy, oy, IH, y0, y1, ly);
      src_index(sx, ox, IW, x0, x1, lx);
      const bool touches = !(y1 < r0 || y0 > r1 - 1 || x1 < c0 || x0 > c1 - 1);
      if (touches) {
        const int64_t gp = (b * OH + oy) * (int64_t)OW + ox;
        const int64_t lab = Lab<LT>::get(labels, gp);
        const bool valid = lab != ignore_label;
        bool kept = valid;
        if (valid && branch != 2) kept = prob_of_nll(nll[gp]) <= thr;
        if (kept) {
          coef[k] = g * (weight ? weight[lab] : 1.f);
          lsv[k] = lse[gp];
          tt[k] = (int)lab;
        }
      }
    }
  }
  __syncthreads();
  for (int cb = 0; cb < C; cb += kChunk) {
    const int nc = (C - cb < kChunk) ? C - cb : kChunk;
    // phase 1
#pragma unroll
    for (int k = 0; k < kKP; ++k) {
      const int p = tid + k * kT;
      if (p < NP) {
        if (coef[k] != 0.f) {
          const int oy = oy_lo + p / ncols, ox = ox_lo + p % ncols;
          int y0, y1, x0, x1; float ly, lx;
          src_index(sy, oy, IH, y0, y1, ly);
          src_index(sx, ox, IW, x0, x1, lx);
          const float hy = 1.f - ly, hx = 1.f - lx;
          const int a0 = (y0 - (r0 - 1)) * ZW, a1 = (y1 - (r0 - 1)) * ZW;
          const int b0 = x0 - (c0 - 1), b1 = x1 - (c0 - 1);
          for (int cl = 0; cl < nc; ++cl) {
            const float* pz = zt + (cb + cl) * zplane;
            const float v = hy * (hx * pz[a0 + b0] + lx * pz[a0 + b1]) + ly * (hx * pz[a1 + b0] + lx * pz[a1 + b1]);
            const float smx = __expf(v - lsv[k]);
            gbuf[cl * NP + p] = coef[k] * (smx - (tt[k] == cb + cl ? 1.f : 0.f));
          }
        } else {
          for (int cl = 0; cl < nc; ++cl) gbuf[cl * NP + p] = 0.f;
        }
      }
    }
    __syncthreads();
    // phase 2a: horizontal
    const int ncw = c1 - c0;
    for (int i = tid; i < nc * nrows * ncw; i += kT) {
      const int ixl = i % ncw;
      const int row = (i / ncw) % nrows;
      const int cl = i / (ncw * nrows);
      const int xs = lo_x[ixl] - ox_lo;            // first tap relative to the region
      const float* gr = gbuf + cl * NP + row * ncols;
      const float* w = wxt + ixl * gm.FX;
      float acc = 0.f;
      for (int j = 0; j < gm.FX; ++j) {
        const int xo = xs + j;
        if (xo >= 0 && xo < ncols) acc += w[j] * gr[xo];
      }
      hsum[(cl * RHf + row) * CW + ixl] = acc;
    }
    __syncthreads();
    // phase 2b: vertical + store
    const int nch = r1 - r0;
    for (int i = tid; i < nc * nch * ncw; i += kT) {
      const int ixl = i % ncw;
      const int iyl = (i / ncw) % nch;
      const int cl = i / (ncw * nch);
      const int ys = lo_y[iyl] - oy_lo;
      const float* w = wyt + iyl * gm.FY;
      float acc = 0.f;
      for (int j = 0; j < gm.FY; ++j) {
        const int yo = ys + j;
        if (yo >= 0 && yo < nrows) acc += w[j] * hsum[(cl * RHf + yo) * CW + ixl];
      }
      st1<T>(dz + ((b * C + cb + cl) * IH + r0 + iyl) * (int64_t)IW + c0 + ixl, acc);
    }
    __syncthreads();
  }
}

// host-side geometry of the fused kernels -------------------------------------
static bool up_fwd_geom(int C, int IH, int IW, int OH, int OW, int bins0, int* RH, int* RW, size_t* sh) {
  if (OH < 2 * IH || OW < 2 * IW) return false;             // only genuine up-sampling is fused
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  *RH = (int)floorf((kFTY - 1) * sy) + 3;
  *RW = (int)floorf((kFTX - 1) * sx) + 3;
  *sh = ((size_t)((bins0 + 3) & ~3) + (size_t)C * *RH * *RW) * sizeof(float);
  return *sh <= 60 * 1024;
}

static size_t up_bwd_lds(int C, const UpBwdGeom& g) {
  return ((size_t)C * (g.CH + 2) * (g.CW + 2) + (size_t)kChunk * g.RHf * g.RWf + (size_t)kChunk * g.RHf * g.CW +
          (size_t)g.CH * g.FY + (size_t)g.CW * g.FX + g.CH + g.CW) * sizeof(float);
}

static bool up_bwd_geom(int C, int IH, int IW, int OH, int OW, UpBwdGeom* out) {
  if (OH < 2 * IH || OW < 2 * IW) return false;
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  if (sy <= 0.f || sx <= 0.f) return false;
  const int FY = (int)floorf(2.f / sy) + 4, FX = (int)floorf(2.f / sx) + 4;
  if (FY > kMaxFoot || FX > kMaxFoot) return false;
  const int cand[8][2] = {{4, 8}, {4, 6}, {2, 8}, {4, 4}, {2, 4}, {2, 2}, {1, 2}, {1, 1}};
  for (int i = 0; i < 8; ++i) {
    UpBwdGeom g;
    g.CH = cand[i][0] < IH ? cand[i][0] : IH;
    g.CW = cand[i][1] < IW ? cand[i][1] : IW;
    g.FY = FY; g.FX = FX;
    g.RHf = (int)ceilf((g.CH + 1) / sy) + 4;
    g.RWf = (int)ceilf((g.CW + 1) / sx) + 4;
    if (g.RHf > OH) g.RHf = OH;
    if (g.RWf > OW) g.RWf = OW;
    if ((int64_t)g.RHf * g.RWf <= (int64_t)kKP * kT && up_bwd_lds(C, g) <= 62 * 1024) { *out = g; return true; }
  }
  return false;
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
    hipLaunchKernelGGL((ohem_pass_c<TSG_I64>), dim3(pl.grid), dim3(kT), 0, st, nll, labels, pl.P,
                       ignore_label, weight, w.st, w.csum, w.cwsum, w.ccnt);
  else
    hipLaunchKernelGGL((ohem_pass_c<TSG_U8>), dim3(pl.grid), dim3(kT), 0, st, nll, labels, pl.P,
                       ignore_label, weight, w.st, w.csum, w.cwsum, w.ccnt);
  TSG_CHECK_LAUNCH();
  hipLaunchKernelGGL(ohem_finish, dim3(1), dim3(kT), 0, st, w.csum, w.cwsum, w.ccnt, pl.grid,
                     weight ? 1 : 0, w.st, loss, sel);
  TSG_CHECK_LAUNCH();
  return 0;
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
  const size_t sh = (size_t)(bins0 > 0 ? bins0 : 1) * sizeof(uint32_t);
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
#define PB(T, VV, LTT)                                                                          \
  hipLaunchKernelGGL((ohem_bwd_k<T, VV, LTT>), dim3(grid), dim3(kT), 0, st, (const T*)logits, labels, \
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
  tsg_ohem_plan pl;
  if (tsg_ohem_make_plan(1, C, (int64_t)OH * OW, thresh, &pl)) return 0;
  int RH, RW; size_t sh;
  UpBwdGeom g;
  return up_fwd_geom(C, IH, IW, OH, OW, pl.levels > 0 ? pl.bins[0] : 0, &RH, &RW, &sh) &&
         up_bwd_geom(C, IH, IW, OH, OW, &g);
}

int tsg_ohem_up_fwd(const void* z, int dtype, const void* labels, int ltype, int64_t B, int C, int IH,
                    int IW, int OH, int OW, int64_t ignore_label, float thresh, int64_t min_kept,
                    const float* weight, float* nll, float* lse, float* loss, int32_t* sel, void* ws,
                    size_t ws_bytes, void* stream) {
  if (!z || !labels || !nll || !lse || !loss || !sel || !ws) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (ltype != TSG_I64 && ltype != TSG_U8) return TSG_E_DTYPE;
  if (IH <= 0 || IW <= 0) return TSG_E_SHAPE;
  tsg_ohem_plan pl;
  int e = tsg_ohem_make_plan(B, C, (int64_t)OH * OW, thresh, &pl);
  if (e) return e;
  if (ws_bytes < pl.ws_bytes) return TSG_E_WS;
  if (!aligned16(ws)) return TSG_E_ALIGN;
  const int bins0 = pl.levels > 0 ? pl.bins[0] : 0;
  int RH, RW; size_t sh;
  if (!up_fwd_geom(C, IH, IW, OH, OW, bins0, &RH, &RW, &sh)) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  OhemWs w = carve(ws, pl);
  TSG_HIP(hipMemsetAsync(ws, 0, w.zero_bytes, st));
  const int64_t tb = thresh_tb(thresh);
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
#define PA(T, LTT)                                                                                    \
  do {                                                                                                \
    TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ohem_up_pass_a<T, LTT>),               \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));                \
    hipLaunchKernelGGL((ohem_up_pass_a<T, LTT>), dim3(pl.grid), dim3(kT), sh, st, (const T*)z, labels, B, C, \
                       IH, IW, OH, OW, sy, sx, RH, RW, ignore_label, thresh, tb, pl.shift[0], bins0,  \
                       weight, nll, lse, w.hist[0], w.part);                                          \
  } while (0)
  if (dtype == TSG_F32) { if (ltype == TSG_I64) PA(float, TSG_I64); else PA(float, TSG_U8); }
  else { if (ltype == TSG_I64) PA(bf16_t, TSG_I64); else PA(bf16_t, TSG_U8); }
#undef PA
  TSG_CHECK_LAUNCH();
  return ohem_select_tail(pl, w, labels, ltype, ignore_label, thresh, min_kept, weight, nll, loss, sel, tb, st);
}

int tsg_ohem_up_bwd(const void* z, int dtype, const void* labels, int ltype, int64_t B, int C, int IH,
                    int IW, int OH, int OW, int64_t ignore_label, const float* weight, const float* nll,
                    const float* lse, const int32_t* sel, const float* gscale, void* dz, void* stream) {
  if (!z || !labels || !nll || !lse || !sel || !gscale || !dz) return TSG_E_NULL;
  if (dtype != TSG_F32 && dtype != TSG_BF16) return TSG_E_DTYPE;
  if (ltype != TSG_I64 && ltype != TSG_U8) return TSG_E_DTYPE;
  if (B <= 0 || C <= 0) return TSG_E_SHAPE;
  UpBwdGeom g;
  if (!up_bwd_geom(C, IH, IW, OH, OW, &g)) return TSG_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const float sy = ac_scale(IH, OH), sx = ac_scale(IW, OW);
  const size_t sh = up_bwd_lds(C, g);
  const int64_t blocks = B * ((IH + g.CH - 1) / g.CH) * (int64_t)((IW + g.CW - 1) / g.CW);
  if (blocks > 0x7fffffffLL) return TSG_E_SHAPE;
#define PB(T, LTT)                                                                                    \
  do {                                                                                                \
    TSG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ohem_up_bwd_k<T, LTT>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));                \
    hipLaunchKernelGGL((ohem_up_bwd_k<T, LTT>), dim3((unsigned)blocks), dim3(kT), sh, st, (const T*)z, labels, \
                       C, IH, IW, OH, OW, sy, sx, g, ignore_label, weight, nll, lse, sel, gscale, (T*)dz); \
  } while (0)
  if (dtype == TSG_F32) { if (ltype == TSG_I64) PB(float, TSG_I64); else PB(float, TSG_U8); }
  else { if (ltype == TSG_I64) PB(bf16_t, TSG_I64); else PB(bf16_t, TSG_U8); }
#undef PB
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
