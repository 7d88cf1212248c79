// Shared device helpers for the gfx950 kernels: 16-byte vector access for
// f32 / bf16, wave64 and block reductions, launch-error plumbing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tsg_hip.h"

#define TSG_WAVE 64

// Every launch of this library first DROPS whatever error an earlier, unrelated HIP call of the calling thread left
// behind (the framework's own probing calls leave e.g. hipErrorNoDevice = 100 in the thread's last-error slot: round 5,
// smoke() after a float64 host pass), so that TSG_CHECK_LAUNCH below reports the error of THIS launch only.
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)          \
  do {                                                                                             \
    (void)hipGetLastError();                                                                       \
    (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);           \
  } while (0)

#define TSG_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

#define TSG_HIP(call)                              \
  do {                                             \
    hipError_t e__ = (call);                       \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

namespace tsg {

typedef uint16_t bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even, NaN kept quiet (matches torch's float->bfloat16)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

// two floats -> two bf16 in one word (low half = a), round to nearest even: v_cvt_pk_bf16_f32 on gfx950
typedef __attribute__((ext_vector_type(2))) float tsg_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 tsg_bf16x2;
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
  const tsg_f32x2 f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, tsg_bf16x2));
}

// Vec<T>: one 16-byte global access worth of elements, unpacked to floats.
template <typename T> struct Vec;

// Raw / ldraw / unpack: the 16 bytes as the load leaves them, and the conversion as a separate step.  A streaming loop that
// wants SEVERAL loads in flight issues all its ldraw() first and unpacks afterwards: load() converts at once, and any
// arithmetic on a loaded value is where the compiler puts the s_waitcnt for it.
template <> struct Vec<float> {
  static constexpr int N = 4;
  float v[4];
  typedef float4 Raw;
  static __device__ __forceinline__ Raw ldraw(const float* p) { return *reinterpret_cast<const float4*>(p); }
  // an operand nobody reads again soon: a NON-TEMPORAL load, so that it does not push what the next kernel will read out of
  // L2 / Infinity Cache (round 6: the BatchNorm apply passes' inputs, step - 0.65 % in four interleaved pairs,
  // profiles/r06_nontemporal_stores.txt; -DTSG_NO_NT_LOAD builds the plain loads)
  static __device__ __forceinline__ Raw ldraw_dead(const float* p) {
#if !defined(TSG_NO_NT_LOAD)
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
#else
    return ldraw(p);
#endif
  }
  __device__ __forceinline__ void unpack(const Raw& t) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  __device__ __forceinline__ void load(const float* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p) const {
#if defined(TSG_NT_STORE)
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const nt_f4 t = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
#else
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
#endif
  }
};

template <> struct Vec<bf16_t> {
  static constexpr int N = 8;
  float v[8];
  typedef uint4 Raw;
  static __device__ __forceinline__ Raw ldraw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ Raw ldraw_dead(const bf16_t* p) {
#if !defined(TSG_NO_NT_LOAD)
    typedef unsigned int nt_u4 __attribute__((ext_vector_type(4)));
    const nt_u4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u4*>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
#else
    return ldraw(p);
#endif
  }
  __device__ __forceinline__ void unpack(const Raw& t) {
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i]     = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void load(const bf16_t* p) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i]     = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void store(bf16_t* p) const {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
#if defined(TSG_NT_STORE)
    // experiment (round 6): the streaming passes' outputs as non-temporal stores — does the NEXT kernel stop paying for
    // this one's write-backs?  Built only with -DTSG_NT_STORE; see profiles/r06_nontemporal_stores.txt
    typedef unsigned int nt_u4 __attribute__((ext_vector_type(4)));
    const nt_u4 t = {w[0], w[1], w[2], w[3]};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_u4*>(p));
#else
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
#endif
  }
};

// Global loads the compiler's waitcnt pass does not see.  Around a prefetch that lives across the back edge of the K loop
// the pass is conservative: it drained EVERY outstanding load before the first MFMA of a tile (s_waitcnt vmcnt(7) .. (0) in
// the ISA of round 3's first fragment-order PSA build), so the tile time was one full memory latency whatever the prefetch depth.  These
// loads are paired with vm_wait<N>, which names the registers it makes valid (the "+v" operands order their consumers
// after it); loads return in order, so "at most N outstanding" with N = the loads issued AFTER the set that is needed.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define TSG_ASM_LD16(dst, ptr, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(dst) : "v"(ptr))
template <int N>
__device__ __forceinline__ void vm_wait(u32x4& a) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N)); }
__device__ __forceinline__ void vm_tie(u32x4& a) { asm volatile("" : "+v"(a)); }

template <typename T> __device__ __forceinline__ float ld1(const T* p);
template <> __device__ __forceinline__ float ld1<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void st1(T* p, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }

// sum across the 64 lanes of a wave; every lane gets the total
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block-wide sum of two floats (fixed order => deterministic). `sm` needs
// 2 * (blockDim.x / 64) floats.  Result valid in thread 0.
__device__ __forceinline__ void block_sum2(float& a, float& b, float* sm) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (lane == 0) { sm[2 * w] = a; sm[2 * w + 1] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ta = 0.f, tb = 0.f;
    for (int i = 0; i < nw; ++i) { ta += sm[2 * i]; tb += sm[2 * i + 1]; }
    a = ta; b = tb;
  }
}

static inline int ceil_div_i(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

}  // namespace tsg
