// Probe of gfx950's ds_read_b64_tr_b16 (LDS transpose read): LDS holds lds16[i] = i; every lane issues the
// instruction at its own byte address addr[lane] and stores the four 16-bit values it receives.  Not part of
// the library: built and run by tools/probes/run_tr_probe.py to pin down the lane/element mapping.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ void tr_probe(const uint32_t* __restrict__ addr, uint16_t* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds16[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds16[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)lds16;          // low 32 bits of a shared pointer = LDS byte offset
  const uint32_t a = base + addr[threadIdx.x];
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = (uint16_t)(v & 0xffff);
  out[threadIdx.x * 4 + 1] = (uint16_t)((v >> 16) & 0xffff);
  out[threadIdx.x * 4 + 2] = (uint16_t)((v >> 32) & 0xffff);
  out[threadIdx.x * 4 + 3] = (uint16_t)((v >> 48) & 0xffff);
}

extern "C" int tr_probe_launch(const uint32_t* addr_dev, uint16_t* out_dev, void* stream) {
  hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, addr_dev, out_dev);
  return (int)hipGetLastError();
}
