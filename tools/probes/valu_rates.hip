// Issue rates of the VALU instructions the fused-head kernels are made of, on gfx950 (one wave per SIMD and four waves per
// SIMD): cycles per wave-instruction, measured with s_memtime around long dependent-free chains.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rates.hip -o gpurun_out/valu_rates && gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(2))) float f2;
#define N_IT 2000
template <int OP>
__global__ void k(float* out, uint64_t* cyc, float seed) {
  float a[8]; f2 p[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i * 0.001f + threadIdx.x * 1e-6f; p[i] = f2{a[i], a[i] * 0.5f}; }
  const f2 c2 = {0.999f, 1.001f};
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < N_IT; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);                       // v_exp_f32
      if (OP == 1) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);               // v_fma_f32
      if (OP == 2) p[i] = __builtin_elementwise_fma(p[i], c2, c2);            // v_pk_fma_f32
      if (OP == 3) p[i] = p[i] + c2;                                          // v_pk_add_f32
      if (OP == 4) a[i] = __builtin_amdgcn_logf(a[i] + 2.f);                  // v_log_f32 (+ add)
      if (OP == 5) a[i] = fmaxf(a[i], a[(i + 1) & 7] * 0.5f);                 // v_max + v_mul
      if (OP == 6) a[i] = __builtin_amdgcn_rcpf(a[i] + 2.f);                  // v_rcp_f32 (+ add)
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8 * 4096);
  const char* names[7] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_log_f32+v_add", "v_max_f32+v_mul", "v_rcp_f32+v_add"};
  for (int waves = 1; waves <= 4; waves *= 2) {          // waves per SIMD: block of 256 * waves threads on one CU
    for (int op = 0; op < 7; ++op) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      dim3 grid(256), block(256 * waves);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        switch (op) { case 0: k<0><<<grid, block>>>(out, cyc, 0.1f); break; case 1: k<1><<<grid, block>>>(out, cyc, 0.1f); break;
          case 2: k<2><<<grid, block>>>(out, cyc, 0.1f); break; case 3: k<3><<<grid, block>>>(out, cyc, 0.1f); break;
          case 4: k<4><<<grid, block>>>(out, cyc, 0.1f); break; case 5: k<5><<<grid, block>>>(out, cyc, 0.1f); break;
          default: k<6><<<grid, block>>>(out, cyc, 0.1f); }
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      uint64_t h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
      const double instr = (double)N_IT * 8 * ((op >= 4) ? 2 : 1);
      // one CU runs the block's 4 * waves waves on 4 SIMDs: per SIMD `waves` waves interleave
      printf("waves/SIMD %d  %-18s  %.1f us  clock-counter ticks per wave-instr (one wave's view) %.2f ; per SIMD-issued instr %.2f\n",
             waves, names[op], ms * 1e3, avg / instr, avg / instr / waves);
    }
  }
  return 0;
}
