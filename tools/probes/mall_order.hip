// Probe: does the order in which a streaming pass walks a tensor matter to the pass that follows it?
// (256 MiB Infinity Cache behind the L2s: a consumer that starts where its producer ENDED finds the
// most recently touched bytes on-die.)  hipcc --offload-arch=gfx950 -O3 mall_order.hip -o mall_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// chunk = 256 threads x 4 x 16 B = 16 KiB per block iteration
__global__ __launch_bounds__(256) void write_k(uint4* p, long nchunk, int rev) {
    for (long c = blockIdx.x; c < nchunk; c += gridDim.x) {
        long cc = rev ? nchunk - 1 - c : c;
        uint4* q = p + cc * 1024 + threadIdx.x;
        uint4 v = make_uint4((unsigned)cc, 1, 2, 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i * 256] = v;
    }
}
__global__ __launch_bounds__(256) void read_k(const uint4* p, long nchunk, int rev, unsigned* out) {
    unsigned acc = 0;
    for (long c = blockIdx.x; c < nchunk; c += gridDim.x) {
        long cc = rev ? nchunk - 1 - c : c;
        const uint4* q = p + cc * 1024 + threadIdx.x;
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = q[i * 256];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    }
    if (acc == 0x12345) out[0] = acc;
}
__global__ __launch_bounds__(256) void copy_k(const uint4* p, uint4* d, long nchunk, int rev) {
    for (long c = blockIdx.x; c < nchunk; c += gridDim.x) {
        long cc = rev ? nchunk - 1 - c : c;
        const uint4* q = p + cc * 1024 + threadIdx.x;
        uint4* w = d + cc * 1024 + threadIdx.x;
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = q[i * 256];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i * 256] = v[i];
    }
}

int main() {
    const long MB = 1 << 20;
    long sizes[] = {64 * MB, 128 * MB, 192 * MB, 256 * MB, 384 * MB, 512 * MB};
    uint4 *a, *b, *flush; unsigned* out;
    CK(hipMalloc(&a, 512 * MB)); CK(hipMalloc(&b, 512 * MB)); CK(hipMalloc(&flush, 1024 * MB)); CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 2048;
    auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("size_MB  producer      consumer      us      GB/s(consumer bytes)\n");
    for (long S : sizes) {
        long nchunk = S / 16384;
        // producer kinds: 0 = write fwd, 1 = read fwd ; consumer: read fwd / read rev / copy fwd / copy rev
        for (int prod = 0; prod < 2; ++prod)
            for (int cons = 0; cons < 4; ++cons) {
                std::vector<float> t;
                for (int rep = 0; rep < 7; ++rep) {
                    // flush the on-die cache with 1 GiB of unrelated traffic
                    hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, flush, 1024 * MB / 16384, 0);
                    if (prod == 0) hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, a, nchunk, 0);
                    else           hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, nchunk, 0, out);
                    CK(hipEventRecord(e0, 0));
                    if (cons < 2) hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, nchunk, cons & 1, out);
                    else          hipLaunchKernelGGL(copy_k, dim3(grid), dim3(256), 0, 0, a, b, nchunk, cons & 1);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep >= 2) t.push_back(ms * 1000.f);
                }
                float us = med(t);
                double bytes = (cons < 2 ? 1.0 : 2.0) * S;
                printf("%6ld   %-12s  %-12s  %7.1f  %7.0f\n", S / MB, prod ? "read fwd" : "write fwd",
                       cons == 0 ? "read fwd" : cons == 1 ? "read rev" : cons == 2 ? "copy fwd" : "copy rev", us, bytes / us * 1e-3);
            }
    }
    // baseline: cold read (after flush) per size
    for (long S : sizes) {
        long nchunk = S / 16384;
        std::vector<float> t;
        for (int rep = 0; rep < 7; ++rep) {
            hipLaunchKernelGGL(write_k, dim3(grid), dim3(256), 0, 0, flush, 1024 * MB / 16384, 0);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(read_k, dim3(grid), dim3(256), 0, 0, a, nchunk, 0, out);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) t.push_back(ms * 1000.f);
        }
        float us = med(t);
        printf("%6ld   cold          read fwd      %7.1f  %7.0f\n", S / MB, us, (double)S / us * 1e-3);
    }
    return 0;
}
