// Micro-benchmark (round 3): does a buffer that was just streamed stay in the memory-side cache (Infinity Cache / MALL,
// 256 MB on MI355X) so that a SECOND sweep over it runs faster than HBM?  The Reinhard path reads every target frame twice
// (statistics sweep, then apply sweep); the pairs-per-step choice decides the re-use distance.
//   read A (S MB) ; touch D MB of other data (read or write) ; read A again  -> GB/s of the second read
// build: make -C tools/ubench mall_probe ; run on the GPU box through gpurun.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void rd(const float4 *__restrict__ p, size_t n4, float *out) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void wr(float4 *__restrict__ p, size_t n4, float s) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = float4{s, s, s, s};
}
__global__ __launch_bounds__(256) void cp(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

int main() {
    const size_t MB = 1 << 20;
    float4 *A, *B, *C;
    float *out;
    hipMalloc(&A, 512 * MB); hipMalloc(&B, 1024 * MB); hipMalloc(&C, 512 * MB); hipMalloc(&out, 4);
    hipMemset(A, 0, 512 * MB); hipMemset(B, 0, 1024 * MB); hipMemset(C, 0, 512 * MB);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int G = 256 * 8;
    const int sizes[] = {25, 50, 100, 200, 400};
    const int dists[] = {0, 50, 100, 200, 400, 800};
    for (int mode = 0; mode < 3; ++mode) {          // 0: first touch of A is a read; 1: a write; 2: A is read, then copy A->C timed
        for (int S : sizes) {
            for (int D : dists) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, (const float4 *)B + 0, (size_t)1024 * MB / 16, out);   // flush
                    if (mode == 1) hipLaunchKernelGGL(wr, dim3(G), dim3(256), 0, 0, A, (size_t)S * MB / 16, 1.0f);
                    else hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, (const float4 *)A, (size_t)S * MB / 16, out);
                    if (D) hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, (const float4 *)B, (size_t)D * MB / 16, out);
                    hipEventRecord(e0);
                    if (mode == 2) hipLaunchKernelGGL(cp, dim3(G), dim3(256), 0, 0, (const float4 *)A, C, (size_t)S * MB / 16);
                    else hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, (const float4 *)A, (size_t)S * MB / 16, out);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("mode %d (%s)  A = %3d MB  distance %3d MB  second pass %7.1f us  %6.0f GB/s%s\n", mode,
                       mode == 0 ? "read, read" : mode == 1 ? "write, read" : "read, copy A->C", S, D, best * 1e3,
                       (double)S * MB * (mode == 2 ? 2 : 1) / (best * 1e-3) / 1e9, mode == 2 ? " (read+write bytes)" : "");
            }
        }
    }
    // reference: cold streaming read / copy of 400 MB
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, (const float4 *)B, (size_t)1024 * MB / 16, out);
        hipEventRecord(e0);
        hipLaunchKernelGGL(rd, dim3(G), dim3(256), 0, 0, (const float4 *)A, (size_t)400 * MB / 16, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("cold read 400 MB: %7.1f us %6.0f GB/s\n", ms * 1e3, 400.0 * MB / (ms * 1e-3) / 1e9);
    }
    return 0;
}
