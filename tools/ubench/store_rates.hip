// Micro-benchmark: issue cost of global stores/loads by shape, 8 waves per CU (2 per SIMD), like the conv epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float *buf, size_t plane, int iters, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nl = lane & 31, hl = lane >> 5;
    const size_t tile = (size_t)blockIdx.x * 4 + wave;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const size_t base = ((tile * iters + it) * 64) % (plane - 4096);
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            if (MODE == 0) buf[(size_t)(2 * r + hl) * plane + base + nl] = (float)r;                 // 2 x 128B segments, planes apart
            if (MODE == 1) buf[(size_t)r * plane + base + lane] = (float)r;                           // 1 x 256B contiguous
            if (MODE == 2) reinterpret_cast<float4 *>(buf + (size_t)r * plane + base * 4)[lane] = make_float4(r, r, r, r);  // 1 KiB contiguous
            if (MODE == 3) { float v = buf[(size_t)(2 * r + hl) * plane + base + nl]; if (v == 123.f) buf[0] = v; }        // loads, 2 x 128B
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}
template <int MODE> void run(const char *name, float *buf, size_t plane, unsigned long long *cyc) {
    const int blocks = 512, iters = 16;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, plane, 2, cyc);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, plane, iters, cyc);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 2048; ++i) s += h[i];
    const double bytes = (double)blocks * 4 * iters * 32 * 64 * (MODE == 2 ? 16 : 4);
    printf("%-34s %8.3f ms  %8.0f cycles/instr/wave  %7.1f GB/s\n", name, ms, s / 2048 / (iters * 32), bytes / ms / 1e6);
}
int main() {
    const size_t plane = 512 * 512;
    float *buf; hipMalloc(&buf, plane * 66 * sizeof(float) * 4); hipMemset(buf, 0, plane * 66 * 4 * 4);
    unsigned long long *cyc; hipMalloc(&cyc, 2048 * 8);
    run<0>("store 4B/lane, 2x128B, planes", buf, plane, cyc);
    run<1>("store 4B/lane, 256B contiguous", buf, plane, cyc);
    run<2>("store 16B/lane, 1KiB contiguous", buf, plane, cyc);
    run<3>("load 4B/lane, 2x128B, planes", buf, plane, cyc);
    return 0;
}
