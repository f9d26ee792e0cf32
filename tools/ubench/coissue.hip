// Micro-benchmark (round 3): do MFMA and VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?
// Each workgroup = 512 threads = 8 waves = 2 per SIMD.  Modes: (0) both waves of a SIMD run N MFMAs; (1) both run M VALU ops;
// (2) one wave runs the MFMAs, its SIMD partner the VALU ops; (3) both run an interleaved stream (MFMA + VALU per iteration).
// If (2) takes max(t_mfma, t_valu) the pipes overlap across waves; if it takes the sum they serialise.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/coissue.hip -o tools/ubench/coissue ; run through gpurun.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(float *out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;                 // waves w and w + 4 share a SIMD (round-robin placement)
    const bool first = wave < 4;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f); b[i] = (_Float16)1.0f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const bool do_mfma = mode == 0 || mode == 3 || (mode == 2 && first);
    const bool do_valu = mode == 1 || mode == 3 || (mode == 2 && !first);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);   // 12 MFMAs = 384 clk
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 20; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(1.0000001f), "v"(1e-7f));   // 160 FMAs ~ 384 clk
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float *out; hipMalloc(&out, 4);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 10, mode);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const char *names[4] = {"both waves: 12 MFMA / iter", "both waves: 160 v_fma / iter", "wave A MFMA, wave B v_fma", "both waves: MFMA + v_fma"};
        printf("mode %d  %-30s %8.3f ms  = %7.1f clk per iteration per SIMD @2.4 GHz\n", mode, names[mode], ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
