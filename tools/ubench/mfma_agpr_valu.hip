// Micro-benchmark (round 6): ONE wave per SIMD (256-thread workgroup, 512 registers per wave) issuing v_mfma_f32_16x16x32_f16 with
// k independent vector instructions after each -- the instruction stream of csrc/conv_wino4.hip's matrix phase.  Questions:
//   (a) cycles per (MFMA + k VALU) slice: max(16, 8 + 4 k) as MI355X_MICROARCH.md prices it, or the sum 16 + 4 k?
//   (b) does it matter that the A operand lives in the accumulation registers (inline asm, "a" constraint) instead of VGPRs?
//   (c) what does one ds_write_b32 per slice add, and how far apart must two MFMAs on the SAME accumulator be?
// In-kernel s_memtime (shader cycles), median over workgroups.  build: hipcc -O3 --offload-arch=gfx950 mfma_agpr_valu.hip -o mfma_agpr_valu
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF_A(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b))
#define MF_V(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define VADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(one))

// K: VALU per slice; AG: A operand in a[] (1) or v[] (0); NACC: accumulators used round-robin; LDSW: ds_write_b32 per slice
template <int K, int AG, int NACC, int LDSW>
__global__ __launch_bounds__(256, 1) void k(const u32x4 *w, unsigned long long *t, float *out, int iters) {
    __shared__ unsigned int sm[16384];
    u32x4 wr[16], b0, b1;
    for (int i = 0; i < 16; ++i) wr[i] = w[i * 64 + (threadIdx.x & 63)];
    b0 = w[1024 + threadIdx.x]; b1 = w[2048 + threadIdx.x];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float one = 1.0f;
    const unsigned int spa = threadIdx.x * 4;
    sm[threadIdx.x] = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            if constexpr (AG) MF_A(acc[c % NACC], wr[c & 15], (c & 1) ? b1 : b0);
            else MF_V(acc[c % NACC], wr[c & 15], (c & 1) ? b1 : b0);
#pragma unroll
            for (int i = 0; i < K; ++i) VADD(v[(c * K + i) & 7]);
            if constexpr (LDSW) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(spa), "v"(v[c & 7]), "n"((c & 15) * 1024) : "memory");   // sm is the only LDS object: offset 0
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (s == 12345.678f) out[0] = s + sm[threadIdx.x + 7];
}

template <int K, int AG, int NACC, int LDSW>
static void run(const u32x4 *w, unsigned long long *t, float *out) {
    const int iters = 200;
    hipLaunchKernelGGL((k<K, AG, NACC, LDSW>), dim3(256), dim3(256), 0, 0, w, t, out, 10);
    hipLaunchKernelGGL((k<K, AG, NACC, LDSW>), dim3(256), dim3(256), 0, 0, w, t, out, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), t, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("K=%d VALU/slice  A in %s  %d accumulator(s)  %d ds_write/slice : %6.1f cycles per slice (median wave)\n", K, AG ? "a[]" : "v[]", NACC, LDSW,
           (double)h[512] / (iters * 32.0));
}

int main() {
    u32x4 *w; unsigned long long *t; float *out;
    hipMalloc(&w, 4096 * 16); hipMemset(w, 0, 4096 * 16); hipMalloc(&t, 1024 * 8); hipMalloc(&out, 4);
    run<0, 1, 2, 0>(w, t, out); run<0, 0, 2, 0>(w, t, out); run<0, 1, 1, 0>(w, t, out); run<0, 1, 4, 0>(w, t, out);
    run<1, 1, 2, 0>(w, t, out); run<2, 1, 2, 0>(w, t, out); run<3, 1, 2, 0>(w, t, out); run<4, 1, 2, 0>(w, t, out); run<6, 1, 2, 0>(w, t, out);
    run<3, 0, 2, 0>(w, t, out); run<4, 0, 2, 0>(w, t, out);
    run<3, 1, 4, 0>(w, t, out); run<3, 1, 2, 1>(w, t, out); run<4, 1, 2, 1>(w, t, out); run<0, 1, 2, 1>(w, t, out); run<2, 1, 2, 1>(w, t, out);
    return 0;
}
