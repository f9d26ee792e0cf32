// Micro-benchmark (round 3): issue cost of individual VALU opcodes on gfx950 (clk per wave-instruction per SIMD), forced
// with inline asm so that the compiler cannot fuse / pack / fold them, at 8 and at 4 waves per SIMD.
// Question it answers: which ops run at the double (2 clk / wave64) rate next to v_fma_f32, and which cost a full 4 clk.
// build: make -C tools/ubench valu_ops ; run on the GPU box through gpurun.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
#define REP 32      // instructions per chain per iteration

#define OPK(NAME, ASM)                                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(float *out, int iters, float seed) {                           \
        float a[CHAINS];                                                                                            \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;                   \
        float b = seed * 0.999f, c = seed * 1e-3f;                                                                  \
        for (int it = 0; it < iters; ++it) {                                                                        \
            _Pragma("unroll") for (int r = 0; r < REP; ++r) {                                                       \
                _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
            }                                                                                                       \
        }                                                                                                           \
        float s = 0;                                                                                                \
        _Pragma("unroll") for (int i = 0; i < CHAINS; ++i) s += a[i];                                               \
        if (s == 12345.678f) out[0] = s;                                                                            \
    }

OPK(fma_f32, "v_fma_f32 %0, %0, %1, %2")
OPK(fmac_f32, "v_fmac_f32 %0, %1, %2")
OPK(add_f32, "v_add_f32 %0, %0, %1")
OPK(sub_f32, "v_sub_f32 %0, %0, %1")
OPK(mul_f32, "v_mul_f32 %0, %0, %1")
OPK(add_f32_e64, "v_add_f32_e64 %0, %0, %1")
OPK(mul_f32_e64, "v_mul_f32_e64 %0, %0, %1")
OPK(max_f32, "v_max_f32 %0, %0, %1")
OPK(med3_f32, "v_med3_f32 %0, %0, %1, %2")
OPK(mov_b32, "v_mov_b32 %0, %1")
OPK(and_b32, "v_and_b32 %0, %0, %1")
OPK(or_b32, "v_or_b32 %0, %0, %1")
OPK(lshrrev_b32, "v_lshrrev_b32 %0, 3, %0")
OPK(lshlrev_b32, "v_lshlrev_b32 %0, 3, %0")
OPK(add_u32, "v_add_u32 %0, %0, %1")
OPK(sub_u32, "v_sub_u32 %0, %0, %1")
OPK(lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %1")
OPK(add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 3")
OPK(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
OPK(bfe_u32, "v_bfe_u32 %0, %0, 3, 11")
OPK(add3_u32, "v_add3_u32 %0, %0, %1, %2")
OPK(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OPK(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
OPK(cvt_f32_u32, "v_cvt_f32_u32 %0, %0")
OPK(cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
OPK(cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
OPK(min_i32, "v_min_i32 %0, %0, %1")
OPK(rcp_f32, "v_rcp_f32 %0, %0")
OPK(log_f32, "v_log_f32 %0, %0")
OPK(exp_f32, "v_exp_f32 %0, %0")
OPK(ldexp_f32, "v_ldexp_f32 %0, %0, %1")
OPK(perm_b32, "v_perm_b32 %0, %0, %1, %2")

// v_cmp writes vcc: chain through a VGPR is impossible, measure a block of compares + one dependent cndmask
__global__ __launch_bounds__(256) void k_cmp_f32(float *out, int iters, float seed) {
    float a[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    float b = seed * 0.999f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i];
    if (s == 12345.678f) out[0] = s;
}

template <typename K>
void run(const char *name, K kern, int wg_per_cu) {
    float *out;
    hipMalloc(&out, 4);
    const int blocks = 256 * wg_per_cu, iters = 64;       // 256 threads = 1 wave per SIMD per workgroup
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 2, 1.0000001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winst = (double)wg_per_cu * iters * REP * CHAINS;   // wave-instructions per SIMD
    printf("%-18s waves/SIMD %d  %7.3f ms  %5.2f clk/wave-instr/SIMD @2.1GHz (%5.2f @2.4)\n", name, wg_per_cu, ms, ms * 1e6 * 2.1 / winst,
           ms * 1e6 * 2.4 / winst);
    hipFree(out);
}

#define RUN(NAME) run(#NAME, k_##NAME, w)
int main() {
    for (int w : {8, 4, 2}) {
        RUN(fma_f32); RUN(fmac_f32); RUN(add_f32); RUN(sub_f32); RUN(mul_f32); RUN(add_f32_e64); RUN(mul_f32_e64); RUN(max_f32); RUN(med3_f32);
        RUN(mov_b32); RUN(and_b32); RUN(or_b32); RUN(lshrrev_b32); RUN(lshlrev_b32); RUN(add_u32); RUN(sub_u32); RUN(lshl_add_u32);
        RUN(add_lshl_u32); RUN(and_or_b32); RUN(bfe_u32); RUN(add3_u32); RUN(mad_u32_u24); RUN(cndmask); RUN(cvt_f32_u32);
        RUN(cvt_f32_ubyte0); RUN(cvt_u32_f32); RUN(min_i32); RUN(rcp_f32); RUN(log_f32); RUN(exp_f32); RUN(ldexp_f32); RUN(perm_b32);
        RUN(cmp_f32);
        printf("\n");
    }
    return 0;
}
