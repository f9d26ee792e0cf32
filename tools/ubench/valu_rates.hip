// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU ops the
// colour kernels are made of, at 8 waves/SIMD (the occupancy those kernels run at).
// build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
#define CHAINS 8

template <int OP>
__global__ __launch_bounds__(256) void k(double *out, int iters, double seed) {
    double d[CHAINS];
    float f[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) { d[i] = seed + threadIdx.x * 1e-3 + i; f[i] = (float)d[i]; }
    const double c1 = seed * 0.999, c2 = seed * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (OP == 0) d[i] = d[i] * c1;                       // v_mul_f64
                if (OP == 1) d[i] = fma(d[i], c1, c2);               // v_fma_f64
                if (OP == 2) d[i] = d[i] + c2;                       // v_add_f64
                if (OP == 3) f[i] = __builtin_amdgcn_logf(f[i]);     // v_log_f32
                if (OP == 4) f[i] = __builtin_amdgcn_exp2f(f[i]);    // v_exp_f32
                if (OP == 5) f[i] = fmaf(f[i], (float)c1, (float)c2);// v_fma_f32
                if (OP == 6) { f[i] = (float)d[i]; d[i] = d[i] + (double)f[i]; }   // cvt_f32_f64 + cvt_f64_f32 + add
                if (OP == 7) d[i] = __builtin_amdgcn_rcp(d[i]);      // v_rcp_f64
                if (OP == 8) d[i] = __builtin_amdgcn_rsq(d[i]);      // v_rsq_f64
                if (OP == 9) d[i] = (d[i] > c1) ? d[i] : c2;         // v_cmp_f64 + 2 cndmask
                if (OP == 10) f[i] = __builtin_amdgcn_rcpf(f[i]);    // v_rcp_f32
                if (OP == 11) f[i] = __builtin_amdgcn_sqrtf(f[i]);   // v_sqrt_f32
                if (OP == 12) d[i] = __builtin_amdgcn_sqrt(d[i]);    // v_sqrt_f64
                if (OP == 13) d[i] = __builtin_amdgcn_ldexp(d[i], 1);// v_ldexp_f64
                if (OP == 14) d[i] = __builtin_amdgcn_fract(d[i]);   // v_fract_f64
                if (OP == 15) d[i] = __builtin_amdgcn_frexp_mant(d[i]); // v_frexp_mant_f64
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += d[i] + f[i];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
double run(const char *name, int ops_per_rep, double seed) {
    double *out;
    hipMalloc(&out, 8);
    const int blocks = 256 * 8, iters = 64;   // 8 waves/SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 2, seed);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, seed);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD = blocks*4 waves / 1024 SIMDs * iters * REP
    double winst = (double)blocks * 4 / 1024.0 * iters * REP * ops_per_rep;
    double ns_per = ms * 1e6 / winst;
    printf("%-28s %8.3f ms  %7.3f ns/wave-instr/SIMD  (= %5.2f cycles @2.4GHz, %5.2f @2.1GHz)\n", name, ms, ns_per,
           ns_per * 2.4, ns_per * 2.1);
    hipFree(out);
    return ns_per;
}

int main() {
    run<0>("v_mul_f64", 1, 1.0000001);
    run<1>("v_fma_f64", 1, 1.0000001);
    run<2>("v_add_f64", 1, 1.0000001);
    run<3>("v_log_f32", 1, 1.5);
    run<4>("v_exp_f32", 1, 0.5);
    run<5>("v_fma_f32", 1, 1.0000001);
    run<6>("cvt_f32_f64+cvt_f64_f32+add", 3, 1.0000001);
    run<7>("v_rcp_f64", 1, 1.0000001);
    run<8>("v_rsq_f64", 1, 1.0000001);
    run<9>("v_cmp_gt_f64+2cndmask", 3, 1.0000001);
    run<10>("v_rcp_f32", 1, 1.0000001);
    run<11>("v_sqrt_f32", 1, 1.0000001);
    run<12>("v_sqrt_f64", 1, 1.0000001);
    run<13>("v_ldexp_f64", 1, 1.0000001);
    run<14>("v_fract_f64", 1, 1.0000001);
    run<15>("v_frexp_mant_f64", 1, 1.0000001);
    return 0;
}
