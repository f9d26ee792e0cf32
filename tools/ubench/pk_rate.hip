// micro-benchmark: issue rate of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 against v_fma_f32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s, int iters) {
    f2 a[8]; 
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i};
    const f2 m = {s, s}, c = {s * 0.5f, s * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { a[i].x = __builtin_fmaf(a[i].x, s, s * 0.5f); }                       // 1 scalar fma
                else if (MODE == 1) { a[i].x = __builtin_fmaf(a[i].x, s, s * 0.5f); a[i].y = __builtin_fmaf(a[i].y, s, s * 0.5f); }   // 2 scalar fma
                else if (MODE == 2) { a[i] = __builtin_elementwise_fma(a[i], m, c); }                // 1 packed fma
                else if (MODE == 3) { a[i] = a[i] * m; }                                             // packed mul
                else if (MODE == 4) { a[i] = a[i] + c; }                                             // packed add
            }
    }
    float r = 0; for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, float *out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<MODE><<<256 * 8, 256>>>(out, 1.0001f, 10);
    hipEventRecord(e0); k<MODE><<<256 * 8, 256>>>(out, 1.0001f, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 8 workgroups x 4 waves / 4 SIMDs = 8 waves per SIMD; instructions per wave = iters * 64 (x2 for MODE 1)
    const double instr = (double)iters * 64 * (MODE == 1 ? 2 : 1) * 8;
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (4 cycles at 2.4 GHz = 1.67 ns)\n", name, ms, ms * 1e6 / instr);
}
int main() {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_fma_f32", out); run<1>("2 x v_fma_f32", out); run<2>("v_pk_fma_f32", out); run<3>("v_pk_mul_f32", out); run<4>("v_pk_add_f32", out);
    return 0;
}
