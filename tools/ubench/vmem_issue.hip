// Micro-benchmark (round 6): what does ONE wave per SIMD pay per vector memory instruction, by access pattern?  csrc/conv_wino4.hip
// issues 16 dwordx4 instructions per wave and step; its counters say they cost ~100 cycles of issue stall each when they come in bursts.
// Patterns (64 lanes, 16 bytes per lane, one "plane" = 8 MB apart like the channel planes of a 1080p tensor):
//   0  aligned, contiguous: lane l at 16 l                                   (1 KB, 8 lines of 128 B)
//   1  conv_wino4's patch load: 4 planes x 16 lanes, lane stride 8 B, start -4 B (4-byte aligned, overlapping windows: 136 B per plane)
//   2  tile-pair layout: 8 planes x 8 lanes, lane stride 16 B, aligned        (128 B per plane)
//   3  4 planes x 16 lanes, lane stride 16 B, aligned                         (256 B per plane)
// Each wave issues NL loads back to back, waits, repeats; the footprint per CU is small (L2 / L1 resident after the first pass) so that
// the address path, not HBM, is what is timed.  s_memtime around the loop; median wave.  Also the same with buffer stores of pattern 2.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int PAT, int NL, int ST, int FILL>
__global__ __launch_bounds__(256, 1) void k(float *buf, unsigned long long *t, float *out, int iters) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t plane = 2 * 1024 * 1024;                 // floats
    size_t off;
    if (PAT == 0) off = 4 * l;
    else if (PAT == 1) off = (size_t)(l >> 4) * plane + 2 * (l & 15) + 3;
    else if (PAT == 2) off = (size_t)(l >> 3) * plane + 4 * (l & 7);
    else off = (size_t)(l >> 4) * plane + 4 * (l & 15);
    float *p = buf + off + (size_t)w * 64 + (size_t)(blockIdx.x % 64) * 4096;      // a few KB per CU: cache resident
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        f32x4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            float *q = p + 8 * 1920 * (i & 3);
            if (ST) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(q), "v"(acc) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(q) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!ST) {
#pragma unroll
            for (int i = 0; i < NL; ++i) acc += v[i];
        }
        if (FILL) {            // vector work between the bursts, as in a real kernel
#pragma unroll
            for (int i = 0; i < FILL; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i & 3]) : "v"(1.0f));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (l == 0) t[blockIdx.x * 4 + w] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

template <int PAT, int NL, int ST, int FILL = 0>
static void run(float *buf, unsigned long long *t, float *out) {
    const int iters = 200;
    hipLaunchKernelGGL((k<PAT, NL, ST, FILL>), dim3(256), dim3(256), 0, 0, buf, t, out, 10);
    hipLaunchKernelGGL((k<PAT, NL, ST, FILL>), dim3(256), dim3(256), 0, 0, buf, t, out, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), t, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("pattern %d  %s  %2d per burst, %3d VALU between: %7.1f cycles per instruction incl. the wait and the fill (median wave, 4 waves per CU issuing)\n", PAT, ST ? "store" : "load ", NL, FILL, (double)h[512] / (iters * (double)NL));
}

int main() {
    float *buf; unsigned long long *t; float *out;
    (void)hipMalloc(&buf, (size_t)9 * 2 * 1024 * 1024 * 4 + (1 << 22)); (void)hipMemset(buf, 0, (size_t)9 * 2 * 1024 * 1024 * 4 + (1 << 22));
    (void)hipMalloc(&t, 1024 * 8); (void)hipMalloc(&out, 4);
    run<0, 8, 0>(buf, t, out); run<1, 8, 0>(buf, t, out); run<2, 8, 0>(buf, t, out); run<3, 8, 0>(buf, t, out);
    run<0, 1, 0>(buf, t, out); run<1, 1, 0>(buf, t, out); run<2, 1, 0>(buf, t, out);
    run<0, 8, 1>(buf, t, out); run<2, 8, 1>(buf, t, out); run<2, 1, 1>(buf, t, out);
    run<1, 16, 0>(buf, t, out); run<2, 16, 0>(buf, t, out); run<2, 4, 1>(buf, t, out); run<2, 4, 1, 200>(buf, t, out); run<1, 8, 0, 400>(buf, t, out); run<2, 4, 0, 200>(buf, t, out);
    return 0;
}
