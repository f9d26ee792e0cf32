// Micro-benchmark (round 3): what per-lane random LDS look-ups cost on gfx950 when the table is REPLICATED so that the
// lanes the LDS serves in one pass own disjoint banks.  A ds_read_b128 wave-instruction is served 16 lanes per pass
// (256 B/clk), a b64 32 lanes, a b32 64 (or 32) lanes; with one replica per lane-of-a-pass every pass is conflict-free.
//   layout: byte address = (entry * R + replica(lane)) * ESZ      replica(lane) = (lane >> SH) & (R - 1)
// Prints CU-cycles per wave-look-up for entry sizes 4 / 8 / 16 B, replication 1..16(32/64), several lane->replica maps.
// build: make -C tools/ubench lds_replica ; run on the GPU box through gpurun.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ESZ>
__global__ __launch_bounds__(512) void lookup(double *out, int iters, int nent, int R, int sh, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tab[];
    for (int i = threadIdx.x; i < nent * R * ESZ / 4; i += 512) reinterpret_cast<float *>(tab)[i] = (float)i * 1e-6f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t rep = (uint32_t)(lane >> sh) & (uint32_t)(R - 1);
    uint32_t h = (threadIdx.x + blockIdx.x * 512) * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    uint32_t idx[4], stp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { idx[j] = (h >> (j * 3)) % nent; stp[j] = ((h >> (j + 5)) | 1u) % nent; }
    float facc = 0.f;
    const uint32_t un = nent;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                idx[j] += stp[j];
                idx[j] = idx[j] >= un ? idx[j] - un : idx[j];
                const unsigned char *e = tab + (idx[j] * R + rep) * ESZ;
                if (ESZ == 4) facc += *reinterpret_cast<const float *>(e);
                if (ESZ == 8) { const float2 v = *reinterpret_cast<const float2 *>(e); facc += v.x; facc += v.y; }
                if (ESZ == 16) { const float4 v = *reinterpret_cast<const float4 *>(e); facc += v.x + v.y; facc += v.z + v.w; }
            }
        }
    }
    if (facc == 12345.678f) out[0] = facc;
}

template <int ESZ>
void run(int nent, int R, int sh, int wg_per_cu) {
    const size_t lds = (size_t)nent * R * ESZ;
    if (lds > 160 * 1024 / wg_per_cu) return;
    double *out;
    hipMalloc(&out, 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(lookup<ESZ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * wg_per_cu, iters = 100;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(lookup<ESZ>, dim3(blocks), dim3(512), lds, 0, out, 2, nent, R, sh, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(lookup<ESZ>, dim3(blocks), dim3(512), lds, 0, out, iters, nent, R, sh, 1u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double wave_lookups_per_cu = (double)wg_per_cu * 8 * iters * 64;
    printf("ESZ %2d  entries %5d  R %2d  replica=(lane>>%d)&(R-1)  %3zu KB  waves/CU %2d  %7.3f ms  %6.2f clk/wave-lookup/CU @2.1GHz\n", ESZ, nent, R,
           sh, lds / 1024, wg_per_cu * 8, ms, ms * 1e6 * 2.1 / wave_lookups_per_cu);
    hipFree(out);
}

int main() {
    for (int wg = 1; wg <= 2; ++wg) {
        // unreplicated baselines
        run<4>(1024, 1, 0, wg); run<8>(1024, 1, 0, wg); run<16>(1024, 1, 0, wg);
        run<16>(128, 1, 0, wg); run<16>(64, 1, 0, wg); run<16>(16, 1, 0, wg); run<4>(64, 1, 0, wg); run<8>(32, 1, 0, wg);
        // b128: replicas 2..16, lane maps lane&.., (lane>>2)&.., (lane>>4)..
        for (int R = 2; R <= 16; R *= 2)
            for (int sh = 0; sh <= 2; ++sh) run<16>(128, R, sh, wg);
        run<16>(256, 16, 0, wg); run<16>(64, 16, 0, wg);
        // b64: replicas up to 32
        for (int R = 4; R <= 32; R *= 2)
            for (int sh = 0; sh <= 1; ++sh) run<8>(128, R, sh, wg);
        // b32: replicas up to 64
        for (int R = 8; R <= 64; R *= 2) run<4>(128, R, 0, wg);
    }
    return 0;
}
