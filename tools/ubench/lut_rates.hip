// Micro-benchmark (round 2): what a table-driven sRGB<->Lab pipeline costs on gfx950.
//   (1) VALU throughput per SIMD with 8 waves/SIMD resident: f32 / packed f32 / f64 / converts / integer / transcendental
//   (2) LDS look-up throughput with per-lane random addresses (bank conflicts included): b64, b128, b128+b64 (24-byte entries)
// build: make -C tools/ubench lut_rates ; run on the GPU box through gpurun.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 256
#define CHAINS 8
typedef float float2v __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void valu(double *out, int iters, double seed) {
    double d[CHAINS];
    float f[CHAINS];
    float2v p[CHAINS];
    uint32_t u[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
        d[i] = seed + threadIdx.x * 1e-3 + i; f[i] = (float)d[i]; p[i] = float2v{f[i], f[i] + 1.f}; u[i] = threadIdx.x * 977 + i;
    }
    const double c1 = seed * 0.999, c2 = seed * 1e-3;
    const float2v pc1 = {(float)c1, (float)c1}, pc2 = {(float)c2, (float)c2};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < REP / CHAINS; ++r) {
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (OP == 0) f[i] = fmaf(f[i], (float)c1, (float)c2);                 // v_fma_f32
                if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], pc1, pc2);        // v_pk_fma_f32
                if (OP == 2) d[i] = fma(d[i], c1, c2);                                // v_fma_f64
                if (OP == 3) d[i] = d[i] + c2;                                        // v_add_f64
                if (OP == 4) { f[i] = (float)d[i]; d[i] = (double)(f[i]) ; asm volatile("" : "+v"(d[i])); }   // 2 converts
                if (OP == 5) f[i] = __builtin_amdgcn_logf(f[i]);                      // v_log_f32
                if (OP == 6) u[i] = (u[i] + 0x4000u) & 0xffff8000u;                   // v_add_u32 + v_and_b32
                if (OP == 7) u[i] = (u[i] >> 11) + u[i];                              // v_lshrrev + v_add (or v_lshl_add)
                if (OP == 8) f[i] = (f[i] > (float)c1) ? f[i] : (float)c2;            // v_cmp_f32 + v_cndmask
                if (OP == 9) d[i] = (d[i] > c1) ? d[i] : c2;                          // v_cmp_f64 + 2 v_cndmask
                if (OP == 10) f[i] = __builtin_amdgcn_fmed3f(f[i], 0.f, (float)c1);   // v_med3_f32
                if (OP == 11) u[i] = __builtin_amdgcn_ubfe(u[i], 3, 11) + u[i];       // v_bfe_u32 + add
                if (OP == 12) u[i] = u[i] * 24u + 7u;                                 // v_mad_u32_u24 / mul_lo
                if (OP == 13) d[i] = fma(d[i], d[(i + 1) % CHAINS], d[(i + 3) % CHAINS]);   // v_fma_f64, three VGPR operands
                if (OP == 14) d[i] = d[i] * d[(i + 1) % CHAINS];                      // v_mul_f64, two VGPR operands
                if (OP == 15) f[i] = fmaf(f[i], f[(i + 1) % CHAINS], f[(i + 3) % CHAINS]);  // v_fma_f32, three VGPR operands
                if (OP == 16) d[i] = fma(d[i], 0.4339463633781182, d[(i + 3) % CHAINS]);    // v_fma_f64, literal + two VGPR
                if (OP == 17) d[i] = d[i] + d[(i + 1) % CHAINS];                      // v_add_f64, two VGPR
                if (OP == 18) f[i] = f[i] * f[(i + 1) % CHAINS];                      // v_mul_f32, two VGPR
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += d[i] + f[i] + p[i].x + p[i].y + u[i];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
void run_valu(const char *name, int ops_per_rep, double seed) {
    double *out;
    hipMalloc(&out, 8);
    const int blocks = 256 * 8, iters = 64;   // 8 waves/SIMD
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(valu<OP>, dim3(blocks), dim3(256), 0, 0, out, 2, seed);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(valu<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, seed);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double winst = (double)blocks * 4 / 1024.0 * iters * REP * ops_per_rep;
    double ns_per = ms * 1e6 / winst;
    printf("VALU %-34s %8.3f ms  %6.3f ns/wave-instr/SIMD (= %5.2f cyc @2.4GHz, %5.2f @2.1GHz)\n", name, ms, ns_per, ns_per * 2.4,
           ns_per * 2.1);
    hipFree(out);
}

// ---- LDS random look-ups ------------------------------------------------------------------------------------------
// MODE 0: ds_read_b64 of 8-byte entries; 1: ds_read_b128 of 16-byte entries; 2: 24-byte entries (b128 + b64);
// MODE 3: 24-byte entries as three b64; 4: ds_read_b32 of 4-byte entries
template <int MODE, int NENT, int THREADS>
__global__ __launch_bounds__(THREADS) void lds_lookup(double *out, int iters, uint32_t seed) {
    constexpr int ESZ = MODE == 0 ? 8 : MODE == 1 ? 16 : MODE == 4 ? 4 : MODE == 2 ? 32 : 24;
    __shared__ __attribute__((aligned(16))) unsigned char tab[NENT * ESZ];
    for (int i = threadIdx.x; i < NENT * ESZ / 4; i += THREADS) reinterpret_cast<float *>(tab)[i] = (float)i * 1e-6f;
    __syncthreads();
    // per-lane odd stride -> the 64 lanes of a wave hit unrelated entries every step
    uint32_t h = (threadIdx.x + blockIdx.x * THREADS) * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    uint32_t idx[4], stp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { idx[j] = (h >> (j * 3)) % NENT; stp[j] = ((h >> (j + 5)) | 1u) % NENT; }
    double acc = 0.0;
    float facc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                idx[j] += stp[j];
                idx[j] = idx[j] >= NENT ? idx[j] - NENT : idx[j];
                const unsigned char *e = tab + idx[j] * ESZ;
                if (MODE == 0) acc += *reinterpret_cast<const double *>(e);
                if (MODE == 1) { const double2 v = *reinterpret_cast<const double2 *>(e); acc += v.x; acc += v.y; }
                if (MODE == 2) {
                    // 24 bytes used of a 32-byte padded entry: b128 + b64
                    const float4 v = *reinterpret_cast<const float4 *>(e);
                    const double w = *reinterpret_cast<const double *>(e + 16);
                    facc += v.x + v.w; acc += w;
                }
                if (MODE == 3) { const double *q = reinterpret_cast<const double *>(e); acc += q[0]; acc += q[1]; acc += q[2]; }
                if (MODE == 4) facc += *reinterpret_cast<const float *>(e);
            }
        }
    }
    if (acc + facc == 12345.678) out[0] = acc;
}

template <int MODE, int NENT, int THREADS>
void run_lds(const char *name, int blocks_per_cu) {
    double *out;
    hipMalloc(&out, 8);
    const int blocks = 256 * blocks_per_cu, iters = 200;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((lds_lookup<MODE, NENT, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, 2, 1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((lds_lookup<MODE, NENT, THREADS>), dim3(blocks), dim3(THREADS), 0, 0, out, iters, 1u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    constexpr int ESZ = MODE == 0 ? 8 : MODE == 1 ? 16 : MODE == 4 ? 4 : MODE == 2 ? 32 : 24;
    const double lookups_per_cu = (double)blocks_per_cu * THREADS * iters * 64;   // lane look-ups
    const double ns = ms * 1e6;
    printf("LDS  %-40s waves/CU %2d  %7.3f ms  %6.2f lane-lookups/ns/CU  %6.1f B/ns/CU (= %5.1f B/clk @2.1GHz)  %5.2f clk/wave-lookup/CU\n", name,
           blocks_per_cu * THREADS / 64, ms, lookups_per_cu / ns, lookups_per_cu * ESZ / ns, lookups_per_cu * ESZ / ns / 2.1,
           ns * 2.1 / (lookups_per_cu / 64));
    hipFree(out);
}

int main() {
    run_valu<0>("v_fma_f32", 1, 1.0000001);
    run_valu<1>("v_pk_fma_f32 (2 fma)", 1, 1.0000001);
    run_valu<2>("v_fma_f64", 1, 1.0000001);
    run_valu<3>("v_add_f64", 1, 1.0000001);
    run_valu<4>("cvt_f32_f64 + cvt_f64_f32", 2, 1.0000001);
    run_valu<5>("v_log_f32", 1, 1.5);
    run_valu<6>("v_add_u32 + v_and_b32", 2, 1.0);
    run_valu<7>("v_lshrrev + v_add_u32", 2, 1.0);
    run_valu<8>("v_cmp_f32 + v_cndmask", 2, 1.0000001);
    run_valu<9>("v_cmp_f64 + 2 v_cndmask", 3, 1.0000001);
    run_valu<10>("v_med3_f32", 1, 1.0000001);
    run_valu<11>("v_bfe_u32 + v_add_u32", 2, 1.0);
    run_valu<12>("u*24+7", 1, 1.0);
    run_valu<13>("v_fma_f64 3 VGPR operands", 1, 0.9999999);
    run_valu<14>("v_mul_f64 2 VGPR operands", 1, 0.9999999);
    run_valu<15>("v_fma_f32 3 VGPR operands", 1, 0.9999999);
    run_valu<16>("v_fma_f64 literal + 2 VGPR", 1, 0.9999999);
    run_valu<17>("v_add_f64 2 VGPR", 1, 0.9999999);
    run_valu<18>("v_mul_f32 2 VGPR", 1, 0.9999999);

    run_lds<4, 2048, 256>("b32, 2048 x 4 B", 8);
    run_lds<0, 2048, 256>("b64, 2048 x 8 B", 8);
    run_lds<0, 2048, 256>("b64, 2048 x 8 B", 4);
    run_lds<0, 256, 256>("b64, 256 x 8 B", 8);
    run_lds<1, 1024, 256>("b128, 1024 x 16 B", 8);
    run_lds<1, 1024, 256>("b128, 1024 x 16 B", 4);
    run_lds<2, 1024, 256>("b128+b64, 1024 x 32 B (24 used)", 6);
    run_lds<3, 1024, 256>("3 x b64, 1024 x 24 B", 6);
    run_lds<0, 2048, 1024>("b64, 2048 x 8 B, 1024-thread WG", 2);
    run_lds<1, 2048, 1024>("b128, 2048 x 16 B, 1024-thread WG", 2);
    return 0;
}
