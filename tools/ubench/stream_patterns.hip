// Micro-benchmark (round 2): how an interleaved HWC float32 image should be streamed on gfx950.
// A lane needs WHOLE pixels (3 floats); the candidates per wave and 256 pixels (3 KB):
//   P1  3 x global_load_dwordx4 per lane at a 48-byte lane stride (lane owns 4 consecutive pixels)      [round-1 kernels]
//   P2  4 x global_load_dwordx3 per lane, lane stride 12 bytes (lane owns pixels l, l+64, l+128, l+192)
//   P3  3 x fully coalesced global_load_dwordx4 (lane i <-> 16 i) + per-wave LDS transpose to 4 pixels per lane
//   P0  3 x fully coalesced global_load_dwordx4, no regrouping (upper bound; lanes do not hold whole pixels)
// read test: sum of all elements; copy test: out = in * 1.0001 with the same pattern on the store side.
// build: make -C tools/ubench stream_patterns ; run on the GPU box through gpurun.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float float3v __attribute__((ext_vector_type(3)));

template <int P, bool COPY>
__global__ __launch_bounds__(256) void k(const float *__restrict__ in, float *__restrict__ out, int64_t n_chunks /* 4-pixel chunks */,
                                         float *__restrict__ sink) {
    __shared__ float xpose[P == 3 ? 256 * 12 : 1];
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 256;
    float acc = 0.f;
    for (int64_t c0 = (int64_t)blockIdx.x * 256; c0 < n_chunks; c0 += stride) {
        const int64_t wave_c0 = c0 + (threadIdx.x & ~63);       // first chunk of this wave: 64 chunks = 256 pixels = 768 floats
        if (wave_c0 + 64 > n_chunks) continue;
        const float *g = in + wave_c0 * 12;
        float *go = out + wave_c0 * 12;
        float v[12];
        if (P == 1) {
            const float4 *q = reinterpret_cast<const float4 *>(g + lane * 12);
            const float4 a = q[0], b = q[1], c = q[2];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
        } else if (P == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float3v a = *reinterpret_cast<const float3v *>(g + (j * 64 + lane) * 3);
                v[3 * j] = a.x; v[3 * j + 1] = a.y; v[3 * j + 2] = a.z;
            }
        } else if (P == 3) {
            float *lw = xpose + (threadIdx.x >> 6) * (64 * 12);
            const float4 *q = reinterpret_cast<const float4 *>(g);
            float4 *l4 = reinterpret_cast<float4 *>(lw);
            l4[lane] = q[lane]; l4[64 + lane] = q[64 + lane]; l4[128 + lane] = q[128 + lane];
            __builtin_amdgcn_wave_barrier();
            const float4 *r4 = reinterpret_cast<const float4 *>(lw + lane * 12);
            const float4 a = r4[0], b = r4[1], c = r4[2];
            __builtin_amdgcn_wave_barrier();
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
        } else {
            const float4 *q = reinterpret_cast<const float4 *>(g);
            const float4 a = q[lane], b = q[64 + lane], c = q[128 + lane];
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
        }
        if (!COPY) {
#pragma unroll
            for (int i = 0; i < 12; ++i) acc += v[i];
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) v[i] *= 1.0001f;
            if (P == 1) {
                float4 *q = reinterpret_cast<float4 *>(go + lane * 12);
                q[0] = make_float4(v[0], v[1], v[2], v[3]); q[1] = make_float4(v[4], v[5], v[6], v[7]); q[2] = make_float4(v[8], v[9], v[10], v[11]);
            } else if (P == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<float3v *>(go + (j * 64 + lane) * 3) = float3v{v[3 * j], v[3 * j + 1], v[3 * j + 2]};
            } else if (P == 3) {
                float *lw = xpose + (threadIdx.x >> 6) * (64 * 12);
                float4 *w4 = reinterpret_cast<float4 *>(lw + lane * 12);
                w4[0] = make_float4(v[0], v[1], v[2], v[3]); w4[1] = make_float4(v[4], v[5], v[6], v[7]); w4[2] = make_float4(v[8], v[9], v[10], v[11]);
                __builtin_amdgcn_wave_barrier();
                float4 *l4 = reinterpret_cast<float4 *>(lw);
                float4 *q = reinterpret_cast<float4 *>(go);
                q[lane] = l4[lane]; q[64 + lane] = l4[64 + lane]; q[128 + lane] = l4[128 + lane];
                __builtin_amdgcn_wave_barrier();
            } else {
                float4 *q = reinterpret_cast<float4 *>(go);
                q[lane] = make_float4(v[0], v[1], v[2], v[3]); q[64 + lane] = make_float4(v[4], v[5], v[6], v[7]); q[128 + lane] = make_float4(v[8], v[9], v[10], v[11]);
            }
        }
    }
    if (!COPY && acc == 12345.678f) sink[0] = acc;
}

template <int P, bool COPY>
void run(const char *name, const float *in, float *out, int64_t n_chunks, float *sink, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<P, COPY>), dim3(blocks), dim3(256), 0, 0, in, out, n_chunks, sink);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<P, COPY>), dim3(blocks), dim3(256), 0, 0, in, out, n_chunks, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)n_chunks * 48 * (COPY ? 2 : 1);
    printf("%-44s blocks %5d  %7.1f us  %6.2f TB/s\n", name, blocks, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const int64_t n_pix = 8LL * 1080 * 1920;      // 8 images of 1080p, like one stats launch of bench.py
    const int64_t n_chunks = n_pix / 4;
    float *in, *out, *sink;
    hipMalloc(&in, n_chunks * 48); hipMalloc(&out, n_chunks * 48); hipMalloc(&sink, 4);
    hipMemset(in, 0x3c, n_chunks * 48);
    for (int blocks : {1024, 2048, 4096}) {
        run<0, false>("read  P0 coalesced x4 (no regroup)", in, out, n_chunks, sink, blocks);
        run<1, false>("read  P1 3 x dwordx4 @48 B lane stride", in, out, n_chunks, sink, blocks);
        run<2, false>("read  P2 4 x dwordx3 (pixel per lane)", in, out, n_chunks, sink, blocks);
        run<3, false>("read  P3 coalesced x4 + LDS transpose", in, out, n_chunks, sink, blocks);
        run<0, true>("copy  P0 coalesced x4 (no regroup)", in, out, n_chunks, sink, blocks);
        run<1, true>("copy  P1 3 x dwordx4 @48 B lane stride", in, out, n_chunks, sink, blocks);
        run<2, true>("copy  P2 4 x dwordx3 (pixel per lane)", in, out, n_chunks, sink, blocks);
        run<3, true>("copy  P3 coalesced x4 + LDS transpose", in, out, n_chunks, sink, blocks);
    }
    return 0;
}
