// conv_direct.hip -- float32 VALU convolutions for the two shapes of the GMFlow matcher on which an implicit-GEMM tile (64 output
// channels x 16-channel K steps) wastes most of its work:
//   conv_smallcout_kernel : cout <= 4, 3x3 / 1x1, stride 1, "same" padding -- the flow head's 256 -> 2 convolution
//                           (unimatch/reg_refine.py:9-22): a 64-channel tile computes 32x the channels that exist;
//   conv_smallcin_kernel  : cin <= 3, up to 7x7, stride 1 / 2 -- the backbone's 7x7 stride-2 stem (backbone.py:49, 3 -> 64) and the
//                           motion encoder's 7x7 on the 2-channel flow (reg_refine.py:58-69, convf1): K = cin * taps <= 147 is a
//                           handful of MFMA K steps against a full LDS halo pipeline.
// Plain fmaf chains in (channel, ky, kx) order, float32 accumulation: exact-f32 class, like conv_generic_kernel.
// Weights arrive in ct_gconv2d_f32's packed layout: wp[coutp/64][kh*kw][ceil(cin/2)][2][64].
#include "ct_common.h"
#include "ct_conv.h"

namespace ct {

__device__ __forceinline__ float direct_act(float v, int act) {
    switch (act) {
        case 1: return v > 0.f ? v : 0.01f * v;
        case 2: return v > 0.f ? v : 0.f;
        case 3: return 1.0f / (1.0f + expf(-v));
        case 4: return tanhf(v);
        case 5: return v / (1.0f + expf(-v));
        default: return v;
    }
}
__device__ __forceinline__ float packed_w(const GConvArgs &a, int co, int ci, int tap) {
    const int cin_pairs = (a.cin + 1) / 2, taps = a.KH * a.KW;
    return a.wp[((((size_t)(co >> 6) * taps + tap) * cin_pairs + (ci >> 1)) * 2 + (ci & 1)) * 64 + (co & 63)];
}

// Both shapes are SMALL problems (57 k pixels at GMFlow's 1/4 scale = 896 waves of pixels for 1024 SIMDs): one wave per 64 pixels
// would leave every SIMD with a single wave and nothing to hide its load -> FMA latency chain behind.  So the reduction (small
// cout) resp. the output channels (small cin) are split across the waves of a workgroup / across workgroups.

// ---- cout <= 4: 8 waves per 2 x 32 pixel tile, wave k convolves input channels [k cin/8, (k+1) cin/8) straight from global
//      memory (coalesced rows, L1 serves the nine taps), weights come as wave-uniform scalar loads; the partial sums of the eight
//      waves meet in LDS and are added in wave order ----
__global__ __launch_bounds__(512) void conv_smallcout_kernel(GConvArgs a, int tiles_x) {
    __shared__ float4 red[8][64];
    extern __shared__ float4 wl[];                               // [cin][taps] x 4 output channels
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = a.KH, P = K / 2, taps = K * K;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x, n = blockIdx.y;
    const int oy = by * 2 + (lane >> 5), ox = bx * 32 + (lane & 31);
    for (int i = tid; i < a.cin * taps; i += 512) {
        const int ci = i / taps, tap = i - ci * taps;
        float4 w4 = make_float4(packed_w(a, 0, ci, tap), 0.f, 0.f, 0.f);
        if (a.cout > 1) w4.y = packed_w(a, 1, ci, tap);
        if (a.cout > 2) w4.z = packed_w(a, 2, ci, tap);
        if (a.cout > 3) w4.w = packed_w(a, 3, ci, tap);
        wl[i] = w4;
    }
    // the nine tap positions of this lane's pixel: clamped offsets into a plane + 0 / 1 masks, computed once
    unsigned int off[9];
    float msk[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - 3 * ky;
        const int gy = oy - P + (K == 3 ? ky : 0), gx = ox - P + (K == 3 ? kx : 0);
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && (K == 3 || t == 0);
        off[t] = ok ? (unsigned int)(gy * a.W + gx) : 0u;
        msk[t] = ok ? 1.f : 0.f;
    }
    const float *in = a.in + (size_t)n * a.in_bstride;
    const size_t plane = (size_t)a.H * a.W;
    const int cper = (a.cin + 7) / 8, c_begin = wave * cper, c_end = min(a.cin, c_begin + cper);
    __syncthreads();
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = c_begin; c < c_end; ++c) {
        const float *ip = in + (size_t)c * plane;                // wave-uniform base, per-lane 32-bit offsets
        const float4 *wp = wl + (size_t)c * taps;
        if (K == 3) {
            float x[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) { const float v = ip[off[t]]; x[t] = msk[t] != 0.f ? v : 0.f; }      // select, not multiply: inf / nan at (0,0) must not leak into the padding
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 w4 = wp[t];
                acc.x = fmaf(x[t], w4.x, acc.x); acc.y = fmaf(x[t], w4.y, acc.y); acc.z = fmaf(x[t], w4.z, acc.z); acc.w = fmaf(x[t], w4.w, acc.w);
            }
        } else {
            const float v = ip[off[0]];
            const float x = msk[0] != 0.f ? v : 0.f;
            const float4 w4 = wp[0];
            acc.x = fmaf(x, w4.x, acc.x); acc.y = fmaf(x, w4.y, acc.y); acc.z = fmaf(x, w4.z, acc.z); acc.w = fmaf(x, w4.w, acc.w);
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && oy < a.Ho && ox < a.Wo) {
        float4 s4 = red[0][lane];
#pragma unroll
        for (int k = 1; k < 8; ++k) { const float4 t = red[k][lane]; s4.x += t.x; s4.y += t.y; s4.z += t.z; s4.w += t.w; }
        float *op = a.out + (size_t)n * a.out_bstride + (size_t)oy * a.Wo + ox;
        const size_t oplane = (size_t)a.Ho * a.Wo;
        const float v[4] = {s4.x, s4.y, s4.z, s4.w};
        for (int co = 0; co < a.cout; ++co) op[co * oplane] = direct_act(v[co] + (a.bias ? a.bias[co] : 0.f), a.act);
    }
}

// ---- cin <= 3: a thread owns one output pixel and keeps its whole receptive field in registers; a workgroup computes a slice of
//      the output channels (gridDim.z = images x slices) whose weights sit in LDS as [channel][T] rows (T = cin * taps padded to 4),
//      read as wave-uniform float4 broadcasts ----
template <int CIN, int K>
__global__ __launch_bounds__(256) void conv_smallcin_kernel(GConvArgs a, int co_slices, int co_per) {
    constexpr int T = CIN * K * K, TP = (T + 3) & ~3;
    extern __shared__ float smem[];
    float4 *wl = reinterpret_cast<float4 *>(smem);               // [co_per][TP / 4]
    const int tid = threadIdx.x;
    const int n = blockIdx.z / co_slices, co0 = (blockIdx.z % co_slices) * co_per, nco = min(co_per, a.cout - co0);
    for (int i = tid; i < nco * TP; i += 256) {
        const int co = i / TP, t = i - co * TP;
        smem[i] = t < T ? packed_w(a, co0 + co, t / (K * K), t % (K * K)) : 0.f;
    }
    const int ox = blockIdx.x * 64 + (tid & 63), oy = blockIdx.y * 4 + (tid >> 6);
    const float *in = a.in + (size_t)n * a.in_bstride;
    const size_t plane = (size_t)a.H * a.W;
    float x[TP];
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int gy = oy * a.stride - a.padH + ky, gx = ox * a.stride - a.padW + kx;
                x[(c * K + ky) * K + kx] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? in[c * plane + (size_t)gy * a.W + gx] : 0.f;
            }
#pragma unroll
    for (int t = T; t < TP; ++t) x[t] = 0.f;
    __syncthreads();
    if (oy >= a.Ho || ox >= a.Wo) return;
    const size_t oplane = (size_t)a.Ho * a.Wo;
    float *op = a.out + (size_t)n * a.out_bstride + (size_t)co0 * oplane + (size_t)oy * a.Wo + ox;
    for (int co = 0; co < nco; ++co) {
        const float4 *w = wl + co * (TP / 4);
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < TP / 4; ++q) {
            const float4 w4 = w[q];
            acc = fmaf(x[4 * q], w4.x, acc); acc = fmaf(x[4 * q + 1], w4.y, acc);
            acc = fmaf(x[4 * q + 2], w4.z, acc); acc = fmaf(x[4 * q + 3], w4.w, acc);
        }
        op[co * oplane] = direct_act(acc + (a.bias ? a.bias[co0 + co] : 0.f), a.act);
    }
}

template <int CIN, int K>
static int launch_smallcin(const GConvArgs &a, int N, hipStream_t s) {
    constexpr int TP = (CIN * K * K + 3) & ~3;
    // enough output-channel slices for ~4 waves per SIMD, at least 8 channels per slice
    const long long waves = (long long)((a.Wo + 63) / 64) * ((a.Ho + 3) / 4) * 4 * N;
    int slices = (int)((4096 + waves - 1) / waves);
    slices = slices < 1 ? 1 : slices;
    const int max_slices = (a.cout + 7) / 8;
    if (slices > max_slices) slices = max_slices;
    const int co_per = (a.cout + slices - 1) / slices;
    slices = (a.cout + co_per - 1) / co_per;
    const size_t lds = (size_t)co_per * TP * sizeof(float);
    if (lds > 64 * 1024 || (long long)N * slices > 65535) return 1;
    dim3 grid((a.Wo + 63) / 64, (a.Ho + 3) / 4, N * slices);
    hipLaunchKernelGGL((conv_smallcin_kernel<CIN, K>), grid, dim3(256), lds, s, a, slices, co_per);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int conv_direct(const GConvArgs &a, int N, hipStream_t s) {
    static const int enabled = [] { const char *e = getenv("CT_HIP_CONV_DIRECT"); return e ? atoi(e) : 1; }();
    if (!enabled) return 1;
    if (a.cout <= 4 && a.KH == a.KW && (a.KH == 3 || a.KH == 1) && a.stride == 1 && a.padH == a.KH / 2 && a.padW == a.KW / 2) {
        const int tiles_x = (a.Wo + 31) / 32, tiles_y = (a.Ho + 1) / 2;
        const size_t lds = (size_t)a.cin * a.KH * a.KW * sizeof(float4);
        if (lds > 48 * 1024 || (long long)a.H * a.W > 0x7fffffffLL) return 1;
        hipLaunchKernelGGL(conv_smallcout_kernel, dim3(tiles_x * tiles_y, N), dim3(512), lds, s, a, tiles_x);
        CT_CHECK_LAUNCH();
        return CT_OK;
    }
    if (a.KH == 7 && a.KW == 7 && (a.stride == 1 || a.stride == 2)) {
        if (a.cin == 2) return launch_smallcin<2, 7>(a, N, s);
        if (a.cin == 3) return launch_smallcin<3, 7>(a, N, s);
    }
    return 1;
}

}  // namespace ct
