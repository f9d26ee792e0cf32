// conv1x1_rows.hip -- the 1x1 convolutions that feed the parallax attention (reference pasmnet/attention.py:39-40,44-45: query / key;
// methods/dcmcs3di.py:58: value), 64 input channels -> cout <= 64 output channels, float32 NCHW in, TOKEN ROWS out
// (out_rows[(n h + y) w + x][c0 + co] of a [n h, w, rows_channels] tensor: what ct_attention_rows64_f32 reads).
//
// Why a kernel of its own (round 6): ct_conv2d_split_rows_f32 ran these on the tile kernel of conv_split.hip -- 8 x 32
// pixel tiles, 16 input channels per barrier-separated stage -- at 632 us per 2-view 1080p launch = 2.7 TB/s on the 1.73 GB it moves
// (MFMA-busy 0.06).  A 1x1 convolution is a STREAMING kernel: 4096 multiply-adds per pixel against 512 bytes, HBM-bound by a factor of
// two even on the exact float32 matrix pipe (v_mfma_f32_32x32x2_f32: 157 TFLOP/s -> 216 us of matrix time at 1080p, 350 us of HBM time
// at 5 TB/s).  So: no fp16 split, no scales, no LDS, no barrier.  A wave keeps the whole 64 x 64 weight matrix in 64 registers (one A
// fragment per lane and 2-channel K step), walks 32-pixel row segments, fetches the segment's 64 channel values per pixel with 32
// coalesced dword loads per lane a tile ahead, runs 64 MFMAs and stores 16 bytes per lane and 4 output channels.  Arithmetic: float32
// products, float32 accumulation in the matrix pipe's order -- the "exact" class of cnn.hip, closer to the reference than the two-piece
// form it replaces.
#include "ct_common.h"
#include <type_traits>

namespace ct {

typedef float f32x16r __attribute__((ext_vector_type(16)));
typedef float f32x4r __attribute__((ext_vector_type(4)));

struct Rows1x1Args {
    const float *in, *w, *bias;      // w: the Conv2d weight [cout][64] (row major), bias [cout]
    float *out;
    int cout, H, W, tiles_x, n_tiles, pitch, c0, act;
    long long in_bstride;
};

__device__ __forceinline__ float rows_act(float v, int act) {
    switch (act) {
    case 1: return fmaxf(v, 0.01f * v);
    case 2: return fmaxf(v, 0.f);
    case 3: return 1.f / (1.f + __expf(-v));
    case 4: return tanhf(v);
    default: return v;
    }
}

// NMB = 32-channel output blocks (1: cout <= 32, 2: cout <= 64)
template <int NMB>
__global__ __launch_bounds__(256, 2) void conv1x1_rows_kernel(Rows1x1Args a) {
    const int lane = threadIdx.x & 63, n = lane & 31, hl = lane >> 5;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    const size_t plane = (size_t)a.H * a.W;
    // A fragments: lane (m = n, k = hl) of K step st holds w[32 mb + m][2 st + hl]
    float wa[NMB][32];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb) {
        const int co = 32 * mb + n;
        const f32x4r *wrow = reinterpret_cast<const f32x4r *>(a.w + (size_t)min(co, a.cout - 1) * 64);      // the lane's weight row, 16 bytes at a time
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4r v = wrow[q];                               // columns 4 q .. 4 q + 3 = K steps 2 q (columns 4 q, 4 q + 1) and 2 q + 1
            wa[mb][2 * q] = co < a.cout ? (hl ? v[1] : v[0]) : 0.f;
            wa[mb][2 * q + 1] = co < a.cout ? (hl ? v[3] : v[2]) : 0.f;
        }
    }
    // the bias of the 4 output channels this lane stores per (mb, g): 32 mb + 8 g + 4 hl + 0..3
    f32x4r bv[NMB][4];
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 32 * mb + 8 * g + 4 * hl;
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[mb][g][i] = co + i < a.cout ? a.bias[co + i] : 0.f;
        }
    auto tile_src = [&](int t, bool &live) -> const float * {      // this lane's pixel of tile t, channel hl (clamped inside the image)
        const int tx = t % a.tiles_x, ry = t / a.tiles_x;         // ry = image * H + row
        const int img = ry / a.H, y = ry - img * a.H;
        const int x = 32 * tx + n;
        live = x < a.W;
        return a.in + (size_t)img * a.in_bstride + (size_t)hl * plane + (size_t)y * a.W + min(x, a.W - 1);
    };
    float xb[2][32];
    bool live[2] = {false, false};
    int t = wave_id;
    if (t < a.n_tiles) {
        const float *p = tile_src(t, live[0]);
#pragma unroll
        for (int st = 0; st < 32; ++st) xb[0][st] = p[(size_t)(2 * st) * plane];
    }
    auto body = [&](auto cur_c) {
        constexpr int CUR = decltype(cur_c)::value;
        const int tn = t + n_waves;
        if (tn < a.n_tiles) {                                      // the next tile's 32 loads, a whole tile of MFMAs ahead
            const float *p = tile_src(tn, live[CUR ^ 1]);
#pragma unroll
            for (int st = 0; st < 32; ++st) xb[CUR ^ 1][st] = p[(size_t)(2 * st) * plane];
        }
        f32x16r acc[NMB];
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            const float xv = live[CUR] ? xb[CUR][st] : 0.f;
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[mb][st], xv, acc[mb], 0, 0, 0);
        }
        // D: lane (pixel n, half hl) holds output channels 32 mb + 8 g + 4 hl + i in acc[mb][4 g + i]
        const int tx = t % a.tiles_x, ry = t / a.tiles_x;
        const int x = 32 * tx + n;
        if (x < a.W) {
            float *o = a.out + ((size_t)ry * a.W + x) * a.pitch + a.c0 + 4 * hl;
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (32 * mb + 8 * g + 4 * hl < a.cout) {
                        f32x4r v;
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = rows_act(acc[mb][4 * g + i] + bv[mb][g][i], a.act);
                        *reinterpret_cast<f32x4r *>(o + 32 * mb + 8 * g) = v;
                    }
                }
        }
        t = tn;
    };
    while (t < a.n_tiles) {
        body(std::integral_constant<int, 0>());
        if (t < a.n_tiles) body(std::integral_constant<int, 1>());
    }
}

}  // namespace ct

extern "C" {

// 1x1 convolution 64 -> cout (<= 64, a multiple of 4) with token-row output, exact float32 arithmetic on the float32 matrix pipe.
// weight: the Conv2d weight itself, float32 [cout][64]; bias [cout]; out_rows: [n h, w, rows_channels], channels rows_c0 .. rows_c0 + cout
// are written (rows_channels, rows_c0 multiples of 4; out_rows and weight 16-byte aligned).  act as ct_conv2d_split_f32 (0 none .. 4 tanh).
int ct_conv1x1_rows_f32(const float *in, const float *weight, const float *bias, float *out_rows, int n, int cin, int cout, int h, int w,
                        long long in_bstride, int rows_channels, int rows_c0, int act, void *stream) {
    if (!in || !weight || !bias || !out_rows || n < 0 || cin != 64 || cout < 1 || cout > 64 || (cout & 3) || h < 0 || w < 0 || act < 0 || act > 4)
        return CT_E_BADARG;
    if (rows_channels < 4 || (rows_channels & 3) || rows_c0 < 0 || (rows_c0 & 3) || rows_c0 + cout > rows_channels) return CT_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(out_rows) & 15) || (reinterpret_cast<uintptr_t>(weight) & 15)) return CT_E_ALIGN;
    if (n == 0 || h == 0 || w == 0) return CT_OK;
    const long long tiles_x = (w + 31) / 32, n_tiles = tiles_x * h * n;
    if (n_tiles > 0x7fffffffLL) return CT_E_BADARG;
    ct::Rows1x1Args a;
    a.in = in; a.w = weight; a.bias = bias; a.out = out_rows;
    a.cout = cout; a.H = h; a.W = w; a.tiles_x = (int)tiles_x; a.n_tiles = (int)n_tiles; a.pitch = rows_channels; a.c0 = rows_c0; a.act = act;
    a.in_bstride = in_bstride;
    long long blocks = (n_tiles + 3) / 4;
    if (blocks > 256 * 2) blocks = 256 * 2;                      // persistent: the two workgroups a CU holds (213 registers per lane); a wave's
                                                                 // prologue fetches the 16 KB of weights, so waves should live long
    if (cout <= 32) hipLaunchKernelGGL(ct::conv1x1_rows_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ct::conv1x1_rows_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // extern "C"
