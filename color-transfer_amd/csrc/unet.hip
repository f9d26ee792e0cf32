// unet.hip -- the layers of DMSCT's colour-correction network (methods/dmsct.py:34-56: segmentation_models_pytorch
// EfficientNet-B2 encoder + U-Net decoder) that the convolution kernels of cnn.hip / conv_split.hip / gmflow.hip do not
// cover, on gfx950.  float32 NCHW.  (smp / efficientnet_pytorch are third-party and absent offline: oracle/smp_unet.py
// restates their published structure, "parity unpinned".)
//
//   dwconv_kernel        depthwise k x k convolution (k = 3 / 5, stride 1 / 2, TF-"SAME" static padding = explicit top/left
//                        offsets, zero fill) with the folded BatchNorm and swish in its epilogue (MBConvBlock.forward:
//                        `swish(bn1(depthwise_conv(x)))`), and the per-tile sums of its output for the squeeze step
//   se_scale_kernel      squeeze-and-excitation gate: plane means from the tile sums (fixed order, float64), `se_reduce`,
//                        swish, `se_expand`, sigmoid  ->  one multiplier per (image, channel)
//   scale_planes_kernel  x[n][c][:] *= gate[n][c]   (`torch.sigmoid(x_squeezed) * x`)
//   upcat_kernel         UnetDecoder's DecoderBlock front: nearest x2 up-sampling of x, channel concatenation with the skip
// All of them are one pass over their tensor: HBM-bound (algorithmic bytes = input + output once).
#include "ct_common.h"

namespace ct {

constexpr int kDwTH = 16, kDwTW = 64;      // output tile of a workgroup (256 threads x 4 consecutive columns)

__device__ __forceinline__ float swishf(float v) { return v / (1.0f + expf(-v)); }

template <int K, int S>
__global__ __launch_bounds__(256) void dwconv_kernel(const float *__restrict__ in, const float *__restrict__ w, const float *__restrict__ bias,
                                                     float *__restrict__ out, int C, int H, int W, int Ho, int Wo, int pad_top, int pad_left,
                                                     int act, float *__restrict__ tile_sums, int tiles_x, int tiles) {
    constexpr int TR = (kDwTH - 1) * S + K, TC = (kDwTW - 1) * S + K;
    constexpr int LD = TC | 1;                                                     // odd row stride: stride-2 column reads stay conflict free
    __shared__ float tin[TR * LD];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const int c = blockIdx.y, n = blockIdx.z;
    const float *src = in + ((size_t)n * C + c) * H * W;
    const int iy0 = ty * kDwTH * S - pad_top, ix0 = tx * kDwTW * S - pad_left;
    for (int idx = tid; idx < TR * TC; idx += 256) {
        const int yy = idx / TC, xx = idx - yy * TC;
        const int gy = iy0 + yy, gx = ix0 + xx;
        tin[yy * LD + xx] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? src[(size_t)gy * W + gx] : 0.f;
    }
    float wk[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wk[i] = w[c * K * K + i];
    const float b = bias[c];
    __syncthreads();
    const int oy = tid >> 4, ox4 = (tid & 15) * 4;
    float acc[4] = {b, b, b, b};
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const float *p = tin + (oy * S + ky) * LD + ox4 * S + kx;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(wk[ky * K + kx], p[j * S], acc[j]);
        }
    const int gy = ty * kDwTH + oy, gx = tx * kDwTW + ox4;
    float s = 0.f;
    float *dst = out + ((size_t)n * C + c) * Ho * Wo + (size_t)gy * Wo + gx;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = act == 5 ? swishf(acc[j]) : acc[j];
        if (gy < Ho && gx + j < Wo) {
            dst[j] = v;
            s += v;
        }
    }
    if (tile_sums) {                                       // fixed-order sum of the tile (lanes, then waves)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((tid & 63) == 0) red[tid >> 6] = s;
        __syncthreads();
        if (tid == 0) tile_sums[((size_t)n * C + c) * tiles + tile] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// one workgroup per image: means -> reduce (swish) -> expand (sigmoid)
__global__ __launch_bounds__(256) void se_scale_kernel(const float *__restrict__ tile_sums, int tiles, float inv_plane,
                                                       const float *__restrict__ w_reduce, const float *__restrict__ b_reduce,
                                                       const float *__restrict__ w_expand, const float *__restrict__ b_expand,
                                                       float *__restrict__ gate, int C, int NSQ) {
    extern __shared__ float sm[];                          // [C] means, [NSQ] squeezed
    float *mean = sm, *sq = sm + C;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < C; c += 256) {
        const float *p = tile_sums + ((size_t)n * C + c) * tiles;
        double s = 0.0;
        for (int t = 0; t < tiles; ++t) s += (double)p[t];
        mean[c] = (float)(s * (double)inv_plane);
    }
    __syncthreads();
    for (int j = wave; j < NSQ; j += 4) {                  // one wave per squeezed channel
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(w_reduce[(size_t)j * C + c], mean[c], s);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) sq[j] = swishf(s + b_reduce[j]);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = b_expand[c];
        for (int j = 0; j < NSQ; ++j) s = fmaf(w_expand[(size_t)c * NSQ + j], sq[j], s);
        gate[(size_t)n * C + c] = 1.0f / (1.0f + expf(-s));
    }
}

__global__ __launch_bounds__(256) void scale_planes_kernel(float *__restrict__ x, const float *__restrict__ gate, int plane4) {
    const float g = gate[blockIdx.y];
    float4 *p = reinterpret_cast<float4 *>(x) + (size_t)blockIdx.y * plane4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < plane4; i += gridDim.x * 256) {
        float4 v = p[i];
        v.x *= g; v.y *= g; v.z *= g; v.w *= g;
        p[i] = v;
    }
}
__global__ __launch_bounds__(256) void scale_planes_scalar_kernel(float *__restrict__ x, const float *__restrict__ gate, int plane) {
    const float g = gate[blockIdx.y];
    float *p = x + (size_t)blockIdx.y * plane;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < plane; i += gridDim.x * 256) p[i] *= g;
}

// out[n][0:cx] = nearest-x2(x[n]), out[n][cx:cx+cs] = skip[n]; one thread per pair of output columns
__global__ __launch_bounds__(256) void upcat_kernel(const float *__restrict__ x, const float *__restrict__ skip, float *__restrict__ out,
                                                    int cx, int cs, int h, int w) {
    const int W2 = 2 * w, H2 = 2 * h;
    const int c = blockIdx.y, n = blockIdx.z;
    const int pairs = H2 * w;
    float *dst = out + ((size_t)n * (cx + cs) + c) * H2 * W2;
    if (c < cx) {
        const float *src = x + ((size_t)n * cx + c) * h * w;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < pairs; i += gridDim.x * 256) {
            const int y2 = i / w, xx = i - y2 * w;
            const float v = src[(y2 >> 1) * w + xx];
            *reinterpret_cast<float2 *>(dst + (size_t)y2 * W2 + 2 * xx) = make_float2(v, v);
        }
    } else {
        const float2 *src = reinterpret_cast<const float2 *>(skip + ((size_t)n * cs + (c - cx)) * H2 * W2);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < pairs; i += gridDim.x * 256) reinterpret_cast<float2 *>(dst)[i] = src[i];
    }
}

template <int K, int S>
static int launch_dw(const float *in, const float *w, const float *bias, float *out, int n, int c, int h, int wd, int ho, int wo, int pt,
                     int pl, int act, float *tile_sums, hipStream_t s) {
    const int tiles_x = (wo + kDwTW - 1) / kDwTW, tiles_y = (ho + kDwTH - 1) / kDwTH;
    hipLaunchKernelGGL((dwconv_kernel<K, S>), dim3(tiles_x * tiles_y, c, n), dim3(256), 0, s, in, w, bias, out, c, h, wd, ho, wo, pt, pl, act,
                       tile_sums, tiles_x, tiles_x * tiles_y);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // namespace ct

extern "C" {

// tile sums a depthwise convolution with an [out_h, out_w] output writes per (image, channel)
int ct_dwconv_tiles(int out_h, int out_w) {
    return out_h < 1 || out_w < 1 ? 0 : ((out_w + ct::kDwTW - 1) / ct::kDwTW) * ((out_h + ct::kDwTH - 1) / ct::kDwTH);
}

// w: [c][k*k] (BatchNorm folded in), bias: [c]; act: 0 none, 5 swish.  tile_sums (nullable): [n][c][ct_dwconv_tiles(out_h, out_w)].
int ct_dwconv_f32(const float *in, const float *w, const float *bias, float *out, int n, int c, int h, int wd, int k, int stride,
                  int pad_top, int pad_left, int out_h, int out_w, int act, float *tile_sums, void *stream) {
    if (!in || !w || !bias || !out || n < 0 || c < 1 || h < 1 || wd < 1 || out_h < 1 || out_w < 1 || pad_top < 0 || pad_left < 0 ||
        (act != 0 && act != 5) || c > 65535 || n > 65535)
        return CT_E_BADARG;
    if ((out_h - 1) * stride - pad_top >= h || (out_w - 1) * stride - pad_left >= wd) return CT_E_BADARG;   // an output row / column of padding only
    if (n == 0) return CT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (k == 3 && stride == 1) return ct::launch_dw<3, 1>(in, w, bias, out, n, c, h, wd, out_h, out_w, pad_top, pad_left, act, tile_sums, s);
    if (k == 3 && stride == 2) return ct::launch_dw<3, 2>(in, w, bias, out, n, c, h, wd, out_h, out_w, pad_top, pad_left, act, tile_sums, s);
    if (k == 5 && stride == 1) return ct::launch_dw<5, 1>(in, w, bias, out, n, c, h, wd, out_h, out_w, pad_top, pad_left, act, tile_sums, s);
    if (k == 5 && stride == 2) return ct::launch_dw<5, 2>(in, w, bias, out, n, c, h, wd, out_h, out_w, pad_top, pad_left, act, tile_sums, s);
    return CT_E_BADARG;
}

// gate[n][c] = sigmoid(b_expand + w_expand . swish(b_reduce + w_reduce . mean)), mean[c] = sum(tile_sums[n][c][:]) / plane
int ct_se_gate_f32(const float *tile_sums, int tiles, int plane, const float *w_reduce, const float *b_reduce, const float *w_expand,
                   const float *b_expand, float *gate, int n, int c, int nsq, void *stream) {
    if (!tile_sums || !w_reduce || !b_reduce || !w_expand || !b_expand || !gate || tiles < 1 || plane < 1 || n < 0 || c < 1 || nsq < 1 ||
        (size_t)(c + nsq) * 4 > 60000)
        return CT_E_BADARG;
    if (n == 0) return CT_OK;
    hipLaunchKernelGGL(ct::se_scale_kernel, dim3(n), dim3(256), (size_t)(c + nsq) * sizeof(float), (hipStream_t)stream, tile_sums, tiles,
                       1.0f / (float)plane, w_reduce, b_reduce, w_expand, b_expand, gate, c, nsq);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

// x[p][:] *= gate[p] for `planes` planes of `plane` elements, in place
int ct_scale_planes_f32(float *x, const float *gate, int planes, int plane, void *stream) {
    if (!x || !gate || planes < 0 || plane < 1 || planes > 65535 * 64) return CT_E_BADARG;
    if (planes == 0) return CT_OK;
    if (planes > 65535) return CT_E_BADARG;
    const bool vec = (plane % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const int work = vec ? plane / 4 : plane;
    int bx = (work + 255) / 256;
    if (bx > 64) bx = 64;
    if (vec) hipLaunchKernelGGL(ct::scale_planes_kernel, dim3(bx, planes), dim3(256), 0, (hipStream_t)stream, x, gate, plane / 4);
    else hipLaunchKernelGGL(ct::scale_planes_scalar_kernel, dim3(bx, planes), dim3(256), 0, (hipStream_t)stream, x, gate, plane);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

// out [n][cx+cs][2h][2w] = cat(nearest-x2(x [n][cx][h][w]), skip [n][cs][2h][2w]); skip may be NULL with cs == 0
int ct_upsample2_concat_f32(const float *x, const float *skip, float *out, int n, int cx, int cs, int h, int w, void *stream) {
    if (!x || !out || n < 0 || cx < 1 || cs < 0 || h < 1 || w < 1 || (cs > 0 && !skip) || cx + cs > 65535 || n > 65535) return CT_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(skip)) & 7) return CT_E_ALIGN;
    if (n == 0) return CT_OK;
    int bx = (2 * h * w + 255) / 256;
    if (bx > 128) bx = 128;
    hipLaunchKernelGGL(ct::upcat_kernel, dim3(bx, cx + cs, n), dim3(256), 0, (hipStream_t)stream, x, skip, out, cx, cs, h, w);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // extern "C"
