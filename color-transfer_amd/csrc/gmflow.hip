// gmflow.hip -- building blocks of the GMFlow / UniMatch matcher forward (DMSCT's matcher) on gfx950.
//
// Replaces the ATen op sequences of the reference's unimatch/*.py for the one configuration DMSCT uses
// (methods/dmsct.py:85-94).  float32 in, float32 accumulate everywhere (exact-f32 MFMA for
// the contractions), so parity against the float32 reference is at rounding level.
//
//   conv_generic_kernel      Conv2d, any kernel / stride / channel count (backbone.py, reg_refine.py)        NCHW
//   inorm_*                  InstanceNorm2d (affine=False, eps 1e-5) [+ReLU] [+skip, ReLU] (backbone.py:34-39) NCHW
//   linear_tokens_kernel     nn.Linear on [tokens][C] (transformer.py:26-41, attention.py:181-182)            NLC
//   layernorm_tokens_kernel  LayerNorm(128) [+ residual]  (transformer.py:32,43,139-147)                        NLC
//   attention_tokens_kernel  softmax(Q K^T / sqrt(C) [+ shift mask]) V, streaming (online softmax), V = 128
//                            channels (swin window attention, attention.py:48-107) or 2 channels (global
//                            correlation -> expected coordinates, matching.py:10-39; flow propagation,
//                            attention.py:199-216)                                                              NLC
//   local_corr_softmax / local_corr_flow / local_attn_prop    matching.py:42-126, attention.py:220-256
//   convex_upsample, bilinear_resize, flow_warp, fb_check, gru ops, elementwise   utils.py:137-155, geometry.py
#include "ct_common.h"
#include "ct_conv.h"
#include "ct_attention16.h"

namespace ct {

typedef float f32x16g __attribute__((ext_vector_type(16)));

// =================================================================================================
// Generic convolution: M = 64 output channels per workgroup, N = 4 rows x 32 columns of output pixels,
// K = (kh*kw) taps x input channels staged through LDS in chunks.
// =================================================================================================
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case 1: return v > 0.f ? v : 0.01f * v;
        case 2: return v > 0.f ? v : 0.f;
        case 3: return 1.0f / (1.0f + expf(-v));
        case 4: return tanhf(v);
        case 5: return v / (1.0f + expf(-v));        // swish (efficientnet_pytorch MemoryEfficientSwish)
        default: return v;
    }
}

__global__ __launch_bounds__(256, 2) void conv_generic_kernel(GConvArgs a, int tiles_x, int tiles_y) {
    extern __shared__ float tin[];     // [cchunk][TR][TC]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int n = blockIdx.z, mt0 = blockIdx.y * 64;           // first output channel of this workgroup
    const int TR = 3 * a.stride + a.KH, TC = 31 * a.stride + a.KW, CS = TR * TC;
    const int oy0 = ty * 4, ox0 = tx * 32;
    const int iy0 = oy0 * a.stride - a.padH, ix0 = ox0 * a.stride - a.padW;
    const size_t iplane = (size_t)a.H * a.W, oplane = (size_t)a.Ho * a.Wo;
    const float *in = a.in + (size_t)n * a.in_bstride;
    const int cin_pairs = (a.cin + 1) >> 1;
    f32x16g acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    for (int c0 = 0; c0 < a.cin; c0 += a.cchunk) {
        const int cc = (a.cin - c0) < a.cchunk ? (a.cin - c0) : a.cchunk;
        const int ccp = (cc + 1) >> 1;
        __syncthreads();
        for (int idx = tid; idx < 2 * ccp * CS; idx += 256) {
            const int c = idx / CS, rem = idx - c * CS;
            const int yy = rem / TC, xx = rem - yy * TC;
            const int gy = iy0 + yy, gx = ix0 + xx;
            float v = 0.f;
            if (c < cc && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) v = in[(size_t)(c0 + c) * iplane + (size_t)gy * a.W + gx];
            tin[idx] = v;
        }
        __syncthreads();
        for (int tap = 0; tap < a.KH * a.KW; ++tap) {
            const int ky = tap / a.KW, kx = tap - ky * a.KW;
            const float *brow = tin + hl * CS + (wave * a.stride + ky) * TC + nl * a.stride + kx;
            const float *wrow = a.wp + (((size_t)blockIdx.y * a.KH * a.KW + tap) * cin_pairs + (c0 >> 1)) * 128 + hl * 64 + nl;
            for (int p = 0; p < ccp; ++p) {
                const float b = brow[p * 2 * CS];
                const float w0 = wrow[p * 128], w1 = wrow[p * 128 + 32];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, b, acc[1], 0, 0, 0);
            }
        }
    }
    const int oy = oy0 + wave, ox = ox0 + nl;
    if (oy < a.Ho && ox < a.Wo) {
        float *out = a.out + (size_t)n * a.out_bstride + (size_t)oy * a.Wo + ox;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = mt0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                if (co < a.cout) {
                    float v = acc[m][r] + (a.bias ? a.bias[co] : 0.f);
                    out[(size_t)co * oplane] = apply_act(v, a.act);
                }
            }
    }
}

// =================================================================================================
// InstanceNorm2d (affine=False, biased variance, eps): one workgroup per (n, c) plane, two passes.
//   mode 0: y = IN(x)          mode 1: y = relu(IN(x))          mode 2: y = relu(skip + relu(IN(x)))   (backbone.py:34-39)
//   mode 3: y = relu(IN(x) + skip')  is not needed: the downsample branch is normalised by its own launch.
// =================================================================================================
__global__ __launch_bounds__(256) void inorm_kernel(const float *__restrict__ x, const float *__restrict__ skip,
                                                    float *__restrict__ y, int plane, float eps, int mode) {
    __shared__ double red[8];
    const size_t base = (size_t)blockIdx.x * plane;
    double s = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < plane; i += 256) {
        const double v = x[base + i];
        s += v;
        ss += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        ss += __shfl_down(ss, off, 64);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[wid] = s; red[4 + wid] = ss; }
    __syncthreads();
    const double ts = red[0] + red[1] + red[2] + red[3], tss = red[4] + red[5] + red[6] + red[7];
    const double mean = ts / plane;
    double var = tss / plane - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float fm = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    for (int i = threadIdx.x; i < plane; i += 256) {
        float v = (x[base + i] - fm) * rstd;
        if (mode >= 1) v = v > 0.f ? v : 0.f;
        if (mode == 2) { v += skip[base + i]; v = v > 0.f ? v : 0.f; }
        y[base + i] = v;
    }
}

// Large planes (the 1/2-resolution backbone stage: 128 planes of 115k pixels) cannot fill 256 CUs with one workgroup
// per plane: split every plane over kInormSplit workgroups -- partial sums first (fixed order, no atomics), then each
// workgroup normalises its own chunk.
constexpr int kInormSplit = 16;

__global__ __launch_bounds__(256) void inorm_partial_kernel(const float *__restrict__ x, int plane, double *__restrict__ part) {
    __shared__ double red[8];
    const int chunk = (((plane + kInormSplit - 1) / kInormSplit) + 3) & ~3;
    const int i0 = blockIdx.y * chunk, i1 = (i0 + chunk < plane) ? i0 + chunk : plane;
    const float *xp = x + (size_t)blockIdx.x * plane;
    double s = 0.0, ss = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const double v = xp[i];
        s += v;
        ss += v * v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        ss += __shfl_down(ss, off, 64);
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { red[wid] = s; red[4 + wid] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *o = part + ((size_t)blockIdx.x * kInormSplit + blockIdx.y) * 2;
        o[0] = red[0] + red[1] + red[2] + red[3];
        o[1] = red[4] + red[5] + red[6] + red[7];
    }
}

__global__ __launch_bounds__(256) void inorm_apply_kernel(const float *__restrict__ x, const float *__restrict__ skip,
                                                          float *__restrict__ y, int plane, float eps, int mode,
                                                          const double *__restrict__ part) {
    const int chunk = (((plane + kInormSplit - 1) / kInormSplit) + 3) & ~3;
    const int i0 = blockIdx.y * chunk, i1 = (i0 + chunk < plane) ? i0 + chunk : plane;
    const double *pp = part + (size_t)blockIdx.x * kInormSplit * 2;
    double ts = 0.0, tss = 0.0;
#pragma unroll
    for (int z = 0; z < kInormSplit; ++z) { ts += pp[2 * z]; tss += pp[2 * z + 1]; }
    const double mean = ts / plane;
    double var = tss / plane - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float fm = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
    const size_t base = (size_t)blockIdx.x * plane;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        float v = (x[base + i] - fm) * rstd;
        if (mode >= 1) v = v > 0.f ? v : 0.f;
        if (mode == 2) { v += skip[base + i]; v = v > 0.f ? v : 0.f; }
        y[base + i] = v;
    }
}

// =================================================================================================
// Elementwise helpers
//   op 0: y = a + b              op 1: y = a * b                  op 2: y = (1 - z) * h + z * q   (a=z, b=h, c=q)
//   op 3: y = (a / 255 - mean[c]) / std[c]   (normalize_img, utils.py:26-34; a: [N,3,H,W], plane given)
//   op 4: y = a * s0             op 5: tanh(a) for channels < split, relu(a) otherwise (refine_proj chunk, unimatch.py:320-323)
// =================================================================================================
// Space-to-depth by 2: out[n][(2 sy + sx) C + c][y][x] = in[n][c][2y + sy][2x + sx] (sub-position major, so the (0, 0) sub-grid --
// what a stride-2 1x1 convolution reads -- is the first C channels).  A stride-2 3x3 "same" convolution is a stride-1 2x2
// convolution over this tensor (block offsets -1 / 0; the (by, sy) pairs (0,1), (1,0), (1,1) are the rows ky = 0, 1, 2 and (0,0)
// carries zero weights), which the MFMA tile kernel computes (conv_split_kernel<2, 2>): backbone.py:14-17,53,67 and the
// stride-2 trident branch (trident_conv.py:64-72) leave the generic kernel.  One thread: 8 input columns -> 4 + 4 outputs.
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float *__restrict__ in, float *__restrict__ out, int C, int H, int W,
                                                              long long in_bstride, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // over [n][c][input row][W / 8]
    if (i >= total) return;
    const int w8 = W >> 3, ho = H >> 1, wo = W >> 1;
    const int xg = (int)(i % w8);
    long long r = i / w8;
    const int yin = (int)(r % H); r /= H;
    const int c = (int)(r % C);
    const long long n = r / C;
    const float *src = in + n * in_bstride + ((size_t)c * H + yin) * W + 8 * xg;
    const float4 a = *reinterpret_cast<const float4 *>(src), b = *reinterpret_cast<const float4 *>(src + 4);
    const int sy = yin & 1, y = yin >> 1;
    float *dst = out + ((size_t)(n * 4 + 2 * sy) * C + c) * ho * wo + (size_t)y * wo + 4 * xg;
    *reinterpret_cast<float4 *>(dst) = make_float4(a.x, a.z, b.x, b.z);                       // sx = 0
    *reinterpret_cast<float4 *>(dst + (size_t)C * ho * wo) = make_float4(a.y, a.w, b.y, b.w);   // sx = 1
}

__global__ void eltwise_kernel(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ c,
                               float *__restrict__ y, long long n, int op, int plane, int chans, int split, float s0) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v;
    switch (op) {
        case 0: v = a[i] + b[i]; break;
        case 1: v = a[i] * b[i]; break;
        case 2: v = (1.0f - a[i]) * b[i] + a[i] * c[i]; break;
        case 3: {
            const int ch = (int)((i / plane) % 3);
            const float mean = ch == 0 ? 0.485f : (ch == 1 ? 0.456f : 0.406f);
            const float sd = ch == 0 ? 0.229f : (ch == 1 ? 0.224f : 0.225f);
            v = (a[i] / 255.0f - mean) / sd;
            break;
        }
        case 4: v = a[i] * s0; break;
        default: {
            const int ch = (int)((i / plane) % chans);
            v = ch < split ? tanhf(a[i]) : (a[i] > 0.f ? a[i] : 0.f);
        }
    }
    y[i] = v;
}

// =================================================================================================
// NCHW <-> token rows: rows[(b*H + y)*W + x][c0 + c] = nchw[b][c][y][x]  (and back).  The streaming parallax attention
// works on channels-last rows; torch's permute().contiguous() moves these tensors at < 1 TB/s.  One workgroup per
// (64 pixels of a row, b*H + y): 32-channel x 64-pixel tiles through LDS, 256-byte runs on the NCHW side, 128-byte
// runs on the row side.  grid = (ceil(W/64), B*H), block 256.
// =================================================================================================
template <bool TO_ROWS>
__global__ __launch_bounds__(256) void rows_transpose_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int H,
                                                             int W, long long nchw_bstride, int row_channels, int c0) {
    __shared__ float t[32][65];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * 64;
    const int by = blockIdx.y, b = by / H, y = by - b * H;
    const size_t plane = (size_t)H * W;
    const float *nsrc = src;
    float *ndst = dst;
    // full 64-pixel tiles with 16-byte aligned rows move as two float4 per thread on the global side (round 3: the scalar
    // form ran at 2.3 TB/s and cost DCMCS3DI 3.4 ms per 1080p pair); everything else takes the element-wise path
    const bool vec = (x0 + 64 <= W) && ((W & 3) == 0) && ((row_channels & 3) == 0) && ((c0 & 3) == 0) && ((nchw_bstride & 3) == 0) &&
                     (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0);
    for (int cb = 0; cb < C; cb += 32) {
        const bool full = vec && (cb + 32 <= C);
        if (TO_ROWS) {
            // NCHW -> LDS: thread (c = tid>>3 [+0], x = 8*(tid&7)..) : 32 channels x 64 pixels, 8 floats per thread
            const int c = tid >> 3, xs = (tid & 7) * 8;
            const float *p = nsrc + (size_t)b * nchw_bstride + (size_t)(cb + c) * plane + (size_t)y * W + x0 + xs;
            if (full) {
                const float4 a = reinterpret_cast<const float4 *>(p)[0], bb = reinterpret_cast<const float4 *>(p)[1];
                t[c][xs] = a.x; t[c][xs + 1] = a.y; t[c][xs + 2] = a.z; t[c][xs + 3] = a.w;
                t[c][xs + 4] = bb.x; t[c][xs + 5] = bb.y; t[c][xs + 6] = bb.z; t[c][xs + 7] = bb.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) t[c][xs + i] = (cb + c < C && x0 + xs + i < W) ? p[i] : 0.f;
            }
            __syncthreads();
            // LDS -> rows: thread (x = tid>>2, channel group g = tid&3 -> 8 channels)
            const int x = tid >> 2, g = (tid & 3) * 8;
            if (x0 + x < W) {
                float *q = ndst + ((size_t)by * W + x0 + x) * row_channels + c0 + cb + g;
                if (full) {
                    reinterpret_cast<float4 *>(q)[0] = make_float4(t[g][x], t[g + 1][x], t[g + 2][x], t[g + 3][x]);
                    reinterpret_cast<float4 *>(q)[1] = make_float4(t[g + 4][x], t[g + 5][x], t[g + 6][x], t[g + 7][x]);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (cb + g + i < C) q[i] = t[g + i][x];
                }
            }
            __syncthreads();
        } else {
            const int x = tid >> 2, g = (tid & 3) * 8;
            if (x0 + x < W) {
                const float *q = nsrc + ((size_t)by * W + x0 + x) * row_channels + c0 + cb + g;
                if (full) {
                    const float4 a = reinterpret_cast<const float4 *>(q)[0], bb = reinterpret_cast<const float4 *>(q)[1];
                    t[g][x] = a.x; t[g + 1][x] = a.y; t[g + 2][x] = a.z; t[g + 3][x] = a.w;
                    t[g + 4][x] = bb.x; t[g + 5][x] = bb.y; t[g + 6][x] = bb.z; t[g + 7][x] = bb.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) t[g + i][x] = (cb + g + i < C) ? q[i] : 0.f;
                }
            }
            __syncthreads();
            const int c = tid >> 3, xs = (tid & 7) * 8;
            if (cb + c < C) {
                float *p = ndst + (size_t)b * nchw_bstride + (size_t)(cb + c) * plane + (size_t)y * W + x0 + xs;
                if (full) {
                    reinterpret_cast<float4 *>(p)[0] = make_float4(t[c][xs], t[c][xs + 1], t[c][xs + 2], t[c][xs + 3]);
                    reinterpret_cast<float4 *>(p)[1] = make_float4(t[c][xs + 4], t[c][xs + 5], t[c][xs + 6], t[c][xs + 7]);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (x0 + xs + i < W) p[i] = t[c][xs + i];
                }
            }
            __syncthreads();
        }
    }
}

// =================================================================================================
// nn.Linear on channels-last tokens: out[t][n] = act( sum_k x[t][k] W[n][k] + bias[n] )
// LDS-tiled "NT" GEMM: a workgroup owns 128 tokens x 128 features, each of its 4 waves a 64 x 64 quarter (2 x 2 MFMA
// tiles).  K is walked in 32-channel chunks: both operands are K-contiguous in memory ([T][K] and PyTorch's [N][K]),
// so a chunk of either is 128 rows x 128 bytes, fetched with fully coalesced 16-byte loads into registers while the
// previous chunk is multiplied, then written to LDS rows of 36 floats (16-byte aligned, conflict-free 16-byte reads).
// The contraction index of the two lane halves is split as k = 16*hl + p inside a chunk (any order is a valid dot
// product), so the MFMA operands come out of LDS as float4.  The result goes through a per-wave 32x32 LDS transpose so
// that it is stored 16 bytes per lane.  K % 16 == 0 (128, 256, 1024 here).  grid = (ceil(T/128), ceil(N/128)).
// =================================================================================================
constexpr int kLinLd = 36;   // LDS row stride in floats

// x2 != null: the input row is the concatenation [x[t][0:K1] | x2[t][0:K-K1]] (K1 % 32 == 0) -- the
// torch.cat([source, message]) in front of the FFN (transformer.py:131) without materialising it.
__global__ __launch_bounds__(256) void linear_tokens_kernel(const float *__restrict__ x, const float *__restrict__ x2, int K1,
                                                            const float *__restrict__ w, const float *__restrict__ bias,
                                                            float *__restrict__ out, long long T, int K, int N,
                                                            int act /*0 none, 6 gelu*/) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 128 * kLinLd];
    float *Xs = lds, *Ws = lds + 128 * kLinLd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    const long long t0 = (long long)blockIdx.x * 128;
    const int n0 = blockIdx.y * 128;

    // staging: thread -> 4 (row, 16-byte column) slots of each operand tile
    const int srow = tid >> 3, sq = tid & 7;
    const float *wg[4];
    long long xr[4];
    const int K2 = K - K1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long tr = t0 + srow + 32 * i;
        const int nr = n0 + srow + 32 * i;
        xr[i] = tr < T ? tr : T - 1;
        wg[i] = w + (size_t)(nr < N ? nr : N - 1) * K + 4 * sq;
    }
    float4 px[4], pw[4];
    auto fetch = [&](int kc) {
        const bool inb = (kc + 4 * sq) < K;    // K % 4 == 0: a float4 is in range or not at all
        const bool second = kc >= K1;          // uniform: a 32-channel chunk never straddles the two sources
        const float *xs = second ? x2 + (kc - K1) + 4 * sq : x + kc + 4 * sq;
        const int ld = second ? K2 : K1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            px[i] = inb ? *reinterpret_cast<const float4 *>(xs + xr[i] * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
            pw[i] = inb ? *reinterpret_cast<const float4 *>(wg[i] + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4 *>(Xs + (srow + 32 * i) * kLinLd + 4 * sq) = px[i];
            *reinterpret_cast<float4 *>(Ws + (srow + 32 * i) * kLinLd + 4 * sq) = pw[i];
        }
    };

    f32x16g acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fetch(0);
    stage();
    const float *xa = Xs + (wm * 64 + nl) * kLinLd + hl * 16;
    const float *wb = Ws + (wn * 64 + nl) * kLinLd + hl * 16;
    for (int kc = 0; kc < K; kc += 32) {
        __syncthreads();                       // this chunk is visible
        const bool more = (kc + 32) < K;
        if (more) fetch(kc + 32);              // in flight under the MFMAs below
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a0 = *reinterpret_cast<const float4 *>(xa + 4 * q);
            const float4 a1 = *reinterpret_cast<const float4 *>(xa + 32 * kLinLd + 4 * q);
            const float4 b0 = *reinterpret_cast<const float4 *>(wb + 4 * q);
            const float4 b1 = *reinterpret_cast<const float4 *>(wb + 32 * kLinLd + 4 * q);
            const float av[2][4] = {{a0.x, a0.y, a0.z, a0.w}, {a1.x, a1.y, a1.z, a1.w}};
            const float bv[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();                       // every wave is done with this chunk
        if (more) stage();
    }

    // D[token][feature]: lane = feature column, registers = token rows (r&3)+8(r>>2)+4hl of the 32x32 tile.
    // Per-wave transpose buffer (the operand tiles are dead after the last barrier): rows = tokens, 32 features each.
    float *stg = lds + wave * (32 * 32);
    const bool wide = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f0 = n0 + wn * 64 + j * 32;
            const long long tt0 = t0 + wm * 64 + i * 32;
            const int nf = f0 + nl;
            const float bb = (bias && nf < N) ? bias[nf] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bb;
                if (act == 6) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));   // exact GELU (nn.GELU default)
                acc[i][j][r] = v;
            }
            if (wide) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * hl) * 32 + nl] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
                const int fc = f0 + 4 * (lane & 7);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = (lane >> 3) + 8 * g;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + row * 32 + 4 * (lane & 7));
                    const long long t = tt0 + row;
                    if (t < T && fc < N) *reinterpret_cast<float4 *>(out + t * N + fc) = v;   // N % 4 == 0: all four or none
                }
                __builtin_amdgcn_wave_barrier();
            } else if (nf < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long t = tt0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                    if (t < T) out[t * N + nf] = acc[i][j][r];
                }
            }
        }
}

// LayerNorm over the last dim (C = 128, eps 1e-5, affine) of [T][128], optional residual: out = res + LN(x).
// One wave per token (2 channels per lane).
__global__ __launch_bounds__(256) void layernorm_tokens_kernel(const float *__restrict__ x, const float *__restrict__ g,
                                                               const float *__restrict__ b, const float *__restrict__ res,
                                                               float *__restrict__ out, long long T, int partials) {
    const int lane = threadIdx.x & 63;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    float2 v = *reinterpret_cast<const float2 *>(x + t * 128 + 2 * lane);
    for (int p = 1; p < partials; ++p) {       // the K-sliced linear's partial slabs [partials][T][128], added in slab order
        const float2 u = *reinterpret_cast<const float2 *>(x + ((long long)p * T + t) * 128 + 2 * lane);
        v.x += u.x; v.y += u.y;
    }
    float s = v.x + v.y;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s * (1.0f / 128.0f);
    const float dx = v.x - mean, dy = v.y - mean;
    float ss = dx * dx + dy * dy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + 1e-5f);
    float o0 = dx * rstd * g[2 * lane] + b[2 * lane], o1 = dy * rstd * g[2 * lane + 1] + b[2 * lane + 1];
    if (res) { o0 += res[t * 128 + 2 * lane]; o1 += res[t * 128 + 2 * lane + 1]; }
    *reinterpret_cast<float2 *>(out + t * 128 + 2 * lane) = make_float2(o0, o1);
}

// Merge of the key splits of attention_tokens_kernel: out = sum_z o_z e^(m_z - M) / sum_z l_z e^(m_z - M).
// grid = ceil(batch*len*CV / 256); one thread per output element.
__global__ __launch_bounds__(256) void attention_combine_kernel(const float *__restrict__ part, const int *__restrict__ rowmap,
                                                                float *__restrict__ out, long long tokens, int CV, int nsplit) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= tokens * CV) return;
    const long long t = e / CV;
    const int ch = (int)(e - t * CV);
    const size_t zs = (size_t)tokens * (CV + 2);
    const float *p = part + (size_t)t * (CV + 2);
    float M = -INFINITY;
    for (int z = 0; z < nsplit; ++z) M = fmaxf(M, p[z * zs + CV]);
    float num = 0.f, den = 0.f;
    for (int z = 0; z < nsplit; ++z) {
        const float m = p[z * zs + CV];
        const float wgt = (m == -INFINITY) ? 0.f : exp2f(m - M);   // the partial maxima are log2-domain scores
        num += wgt * p[z * zs + ch];
        den += wgt * p[z * zs + CV + 1];
    }
    const size_t row = rowmap ? (size_t)rowmap[t] : (size_t)t;
    out[row * CV + ch] = num / den;
}

// =================================================================================================
// Streaming single-head attention on channels-last tokens (C = 128):
//   out[b][i][:] = sum_j softmax_j( q[b][i].k[b][j] / sqrt(C) + mask(i,j) ) v[b][j][:]
// One wave = 32 queries; keys are visited 32 at a time with an online softmax.  The score tile is computed
// TRANSPOSED (M = keys, N = queries), so a lane owns ONE query column: its running max / sum / rescale are
// per-lane scalars and the accumulator (also query-on-lane) rescales without any cross-lane traffic.
//   CV == 128: P.V on MFMA (M = value channels, N = queries, K = keys, P taken from the score registers)
//   CV == 2  : the two value channels (coordinates / flow) are accumulated by the VALU
//   region != null: additive -100 where region[i] != region[j] (shifted-window mask, utils.py:87-111)
// q/k/v rows are fetched with 16-byte loads (a lane needs C/2 consecutive channels of one token row).
// grid = (ceil(L/128), B); block = 4 waves.
// =================================================================================================
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

// ---- split-bf16 scores (SS = true): the q.k products of the 64-channel row attention on the bf16 matrix pipe with
// float32-grade accuracy -- three bf16 pieces per operand, six v_mfma_f32_32x32x16_bf16 per K step of 16 channels, float32
// accumulation (arithmetic and error bound as in conv_split.hip).  The P.V product stays on the f32 MFMA.
typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2g __attribute__((ext_vector_type(2)));
typedef float f32x2g __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16g(float a, float b) {
    f32x2g v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2g));
}
__device__ __forceinline__ void split3x2g(float x0, float x1, unsigned int &hw, unsigned int &mw, unsigned int &lw) {
    hw = pack_bf16g(x0, x1);
    const float r0 = x0 - __uint_as_float(hw << 16), r1 = x1 - __uint_as_float(hw & 0xffff0000u);
    mw = pack_bf16g(r0, r1);
    lw = pack_bf16g(r0 - __uint_as_float(mw << 16), r1 - __uint_as_float(mw & 0xffff0000u));
}
// eight consecutive floats -> the three 16-byte bf16 fragments (hi, mid, lo)
__device__ __forceinline__ void split3x8g(const float (&x)[8], uint4 &h, uint4 &m, uint4 &l) {
    unsigned int hw[4], mw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split3x2g(x[2 * i], x[2 * i + 1], hw[i], mw[i], lw[i]);
    h = make_uint4(hw[0], hw[1], hw[2], hw[3]); m = make_uint4(mw[0], mw[1], mw[2], mw[3]); l = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}
// s += A . B with A, B given as (hi, mid, lo) fragments; small terms first
__device__ __forceinline__ void mfma_split6(f32x16g &s, const uint4 (&a)[3], const uint4 (&b)[3]) {
    const bf16x8g ah = __builtin_bit_cast(bf16x8g, a[0]), am = __builtin_bit_cast(bf16x8g, a[1]), al = __builtin_bit_cast(bf16x8g, a[2]);
    const bf16x8g bh = __builtin_bit_cast(bf16x8g, b[0]), bm = __builtin_bit_cast(bf16x8g, b[1]), bl = __builtin_bit_cast(bf16x8g, b[2]);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, s, 0, 0, 0);
}
// =================================================================================================
// nn.Linear on channels-last tokens on the bf16 matrix pipe ("split", float32-grade accuracy: conv_split.hip's arithmetic).
// A workgroup owns MT (128 or 64) tokens x 128 features, its 4 waves MT/2 x 64 quarters; K is walked in chunks of 32
// channels = two v_mfma_f32_32x32x16_bf16 K steps, six MFMAs per product.
//   W: pre-split on the host in exactly the LDS image of a chunk ([piece][8-channel group][feature row] x 16 bytes,
//      ct_hip.pack_linear_weight_split): staging is a linear 24 KiB copy through registers, fetched one chunk ahead.
//   X: split in registers while it is staged (a thread owns 8 consecutive channels of a token = one MFMA fragment per
//      piece), into a DOUBLE-buffered LDS image: the split of chunk c+1 (VALU) runs between the MFMAs of chunk c, the raw
//      loads are issued two chunks ahead.  Group stride MT+4 rows: the 16 lanes of a ds_write_b128 pass hit 16 distinct
//      bank groups; fragment reads are 32 consecutive rows of one group = conflict free.
// Two barriers per chunk (X/W of the chunk visible; W consumed), only the short W copy sits between them.  LDS 73.5 KiB
// (MT = 128): two workgroups per CU, so one's prologue / epilogue (bias, GELU, transpose, stores) runs under the other's MFMAs.
// Several feature tiles (N > 128): 1-D grid ordered so that the tiles of one token block run at the same time on the same
// XCD (workgroup b is placed on XCD b % 8) -- the block's tokens come from HBM once and from that XCD's L2 afterwards.
// K % 32 == 0.
// =================================================================================================
constexpr int kLsW = 3 * 4 * 128;                 // uint4 entries of the W image (= one packed chunk)
#ifdef CT_LS_PROFILE
// diagnostic build (tools/build_variant.sh, never shipped): per-phase s_memtime totals of wave 0 of every workgroup
__device__ unsigned long long g_ls_prof[8];
#define LS_STAMP(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define LS_PHASE(i) do { unsigned long long t__; LS_STAMP(t__); pt[i] += t__ - pt0; pt0 = t__; } while (0)
#else
#define LS_PHASE(i) do { } while (0)
#endif

// GELU(v) = v Phi(v) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. float32 rounding level of
// the result; branch free: one v_rcp_f32, one v_exp_f32, seven FMAs) instead of the library erff (two branches, both of
// which a wave executes) -- the epilogue of the 1024-wide FFN layer evaluates it 16384 times per workgroup.
__device__ __forceinline__ float gelu_as(float v) {
    const float z = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    const float erfc_half = 0.5f * p * t * e;                       // erfc(z) / 2
    return v > 0.f ? v - v * erfc_half : v * erfc_half;           // v Phi(v), Phi(-z sqrt2) = erfc(z) / 2
}

template <int MT>
__global__ __launch_bounds__(256, 2) void linear_split_kernel(const float *__restrict__ x, const float *__restrict__ x2, int K1,
                                                              const uint4 *__restrict__ wp, const float *__restrict__ bias,
                                                              float *__restrict__ out, long long T, int K, int N, int act,
                                                              int n_nt) {
    constexpr int MI = MT / 64;                   // 32-token MFMA tiles per wave (and X staging units per thread)
    constexpr int XR = MT + 4;                    // rows per (piece, group) of an X image
    constexpr int XIMG = 3 * 4 * XR;              // uint4 entries of one X image
    __shared__ uint4 lds[2 * XIMG + kLsW];
    uint4 *Ws = lds + 2 * XIMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const int wm = wave & 1, wn = wave >> 1;
    // tile of this workgroup: logical ids run XCD-major, feature tile fastest
    unsigned int lid = blockIdx.x;
    if (n_nt > 1 && (gridDim.x & 7) == 0) lid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int nt = (int)(lid % (unsigned)n_nt);
    const long long t0 = (long long)(lid / (unsigned)n_nt) * MT;
    const int n0 = nt * 128;
    const int n_chunks = K >> 5;
    const uint4 *wsrc = wp + (size_t)nt * n_chunks * kLsW + tid;
    // X staging: unit u = tid + 256 i -> (token row u >> 2, 8-channel group u & 3): four lanes read one token's 128 bytes
    const int srow = tid >> 2, sg = tid & 3;
    long long xr[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const long long tr = t0 + srow + 64 * i;
        xr[i] = tr < T ? tr : T - 1;
    }
    const int K2 = K - K1;
    float4 pa[MI][2], pb[MI][2];                  // raw X of chunk c+1 (being split) / chunk c+2 (in flight)
    uint4 pw[6];
    auto fetch_x = [&](int c, float4 (&px)[MI][2]) {
        const int kc = c << 5;
        const bool second = kc >= K1;          // uniform: a 32-channel chunk never straddles the two sources
        const float *xs = second ? x2 + (kc - K1) + 8 * sg : x + kc + 8 * sg;
        const int ld = second ? K2 : K1;
#ifdef CT_LS_NOXLOAD
        if (c > 1) return;
#endif
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float4 *p = reinterpret_cast<const float4 *>(xs + xr[i] * ld);
            px[i][0] = p[0];
            px[i][1] = p[1];
        }
    };
    auto fetch_w = [&](int c) {
#ifdef CT_LS_NOWLOAD
        if (c > 1) return;
#endif
#pragma unroll
        for (int j = 0; j < 6; ++j) pw[j] = wsrc[(size_t)c * kLsW + 256 * j];
    };
    auto split_unit = [&](const float4 (&px)[MI][2], int i, uint4 *img) {
        const float v[8] = {px[i][0].x, px[i][0].y, px[i][0].z, px[i][0].w, px[i][1].x, px[i][1].y, px[i][1].z, px[i][1].w};
        uint4 h, m, l;
        split3x8g(v, h, m, l);
        uint4 *d = img + sg * XR + srow + 64 * i;
        d[0] = h;
        d[4 * XR] = m;
        d[8 * XR] = l;
    };
    auto store_w = [&]() {
#ifdef CT_LS_NOWSTORE
        if (pw[0].x != 0x12345u) return;
#endif
#pragma unroll
        for (int j = 0; j < 6; ++j) Ws[tid + 256 * j] = pw[j];
    };

    f32x16g acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#ifdef CT_LS_PROFILE
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0;
    LS_STAMP(pt0);
#endif
    fetch_x(0, pb);
    fetch_w(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) split_unit(pb, i, lds);
    store_w();
    if (n_chunks > 1) {
        fetch_x(1, pa);
        fetch_w(1);
    }
    LS_PHASE(0);                               // prologue
    const int xoff = hl * XR + wm * (MT / 2) + nl;
    const uint4 *wb = Ws + hl * 128 + wn * 64 + nl;
    // one chunk: `cur` holds the raw X of chunk c+1 (loaded a chunk ago), `nxt` receives chunk c+2
    auto chunk = [&](int c, float4 (&cur)[MI][2], float4 (&nxt)[MI][2]) {
#ifndef CT_LS_NOBAR
        __syncthreads();                       // X image c & 1 and the W image hold chunk c
#endif
        LS_PHASE(1);
        if (c + 2 < n_chunks) fetch_x(c + 2, nxt);
        const uint4 *xa = lds + (c & 1) * XIMG + xoff;
        uint4 *xn = lds + ((c + 1) & 1) * XIMG;
#pragma unroll
        for (int s = 0; s < 2; ++s) {          // K step: channels 16 s + 8 hl + 0..7 of the chunk
            uint4 a[MI][3], b[2][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i][p] = xa[(4 * p + 2 * s) * XR + 32 * i];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j][p] = wb[(4 * p + 2 * s) * 128 + 32 * j];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_split6(acc[i][j], a[i], b[j]);
            // the split of chunk c+1 (one staging unit per K step) goes to the other image, its VALU work between the MFMAs
            // above (after the last chunk it re-splits stale registers into the dead image: no branch in the schedule region)
            if (s < MI) {
                split_unit(cur, s, xn);
#pragma unroll
                for (int q = 0; q < 12 * MI; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, MI == 2 ? 3 : 6, 0);
                }
            }
        }
        LS_PHASE(2);
#ifndef CT_LS_NOBAR
        __syncthreads();                       // every wave is done with the W image
#endif
        LS_PHASE(3);
        if (c + 1 < n_chunks) {
            store_w();
            if (c + 2 < n_chunks) fetch_w(c + 2);
        }
        LS_PHASE(4);
    };
    for (int c = 0; c < n_chunks; c += 2) {    // unrolled by two: the raw-X register sets swap roles without copies
        chunk(c, pa, pb);
        if (c + 1 < n_chunks) chunk(c + 1, pb, pa);
    }

    // epilogue: per-wave LDS transpose (the operand images are dead after the last barrier), 16-byte stores
    float *stg = reinterpret_cast<float *>(lds) + wave * (32 * 32);
    const bool wide = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f0 = n0 + wn * 64 + j * 32;
            const long long tt0 = t0 + wm * (MT / 2) + i * 32;
            const int nf = f0 + nl;
            const float bb = (bias && nf < N) ? bias[nf] : 0.f;
            if (act == 6) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = gelu_as(acc[i][j][r] + bb);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += bb;
            }
            if (wide) {
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * hl) * 32 + nl] = acc[i][j][r];
                __builtin_amdgcn_wave_barrier();
                const int fc = f0 + 4 * (lane & 7);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int row = (lane >> 3) + 8 * g;
                    const float4 v = *reinterpret_cast<const float4 *>(stg + row * 32 + 4 * (lane & 7));
                    const long long t = tt0 + row;
                    if (t < T && fc < N) *reinterpret_cast<float4 *>(out + t * N + fc) = v;   // N % 4 == 0: all four or none
                }
                __builtin_amdgcn_wave_barrier();
            } else if (nf < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long t = tt0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                    if (t < T) out[t * N + nf] = acc[i][j][r];
                }
            }
        }
#ifdef CT_LS_PROFILE
    LS_PHASE(5);                               // epilogue
    if (tid == 0) {
        for (int i = 0; i < 7; ++i) atomicAdd(&g_ls_prof[i], pt[i]);
        atomicAdd(&g_ls_prof[7], 1ull);
    }
#endif
}

// -------------------------------------------------------------------------------------------------
// K = 128, N <= 128 (the q / k / v / merge projections: 8 of the 10 linears of a transformer layer): the whole pre-split W
// (4 chunks x 24 KiB = 96 KiB) stays RESIDENT in LDS for the lifetime of a persistent 8-wave workgroup, and the activations
// never touch LDS: a lane of v_mfma_f32_32x32x16_bf16 holds 8 consecutive channels of ONE token, which is what it gets from
// two 16-byte loads of a row-major [T][K] row.  Each wave walks 32-token tiles on its own (32 tokens x all 128 features:
// no wave needs another wave's tokens, so there is no barrier after the prologue); per chunk a lane loads its token's 64
// contiguous bytes (channels 16 hl .. 16 hl + 15: K step s of lane half hl is channels 16 hl + 8 s + j -- any bijection is
// a valid contraction order as long as W is read with the same one, group 2 hl + s), the next chunk's / tile's loads are
// in flight under the 48 MFMAs of the current chunk, and its split (VALU) is scheduled between them.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void linear_split_wres_kernel(const float *__restrict__ x, const uint4 *__restrict__ wp,
                                                                    const float *__restrict__ bias, float *__restrict__ out, long long T,
                                                                    int N, int act, int n_tiles) {
    constexpr int NC = 4;                          // K = 128
    __shared__ uint4 Ws[NC * kLsW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
#pragma unroll
    for (int j = 0; j < NC * kLsW / 512; ++j) Ws[tid + 512 * j] = wp[tid + 512 * j];
    const uint4 *wb = Ws + 2 * hl * 128 + nl;
    float bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bb[j] = (bias && 32 * j + nl < N) ? bias[32 * j + nl] : 0.f;
    __syncthreads();

    const int stride = gridDim.x * 8;
    int tile = blockIdx.x * 8 + wave;
    if (tile >= n_tiles) return;
    auto row_ptr = [&](int t) {
        const long long tr = (long long)t * 32 + nl;
        return reinterpret_cast<const float4 *>(x + (tr < T ? tr : T - 1) * 128 + 16 * hl);
    };
    float4 raw[2][4];                              // raw X of the chunk after the current one (ring of two)
    uint4 fa[2][2][3];                             // A fragments of the current / next chunk
    auto fetch = [&](const float4 *p, int c, float4 (&r)[4]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = p[8 * c + q];
    };
    auto split_x = [&](const float4 (&px)[4], uint4 (&f)[2][3]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float v[8] = {px[2 * s].x, px[2 * s].y, px[2 * s].z, px[2 * s].w, px[2 * s + 1].x, px[2 * s + 1].y, px[2 * s + 1].z, px[2 * s + 1].w};
            split3x8g(v, f[s][0], f[s][1], f[s][2]);
        }
    };
    const float4 *xp = row_ptr(tile);
    fetch(xp, 0, raw[0]);
    fetch(xp, 1, raw[1]);
    split_x(raw[0], fa[0]);
    for (; tile < n_tiles; tile += stride) {
        const int next = tile + stride;
        const float4 *xn = row_ptr(next < n_tiles ? next : tile);
        f32x16g acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // raw[(c + 1) & 1] holds chunk c+1 (of this tile, or chunk 0 of the next tile): split it under the MFMAs of chunk c;
            // raw[c & 1] is free: fetch chunk c+2 into it
            if (c + 2 < NC) fetch(xp, c + 2, raw[c & 1]);
            else fetch(xn, c + 2 - NC, raw[c & 1]);
            const uint4 *wc = wb + c * kLsW;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                uint4 b[4][3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j][p] = wc[(4 * p + s) * 128 + 32 * j];
#pragma unroll
                for (int j = 0; j < 4; ++j) mfma_split6(acc[j], fa[c & 1][s], b[j]);
            }
            split_x(raw[(c + 1) & 1], fa[(c + 1) & 1]);
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            }
        }
        // epilogue: lane = feature column, registers = token rows; 32 lanes store 128 contiguous bytes of a token
        const long long t0 = (long long)tile * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nf = 32 * j + nl;
            if (nf < N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long t = t0 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                    const float v = acc[j][r] + bb[j];
                    if (t < T) out[t * N + nf] = act == 6 ? gelu_as(v) : v;
                }
            }
        }
        xp = xn;
    }
}

constexpr int kSsRow(int C) { return 2 * C + 16; }   // bytes per LDS row of a split tile: 16-byte fragment reads are conflict free

// two workgroups per CU where the registers allow it without spilling (the 64-channel parallax attention: 238 VGPRs; the
// 128-channel instances need 256 + 127 and stay at one): 99.1 -> 98.1 ms on DCMCS3DI at 1080p
template <int C, int CV, bool MAP, bool SS>
__global__ __launch_bounds__(256, C == 64 ? 2 : 1) void attention_tokens_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                               const float *__restrict__ v, const int *__restrict__ region,
                                                               const int *__restrict__ rowmap, float *__restrict__ out,
                                                               float *__restrict__ stats, int L, float scale,
                                                               float *__restrict__ part, long long kv_shift = 0, long long kv_total = 0) {
    // kv_shift (rowmap launches only): the keys / values of token (b, i) live kv_shift rows further (mod kv_total) than its
    // query -- the cross attention of transformer.py:281-287 attends every image to the OTHER half of the batch, which the
    // reference materialises as torch.cat(chunk(2)[::-1]) after every layer
    constexpr int CH = C / 2;                       // channels per lane half
    constexpr int NVT = CV >= 32 ? CV / 32 : 1;     // 32-channel value tiles on the MFMA path
    constexpr int KLD = C + 4;                      // padded LDS rows: the 16-lane column reads (b128) are conflict free
    constexpr int VLD = CV >= 32 ? CV + 4 : 4;
    constexpr int KV4 = (32 * C / 4) / 256;         // float4 per thread of one K tile (4 / 2)
    constexpr int VV4 = CV >= 32 ? (32 * CV / 4) / 256 : 1;   // V tile (4 / 3), or the 16 float4 of a 2-channel tile
    // The 4 waves of a workgroup attend 4 x 32 queries of the SAME batch item to the same keys: every 32-key tile
    // of K and V is fetched once per workgroup with 16-byte loads (next tile in flight in registers while the current
    // one is multiplied), staged in LDS, and read from there as MFMA operands.
    constexpr int SROW = kSsRow(C);                 // SS: K tile as [piece][key][C] bf16, rows padded to SROW bytes
    __shared__ __attribute__((aligned(16))) float Ks[SS ? 3 * 32 * SROW / 4 : 32 * KLD];
    // SS with an MFMA value path: V as [piece][key][CV] bf16 too, rows of VROWB bytes with (VROWB / 4) % 64 == 16 or 48, so the
    // four rows a transposed read gathers lie in four disjoint 16-bank windows (cdna_hip_programming.md T10)
    constexpr bool PVS = SS && CV >= 32;
    constexpr int VROWB = (CV * 2) % 256 == 0 ? CV * 2 + 64 : CV * 2;
    static_assert(!PVS || ((VROWB / 4) % 64 == 16 || (VROWB / 4) % 64 == 48), "V image rows must not share banks");
    __shared__ __attribute__((aligned(16))) float Vs[PVS ? 3 * 32 * VROWB / 4 : 32 * VLD];
    __shared__ int Rs[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const int b = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    const size_t tb = (size_t)b * L;
    const int qi = q0 + nl;
    const bool qlive = qi < L;
    const int qclamp = qlive ? qi : L - 1;
    // rowmap (optional): token (b, i) of this launch lives in row rowmap[b*L + i] of q / k / v / out -- the shifted-window
    // partition of attention.py:60-92 as an index table instead of roll + permute copies
    auto row = [&](int i) -> size_t {
        if constexpr (MAP) return (size_t)rowmap[tb + i];
        else return tb + i;
    };
    auto kvrow = [&](int i) -> size_t {
        size_t r = row(i);
        if constexpr (MAP) {
            r += (size_t)kv_shift;
            if (r >= (size_t)kv_total && kv_total > 0) r -= (size_t)kv_total;
        }
        return r;
    };
    // B operand of S^T = K Q^T : lane (query nl, half hl) holds q[query][hl*C/2 + p], p < C/2
    float qb[SS ? 1 : CH];
    uint4 qf[SS ? C / 16 : 1][3];      // SS: B fragments, K step s = channels 16 s + 8 hl + j
    const float qs = scale * kLog2e;   // scores live in the log2 domain: softmax through v_exp_f32 (2^x) directly
    if constexpr (SS) {
        const float *qp = q + row(qclamp) * C + 8 * hl;
#pragma unroll
        for (int st = 0; st < C / 16; ++st) {
            const float4 t0 = *reinterpret_cast<const float4 *>(qp + 16 * st), t1 = *reinterpret_cast<const float4 *>(qp + 16 * st + 4);
            const float x[8] = {t0.x * qs, t0.y * qs, t0.z * qs, t0.w * qs, t1.x * qs, t1.y * qs, t1.z * qs, t1.w * qs};
            split3x8g(x, qf[st][0], qf[st][1], qf[st][2]);
        }
    } else {
        const float *qp = q + row(qclamp) * C + hl * CH;
#pragma unroll
        for (int i = 0; i < CH / 4; ++i) {
            const float4 t = *reinterpret_cast<const float4 *>(qp + 4 * i);
            qb[4 * i] = t.x * qs; qb[4 * i + 1] = t.y * qs; qb[4 * i + 2] = t.z * qs; qb[4 * i + 3] = t.w * qs;
        }
    }
    const int qreg = region ? region[tb + qclamp] : 0;

    float4 kpre[KV4], vpre[VV4];
    int rpre = 0;
    // rows of the keys this thread stages, looked up ONE TILE AHEAD of the data loads that use them (a dependent
    // index -> data load pair inside fetch() would double the latency the MFMAs of one tile have to hide)
    size_t krow[KV4], vrow[CV >= 32 ? VV4 : 1];
    auto fetch_rows = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KV4; ++i) {
            const int key = (tid + i * 256) / (C / 4);
            krow[i] = kvrow(j0 + key < L ? j0 + key : L - 1);
        }
        if constexpr (CV >= 32) {
#pragma unroll
            for (int i = 0; i < VV4; ++i) {
                const int key = (tid + i * 256) / (CV / 4);
                vrow[i] = kvrow(j0 + key < L ? j0 + key : L - 1);
            }
        }
    };
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KV4; ++i) {
            const int f = tid + i * 256, key = f / (C / 4), c4 = f - key * (C / 4);
            kpre[i] = (j0 + key < L) ? *reinterpret_cast<const float4 *>(k + krow[i] * C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (CV >= 32) {
#pragma unroll
            for (int i = 0; i < VV4; ++i) {
                const int f = tid + i * 256, key = f / (CV / 4), c4 = f - key * (CV / 4);
                vpre[i] = (j0 + key < L) ? *reinterpret_cast<const float4 *>(v + vrow[i] * CV + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if constexpr (CV == 2) {
            if (tid < 16) {   // 32 keys x 2 channels = 16 float4
                const int key = 2 * tid;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (MAP) {
                    if (j0 + key < L) { const float2 u = *reinterpret_cast<const float2 *>(v + kvrow(j0 + key) * 2); t.x = u.x; t.y = u.y; }
                    if (j0 + key + 1 < L) { const float2 u = *reinterpret_cast<const float2 *>(v + kvrow(j0 + key + 1) * 2); t.z = u.x; t.w = u.y; }
                } else if (j0 + key + 1 < L) t = *reinterpret_cast<const float4 *>(v + (tb + j0 + key) * 2);
                else if (j0 + key < L) { const float2 u = *reinterpret_cast<const float2 *>(v + (tb + j0 + key) * 2); t.x = u.x; t.y = u.y; }
                vpre[0] = t;
            }
        }
        if (region && tid < 32) rpre = (j0 + tid < L) ? region[tb + j0 + tid] : 0;
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < KV4; ++i) {
            const int f = tid + i * 256, key = f / (C / 4), c4 = f - key * (C / 4);
            if constexpr (SS) {
                unsigned int h0, m0, l0, h1, m1, l1;
                split3x2g(kpre[i].x, kpre[i].y, h0, m0, l0);
                split3x2g(kpre[i].z, kpre[i].w, h1, m1, l1);
                unsigned char *kd = reinterpret_cast<unsigned char *>(Ks) + key * SROW + 8 * c4;
                *reinterpret_cast<uint2 *>(kd) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(kd + 32 * SROW) = make_uint2(m0, m1);
                *reinterpret_cast<uint2 *>(kd + 64 * SROW) = make_uint2(l0, l1);
            } else {
                *reinterpret_cast<float4 *>(Ks + key * KLD + 4 * c4) = kpre[i];
            }
        }
        if constexpr (PVS) {
#pragma unroll
            for (int i = 0; i < VV4; ++i) {
                const int f = tid + i * 256, key = f / (CV / 4), c4 = f - key * (CV / 4);
                unsigned int h0, m0, l0, h1, m1, l1;
                split3x2g(vpre[i].x, vpre[i].y, h0, m0, l0);
                split3x2g(vpre[i].z, vpre[i].w, h1, m1, l1);
                unsigned char *vd = reinterpret_cast<unsigned char *>(Vs) + key * VROWB + 8 * c4;
                *reinterpret_cast<uint2 *>(vd) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(vd + 32 * VROWB) = make_uint2(m0, m1);
                *reinterpret_cast<uint2 *>(vd + 64 * VROWB) = make_uint2(l0, l1);
            }
        } else if constexpr (CV >= 32) {
#pragma unroll
            for (int i = 0; i < VV4; ++i) {
                const int f = tid + i * 256, key = f / (CV / 4), c4 = f - key * (CV / 4);
                *reinterpret_cast<float4 *>(Vs + key * VLD + 4 * c4) = vpre[i];
            }
        } else if constexpr (CV == 2) {
            if (tid < 16) *reinterpret_cast<float4 *>(Vs + 4 * tid) = vpre[0];   // Vs[key*2 + ch]
        }
        if (region && tid < 32) Rs[tid] = rpre;
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16g o[NVT];
    float o2x = 0.f, o2y = 0.f;
    if constexpr (CV >= 32) {
#pragma unroll
        for (int j = 0; j < NVT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    }
    // key split (gridDim.z > 1): this workgroup attends its queries to keys [jb, je) only and leaves the unnormalised
    // partial (o, max, sum) in `part`; attention_combine_kernel merges the splits.  Fills the chip when batch*len/128
    // workgroups would not (global matching: 56 of them on 256 CUs).
    const int nsplit = gridDim.z, split = blockIdx.z;
    const int kchunk = ((L + nsplit - 1) / nsplit + 31) & ~31;
    const int jb = split * kchunk, je = (jb + kchunk < L) ? jb + kchunk : L;
    fetch_rows(jb);
    fetch(jb);
    fetch_rows(jb + 32);
    stage();
    __syncthreads();
    for (int j0 = jb; j0 < je; j0 += 32) {
        const bool more = j0 + 32 < je;
        if (more) {
            fetch(j0 + 32);
            fetch_rows(j0 + 64);
        }
        // ---- S^T tile: A = K rows (key nl) from LDS, B = Q.  LDS operand reads run one group ahead of the MFMAs that
        //      consume them (a single wave per SIMD lives here: nothing else would hide the LDS latency) ----
        f32x16g s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        if constexpr (SS) {
            const unsigned char *kp = reinterpret_cast<const unsigned char *>(Ks) + nl * SROW + 16 * hl;
            uint4 ac[3], an[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) ac[p] = *reinterpret_cast<const uint4 *>(kp + p * 32 * SROW);
#pragma unroll
            for (int st = 0; st < C / 16; ++st) {
                if (st + 1 < C / 16) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) an[p] = *reinterpret_cast<const uint4 *>(kp + p * 32 * SROW + 32 * (st + 1));
                }
                mfma_split6(s, ac, qf[st]);
#pragma unroll
                for (int p = 0; p < 3; ++p) ac[p] = an[p];
            }
        } else {
        const float *kp = Ks + nl * KLD + hl * CH;
        float4 tc = *reinterpret_cast<const float4 *>(kp), tn = tc;
#pragma unroll
        for (int i = 0; i < CH / 4; ++i) {
            if (i + 1 < CH / 4) tn = *reinterpret_cast<const float4 *>(kp + 4 * (i + 1));
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(tc.x, qb[4 * i], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(tc.y, qb[4 * i + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(tc.z, qb[4 * i + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(tc.w, qb[4 * i + 3], s, 0, 0, 0);
            tc = tn;
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // the LDS read of group i+1
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // the MFMAs of group i
        }
        }
        // lane: query nl; s[r] = log2-domain score of key j0 + (r&3)+8(r>>2)+4hl (q carries scale * log2(e))
        if (region) {                                    // shifted-window mask, one uniform branch per tile
            int rk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rk[r] = Rs[(r & 3) + 8 * (r >> 2) + 4 * hl];
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += (rk[r] != qreg) ? -100.0f * kLog2e : 0.0f;
        }
        if (j0 + 32 > L) {                               // ragged last tile
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (j0 + (r & 3) + 8 * (r >> 2) + 4 * hl < L) ? s[r] : -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          // the other half of the keys of this query
        const float m_new = fmaxf(m_run, mx);            // finite: key j0 exists and a mask only subtracts 100
        const float corr = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(s[r] - m_new);
            s[r] = p;
            psum += p;
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * corr + psum;
        m_run = m_new;
        if constexpr (PVS) {
#pragma unroll
            for (int j = 0; j < NVT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[j][r] *= corr;
            // O^T[c][query] += sum_key V[key][c] P[key][query] on the bf16 pipe (three pieces each, six MFMAs per product).
            // B = P: the score tile already has its column (query) on the lane and its rows (keys) in the registers, so the
            // registers 8t .. 8t+7 of a lane ARE its B fragment of K step t (keys 16t + 8(j>>2) + 4hl + (j&3), j = 0..7) --
            // no lane movement.  A = V^T: the same 8 keys of channel nl, i.e. two 4-key column gathers from the row-major
            // [key][channel] image: ds_read_b64_tr_b16 (lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3, and
            // receives its own column of the four rows).
            typedef short s16x4g __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4g *lds_s16x4;
            const unsigned char *vb = reinterpret_cast<const unsigned char *>(Vs) + (4 * hl + ((lane & 15) >> 2)) * VROWB +
                                      (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float x[8] = {s[8 * t], s[8 * t + 1], s[8 * t + 2], s[8 * t + 3], s[8 * t + 4], s[8 * t + 5], s[8 * t + 6], s[8 * t + 7]};
                uint4 pf[3];
                split3x8g(x, pf[0], pf[1], pf[2]);
#pragma unroll
                for (int j = 0; j < NVT; ++j) {
                    uint4 vf[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const unsigned char *a = vb + (p * 32 + 16 * t) * VROWB + 64 * j;
                        const s16x4g lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a));
                        const s16x4g hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 8 * VROWB));
                        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                        vf[p] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                    }
                    mfma_split6(o[j], vf, pf);
                }
            }
        } else if constexpr (CV >= 32) {
#pragma unroll
            for (int j = 0; j < NVT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[j][r] *= corr;
            // O^T[c][query] += sum_key V[key][c] P[key][query]: k-step r pairs the keys held by the two lane halves
            // in register r (keys kk and kk+4); A = V[key][channel nl of each 32-channel tile] from LDS
            float vc[NVT], vn[NVT];
#pragma unroll
            for (int j = 0; j < NVT; ++j) vc[j] = Vs[(4 * hl) * VLD + nl + 32 * j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + 1 < 16) {
                    const float *vp = Vs + (((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * hl) * VLD + nl;
#pragma unroll
                    for (int j = 0; j < NVT; ++j) vn[j] = vp[32 * j];
                }
#pragma unroll
                for (int j = 0; j < NVT; ++j) o[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(vc[j], s[r], o[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NVT; ++j) vc[j] = vn[j];
                __builtin_amdgcn_sched_group_barrier(0x100, (NVT + 1) / 2, 0);   // ds_read2_b32 pairs of step r+1
                __builtin_amdgcn_sched_group_barrier(0x008, NVT, 0);             // the MFMAs of step r
            }
        } else if constexpr (CV == 2) {
            float ax = 0.f, ay = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 vv = *reinterpret_cast<const float2 *>(Vs + ((r & 3) + 8 * (r >> 2) + 4 * hl) * 2);
                ax += s[r] * vv.x;
                ay += s[r] * vv.y;
            }
            ax += __shfl_xor(ax, 32, 64);
            ay += __shfl_xor(ay, 32, 64);
            o2x = o2x * corr + ax;
            o2y = o2y * corr + ay;
        }
        __syncthreads();                 // every wave is done with this tile
        if (more) {
            stage();
            __syncthreads();
        }
    }
    if (nsplit > 1) {
        // part: [split][batch][len][CV + 2]
        float *pp = part + (((size_t)split * gridDim.y + b) * L + qclamp) * (CV + 2);
        if (qlive) {
            if constexpr (CV >= 32) {
#pragma unroll
                for (int j = 0; j < NVT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pp[j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl] = o[j][r];
            } else if (CV == 2 && hl == 0) {
                pp[0] = o2x; pp[1] = o2y;
            }
            if (hl == 0) { pp[CV] = m_run; pp[CV + 1] = l_run; }
        }
        return;
    }
    const float inv = 1.0f / l_run;
    if (stats && qlive && hl == 0) {     // row statistics of the softmax (max, sum): used by the column-sum pass
        stats[(tb + qi) * 2] = m_run * kLn2;   // natural-log units for attention_colsum_kernel
        stats[(tb + qi) * 2 + 1] = l_run;
    }
    if constexpr (CV >= 32) {
        // O^T[c][query]: lane = query nl, registers = channels (r&3)+8(r>>2)+4hl of tile j
        if (qlive) {
            float *op = out + row(qi) * CV;
#pragma unroll
            for (int j = 0; j < NVT; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 4)   // registers r..r+3 are four consecutive channels
                    *reinterpret_cast<float4 *>(op + j * 32 + 8 * (r >> 2) + 4 * hl) =
                        make_float4(o[j][r] * inv, o[j][r + 1] * inv, o[j][r + 2] * inv, o[j][r + 3] * inv);
        }
    } else if constexpr (CV == 2) {
        if (qlive && hl == 0) *reinterpret_cast<float2 *>(out + row(qi) * 2) = make_float2(o2x * inv, o2y * inv);
    }
}

// Column sums of a row-softmax, given its row statistics: colsum[b][j] = sum_i exp(scale q_i.k_j - m_i) / l_i
// (pasmnet/utils.py:31,34: att_left2right.sum(dim=-2)).  One wave per 32 keys; the score tile is computed in the
// natural orientation (queries on the MFMA rows = registers, the key on the lane), so the sum over queries is a
// sum over registers plus one cross-half shuffle, in a fixed order (deterministic).
template <int C>
__global__ __launch_bounds__(256) void attention_colsum_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                               const float *__restrict__ stats, float *__restrict__ colsum, int L,
                                                               float scale) {
    constexpr int QV4 = (32 * C / 4) / 256;         // float4 per thread of one query tile
    // The 4 waves of a workgroup hold 4 x 32 keys of the SAME row and sweep the same queries: every 32-query tile of Q
    // and its (max, 1/sum) statistics are fetched once per workgroup (next tile in flight in registers), staged in LDS
    // and read from there as the MFMA A operand.  Scores live in the log2 domain: p = exp2(s' - m') * (1/l).
    constexpr int SROW = kSsRow(C);                 // Q tile as [piece][query][C] bf16 (split-bf16 scores, see mfma_split6)
    __shared__ __attribute__((aligned(16))) unsigned char Qs[3 * 32 * SROW];
    __shared__ float2 Ms[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const int b = blockIdx.y;
    const int j0 = (blockIdx.x * 4 + wave) * 32;
    const size_t tb = (size_t)b * L;
    const int kj = j0 + nl;
    uint4 kf[C / 16][3];                            // B fragments: this lane's key row (a workgroup's surplus waves clamp)
    {
        const float qs = scale * kLog2e;
        const float *kp = k + (tb + (kj < L ? kj : L - 1)) * C + 8 * hl;
#pragma unroll
        for (int st = 0; st < C / 16; ++st) {
            const float4 t0 = *reinterpret_cast<const float4 *>(kp + 16 * st), t1 = *reinterpret_cast<const float4 *>(kp + 16 * st + 4);
            const float x[8] = {t0.x * qs, t0.y * qs, t0.z * qs, t0.w * qs, t1.x * qs, t1.y * qs, t1.z * qs, t1.w * qs};
            split3x8g(x, kf[st][0], kf[st][1], kf[st][2]);
        }
    }
    float4 qpre[QV4];
    float2 mpre = make_float2(0.f, 0.f);
    auto fetch = [&](int i0) {
#pragma unroll
        for (int i = 0; i < QV4; ++i) {
            const int f = tid + i * 256, qq = f / (C / 4), c4 = f - qq * (C / 4);
            qpre[i] = (i0 + qq < L) ? *reinterpret_cast<const float4 *>(q + (tb + i0 + qq) * C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < 32) {
            // an out-of-range query contributes exp2(0 - inf) * 0 = 0
            mpre = (i0 + tid < L) ? *reinterpret_cast<const float2 *>(stats + (tb + i0 + tid) * 2) : make_float2(INFINITY, INFINITY);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < QV4; ++i) {
            const int f = tid + i * 256, qq = f / (C / 4), c4 = f - qq * (C / 4);
            unsigned int h0, m0, l0, h1, m1, l1;
            split3x2g(qpre[i].x, qpre[i].y, h0, m0, l0);
            split3x2g(qpre[i].z, qpre[i].w, h1, m1, l1);
            unsigned char *qd = Qs + qq * SROW + 8 * c4;
            *reinterpret_cast<uint2 *>(qd) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(qd + 32 * SROW) = make_uint2(m0, m1);
            *reinterpret_cast<uint2 *>(qd + 64 * SROW) = make_uint2(l0, l1);
        }
        if (tid < 32) Ms[tid] = make_float2(mpre.x * kLog2e, 1.0f / mpre.y);   // (max in log2 units, 1 / sum)
    };
    float acc = 0.f;
    fetch(0);
    stage();
    __syncthreads();
    for (int i0 = 0; i0 < L; i0 += 32) {
        const bool more = i0 + 32 < L;
        if (more) fetch(i0 + 32);
        f32x16g s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const unsigned char *qp = Qs + nl * SROW + 16 * hl;
        uint4 ac[3], an[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) ac[p] = *reinterpret_cast<const uint4 *>(qp + p * 32 * SROW);
#pragma unroll
        for (int st = 0; st < C / 16; ++st) {
            if (st + 1 < C / 16) {
#pragma unroll
                for (int p = 0; p < 3; ++p) an[p] = *reinterpret_cast<const uint4 *>(qp + p * 32 * SROW + 32 * (st + 1));
            }
            mfma_split6(s, ac, kf[st]);
#pragma unroll
            for (int p = 0; p < 3; ++p) ac[p] = an[p];
        }
        // lane: key nl; s[r] = log2-domain score of query i0 + (r&3)+8(r>>2)+4hl
        float2 ml[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ml[r] = Ms[(r & 3) + 8 * (r >> 2) + 4 * hl];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc = fmaf(__builtin_amdgcn_exp2f(s[r] - ml[r].x), ml[r].y, acc);
        __syncthreads();                 // every wave is done with this tile
        if (more) {
            stage();
            __syncthreads();
        }
    }
    acc += __shfl_xor(acc, 32, 64);
    if (hl == 0 && kj < L) colsum[tb + kj] = acc;
}

// =================================================================================================
// Local correlation kernels on channels-last features f0, f1: [B][H*W][128]
// =================================================================================================
// matching.py:42-86: softmax over the (2R+1)^2 integer neighbours (out-of-image taps masked to -1e9), expected
// coordinate minus own coordinate -> flow [B][2][H][W].  One wave per pixel, lanes over taps.
__global__ __launch_bounds__(256) void local_corr_softmax_kernel(const float *__restrict__ f0, const float *__restrict__ f1,
                                                                 float *__restrict__ flow, int H, int W, int R, float scale) {
    constexpr int C = 128;
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (pix >= (long long)H * W) return;
    const int y = (int)(pix / W), x = (int)(pix % W);
    const int D = 2 * R + 1, NT = D * D;
    const float *a = f0 + ((size_t)b * H * W + pix) * C;
    float mx = -INFINITY;
    float sc[2];
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int t = lane + rep * 64;
        float s = -INFINITY;
        if (t < NT) {
            const int dy = t / D - R, dx = t % D - R;
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const float *bp = f1 + ((size_t)b * H * W + (size_t)yy * W + xx) * C;
                float acc = 0.f;
                for (int c = 0; c < C; c += 4) {
                    const float4 u = *reinterpret_cast<const float4 *>(a + c), w4 = *reinterpret_cast<const float4 *>(bp + c);
                    acc += u.x * w4.x + u.y * w4.y + u.z * w4.z + u.w * w4.w;
                }
                s = acc * scale;
            } else {
                s = -1e9f;                                  // matching.py:76
            }
        }
        sc[rep] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float sum = 0.f, ex = 0.f, ey = 0.f;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int t = lane + rep * 64;
        if (t < NT) {
            const float p = expf(sc[rep] - mx);
            sum += p;
            ex += p * (float)(x + t % D - R);
            ey += p * (float)(y + t / D - R);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        ex += __shfl_xor(ex, off, 64);
        ey += __shfl_xor(ey, off, 64);
    }
    if (lane == 0) {
        flow[((size_t)b * 2 + 0) * H * W + pix] = ex / sum - (float)x;
        flow[((size_t)b * 2 + 1) * H * W + pix] = ey / sum - (float)y;
    }
}

// matching.py:89-126: corr[b][t][y][x] = f0(y,x) . bilinear(f1, (x,y) + window[t] + flow) / sqrt(C), zeros padding,
// align_corners=True (grid_sample of exactly representable pixel coordinates).  One wave per pixel, lanes over taps.
// All (2R+1)^2 window taps of a pixel sample feature1 at integer offsets from ONE point (x + flow), so they share the
// bilinear weights: corr(dx,dy) = w00 D(dx,dy) + w01 D(dx+1,dy) + w10 D(dx,dy+1) + w11 D(dx+1,dy+1) with the (2R+2)^2
// integer-offset dots D(i,j) = <f0[pix], f1[y0-R+j][x0-R+i]> (zero outside the image = grid_sample's zero padding).
// That is 100 dots per pixel instead of 4 x 81.  One wave per pixel at a time: eight lanes per window position, each with 16
// channels of the position's 512-byte vector (whole cache lines per load instruction) against its part of the pixel's f0 vector.  A workgroup covers 32
// consecutive pixels (4 waves x 8) and writes each of the 81 correlation planes as one 128-byte segment.
constexpr int kLcfPix = 32;    // pixels per workgroup

__global__ __launch_bounds__(256) void local_corr_flow_kernel(const float *__restrict__ f0, const float *__restrict__ f1,
                                                              const float *__restrict__ flow, float *__restrict__ corr, int H,
                                                              int W, int R, float scale) {
    constexpr int C = 128;
    __shared__ __attribute__((aligned(16))) float a_s[4][C];   // f0 vector of the pixel a wave is working on
    __shared__ float d_s[4][128];                               // its integer-offset dots, (2R+2)^2 <= 100
    __shared__ float o_s[81][kLcfPix + 1];                      // results [tap][pixel]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int D = 2 * R + 1, NT = D * D, DP = D + 1, NP = DP * DP;
    const size_t hw = (size_t)H * W;
    const long long pix0 = (long long)blockIdx.x * kLcfPix;
    const float cx = (W - 1) * 0.5f, cy = (H - 1) * 0.5f;
    for (int i = 0; i < kLcfPix / 4; ++i) {
        const int pl = wave * (kLcfPix / 4) + i;
        const long long pix = pix0 + pl;
        if (pix >= (long long)hw) break;                        // wave-uniform
        const int y = (int)(pix / W), x = (int)(pix % W);
        const float fx = flow[((size_t)b * 2 + 0) * hw + pix], fy = flow[((size_t)b * 2 + 1) * hw + pix];
        // the reference normalises to [-1,1] and grid_sample maps back: ((g + 1) / 2) * (size - 1)
        const float gx = (((float)x + fx) - cx) / cx, gy = (((float)y + fy) - cy) / cy;
        const float px = ((gx + 1.0f) * 0.5f) * (float)(W - 1), py = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
        const float x0f = floorf(px), y0f = floorf(py);
        const float wx1 = px - x0f, wy1 = py - y0f, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        // clamp far-out-of-range bases (NaN/inf flows included) so that the int arithmetic below cannot overflow; every
        // position is then outside the image and the row is all zeros, like grid_sample's
        const int x0 = (x0f > -1e6f && x0f < 1e6f) ? (int)x0f : -1000000, y0 = (y0f > -1e6f && y0f < 1e6f) ? (int)y0f : -1000000;
        *reinterpret_cast<float2 *>(&a_s[wave][2 * lane]) = *reinterpret_cast<const float2 *>(f0 + ((size_t)b * hw + pix) * C + 2 * lane);
        __builtin_amdgcn_wave_barrier();
        // eight lanes per window position, 16 channels each: a load instruction reads eight whole 128-byte lines (one lane per
        // position would touch 64 lines per instruction, a quarter of each)
        {
            const int ps = lane >> 3, oc = lane & 7;
            float4 u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) u[i] = *reinterpret_cast<const float4 *>(&a_s[wave][16 * oc + 4 * i]);
            for (int p0 = 0; p0 < NP; p0 += 8) {
                const int p = p0 + ps;
                const int jj = p / DP, ii = p - jj * DP;
                const int xx = x0 - R + ii, yy = y0 - R + jj;
                float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
                if (p < NP && xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    const float *bp = f1 + ((size_t)b * hw + (size_t)yy * W + xx) * C + 16 * oc;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 w4 = *reinterpret_cast<const float4 *>(bp + 4 * i);
                        d0 = fmaf(u[i].x, w4.x, d0); d1 = fmaf(u[i].y, w4.y, d1); d2 = fmaf(u[i].z, w4.z, d2); d3 = fmaf(u[i].w, w4.w, d3);
                    }
                }
                float d = (d0 + d1) + (d2 + d3);
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                if (oc == 0 && p < NP) d_s[wave][p] = d;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < NT; t += 64) {
            const int dy = t / D, dx = t - dy * D;
            const float *dp = &d_s[wave][dy * DP + dx];
            const float v = (wx0 * wy0) * dp[0] + (wx1 * wy0) * dp[1] + (wx0 * wy1) * dp[DP] + (wx1 * wy1) * dp[DP + 1];
            o_s[t][pl] = v * scale;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NT * kLcfPix; e += 256) {
        const int t = e / kLcfPix, pl = e - t * kLcfPix;
        const long long pix = pix0 + pl;
        if (pix < (long long)hw) corr[((size_t)b * NT + t) * hw + pix] = o_s[t][pl];
    }
}

// The same correlation for a TILE of 4 x 8 pixels at once (round 4).  Neighbouring pixels of a smooth flow field sample almost the
// same neighbourhood of feature1: the union of the tile's (2R+2)^2 windows, clipped to the image, is a box of P positions (13 x 17
// = 221 for R = 4 and a constant flow) against 32 x 100 position reads of the per-pixel form.  The box is staged in LDS once per
// 32-channel chunk (float32 as it is in memory: rows of 32 floats padded to 36, so that the MFMA operand reads are conflict-free
// 16-byte reads), and D[position][pixel] = <f1[position], f0[pixel]> is ONE small GEMM, P x 32 x 128, on v_mfma_f32_32x32x2_f32:
// exact float32 products, float32 accumulation, no operand conversion, no scales (a lane of k-half h takes channels 16 h + s of
// the chunk at step s -- both operands alike, a dot product does not care about the order).  That pipe has 1/16 of the 16-bit
// rate and is still 10x what this problem needs (57 k pixels x 221 x 128 MACs = 24 us chip-wide).  D goes back to LDS ([position]
// [33]), and every (pixel, tap) combines its four integer-offset dots with the pixel's bilinear weights exactly as the per-pixel
// form does (positions outside the image: zero, grid_sample's padding).  A tile whose box exceeds kLcMaxP positions (a flow
// discontinuity inside the tile, NaN / huge flows) runs the per-pixel form, one wave per pixel, inside the same launch.
constexpr int kLcTY = 4, kLcTX = 8, kLcTile = kLcTY * kLcTX, kLcMaxP = 320, kLcRow = 36, kLcChunk = 32;

__device__ __forceinline__ void lcf_window(const float *__restrict__ flow, int b, size_t hw, long long pix, int x, int y, int H, int W,
                                           int &x0, int &y0, float &wx1, float &wy1) {
    const float cx = (W - 1) * 0.5f, cy = (H - 1) * 0.5f;
    const float fx = flow[((size_t)b * 2 + 0) * hw + pix], fy = flow[((size_t)b * 2 + 1) * hw + pix];
    // the reference normalises to [-1,1] and grid_sample maps back: ((g + 1) / 2) * (size - 1)
    const float gx = (((float)x + fx) - cx) / cx, gy = (((float)y + fy) - cy) / cy;
    const float px = ((gx + 1.0f) * 0.5f) * (float)(W - 1), py = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
    const float x0f = floorf(px), y0f = floorf(py);
    wx1 = px - x0f; wy1 = py - y0f;
    // far-out-of-range bases (NaN / inf flows included) are clamped so that the int arithmetic cannot overflow; every position
    // is then outside the image and the row is all zeros, like grid_sample's
    x0 = (x0f > -1e6f && x0f < 1e6f) ? (int)x0f : -1000000;
    y0 = (y0f > -1e6f && y0f < 1e6f) ? (int)y0f : -1000000;
}

// D[position][pixel] = <f1[box position], f0[tile pixel]> for the P <= kLcMaxP positions of the box (bx0, by0, width BW; inside the
// image) and the 4 x 8 pixels of the tile at (tx0, ty0); result in fa as [position][33].  All 256 threads; ends with a barrier.
__device__ __forceinline__ void lc_box_gemm(const float *__restrict__ f0, const float *__restrict__ f1, int b, size_t hw, int H, int W, int ty0,
                                            int tx0, int bx0, int by0, int BW, int P, float *fa, float *fb) {
    constexpr int C = 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = (P + 31) >> 5;
    f32x16g acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const int m = lane & 31, kh = lane >> 5;
    for (int c0 = 0; c0 < C; c0 += kLcChunk) {
        // stage the chunk: 8 float4 per position (all loads first, then the LDS writes), one float4 per thread of the pixels
        float4 st[(kLcMaxP * 8 + 255) / 256];
#pragma unroll
        for (int k = 0; k < (kLcMaxP * 8 + 255) / 256; ++k) {
            const int idx = tid + k * 256, pos = idx >> 3, q = idx & 7;
            st[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pos < P) {
                const int j = pos / BW, i = pos - j * BW;
                st[k] = *reinterpret_cast<const float4 *>(f1 + ((size_t)b * hw + (size_t)(by0 + j) * W + (bx0 + i)) * C + c0 + 4 * q);
            }
        }
        {
            const int pl = tid >> 3, q = tid & 7;
            const int y = ty0 + pl / kLcTX, x = tx0 + pl % kLcTX;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < H && x < W) v = *reinterpret_cast<const float4 *>(f0 + ((size_t)b * hw + (size_t)y * W + x) * C + c0 + 4 * q);
            *reinterpret_cast<float4 *>(&fb[pl * kLcRow + 4 * q]) = v;
        }
#pragma unroll
        for (int k = 0; k < (kLcMaxP * 8 + 255) / 256; ++k) {
            const int idx = tid + k * 256, pos = idx >> 3, q = idx & 7;
            if (pos < P) *reinterpret_cast<float4 *>(&fa[pos * kLcRow + 4 * q]) = st[k];
        }
        __syncthreads();
        float4 bv[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) bv[s4] = *reinterpret_cast<const float4 *>(&fb[m * kLcRow + 16 * kh + 4 * s4]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int mb = wave + 4 * k;
            if (mb < nblk) {                                                  // wave-uniform
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    // rows >= P of the last block hold stale LDS: their products stay in their own (never read) rows of D
                    const float4 av = *reinterpret_cast<const float4 *>(&fa[(mb * 32 + m) * kLcRow + 16 * kh + 4 * s4]);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[s4].x, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[s4].y, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[s4].z, acc[k], 0, 0, 0);
                    acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[s4].w, acc[k], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                                      // the chunk is consumed
    }
    // D[position][pixel] -> LDS (over the chunk buffer): lane holds pixel n = lane % 32, rows (r & 3) + 8 (r >> 2) + 4 (lane / 32)
    float *dl = fa;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int mb = wave + 4 * k;
        if (mb < nblk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dl[(mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 33 + m] = acc[k][r];
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void local_corr_flow_tile_kernel(const float *__restrict__ f0, const float *__restrict__ f1,
                                                                   const float *__restrict__ flow, float *__restrict__ corr, int H,
                                                                   int W, int R, float scale, int tiles_x) {
    constexpr int C = 128;
    __shared__ __attribute__((aligned(16))) float fa[kLcMaxP * kLcRow];      // the box's 32-channel chunk; later D [position][33]
    __shared__ __attribute__((aligned(16))) float fb[kLcTile * kLcRow];      // the tile's pixels, same chunk
    __shared__ int sx0[kLcTile], sy0[kLcTile], slive[kLcTile];
    __shared__ float swx[kLcTile], swy[kLcTile];
    __shared__ int sbox[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int ty0 = ((int)blockIdx.x / tiles_x) * kLcTY, tx0 = ((int)blockIdx.x % tiles_x) * kLcTX;
    const int D = 2 * R + 1, NT = D * D, DP = D + 1;
    const size_t hw = (size_t)H * W;
    if (tid < kLcTile) {
        const int y = ty0 + tid / kLcTX, x = tx0 + tid % kLcTX;
        int x0 = 0, y0 = 0, live = 0;
        float wx1 = 0.f, wy1 = 0.f;
        if (y < H && x < W) {
            lcf_window(flow, b, hw, (long long)y * W + x, x, y, H, W, x0, y0, wx1, wy1);
            // live: the window [x0 - R, x0 + R + 1] x [y0 - R, y0 + R + 1] meets the image (otherwise the row is all zeros)
            live = (x0 + R + 1 >= 0 && x0 - R < W && y0 + R + 1 >= 0 && y0 - R < H) ? 1 : 0;
        }
        sx0[tid] = x0; sy0[tid] = y0; swx[tid] = wx1; swy[tid] = wy1; slive[tid] = live;
    }
    __syncthreads();
    if (tid == 0) {
        int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -0x7fffffff, by1 = -0x7fffffff;
        for (int i = 0; i < kLcTile; ++i)
            if (slive[i]) {
                bx0 = min(bx0, sx0[i] - R); bx1 = max(bx1, sx0[i] + R + 1);
                by0 = min(by0, sy0[i] - R); by1 = max(by1, sy0[i] + R + 1);
            }
        const bool some = bx1 >= bx0;                                         // a live pixel exists (its window meets the image)
        sbox[0] = some ? max(bx0, 0) : 0; sbox[1] = some ? max(by0, 0) : 0;
        sbox[2] = some ? min(bx1, W - 1) - max(bx0, 0) + 1 : 0;              // 0: nothing to stage, every output of the tile is zero
        sbox[3] = some ? min(by1, H - 1) - max(by0, 0) + 1 : 0;
    }
    __syncthreads();
    const int bx0 = sbox[0], by0 = sbox[1], BW = sbox[2], BH = sbox[3];
    const bool any = BW > 0 && BH > 0;
    const long long P64 = any ? (long long)BW * BH : 0;
    if (P64 > kLcMaxP) {
        // ---- per-pixel form for this tile: one wave per pixel, eight lanes per position (local_corr_flow_kernel) ----
        float *a_s = fa + wave * 256, *d_s = a_s + 128, *o_s = fa + 1024;      // o_s [tap][33]: the tile's results, stored by plane below
        for (int i = 0; i < kLcTile / 4; ++i) {
            const int pl = wave * (kLcTile / 4) + i;
            const int y = ty0 + pl / kLcTX, x = tx0 + pl % kLcTX;
            if (y >= H || x >= W) continue;                                  // wave-uniform
            const long long pix = (long long)y * W + x;
            const int x0 = sx0[pl], y0 = sy0[pl];
            const float wx1 = swx[pl], wy1 = swy[pl], wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
            *reinterpret_cast<float2 *>(&a_s[2 * lane]) = *reinterpret_cast<const float2 *>(f0 + ((size_t)b * hw + pix) * C + 2 * lane);
            __builtin_amdgcn_wave_barrier();
            const int ps = lane >> 3, oc = lane & 7, NPp = DP * DP;
            float4 u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = *reinterpret_cast<const float4 *>(&a_s[16 * oc + 4 * q]);
            for (int p0 = 0; p0 < NPp; p0 += 8) {
                const int p = p0 + ps;
                const int jj = p / DP, ii = p - jj * DP;
                const int xx = x0 - R + ii, yy = y0 - R + jj;
                float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
                if (p < NPp && xx >= 0 && xx < W && yy >= 0 && yy < H) {
                    const float *bp = f1 + ((size_t)b * hw + (size_t)yy * W + xx) * C + 16 * oc;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 w4 = *reinterpret_cast<const float4 *>(bp + 4 * q);
                        d0 = fmaf(u[q].x, w4.x, d0); d1 = fmaf(u[q].y, w4.y, d1); d2 = fmaf(u[q].z, w4.z, d2); d3 = fmaf(u[q].w, w4.w, d3);
                    }
                }
                float d = (d0 + d1) + (d2 + d3);
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                if (oc == 0 && p < NPp) d_s[p] = d;
            }
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t < NT; t += 64) {
                const int dy = t / D, dx = t - dy * D;
                const float *dp = &d_s[dy * DP + dx];
                const float v = (wx0 * wy0) * dp[0] + (wx1 * wy0) * dp[1] + (wx0 * wy1) * dp[DP] + (wx1 * wy1) * dp[DP + 1];
                o_s[t * 33 + pl] = v * scale;
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        for (int e = tid; e < NT * kLcTile; e += 256) {
            const int t = e / kLcTile, pl = e - t * kLcTile;
            const int y = ty0 + pl / kLcTX, x = tx0 + pl % kLcTX;
            if (y < H && x < W) corr[((size_t)b * NT + t) * hw + (size_t)y * W + x] = o_s[t * 33 + pl];
        }
        return;
    }
    const int P = (int)P64;
    if (any) lc_box_gemm(f0, f1, b, hw, H, W, ty0, tx0, bx0, by0, BW, P, fa, fb);
    const float *dl = fa;
    for (int e = tid; e < NT * kLcTile; e += 256) {
        const int t = e / kLcTile, pl = e - t * kLcTile;
        const int y = ty0 + pl / kLcTX, x = tx0 + pl % kLcTX;
        if (y >= H || x >= W) continue;
        float v = 0.f;
        if (slive[pl]) {
            const int dy = t / D, dx = t - dy * D;
            const int xx = sx0[pl] - R + dx, yy = sy0[pl] - R + dy;         // the tap's four positions: (xx, yy) .. (xx + 1, yy + 1)
            const float wx1 = swx[pl], wy1 = swy[pl], wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
            const bool xa = xx >= 0 && xx < W, xb = xx + 1 >= 0 && xx + 1 < W, ya = yy >= 0 && yy < H, yb = yy + 1 >= 0 && yy + 1 < H;
            const int base = ((yy - by0) * BW + (xx - bx0)) * 33 + pl;
            const float d00 = (xa && ya) ? dl[base] : 0.f, d01 = (xb && ya) ? dl[base + 33] : 0.f;
            const float d10 = (xa && yb) ? dl[base + BW * 33] : 0.f, d11 = (xb && yb) ? dl[base + (BW + 1) * 33] : 0.f;
            v = ((wx0 * wy0) * d00 + (wx1 * wy0) * d01 + (wx0 * wy1) * d10 + (wx1 * wy1) * d11) * scale;
        }
        corr[((size_t)b * NT + t) * hw + (size_t)y * W + x] = v;
    }
}

// local_corr_softmax_kernel for a 4 x 8 tile at once: the tile's (2R+1)^2 neighbourhoods are the box [tx0 - R, tx0 + 7 + R] x
// [ty0 - R, ty0 + 3 + R] clipped to the image (192 positions for R = 4), one float32-MFMA GEMM (lc_box_gemm); then one wave per
// pixel, lanes over taps, with the softmax and the expectation in the per-pixel kernel's own expressions and reduction order.
__global__ __launch_bounds__(256) void local_corr_softmax_tile_kernel(const float *__restrict__ f0, const float *__restrict__ f1,
                                                                      float *__restrict__ flow, int H, int W, int R, float scale, int tiles_x) {
    __shared__ __attribute__((aligned(16))) float fa[kLcMaxP * kLcRow];
    __shared__ __attribute__((aligned(16))) float fb[kLcTile * kLcRow];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int ty0 = ((int)blockIdx.x / tiles_x) * kLcTY, tx0 = ((int)blockIdx.x % tiles_x) * kLcTX;
    const int D = 2 * R + 1, NT = D * D;
    const size_t hw = (size_t)H * W;
    const int bx0 = max(tx0 - R, 0), by0 = max(ty0 - R, 0);
    const int BW = min(tx0 + kLcTX - 1 + R, W - 1) - bx0 + 1, BH = min(ty0 + kLcTY - 1 + R, H - 1) - by0 + 1;
    lc_box_gemm(f0, f1, b, hw, H, W, ty0, tx0, bx0, by0, BW, BW * BH, fa, fb);      // (2R + 4) (2R + 8) <= kLcMaxP: checked by the launcher
    const float *dl = fa;
    for (int i = 0; i < kLcTile / 4; ++i) {
        const int pl = wave * (kLcTile / 4) + i;
        const int y = ty0 + pl / kLcTX, x = tx0 + pl % kLcTX;
        if (y >= H || x >= W) continue;                                              // wave-uniform
        float mx = -INFINITY;
        float sc[2];
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int t = lane + rep * 64;
            float sv = -INFINITY;
            if (t < NT) {
                const int yy = y + t / D - R, xx = x + t % D - R;
                sv = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? dl[((yy - by0) * BW + (xx - bx0)) * 33 + pl] * scale : -1e9f;   // matching.py:76
            }
            sc[rep] = sv;
            mx = fmaxf(mx, sv);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float sum = 0.f, ex = 0.f, ey = 0.f;
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int t = lane + rep * 64;
            if (t < NT) {
                const float pr = expf(sc[rep] - mx);
                sum += pr;
                ex += pr * (float)(x + t % D - R);
                ey += pr * (float)(y + t / D - R);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            sum += __shfl_xor(sum, off, 64);
            ex += __shfl_xor(ex, off, 64);
            ey += __shfl_xor(ey, off, 64);
        }
        if (lane == 0) {
            const size_t pix = (size_t)y * W + x;
            flow[((size_t)b * 2 + 0) * hw + pix] = ex / sum - (float)x;
            flow[((size_t)b * 2 + 1) * hw + pix] = ey / sum - (float)y;
        }
    }
}

// attention.py:220-256: 3x3 (radius r) local window attention: q = q_proj(f) (tokens [B][HW][128]), kp = k_proj(f) (same
// layout), value = flow [B][2][H][W]; zero padding of both keys and values (F.unfold).  One wave per pixel.
__global__ __launch_bounds__(256) void local_attn_prop_kernel(const float *__restrict__ qf, const float *__restrict__ kf,
                                                              const float *__restrict__ flow, float *__restrict__ out, int H,
                                                              int W, int R, float scale) {
    constexpr int C = 128;
    const int lane = threadIdx.x & 63;
    const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.y;
    if (pix >= (long long)H * W) return;
    const int y = (int)(pix / W), x = (int)(pix % W);
    const int D = 2 * R + 1, NT = D * D;
    const size_t hw = (size_t)H * W;
    const float *a = qf + ((size_t)b * hw + pix) * C;
    float s = -INFINITY, vx = 0.f, vy = 0.f;
    if (lane < NT) {
        const int yy = y + lane / D - R, xx = x + lane % D - R;
        s = 0.f;                                            // out-of-image key = zero vector (unfold padding) -> score 0
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const float *bp = kf + ((size_t)b * hw + (size_t)yy * W + xx) * C;
            float acc = 0.f;
            for (int c = 0; c < C; c += 4) {
                const float4 u = *reinterpret_cast<const float4 *>(a + c), w4 = *reinterpret_cast<const float4 *>(bp + c);
                acc += u.x * w4.x + u.y * w4.y + u.z * w4.z + u.w * w4.w;
            }
            s = acc * scale;
            vx = flow[((size_t)b * 2 + 0) * hw + (size_t)yy * W + xx];
            vy = flow[((size_t)b * 2 + 1) * hw + (size_t)yy * W + xx];
        }
    }
    float mx = s;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float p = lane < NT ? expf(s - mx) : 0.f;
    float sum = p, ox = p * vx, oy = p * vy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        ox += __shfl_xor(ox, off, 64);
        oy += __shfl_xor(oy, off, 64);
    }
    if (lane == 0) {
        out[((size_t)b * 2 + 0) * hw + pix] = ox / sum;
        out[((size_t)b * 2 + 1) * hw + pix] = oy / sum;
    }
}

// =================================================================================================
// Sampling / resampling (NCHW)
// =================================================================================================
// F.interpolate(mode='bilinear', align_corners=True) to (Ho, Wo), times `mul[c]` (mul0 for channel 0, mul1 otherwise)
__global__ void bilinear_resize_kernel(const float *__restrict__ in, float *__restrict__ out, int NC, int C, int H, int W, int Ho,
                                       int Wo, float mul0, float mul1) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)NC * Ho * Wo;
    if (i >= total) return;
    const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
    const long long nc = i / ((long long)Ho * Wo);
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float fy = sy * yo, fx = sx * xo;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < H - 1 ? y0 : H - 1; x0 = x0 < W - 1 ? x0 : W - 1;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly = fy - y0, lx = fx - x0;
    const float *p = in + nc * (long long)H * W;
    const float v = (1.f - ly) * ((1.f - lx) * p[(size_t)y0 * W + x0] + lx * p[(size_t)y0 * W + x1]) +
                    ly * ((1.f - lx) * p[(size_t)y1 * W + x0] + lx * p[(size_t)y1 * W + x1]);
    out[i] = v * (((int)(nc % C) == 0) ? mul0 : mul1);
}

// geometry.py:43-75: out = grid_sample(img, (x + flow_x, y + flow_y)), bilinear, zeros padding, align_corners=True
__global__ void flow_warp_kernel(const float *__restrict__ img, const float *__restrict__ flow, float *__restrict__ out, int N, int C,
                                 int H, int W) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long hw = (long long)H * W;
    if (i >= (long long)N * hw) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), n = (int)(i / hw);
    const float sx = (float)x + flow[((size_t)n * 2 + 0) * hw + (size_t)y * W + x];
    const float sy = (float)y + flow[((size_t)n * 2 + 1) * hw + (size_t)y * W + x];
    // bilinear_sample normalises (2 x / (w-1) - 1) and grid_sample maps back ((g + 1) / 2 * (w - 1))
    const float gx = 2.0f * sx / (float)(W - 1) - 1.0f, gy = 2.0f * sy / (float)(H - 1) - 1.0f;
    const float px = ((gx + 1.0f) * 0.5f) * (float)(W - 1), py = ((gy + 1.0f) * 0.5f) * (float)(H - 1);
    const float x0f = floorf(px), y0f = floorf(py);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float wx1 = px - x0f, wy1 = py - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
    for (int c = 0; c < C; ++c) {
        const float *p = img + ((size_t)n * C + c) * hw;
        float v = 0.f;
        if (vy0 && vx0) v += wy0 * wx0 * p[(size_t)y0 * W + x0];
        if (vy0 && vx1) v += wy0 * wx1 * p[(size_t)y0 * W + x0 + 1];
        if (vy1 && vx0) v += wy1 * wx0 * p[(size_t)(y0 + 1) * W + x0];
        if (vy1 && vx1) v += wy1 * wx1 * p[(size_t)(y0 + 1) * W + x0 + 1];
        out[((size_t)n * C + c) * hw + (size_t)y * W + x] = v;
    }
}

// utils.py:137-155: convex upsampling by `f` (mask [B][9*f*f][H][W], flow [B][2][H][W] -> [B][2][fH][fW])
__global__ void convex_upsample_kernel(const float *__restrict__ flow, const float *__restrict__ mask, float *__restrict__ out, int B,
                                       int H, int W, int f) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Ho = H * f, Wo = W * f;
    if (i >= (long long)B * Ho * Wo) return;
    const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho), b = (int)(i / ((long long)Ho * Wo));
    const int x = xo / f, y = yo / f, kx = xo % f, ky = yo % f;
    const size_t hw = (size_t)H * W;
    float m[9], mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        m[t] = mask[((size_t)b * 9 * f * f + (size_t)t * f * f + ky * f + kx) * hw + (size_t)y * W + x];
        mx = fmaxf(mx, m[t]);
    }
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) { m[t] = expf(m[t] - mx); sum += m[t]; }
    float ox = 0.f, oy = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
            const float p = m[t] / sum;
            ox += p * (float)f * flow[((size_t)b * 2 + 0) * hw + (size_t)yy * W + xx];
            oy += p * (float)f * flow[((size_t)b * 2 + 1) * hw + (size_t)yy * W + xx];
        }
    }
    out[((size_t)b * 2 + 0) * Ho * Wo + (size_t)yo * Wo + xo] = ox;
    out[((size_t)b * 2 + 1) * Ho * Wo + (size_t)yo * Wo + xo] = oy;
}

// geometry.py:78-99 given the two warped flows: occ = (|a + wa| > alpha (|fwd| + |bwd|) + beta) as 0/1
__global__ void fb_check_kernel(const float *__restrict__ fwd, const float *__restrict__ bwd, const float *__restrict__ wbwd,
                                const float *__restrict__ wfwd, float *__restrict__ focc, float *__restrict__ bocc, int B, int H, int W,
                                float alpha, float beta) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long hw = (long long)H * W;
    if (i >= (long long)B * hw) return;
    const long long b = i / hw, p = i % hw;
    const size_t i0 = (size_t)(b * 2) * hw + p, i1 = i0 + hw;
    const float mag = sqrtf(fwd[i0] * fwd[i0] + fwd[i1] * fwd[i1]) + sqrtf(bwd[i0] * bwd[i0] + bwd[i1] * bwd[i1]);
    const float dfx = fwd[i0] + wbwd[i0], dfy = fwd[i1] + wbwd[i1], dbx = bwd[i0] + wfwd[i0], dby = bwd[i1] + wfwd[i1];
    const float thr = alpha * mag + beta;
    focc[i] = sqrtf(dfx * dfx + dfy * dfy) > thr ? 1.0f : 0.0f;
    bocc[i] = sqrtf(dbx * dbx + dby * dby) > thr ? 1.0f : 0.0f;
}

}  // namespace ct

// -------------------------------------------------------------------------------------------------
// C ABI (include/ct_hip.h)
// -------------------------------------------------------------------------------------------------
extern "C" {

int ct_gconv2d_f32(const float *in, const float *wp, const float *bias, float *out, int n, int cin, int cout, int h, int w,
                   int kh, int kw, int stride, int pad_h, int pad_w, long long in_bstride, long long out_bstride, int act,
                   void *stream) {
    if (!in || !wp || !out || n < 0 || cin < 1 || cout < 1 || h < 1 || w < 1 || kh < 1 || kw < 1 || stride < 1) return CT_E_BADARG;
    if (n == 0) return CT_OK;
    ct::GConvArgs a;
    a.in = in; a.wp = wp; a.bias = bias; a.out = out;
    a.cin = cin; a.cout = cout; a.coutp = 64 * ((cout + 63) / 64);
    a.H = h; a.W = w; a.KH = kh; a.KW = kw; a.stride = stride; a.padH = pad_h; a.padW = pad_w;
    a.Ho = (h + 2 * pad_h - kh) / stride + 1; a.Wo = (w + 2 * pad_w - kw) / stride + 1;
    if (a.Ho < 1 || a.Wo < 1) return CT_E_BADARG;
    a.in_bstride = in_bstride; a.out_bstride = out_bstride; a.act = act;
    {
        const int rc = ct::conv_direct(a, n, (hipStream_t)stream);      // the shapes an implicit-GEMM tile wastes (conv_direct.hip)
        if (rc != 1) return rc;
    }
    if (stride == 1 && pad_h == kh / 2 && pad_w == kw / 2 && (kh & 1) && (kw & 1) && bias) {
        // stride-1 "same" convolution: the LDS-tiled persistent kernel of cnn.hip (same weight layout)
        ct::ConvArgs f;
        f.in = in; f.in2 = nullptr; f.cin1 = cin; f.in2_bstride = 0; f.wp = wp; f.bias = bias; f.residual = nullptr; f.out = out;
        f.cin = cin; f.cout = cout; f.H = h; f.W = w;
        f.in_bstride = in_bstride; f.out_bstride = out_bstride; f.res_bstride = 0;
        f.act = act; f.clamp = 0; f.groups = a.coutp / 64; f.prof = nullptr;
        const int rc = ct::conv_fast(f, n, kh, kw, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    const int TR = 3 * stride + kh, TC = 31 * stride + kw;
    int cchunk = (48 * 1024) / (TR * TC * 4);
    cchunk &= ~1;
    if (cchunk > 32) cchunk = 32;
    if (cchunk < 2) return CT_E_BADARG;
    a.cchunk = cchunk;
    const size_t lds = (size_t)cchunk * TR * TC * sizeof(float);
    const int tiles_x = (a.Wo + 31) / 32, tiles_y = (a.Ho + 3) / 4;
    dim3 grid(tiles_x * tiles_y, a.coutp / 64, n);
    hipLaunchKernelGGL(ct::conv_generic_kernel, grid, dim3(256), lds, (hipStream_t)stream, a, tiles_x, tiles_y);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

// The generic convolution with explicit top / left zero padding and output size (bottom / right padding is whatever the
// output size implies) -- TF-"SAME" static padding of efficientnet_pytorch's stride-2 stem: pad (0, 1).
int ct_gconv2d_pad_f32(const float *in, const float *wp, const float *bias, float *out, int n, int cin, int cout, int h, int w, int kh,
                       int kw, int stride, int pad_top, int pad_left, int out_h, int out_w, long long in_bstride, long long out_bstride,
                       int act, void *stream) {
    if (!in || !wp || !out || n < 0 || cin < 1 || cout < 1 || h < 1 || w < 1 || kh < 1 || kw < 1 || stride < 1 || pad_top < 0 ||
        pad_left < 0 || out_h < 1 || out_w < 1)
        return CT_E_BADARG;
    if ((out_h - 1) * stride - pad_top >= h || (out_w - 1) * stride - pad_left >= w) return CT_E_BADARG;
    if (n == 0) return CT_OK;
    ct::GConvArgs a;
    a.in = in; a.wp = wp; a.bias = bias; a.out = out;
    a.cin = cin; a.cout = cout; a.coutp = 64 * ((cout + 63) / 64);
    a.H = h; a.W = w; a.KH = kh; a.KW = kw; a.stride = stride; a.padH = pad_top; a.padW = pad_left;
    a.Ho = out_h; a.Wo = out_w;
    a.in_bstride = in_bstride; a.out_bstride = out_bstride; a.act = act;
    const int TR = 3 * stride + kh, TC = 31 * stride + kw;
    int cchunk = (48 * 1024) / (TR * TC * 4);
    cchunk &= ~1;
    if (cchunk > 32) cchunk = 32;
    if (cchunk < 2) return CT_E_BADARG;
    a.cchunk = cchunk;
    const size_t lds = (size_t)cchunk * TR * TC * sizeof(float);
    const int tiles_x = (a.Wo + 31) / 32, tiles_y = (a.Ho + 3) / 4;
    hipLaunchKernelGGL(ct::conv_generic_kernel, dim3(tiles_x * tiles_y, a.coutp / 64, n), dim3(256), lds, (hipStream_t)stream, a, tiles_x,
                       tiles_y);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

size_t ct_instance_norm_workspace_bytes(int planes) {
    return planes > 0 ? (size_t)planes * ct::kInormSplit * 2 * sizeof(double) : 0;
}

int ct_instance_norm_f32(const float *x, const float *skip, float *y, int planes, int plane, float eps, int mode, void *ws,
                         size_t ws_bytes, void *stream) {
    if (!x || !y || planes < 0 || plane < 1 || (mode == 2 && !skip)) return CT_E_BADARG;
    if (planes == 0) return CT_OK;
    if (ws && planes < 1024 && plane >= 16384) {   // few, large planes: split them (see inorm_partial_kernel)
        if (ws_bytes < ct_instance_norm_workspace_bytes(planes)) return CT_E_WORKSPACE;
        if (reinterpret_cast<uintptr_t>(ws) & 7) return CT_E_ALIGN;
        dim3 grid(planes, ct::kInormSplit);
        hipLaunchKernelGGL(ct::inorm_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, plane, (double *)ws);
        CT_CHECK_LAUNCH();
        hipLaunchKernelGGL(ct::inorm_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, skip, y, plane, eps, mode,
                           (const double *)ws);
        CT_CHECK_LAUNCH();
        return CT_OK;
    }
    hipLaunchKernelGGL(ct::inorm_kernel, dim3(planes), dim3(256), 0, (hipStream_t)stream, x, skip, y, plane, eps, mode);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_space_to_depth2_f32(const float *in, float *out, int n, int c, int h, int w, long long in_bstride, void *stream) {
    if (!in || !out || n < 0 || c < 1 || h < 2 || w < 8 || (h & 1) || (w & 7) || (in_bstride & 3)) return CT_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) return CT_E_ALIGN;
    if (n == 0) return CT_OK;
    const long long total = (long long)n * c * h * (w >> 3);
    if ((total + 255) / 256 > 0x7fffffffLL) return CT_E_BADARG;
    hipLaunchKernelGGL(ct::space_to_depth2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, c, h, w,
                       in_bstride, total);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_eltwise_f32(const float *a, const float *b, const float *c, float *y, long long n, int op, int plane, int chans, int split,
                   float s0, void *stream) {
    if (!a || !y || n < 0) return CT_E_BADARG;
    if (n == 0) return CT_OK;
    hipLaunchKernelGGL(ct::eltwise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, y, n, op,
                       plane > 0 ? plane : 1, chans > 0 ? chans : 1, split, s0);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_linear_tokens_f32(const float *x, const float *x2, int k1, const float *w, const float *bias, float *out, long long tokens,
                         int k, int n, int act, void *stream) {
    if (!x || !w || !out || tokens < 0 || k < 16 || (k % 16) || n < 1) return CT_E_BADARG;
    if (x2 ? (k1 < 32 || k1 >= k || (k1 % 32)) : (k1 != k)) return CT_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(x2)) & 15) return CT_E_ALIGN;
    if (tokens == 0) return CT_OK;
    dim3 grid((unsigned)((tokens + 127) / 128), (n + 127) / 128);
    hipLaunchKernelGGL(ct::linear_tokens_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, x2, k1, w, bias, out, tokens, k, n, act);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

#ifdef CT_LS_PROFILE
int ct_debug_ls_prof(unsigned long long *host8, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return CT_E_BADARG;
    if (host8 && hipMemcpyFromSymbol(host8, HIP_SYMBOL(ct::g_ls_prof), 64) != hipSuccess) return CT_E_BADARG;
    if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ct::g_ls_prof), z, 64) != hipSuccess) return CT_E_BADARG; }
    return CT_OK;
}
#endif
// wp: ct_hip.pack_linear_weight_split(weight): bf16 bit patterns [ceil(n/128)][k/32][piece hi/mid/lo][8-channel group 0..3]
// [feature row 0..127][8 channels], zero rows beyond n.  k % 32 == 0 (k1 % 32 == 0 with x2).
int ct_linear_tokens_split_f32(const float *x, const float *x2, int k1, const void *wp, const float *bias, float *out, long long tokens,
                               int k, int n, int act, void *stream) {
    if (!x || !wp || !out || tokens < 0 || k < 32 || (k % 32) || n < 1) return CT_E_BADARG;
    if (x2 ? (k1 < 32 || k1 >= k || (k1 % 32)) : (k1 != k)) return CT_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wp) | reinterpret_cast<uintptr_t>(x2)) & 15) return CT_E_ALIGN;
    if (tokens == 0) return CT_OK;
    const int n_nt = (n + 127) / 128;
    // 64-token tiles while 128-token tiles would leave CUs without a workgroup (2 per CU are resident)
    const long long tiles128 = (tokens + 127) / 128 * n_nt;
#ifndef CT_LS_NOWRES
    if (k == 128 && n <= 128 && !x2) {              // q / k / v / merge projections: W resident in LDS, X straight into MFMA fragments
        const long long tiles32 = (tokens + 31) / 32;
        long long g = (tiles32 + 7) / 8;
        if (g > 256) g = 256;
        hipLaunchKernelGGL(ct::linear_split_wres_kernel, dim3((unsigned)g), dim3(512), 0, (hipStream_t)stream, x, (const uint4 *)wp, bias, out,
                           tokens, n, act, (int)tiles32);
        CT_CHECK_LAUNCH();
        return CT_OK;
    }
#endif
    if (tiles128 >= 2 * 256) {
        hipLaunchKernelGGL(ct::linear_split_kernel<128>, dim3((unsigned)tiles128), dim3(256), 0, (hipStream_t)stream, x, x2, k1,
                           (const uint4 *)wp, bias, out, tokens, k, n, act, n_nt);
    } else {
        hipLaunchKernelGGL(ct::linear_split_kernel<64>, dim3((unsigned)((tokens + 63) / 64 * n_nt)), dim3(256), 0, (hipStream_t)stream, x,
                           x2, k1, (const uint4 *)wp, bias, out, tokens, k, n, act, n_nt);
    }
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_nchw_to_rows_f32(const float *nchw, float *rows, int batch, int c, int h, int w, long long nchw_bstride, int row_channels,
                        int c0, void *stream) {
    if (!nchw || !rows || batch < 0 || c < 1 || h < 1 || w < 1 || c0 < 0 || c0 + c > row_channels) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    hipLaunchKernelGGL((ct::rows_transpose_kernel<true>), dim3((w + 63) / 64, batch * h), dim3(256), 0, (hipStream_t)stream, nchw,
                       rows, c, h, w, nchw_bstride, row_channels, c0);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_rows_to_nchw_f32(const float *rows, float *nchw, int batch, int c, int h, int w, long long nchw_bstride, int row_channels,
                        int c0, void *stream) {
    if (!nchw || !rows || batch < 0 || c < 1 || h < 1 || w < 1 || c0 < 0 || c0 + c > row_channels) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    hipLaunchKernelGGL((ct::rows_transpose_kernel<false>), dim3((w + 63) / 64, batch * h), dim3(256), 0, (hipStream_t)stream, rows,
                       nchw, c, h, w, nchw_bstride, row_channels, c0);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_layernorm128_f32(const float *x, const float *gamma, const float *beta, const float *residual, float *out, long long tokens,
                        int partials, void *stream) {
    if (!x || !gamma || !beta || !out || tokens < 0 || partials < 1 || partials > 64) return CT_E_BADARG;
    if (tokens == 0) return CT_OK;
    hipLaunchKernelGGL(ct::layernorm_tokens_kernel, dim3((unsigned)((tokens + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                       beta, residual, out, tokens, partials);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

size_t ct_attention_workspace_bytes(int batch, int len, int cv, int nsplit) {
    if (batch < 0 || len < 0 || cv < 0 || nsplit < 2) return 0;
    return (size_t)nsplit * batch * len * (cv + 2) * sizeof(float);
}

int ct_attention_tokens_f32(const float *q, const float *k, const float *v, const int *region, const int *rowmap, float *out,
                            int batch, int len, int cv, float scale, int nsplit, float *ws, size_t ws_bytes, long long kv_shift,
                            void *stream) {
    if (!q || !k || !v || !out || batch < 0 || len < 1 || (cv != 2 && cv != 128) || nsplit < 1 || nsplit > 64) return CT_E_BADARG;
    const long long kv_total = (long long)batch * len;
    if (kv_shift < 0 || kv_shift >= (kv_total > 0 ? kv_total : 1) || (kv_shift != 0 && !rowmap)) return CT_E_BADARG;
    if (nsplit > 1 && (!ws || ws_bytes < ct_attention_workspace_bytes(batch, len, cv, nsplit))) return CT_E_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out)) & 15) return CT_E_ALIGN;
    if (batch == 0) return CT_OK;
    dim3 grid((len + 127) / 128, batch, nsplit);
    float *nostats = nullptr;
#define CT_ATT(CVV, MAPPED) hipLaunchKernelGGL((ct::attention_tokens_kernel<128, CVV, MAPPED, true>), grid, dim3(256), 0, (hipStream_t)stream, q, k, v, region, rowmap, out, nostats, len, scale, ws, kv_shift, kv_total)
    if (ct::attention16_enabled()) ct::attention16_tokens128(q, k, v, region, rowmap, out, batch, len, cv, scale, nsplit, ws, kv_shift, kv_total, (hipStream_t)stream);
    else if (cv == 128) { if (rowmap) CT_ATT(128, true); else CT_ATT(128, false); }
    else { if (rowmap) CT_ATT(2, true); else CT_ATT(2, false); }
#undef CT_ATT
    CT_CHECK_LAUNCH();
    if (nsplit > 1) {
        const long long tokens = (long long)batch * len;
        hipLaunchKernelGGL(ct::attention_combine_kernel, dim3((unsigned)((tokens * cv + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           ws, rowmap, out, tokens, cv, nsplit);
        CT_CHECK_LAUNCH();
    }
    return CT_OK;
}

int ct_attention_rows64_f32(const float *q, const float *k, const float *v, float *out, float *stats, int batch, int len, float scale,
                            void *stream) {
    if (!q || !k || batch < 0 || len < 1 || (v && !out) || (!v && !stats)) return CT_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) return CT_E_ALIGN;
    if (batch == 0) return CT_OK;
    dim3 grid((len + 127) / 128, batch);
    const int *noreg = nullptr;
    if (v && (reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(out)) & 15) return CT_E_ALIGN;
    if (ct::attention16_enabled()) ct::attention16_rows64(q, k, v, out, stats, batch, len, scale, (hipStream_t)stream);
    else if (v) hipLaunchKernelGGL((ct::attention_tokens_kernel<64, 96, false, true>), grid, dim3(256), 0, (hipStream_t)stream, q, k, v, noreg, noreg, out, stats, len, scale, (float *)nullptr);
    else hipLaunchKernelGGL((ct::attention_tokens_kernel<64, 0, false, true>), grid, dim3(256), 0, (hipStream_t)stream, q, k, v, noreg, noreg, out, stats, len, scale, (float *)nullptr);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_attention_colsum64_f32(const float *q, const float *k, const float *stats, float *colsum, int batch, int len, float scale,
                              void *stream) {
    if (!q || !k || !stats || !colsum || batch < 0 || len < 1) return CT_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) return CT_E_ALIGN;
    if (batch == 0) return CT_OK;
    dim3 grid((len + 127) / 128, batch);
    if (ct::attention16_enabled()) ct::attention16_colsum64(q, k, stats, colsum, batch, len, scale, (hipStream_t)stream);
    else hipLaunchKernelGGL((ct::attention_colsum_kernel<64>), grid, dim3(256), 0, (hipStream_t)stream, q, k, stats, colsum, len, scale);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_local_corr_softmax_f32(const float *f0, const float *f1, float *flow, int batch, int h, int w, int radius, void *stream) {
    if (!f0 || !f1 || !flow || batch < 0 || h < 1 || w < 1 || radius < 0 || (2 * radius + 1) * (2 * radius + 1) > 128) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    static const bool tile_form = [] { const char *e = getenv("CT_HIP_LCF_TILE"); return !(e && atoi(e) == 0); }();
    if (tile_form && (2 * radius + ct::kLcTY) * (2 * radius + ct::kLcTX) <= ct::kLcMaxP && (reinterpret_cast<uintptr_t>(f0) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(f1) & 15) == 0) {
        const int tiles_x = (w + ct::kLcTX - 1) / ct::kLcTX, tiles_y = (h + ct::kLcTY - 1) / ct::kLcTY;
        hipLaunchKernelGGL(ct::local_corr_softmax_tile_kernel, dim3((unsigned)(tiles_x * tiles_y), batch), dim3(256), 0, (hipStream_t)stream, f0, f1,
                           flow, h, w, radius, 1.0f / sqrtf(128.0f), tiles_x);
        CT_CHECK_LAUNCH();
        return CT_OK;
    }
    dim3 grid((unsigned)(((long long)h * w + 3) / 4), batch);
    hipLaunchKernelGGL(ct::local_corr_softmax_kernel, grid, dim3(256), 0, (hipStream_t)stream, f0, f1, flow, h, w, radius,
                       1.0f / sqrtf(128.0f));
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_local_corr_flow_f32(const float *f0, const float *f1, const float *flow, float *corr, int batch, int h, int w, int radius,
                           void *stream) {
    if (!f0 || !f1 || !flow || !corr || batch < 0 || h < 2 || w < 2 || radius < 0 || radius > 4) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    static const bool tile_form = [] { const char *e = getenv("CT_HIP_LCF_TILE"); return !(e && atoi(e) == 0); }();
    if (tile_form && (reinterpret_cast<uintptr_t>(f0) & 15) == 0 && (reinterpret_cast<uintptr_t>(f1) & 15) == 0) {
        const int tiles_x = (w + ct::kLcTX - 1) / ct::kLcTX, tiles_y = (h + ct::kLcTY - 1) / ct::kLcTY;
        hipLaunchKernelGGL(ct::local_corr_flow_tile_kernel, dim3((unsigned)(tiles_x * tiles_y), batch), dim3(256), 0, (hipStream_t)stream, f0, f1,
                           flow, corr, h, w, radius, 1.0f / sqrtf(128.0f), tiles_x);
        CT_CHECK_LAUNCH();
        return CT_OK;
    }
    dim3 grid((unsigned)(((long long)h * w + ct::kLcfPix - 1) / ct::kLcfPix), batch);
    hipLaunchKernelGGL(ct::local_corr_flow_kernel, grid, dim3(256), 0, (hipStream_t)stream, f0, f1, flow, corr, h, w, radius,
                       1.0f / sqrtf(128.0f));
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_local_attn_prop_f32(const float *q, const float *k, const float *flow, float *out, int batch, int h, int w, int radius,
                           void *stream) {
    if (!q || !k || !flow || !out || batch < 0 || h < 1 || w < 1 || radius < 1 || (2 * radius + 1) * (2 * radius + 1) > 64) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    dim3 grid((unsigned)(((long long)h * w + 3) / 4), batch);
    hipLaunchKernelGGL(ct::local_attn_prop_kernel, grid, dim3(256), 0, (hipStream_t)stream, q, k, flow, out, h, w, radius,
                       1.0f / sqrtf(128.0f));
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_bilinear_resize_f32(const float *in, float *out, int n, int c, int h, int w, int ho, int wo, float mul0, float mul1, void *stream) {
    if (!in || !out || n < 0 || c < 1 || h < 1 || w < 1 || ho < 1 || wo < 1) return CT_E_BADARG;
    const long long total = (long long)n * c * ho * wo;
    if (total == 0) return CT_OK;
    hipLaunchKernelGGL(ct::bilinear_resize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, n * c,
                       c, h, w, ho, wo, mul0, mul1);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_flow_warp_f32(const float *img, const float *flow, float *out, int n, int c, int h, int w, void *stream) {
    if (!img || !flow || !out || n < 0 || c < 1 || h < 2 || w < 2) return CT_E_BADARG;
    const long long total = (long long)n * h * w;
    if (total == 0) return CT_OK;
    hipLaunchKernelGGL(ct::flow_warp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, flow, out, n, c, h, w);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_convex_upsample_f32(const float *flow, const float *mask, float *out, int b, int h, int w, int factor, void *stream) {
    if (!flow || !mask || !out || b < 0 || h < 1 || w < 1 || factor < 1) return CT_E_BADARG;
    const long long total = (long long)b * h * factor * w * factor;
    if (total == 0) return CT_OK;
    hipLaunchKernelGGL(ct::convex_upsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flow, mask, out, b, h, w, factor);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_fb_check_f32(const float *fwd, const float *bwd, const float *warped_bwd, const float *warped_fwd, float *fwd_occ, float *bwd_occ,
                    int b, int h, int w, float alpha, float beta, void *stream) {
    if (!fwd || !bwd || !warped_bwd || !warped_fwd || !fwd_occ || !bwd_occ || b < 0) return CT_E_BADARG;
    const long long total = (long long)b * h * w;
    if (total == 0) return CT_OK;
    hipLaunchKernelGGL(ct::fb_check_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fwd, bwd, warped_bwd, warped_fwd,
                       fwd_occ, bwd_occ, b, h, w, alpha, beta);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // extern "C"
