// linear_ws16.hip -- nn.Linear on channels-last tokens with the weight slice RESIDENT in LDS and float32 operands as two fp16 pieces
// (three v_mfma_f32_32x32x16_f16 per 16-wide product, float32 accumulation; arithmetic and error model of conv_ws.hip /
// attention16.hip).  Built for the FFN of a GMFlow transformer layer (unimatch/transformer.py:12-43,
// mlp = Linear(256 -> 1024, no bias) . GELU . Linear(1024 -> 128, no bias)), whose two GEMMs were bound by operand delivery in the
// LDS-tiled kernel of gmflow.hip (every 128 x 128 output tile re-staged its X tile and streamed its W slice):
//   * a slice = 256 input channels x 128 output features of W = 128 KiB as fp16 (hi, lo) pieces: one persistent 8-wave workgroup
//     per CU keeps ONE slice in LDS for its whole life and walks token tiles; no barrier after the prologue;
//   * the activations never touch LDS: the B operand of D[feature][token] = W X^T wants, per lane, 8 consecutive channels of one
//     token -- what a lane gets from 16-byte loads of a row-major row.  Lane half hl reads channels 128 hl .. 128 hl + 127 of the
//     slice (any bijection is a valid contraction order as long as W is packed with the same one), so with two sources
//     (cat([source, message]) of the FFN) the halves simply read different tensors;
//   * "N slices" (FFN1: 1024 features = 8 slices, every slice reads the same tokens: the 8 workgroups of a token group sit on one XCD
//     and share them through its L2) or "K slices" (FFN2: 1024 channels = 4 slices, each leaves a float32 partial slab
//     [slice][T][128]; ct_layernorm128_f32 adds the slabs in a fixed order on its way in);
//   * fp16 range: the weights carry one power of two per layer (host), the activations a RUNNING power of two per TOKEN (the
//     lane's own: the maximum of its channels seen so far, one cross-half shuffle per 16-channel chunk): when it drops for any
//     token of the wave, the 64 accumulators are rescaled between two chunks (a few times per tile); a channel smaller than its
//     token's running maximum keeps an absolute error below 2^-36 of it -- the result does not depend on the magnitude of a token.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>
#include "ct_common.h"
#include "../../include/ct_hip.h"

namespace ct {

typedef float f32x16w __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2w __attribute__((ext_vector_type(2)));

// uint4 entries of one slice image: [piece][k step][lane half][feature] x 8 channels; NCH = 16-channel chunks per lane half
// (8: slices of 256 input channels; 4: slices of 128 -- the q / k / v / merge projections, several of them in one launch)
constexpr int w16_img(int NCH) { return 2 * (2 * NCH) * 2 * 128; }

__device__ __forceinline__ unsigned int cvt_pk_f16w(float a, float b) {
    unsigned int r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split2x2w(float x0, float x1, unsigned int &hw, unsigned int &lw) {
    hw = cvt_pk_f16w(x0, x1);
    const f16x2w h = __builtin_bit_cast(f16x2w, hw);
    lw = cvt_pk_f16w(x0 - (float)h.x, x1 - (float)h.y);
}
__device__ __forceinline__ int scale_exp_w(float mx, int none) {      // 2^e * mx in [2^11, 2^12)
    const int fld = (int)(__float_as_uint(mx) >> 23);
    const int ex = fld == 0 ? none : fld == 255 ? 0 : 138 - fld;
    return min(max(ex, -100), 100);
}
__device__ __forceinline__ float pow2i_w(int e) { return __uint_as_float((unsigned int)(127 + e) << 23); }

__device__ __forceinline__ float gelu_w(float v) {                    // gmflow.hip: gelu_as (A&S 7.1.26, branch free)
    const float z = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
    const float erfc_half = 0.5f * p * t * e;
    return v > 0.f ? v - v * erfc_half : v * erfc_half;
}

struct Ws16Args {
    const float *xa, *xb;          // token rows read by lane half 0 / 1 (slice 0): 128 channels each
    long long lda, ldb;            // row strides (floats)
    long long x_slice;             // + slice * x_slice floats on both (K slices), 0 for N slices
    const uint4 *wp;               // [slice][w16_img(NCH)]
    int w_exp;
    const float *bias;             // nullable; N slices: bias + slice * 128; K slices: added by slice 0 only
    float *out;
    long long ldo;                 // row stride of out
    long long out_slice;           // + slice * out_slice floats (N slices: 128 columns; K slices: T * 128)
    long long T;
    int n_tiles;                   // 32-token tiles
    int S;                         // slices (1, 2, 3, 4, 8, 16, 32: floor(32 / S) token groups per XCD)
    int kslices;                   // 1 = K slices (partials), 0 = N slices
    int act;                       // 0 none, 6 GELU
    const float *ln_g, *ln_b;      // non-NULL (one slice of 128 features only): out = [ln_res +] LayerNorm_128(result) (eps 1e-5)
    const float *ln_res;           // nullable: the residual rows [T][128]
};

template <int NCH>
__global__ __launch_bounds__(512, 1) void linear_ws16_kernel(Ws16Args a) {
    constexpr int kW16Img = w16_img(NCH), KS = 2 * NCH;
    extern __shared__ uint4 Ws[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    // workgroup b sits on XCD b % 8: the S slices of one token group share an XCD (and, N slices, their tokens through its L2)
    const int xcd = blockIdx.x & 7, m = blockIdx.x >> 3;
    const int per_xcd = gridDim.x >> 3;
    const int gpx = per_xcd / a.S;                 // token groups per XCD (S = 3: two workgroups per XCD stay idle)
    if (m >= gpx * a.S) return;
    const int slice = m % a.S;
    const int group = m / a.S + gpx * xcd, n_groups = gpx * 8;
    {
        const uint4 *src = a.wp + (size_t)slice * kW16Img;
#pragma unroll
        for (int j = 0; j < kW16Img / 512; ++j) Ws[tid + 512 * j] = src[tid + 512 * j];
    }
    const float *xrow = (hl ? a.xb : a.xa) + (size_t)slice * a.x_slice;
    const long long ld = hl ? a.ldb : a.lda;
    float *out = a.out + (size_t)slice * a.out_slice;
    const float *bias = a.bias ? (a.kslices ? (slice == 0 ? a.bias : nullptr) : a.bias + slice * 128) : nullptr;
    __syncthreads();

    const int tpg = (a.n_tiles + n_groups - 1) / n_groups;
    const int t_begin = group * tpg, t_end = min(a.n_tiles, t_begin + tpg);
    int tile = t_begin + wave;
    if (tile >= t_end) return;
    auto row_ptr = [&](int t) {
        const long long tr = (long long)t * 32 + nl;
        return reinterpret_cast<const float4 *>(xrow + (tr < a.T ? tr : a.T - 1) * ld);
    };
    // chunk = 16 channels per lane half = two K steps = four float4; ring of four raw chunks (loads three chunks ahead)
    float4 raw[4][4];
    uint4 fb[2][2][2];                             // B fragments [chunk parity][K step][piece]
    auto fetch = [&](const float4 *p, int c, float4 (&r)[4]) {
#ifdef CT_W16_ABL_NOLOAD
        if (tile != t_begin + wave) return;
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = p[4 * c + q];
    };
    auto chunk_max = [&](const float4 (&r)[4]) {
        float mx = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(r[q].x), fabsf(r[q].y)), fmaxf(fabsf(r[q].z), fabsf(r[q].w))));
        return scale_exp_w(fmaxf(mx, __shfl_xor(mx, 32, 64)), 100);       // the token's two channel halves sit 32 lanes apart
    };
    auto split_x = [&](const float4 (&r)[4], float sc, uint4 (&f)[2][2]) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            unsigned int h[4], l[4];
            split2x2w(r[2 * s].x * sc, r[2 * s].y * sc, h[0], l[0]);
            split2x2w(r[2 * s].z * sc, r[2 * s].w * sc, h[1], l[1]);
            split2x2w(r[2 * s + 1].x * sc, r[2 * s + 1].y * sc, h[2], l[2]);
            split2x2w(r[2 * s + 1].z * sc, r[2 * s + 1].w * sc, h[3], l[3]);
            f[s][0] = make_uint4(h[0], h[1], h[2], h[3]);
            f[s][1] = make_uint4(l[0], l[1], l[2], l[3]);
        }
    };
    const float4 *xp = row_ptr(tile);
    fetch(xp, 0, raw[0]);
    fetch(xp, 1, raw[1]);
    fetch(xp, 2, raw[2]);
    int e_cur = chunk_max(raw[0]), e_nxt = e_cur;  // per lane (= token): domain of its accumulators / scale of the fragments converted last
    split_x(raw[0], pow2i_w(e_cur), fb[0]);
    const uint4 *wb = Ws + hl * 128 + nl;
    uint4 wf0[4][2], wf1[4][2];                    // W fragments of the even / odd K steps
    auto load_w = [&](int ks, uint4 (&wf)[4][2]) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j][p] = wb[((p * KS + ks) * 2) * 128 + 32 * j];
    };
    float4 *stg = reinterpret_cast<float4 *>(Ws + kW16Img) + wave * (32 * 8);      // per-wave output transpose, 4 KiB
    load_w(0, wf0);
    for (; tile < t_end; tile += 8) {
        const int next = tile + 8;
        const float4 *xn = row_ptr(next < t_end ? next : tile);
        f32x16w acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        auto mfma_step = [&](const uint4 (&wf)[4][2], const uint4 (&xf)[2]) {
            const f16x8w xh = __builtin_bit_cast(f16x8w, xf[0]), xl = __builtin_bit_cast(f16x8w, xf[1]);
            // small terms first; the four accumulators take turns, so no MFMA waits for the one issued just before it
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, wf[j][1]), xh, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, wf[j][0]), xl, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8w, wf[j][0]), xh, acc[j], 0, 0, 0);
        };
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // raw[(c + 1) & 3] holds chunk c+1 (of this tile, or chunk 0 of the next): converted under the MFMAs of chunk c;
            // raw[(c + 3) & 3] is free: chunk c+3 goes there
            if (c + 3 < NCH) fetch(xp, c + 3, raw[(c + 3) & 3]);
            else fetch(xn, c + 3 - NCH, raw[(c + 3) & 3]);
            const int e_chunk = chunk_max(raw[(c + 1) & 3]);
            e_nxt = (c == NCH - 1) ? e_chunk : min(e_cur, e_chunk);       // a new tile starts its own running scale
            const float sc = pow2i_w(e_nxt);
            // W fragments double buffered in registers: the eight LDS reads of K step ks + 1 are issued under the MFMAs of step ks
            // (left to itself the compiler keeps ONE fragment register and waits out the LDS latency before every MFMA pair)
            load_w(2 * c + 1, wf1);
            mfma_step(wf0, fb[c & 1][0]);
            load_w((2 * c + 2) % KS, wf0);
            mfma_step(wf1, fb[c & 1][1]);
            split_x(raw[(c + 1) & 3], sc, fb[(c + 1) & 1]);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // two of the sixteen fragment reads
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);      // three of the 24 MFMAs
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);      // conversion of the next chunk
            }
            if (c < NCH - 1 && __builtin_amdgcn_ballot_w64(e_nxt != e_cur) != 0) {   // a token's running scale dropped: new domain
                const float rs = __builtin_amdgcn_ldexpf(1.0f, e_nxt - e_cur);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] *= rs;
            }
            if (c < NCH - 1) e_cur = e_nxt;
        }
        // epilogue: lane = token nl, registers = features 32 j + 8 g + 4 hl + (0..3).  Through a per-wave LDS transpose
        // ([token][8 float4], float4 slot XOR-swizzled by the token so that neither side conflicts) so that 8 lanes store the 128
        // contiguous bytes of one token's 32 features: a direct store would scatter 32-byte pieces at the row stride.
        const int un = -(e_cur + a.w_exp);
        const long long tbase = (long long)tile * 32;
        float ln_mean = 0.f, ln_rstd = 1.f;
        if (NCH == 4 && a.ln_g) {
            // LayerNorm over the token's 128 features (transformer.py:139-147): they are this lane's 64 registers and its
            // partner's (32 lanes apart); two passes over registers like layernorm_tokens_kernel
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[j][r] = __builtin_amdgcn_ldexpf(acc[j][r], un);
                    if (bias) acc[j][r] += bias[32 * j + 8 * (r >> 2) + 4 * hl + (r & 3)];
                    sum += acc[j][r];
                }
            sum += __shfl_xor(sum, 32, 64);
            ln_mean = sum * (1.0f / 128.0f);
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = acc[j][r] - ln_mean; ss = fmaf(d, d, ss); }
            ss += __shfl_xor(ss, 32, 64);
            ln_rstd = 1.0f / sqrtf(ss * (1.0f / 128.0f) + 1e-5f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v;
                if (NCH == 4 && a.ln_g) {
                    const float4 g4 = *reinterpret_cast<const float4 *>(a.ln_g + 32 * j + 8 * g + 4 * hl);
                    const float4 b4 = *reinterpret_cast<const float4 *>(a.ln_b + 32 * j + 8 * g + 4 * hl);
                    v = make_float4((acc[j][4 * g] - ln_mean) * ln_rstd * g4.x + b4.x, (acc[j][4 * g + 1] - ln_mean) * ln_rstd * g4.y + b4.y,
                                    (acc[j][4 * g + 2] - ln_mean) * ln_rstd * g4.z + b4.z, (acc[j][4 * g + 3] - ln_mean) * ln_rstd * g4.w + b4.w);
                } else {
                    v = make_float4(__builtin_amdgcn_ldexpf(acc[j][4 * g], un), __builtin_amdgcn_ldexpf(acc[j][4 * g + 1], un),
                                    __builtin_amdgcn_ldexpf(acc[j][4 * g + 2], un), __builtin_amdgcn_ldexpf(acc[j][4 * g + 3], un));
                    if (bias) {
                        const float4 b4 = *reinterpret_cast<const float4 *>(bias + 32 * j + 8 * g + 4 * hl);
                        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
                    }
#ifndef CT_W16_ABL_NOGELU
                    if (a.act == 6) { v.x = gelu_w(v.x); v.y = gelu_w(v.y); v.z = gelu_w(v.z); v.w = gelu_w(v.w); }
#endif
                }
                stg[nl * 8 + ((2 * g + hl) ^ (nl & 7))] = v;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int tk = (lane >> 3) + 8 * q, c4 = lane & 7;
                float4 v = stg[tk * 8 + (c4 ^ (tk & 7))];
#ifdef CT_W16_ABL_NOSTORE
                if (tbase + tk < a.T && v.x == 123.456f) {
#else
                if (tbase + tk < a.T) {
#endif
                    if (NCH == 4 && a.ln_res) {
                        const float4 r4 = *reinterpret_cast<const float4 *>(a.ln_res + (tbase + tk) * 128 + 32 * j + 4 * c4);
                        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
                    }
                    *reinterpret_cast<float4 *>(out + (tbase + tk) * a.ldo + 32 * j + 4 * c4) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        e_cur = e_nxt;
        xp = xn;
    }
}

}  // namespace ct

extern "C" {

// see include/ct_hip.h
int ct_linear_ws16_f32(const float *x, const float *x2, int k1, const void *wp16, int w_exp, const float *bias, float *out,
                       long long tokens, int k, int n, int act, const float *ln_gamma, const float *ln_beta, const float *ln_residual,
                       void *stream) {
    if (!x || !wp16 || !out || tokens < 0 || (act != 0 && act != 6)) return CT_E_BADARG;
    if ((ln_gamma != nullptr) != (ln_beta != nullptr) || (ln_residual && !ln_gamma)) return CT_E_BADARG;
    if (ln_gamma && (k != 128 || n != 128 || x2 || act != 0)) return CT_E_BADARG;    // the fused LayerNorm needs a token's 128 features in one slice
    if ((reinterpret_cast<uintptr_t>(ln_gamma) | reinterpret_cast<uintptr_t>(ln_beta) | reinterpret_cast<uintptr_t>(ln_residual)) & 15) return CT_E_ALIGN;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x2) | reinterpret_cast<uintptr_t>(wp16) | reinterpret_cast<uintptr_t>(out) |
         reinterpret_cast<uintptr_t>(bias)) & 15) return CT_E_ALIGN;
    ct::Ws16Args a;
    const bool nsl = (k == 256) && (n % 128 == 0);
    const bool ksl = !nsl && (n == 128) && (k % 256 == 0) && !x2;
    const bool slab = !nsl && !ksl && (k == 128) && (n % 128 == 0) && !x2;     // out = [n/128][tokens][128]
    if (!nsl && !ksl && !slab) return CT_E_BADARG;
    a.S = ksl ? k / 256 : n / 128;
    if (a.S < 1 || a.S > 32 || (a.S != 3 && (32 % a.S))) return CT_E_BADARG;
    if (nsl) {
        if (x2 ? (k1 != 128) : (k1 != 256)) return CT_E_BADARG;
        a.xa = x; a.lda = x2 ? 128 : 256;
        a.xb = x2 ? x2 : x + 128; a.ldb = x2 ? 128 : 256;
        a.x_slice = 0; a.ldo = n; a.out_slice = 128; a.kslices = 0;
    } else if (ksl) {
        if (k1 != k || act != 0) return CT_E_BADARG;
        a.xa = x; a.xb = x + 128; a.lda = a.ldb = k;
        a.x_slice = 256; a.ldo = 128; a.out_slice = tokens * 128; a.kslices = 1;
    } else {
        if (k1 != k) return CT_E_BADARG;
        a.xa = x; a.xb = x + 64; a.lda = a.ldb = 128;
        a.x_slice = 0; a.ldo = 128; a.out_slice = tokens * 128; a.kslices = 0;
    }
    if (tokens == 0) return CT_OK;
    a.wp = reinterpret_cast<const uint4 *>(wp16); a.w_exp = w_exp; a.bias = bias; a.out = out; a.T = tokens;
    const long long nt = (tokens + 31) / 32;
    if (nt > 0x7fffffffLL) return CT_E_BADARG;
    a.n_tiles = (int)nt; a.act = act;
    a.ln_g = ln_gamma; a.ln_b = ln_beta; a.ln_res = ln_residual;
    static ct::DynLdsAttr attr8, attr4;  // per device
    if (attr8.ensure(reinterpret_cast<const void *>(ct::linear_ws16_kernel<8>), ct::w16_img(8) * 16 + 32768) != hipSuccess ||
        attr4.ensure(reinterpret_cast<const void *>(ct::linear_ws16_kernel<4>), ct::w16_img(4) * 16 + 32768) != hipSuccess)
        return CT_E_BADARG;
    if (slab) hipLaunchKernelGGL(ct::linear_ws16_kernel<4>, dim3(256), dim3(512), ct::w16_img(4) * 16 + 32768, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ct::linear_ws16_kernel<8>, dim3(256), dim3(512), ct::w16_img(8) * 16 + 32768, (hipStream_t)stream, a);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // extern "C"
