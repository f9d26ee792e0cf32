// ct_color_lut.h -- table-driven sRGB <-> Lab arithmetic for float32 images on gfx950, float32 throughout.
//
// Same functions as ct_color.h (scikit-image 0.18.3 rgb2lab / lab2rgb as called at methods/linear.py:25,26,40).  Round 5: the
// Reinhard sweeps were bound by vector-instruction issue (~300 instructions per pixel pair, a third of them float64
// at twice the issue cost), so this path now issues NO float64 instruction per pixel and keeps the precision where the 1e-4
// Lab gate needs it with float32 "difference forms" instead:
//
//  * every power function is ONE 16-byte LDS look-up {a0, a1, a2, node} + d = x - node + two fma (tables E, F, G of
//    ct_lab_tables.h; tools/gen_lab_tables.py carries the numpy model of everything below and verifies it against mpmath);
//  * the entry address costs two instructions (E: fma with a magic number + and; F, G: shift + and on the float's own bits);
//  * Lab's f() is tabulated on u = v + c0: the linear toe (v <= 0.008856) is part of the table -- no toe test, no selects -- and
//    c0 rides in the first fma of the matrix row; gamma compression likewise on w = u + c1, the clip to [0,1] is one v_med3_f32;
//  * a0 of table F is a multiple of 2^-24: fx - fy = (a0x - a0y) + (rx - ry) with an EXACT first difference, and the arguments of
//    the X and Y rows are two-term values (the rounding error of the last fma of the row, recovered with one more fma), because
//    a* = 500 (fx - fy) amplifies the cube roots' errors 500 times;
//  * the inverse carries y = gy^3, x - y and y - z (difference form of the cubes) into a matrix whose rows sum to ~1:
//    lin = rho y + i0 (x - y) + i2 (y - z): two fma less per channel AND no cancellation of big terms; blue is built on z (a small
//    z -- yellow -- must not inherit the rounding of y).
//
// Accuracy (model: tools/gen_lab_tables.py, tools/model_reinhard_f32.py; measured: tests/test_linear_gpu.py): Lab of the
// transferred image <= 3e-5 and Lab of the final RGB <= 4e-5 against the float64 oracle for affine scales up to 2 (gate 1e-4;
// the forward error grows with the scale, so the kernels send scales above 4 through the exact float64 code).  float32 RGB
// within 5e-6 (saturated colours: a channel that is the small difference of big terms; 1/800 of an 8-bit step).
//
// The reference's f() and its inverse JUMP at their kinks (0.008856 is not (6/29)^3: f jumps by 3.4e-7 = 1.7e-4 in a*), so a
// pixel whose argument lies within the rounding error of a kink must be classified exactly like the float64 reference: both
// transforms flag such pixels (a handful per dark frame) and the kernels redo their tile with the exact code of ct_color.h.
// Inputs outside [0,1] and NaNs never enter this path: the kernels test for them (wave-uniform) and fall back as well.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ct_color.h"
#include "ct_lab_tables.h"

namespace ct {
namespace lut {

// LDS images of the tables (byte offsets)
constexpr int kLdsE = 0;
constexpr int kLdsF = kLdsE + kEEntries * 16;
constexpr int kLdsG = kLdsF + kFEntries * 16;
constexpr int kLdsBytesFwd = kLdsG;                         // forward transform only (E, F): statistics, Lab output
constexpr int kLdsBytesAll = kLdsG + kGEntries * 16;
constexpr int kLdsGBias = kLdsG - kGFirst * 16;             // table G is addressed with (exponent & 15): a negative bias
static_assert(kLdsGBias >= 0 && kLdsGBias + (((16 << kGBits) - 1) << 4) < 65536, "table G addressing");
static_assert(kLdsBytesAll % 16 == 0, "16-byte aligned tables");

// cooperative copy global (L2-resident) -> LDS: every thread first issues all its 16-byte loads, then stores them (one
// memory round trip for the whole image); caller synchronises
template <int THREADS, int N>
__device__ __forceinline__ void copy16(const uint4 *__restrict__ src, uint4 *dst) {
    constexpr int STEPS = (N + THREADS - 1) / THREADS;
    static_assert(STEPS <= 4, "tables of at most 4 x THREADS x 16 bytes");
    const int t = threadIdx.x;
    // named registers (an indexed local array ends up in scratch memory here)
    uint4 v0 = {}, v1 = {}, v2 = {}, v3 = {};
    if (STEPS > 0 && t < N) v0 = src[t];
    if (STEPS > 1 && THREADS + t < N) v1 = src[THREADS + t];
    if (STEPS > 2 && 2 * THREADS + t < N) v2 = src[2 * THREADS + t];
    if (STEPS > 3 && 3 * THREADS + t < N) v3 = src[3 * THREADS + t];
    if (STEPS > 0 && t < N) dst[t] = v0;
    if (STEPS > 1 && THREADS + t < N) dst[THREADS + t] = v1;
    if (STEPS > 2 && 2 * THREADS + t < N) dst[2 * THREADS + t] = v2;
    if (STEPS > 3 && 3 * THREADS + t < N) dst[3 * THREADS + t] = v3;
}
template <int THREADS, bool WITH_G>
__device__ __forceinline__ void load_tables(unsigned char *lds) {
    copy16<THREADS, kEEntries>(reinterpret_cast<const uint4 *>(kTableE), reinterpret_cast<uint4 *>(lds + kLdsE));
    copy16<THREADS, kFEntries>(reinterpret_cast<const uint4 *>(kTableF), reinterpret_cast<uint4 *>(lds + kLdsF));
    if (WITH_G) copy16<THREADS, kGEntries>(reinterpret_cast<const uint4 *>(kTableG), reinterpret_cast<uint4 *>(lds + kLdsG));
}

// float32 bit pattern test: every value of a tile in [0,1] <=> max of the patterns (unsigned) <= bits(1.0f)
// (negative numbers, -0, NaN and inf all have larger patterns)
constexpr uint32_t kOneBits = 0x3f800000u;

// diagnostic builds only (-DCT_LUT_ABLATE_LDS): every look-up reads one of two entries -> no bank conflicts, wrong
// results; the time difference to the real build is what the conflicts cost
#ifdef CT_LUT_ABLATE_LDS
#define CT_LUT_OFF(off) ((off) & 16)
#else
#define CT_LUT_OFF(off) (off)
#endif

// ---- the three look-ups ------------------------------------------------------------------------------------------------------
// sRGB gamma expansion of a float32 in [0,1]; a0 is stored exactly, so the result is within 0.15 ulp of correctly rounded
__device__ __forceinline__ float expand(const unsigned char *lds, float c) {
    const uint32_t off = __float_as_uint(fmaf(c, kEScale, kEMagic)) & kEMask;
    const float4 e = *reinterpret_cast<const float4 *>(lds + kLdsE + CT_LUT_OFF(off));
    const float d = c - e.w;
    return fmaf(d, fmaf(d, e.z, e.y), e.x);
}

__device__ __forceinline__ float4 f_entry(const unsigned char *lds, float u) {
    const uint32_t off = (__float_as_uint(u) >> (19 - kFBits)) & (((8u << kFBits) - 1u) << 4);
    return *reinterpret_cast<const float4 *>(lds + kLdsF + CT_LUT_OFF(off));
}
// Lab's f(u - c0) for u in [c0, 1 + c0], rounded once (statistics: unbiased per-pixel values)
__device__ __forceinline__ float lab_f(const unsigned char *lds, float u) {
    const float4 e = f_entry(lds, u);
    const float d = u - e.w;
    return fmaf(d, fmaf(d, e.z, e.y), e.x);
}
// the same as grid part + remainder: f = a0 + r, a0 a multiple of 2^-24, |r| < 2^-9; ul: low part of a two-term argument
__device__ __forceinline__ void lab_f_parts(const unsigned char *lds, float u, float ul, float &a0, float &r) {
    const float4 e = f_entry(lds, u);
    const float d = (u - e.w) + ul;
    a0 = e.x;
    r = d * fmaf(d, e.z, e.y);
}

// sRGB gamma compression + clip to [0,1] of w = u + c1 (any finite float)
__device__ __forceinline__ float compress_clip(const unsigned char *lds, float w) {
    const float wc = __builtin_amdgcn_fmed3f(w, kGShift, kGHi);
    const uint32_t off = (__float_as_uint(wc) >> (19 - kGBits)) & (((16u << kGBits) - 1u) << 4);
    const float4 e = *reinterpret_cast<const float4 *>(lds + kLdsGBias + CT_LUT_OFF(off));
    const float d = wc - e.w;
    return fmaf(d, fmaf(d, e.z, e.y), e.x);
}

// ---- forward: sRGB (all three in [0,1]) -> (fy, fx - fy, fy - fz) -------------------------------------------------------------------
// one matrix row on the shifted argument, smallest weight first (the early roundings happen on small partial sums)
#define CT_ROW_P(l, W, O) fmaf(l[kOrd[O][1]], W[1], fmaf(l[kOrd[O][0]], W[0], kFShift))
#define CT_ROW_U(l, W, O, p) fmaf(l[kOrd[O][2]], W[2], p)

// statistics: plain float32 values of f().  Means and variances over ~2 M pixels only need UNBIASED per-pixel values: correctly
// rounded float32 arithmetic (errors ~6e-8, symmetric) moves the Lab statistics by ~1e-6 (the float32 matrix weights carry a
// relative rounding of up to 3e-8 each, the same for every pixel), two orders below the gate.
__device__ __forceinline__ void rgb_to_f_stats(const unsigned char *lds, float r, float g, float b, float &fy, float &dxy, float &dyz) {
    const float l[3] = {expand(lds, r), expand(lds, g), expand(lds, b)};
    const float fx = lab_f(lds, CT_ROW_U(l, kMX, 0, CT_ROW_P(l, kMX, 0)));
    fy = lab_f(lds, CT_ROW_U(l, kMY, 1, CT_ROW_P(l, kMY, 1)));
    const float fz = lab_f(lds, CT_ROW_U(l, kMZ, 2, CT_ROW_P(l, kMZ, 2)));
    dxy = fx - fy;
    dyz = fy - fz;
}
__device__ __forceinline__ void lin_to_f_stats(const unsigned char *lds, const float (&l)[3], float &fy, float &dxy, float &dyz) {
    const float fx = lab_f(lds, CT_ROW_U(l, kMX, 0, CT_ROW_P(l, kMX, 0)));
    fy = lab_f(lds, CT_ROW_U(l, kMY, 1, CT_ROW_P(l, kMY, 1)));
    const float fz = lab_f(lds, CT_ROW_U(l, kMZ, 2, CT_ROW_P(l, kMZ, 2)));
    dxy = fx - fy;
    dyz = fy - fz;
}

// The same without table F: the statistics sweep is bound by the LDS pipe (a 16-byte look-up per lane costs the CU ~12 cycles,
// six of them per pixel), not by vector issue, so its three cube roots come from the transcendental unit instead: r ~ v^(-1/3)
// from v_log_f32 / v_exp_f32 (~5e-7 relative), y0 = v r^2, one correction step in the residual g = y0 r = v r^3 (exact out of
// one fma): y = y0 (1 + 2/3 (1 - g)), < 8e-8 relative, symmetric.  The linear toe sits behind one test per pixel.
__device__ __forceinline__ float cbrt_hw(float v) {
    const float r = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v) * (-1.0f / 3.0f));
    const float u = (v * r) * r;
    const float e = fmaf(-u, r, 1.0f);
    return fmaf(u * (2.0f / 3.0f), e, u);
}
#define CT_ROW0(l, W, O) fmaf(l[kOrd[O][2]], W[2], fmaf(l[kOrd[O][1]], W[1], l[kOrd[O][0]] * W[0]))
__device__ __forceinline__ void lin_to_f_stats_hw(const float (&l)[3], float &fy, float &dxy, float &dyz) {
    const float x = CT_ROW0(l, kMX, 0), y = CT_ROW0(l, kMY, 1), z = CT_ROW0(l, kMZ, 2);
    float fx = cbrt_hw(x);
    fy = cbrt_hw(y);
    float fz = cbrt_hw(z);
    if (__builtin_amdgcn_ballot_w64(fminf(fminf(x, y), z) <= 0.008856f)) {
        asm volatile("; lab toe" : "+v"(fx));
        fx = (x > 0.008856f) ? fx : fmaf(7.787f, x, (float)(16.0 / 116.0));
        fy = (y > 0.008856f) ? fy : fmaf(7.787f, y, (float)(16.0 / 116.0));
        fz = (z > 0.008856f) ? fz : fmaf(7.787f, z, (float)(16.0 / 116.0));
    }
    dxy = fx - fy;
    dyz = fy - fz;
}
__device__ __forceinline__ void rgb_to_f_stats_hw(const unsigned char *lds, float r, float g, float b, float &fy, float &dxy, float &dyz) {
    const float l[3] = {expand(lds, r), expand(lds, g), expand(lds, b)};
    lin_to_f_stats_hw(l, fy, dxy, dyz);
}

// apply: per-pixel accuracy.  Returns true (per lane) when an argument lies within the rounding error of the 0.008856 kink of
// f(), where the reference's function jumps: the caller redoes the tile with the exact code.
constexpr float kFBandF = 1.2e-8f;           // ~6 float32 ulps of u on either side of the kink (the argument is good to ~2)
constexpr float kFKinkHi = kFKink + kFBandF;
__device__ __forceinline__ bool lin_to_f(const unsigned char *lds, const float (&l)[3], float &fy, float &dxy, float &dyz) {
    const float px = CT_ROW_P(l, kMX, 0), py = CT_ROW_P(l, kMY, 1);
    const float ux = CT_ROW_U(l, kMX, 0, px), uy = CT_ROW_U(l, kMY, 1, py);
    const float uxl = fmaf(l[kOrd[0][2]], kMX[2], px - ux), uyl = fmaf(l[kOrd[1][2]], kMY[2], py - uy);     // the rounding error of the last fma
    const float uz = CT_ROW_U(l, kMZ, 2, CT_ROW_P(l, kMZ, 2));
    float ax, rx, ay, ry, az, rz;
    lab_f_parts(lds, ux, uxl, ax, rx);
    lab_f_parts(lds, uy, uyl, ay, ry);
    lab_f_parts(lds, uz, 0.0f, az, rz);
    fy = ay + ry;
    dxy = (ax - ay) + (rx - ry);             // the first difference is exact (multiples of 2^-24 below 1)
    dyz = (ay - az) + (ry - rz);
    bool near = false;
    const float mn = fminf(fminf(ux, uy), uz);
    if (__builtin_amdgcn_ballot_w64(mn <= kFKinkHi)) {                                   // some lane of the wave is in the toe
        asm volatile("; kink band" : "+v"(dxy));                                        // keeps this a real branch
        near = fminf(fminf(fabsf(ux - kFKink), fabsf(uy - kFKink)), fabsf(uz - kFKink)) <= kFBandF;
    }
    return near;
}
__device__ __forceinline__ bool rgb_to_f(const unsigned char *lds, float r, float g, float b, float &fy, float &dxy, float &dyz) {
    const float l[3] = {expand(lds, r), expand(lds, g), expand(lds, b)};
    return lin_to_f(lds, l, fy, dxy, dyz);
}

// ---- inverse: (gy, gx - gy, gy - gz), all finite and moderate -> clipped float32 sRGB --------------------------------------------------
// Returns true (per lane) when gx, gy or gz lies within kInvBand of the 0.2068966 kink of the inverse (the same jump).
constexpr float kInvBand = 4e-7f;            // |g| <~ 2 carries ~1e-7 of rounding
__device__ __forceinline__ bool f_to_rgb_clip(const unsigned char *lds, float gy, float dx, float dz, float &r, float &g, float &b) {
    const float gy2 = gy * gy;
    float y = gy2 * gy;
    const float t3 = 3.0f * gy2;
    float ux = dx * fmaf(dx, fmaf(3.0f, gy, dx), t3);           // x - y = dx (3 gy^2 + dx (3 gy + dx))
    float uz = dz * fmaf(-dz, fmaf(3.0f, gy, -dz), t3);         // y - z = dz (3 gy^2 - dz (3 gy - dz))
    const float gx = gy + dx, gz = gy - dz;
    float z = (gz * gz) * gz;
    bool near = false;
    const float mn = fminf(fminf(gx, gy), gz);
    if (__builtin_amdgcn_ballot_w64(mn <= kToeInv + kInvBand)) {
        // Some lane of the wave is in the linear toe of the inverse (t <= 0.2068966 -> (t - 16/116) / 7.787).  On uniform random
        // frames that is nearly every wave (4 % of the pixels), so this block is kept short: the cubes are already there in
        // difference form -- gx^3 = y + ux, gz^3 = y - uz -- and a component on the linear branch is tiny (<= 0.0089), where plain
        // differences are exact enough.
        asm volatile("; lab toe" : "+v"(y));
        near = fminf(fminf(fabsf(gx - kToeInv), fabsf(gy - kToeInv)), fabsf(gz - kToeInv)) <= kInvBand;
        const float gzc = fmaxf(gz, 0.0f);                      // lab2xyz: z < 0 -> 0
        const float xl = fmaf(gx, kToeA, kToeB), yl = fmaf(gy, kToeA, kToeB), zl = fmaf(gzc, kToeA, kToeB);
        const bool bx = gx > kToeInv, by = gy > kToeInv, bz = gz > kToeInv;
        const float yn = by ? y : yl;
        const float dyc = y - yn;                               // 0 where gy is on the cube branch
        ux = bx ? ux + dyc : xl - yn;                           // h(gx) - h(gy)
        uz = bz ? uz - dyc : yn - zl;                           // h(gy) - h(gz)
        z = bz ? z : zl;
        y = yn;
    }
    r = compress_clip(lds, fmaf(ux, kInvR[1], fmaf(uz, kInvR[2], fmaf(y, kInvR[0], kGShift))));
    g = compress_clip(lds, fmaf(ux, kInvG[1], fmaf(uz, kInvG[2], fmaf(y, kInvG[0], kGShift))));
    b = compress_clip(lds, fmaf(ux, kInvB[1], fmaf(uz, kInvB[2], fmaf(z, kInvB[0], kGShift))));
    return near;
}

}  // namespace lut
}  // namespace ct
