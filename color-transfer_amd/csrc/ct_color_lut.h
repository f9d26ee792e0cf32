// ct_color_lut.h -- table-driven sRGB <-> Lab arithmetic for float32 images on gfx950.
//
// Same functions as ct_color.h (scikit-image 0.18.3 rgb2lab / lab2rgb as called at methods/linear.py:25,26,40), but every
// power function is ONE look-up in an LDS-resident table plus a short polynomial instead of a v_log_f32 / v_exp_f32 seed
// and a float64 Newton correction.  Why (measured, tools/ubench/lut_rates.hip): on this chip a float64 VALU op costs
// ~2.1 ns per wave and SIMD, a float32 op ~1.2 ns, a transcendental ~3.6 ns, a float64 select ~8 ns, while a random
// 8-byte / 16-byte LDS look-up costs the CU 7.3 / 11.9 cycles per wave (bank conflicts included).  The exact path spends
// ~74 cycles per gamma expansion and ~46 per cube root; the table path ~24 and ~30.
//
// Accuracy (tools/gen_lab_tables.py verifies each table against 40-digit arithmetic): linear values <= 1.6e-10 absolute,
// cube roots <= 1e-9 relative, gamma compression <= 6e-8 absolute (the float32 output rounding is 3e-8).  In Lab that is
// ~5e-7 before the float32 output rounding -- two orders below the 1e-4 gate; the statistics agree with the float64
// path to ~1e-7.  Inputs outside [0,1], NaNs and non-finite statistics never enter this path: the kernels test for them
// (wave-uniform) and fall back to ct_color.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ct_color.h"
#include "ct_lab_tables.h"

namespace ct {
namespace lut {

// LDS images of the tables (byte offsets).  float64-grade image (apply sweep): A {double a0; float a1, a2} 16 B,
// B double r = v^(-1/3) 8 B, C {float a0..a3} 16 B.  float32 image (statistics sweep): A32 and B32, {c, s1, s2, node} 16 B.
constexpr int kLdsA = 0;
constexpr int kLdsB = kLdsA + kAEntries * 16;
constexpr int kLdsC = kLdsB + kBEntries * 8;
constexpr int kCFirst = 6 << kCBits;                        // first entry of table C that is ever read (2^-9)
constexpr int kLdsBytesFwd = kLdsC;                         // forward transform only (A, B)
constexpr int kLdsBytesAll = kLdsC + (kCEntries - kCFirst) * 16;
constexpr int kLdsBytesF32 = kLdsB + kB32Entries * 16;
static_assert(kLdsB % 16 == 0 && kLdsC % 16 == 0, "16-byte aligned tables");
static_assert(kLdsC >= kCFirst * 16, "table C is addressed with a negative bias");

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
template <int THREADS, bool WITH_C>
__device__ __forceinline__ void load_tables(unsigned char *lds) {
    copy16<THREADS, kAEntries>(reinterpret_cast<const uint4 *>(kTableA), reinterpret_cast<uint4 *>(lds + kLdsA));
    copy16<THREADS, kBEntries / 2>(reinterpret_cast<const uint4 *>(kTableB), reinterpret_cast<uint4 *>(lds + kLdsB));
    if (WITH_C)
        copy16<THREADS, kCEntries - kCFirst>(reinterpret_cast<const uint4 *>(kTableC) + kCFirst, reinterpret_cast<uint4 *>(lds + kLdsC));
}
template <int THREADS>
__device__ __forceinline__ void load_tables_f32(unsigned char *lds) {
    copy16<THREADS, kAEntries>(reinterpret_cast<const uint4 *>(kTableA32), reinterpret_cast<uint4 *>(lds + kLdsA));
    copy16<THREADS, kB32Entries>(reinterpret_cast<const uint4 *>(kTableB32), reinterpret_cast<uint4 *>(lds + kLdsB));
}

// float32 bit pattern test: every value of a tile in [0,1] <=> max of the patterns (unsigned) <= bits(1.0f)
// (negative numbers, -0, NaN and inf all have larger patterns)
constexpr uint32_t kOneBits = 0x3f800000u;

// diagnostic builds only (-DCT_LUT_ABLATE_LDS): every look-up reads one of two entries -> no bank conflicts, wrong
// results; the time difference to the real build is what the conflicts cost
#ifdef CT_LUT_ABLATE_LDS
#define CT_LUT_OFF(off, unit) ((off) & (unit))
#else
#define CT_LUT_OFF(off, unit) (off)
#endif

// ---- float64-grade pieces (apply sweep) ---------------------------------------------------------------------------
// sRGB gamma expansion of a float32 in [0,1] (c slightly outside extrapolates the end segments)
__device__ __forceinline__ double expand(const unsigned char *lds, float c) {
    const float y = fmaf(c, kAScale, kMagic);                                   // MAGIC + round(c * S)
    const uint32_t off = (__float_as_uint(y) << 4) - (kMagicBits << 4);         // 16 * index
    const float d = fmaf(y - kMagic, kANegInv, c);                              // c - index / S
    const uint4 e = *reinterpret_cast<const uint4 *>(lds + kLdsA + CT_LUT_OFF(off, 16));        // one ds_read_b128: {a0 (double), a1, a2}
    const double a0 = __hiloint2double((int)e.y, (int)e.x);
    return a0 + (double)(d * fmaf(d, __uint_as_float(e.w), __uint_as_float(e.z)));
}

// cube root of a float64 in [2^-7, 2): r = v^(-1/3) at the nearest of 256 nodes per octave, then one quadratic in v r^3
// (one 8-byte look-up: the apply sweep is short of LDS bandwidth, not of float64 multiplies)
__device__ __forceinline__ double cbrt_lut(const unsigned char *lds, double v) {
    const uint32_t hi = (uint32_t)__double2hiint(v);
    const uint32_t off = ((hi + (1u << (19 - kBBits))) >> (17 - kBBits)) & (((8u << kBBits) - 1u) << 3);
    const double r = *reinterpret_cast<const double *>(lds + kLdsB + CT_LUT_OFF(off, 8));
    const double t = v * r, b = t * r, e = b * r;
    return b * fma(fma(kBQ2, e, kBQ1), e, kBQ0);
}

constexpr int32_t kToeHiFwd = 0x3f822318;    // high word of 0.008856: hi(v) > this  =>  v > 0.008856
constexpr int32_t kToeHiInv = 0x3fca7b96;    // high word of 0.2068966

// one pixel (float32, all three in [0,1]) -> (fx, fy, fz).  The linear toe of Lab's f() is patched behind ONE integer
// test on the high words (positive doubles order like their bit patterns): the float64 selects run only when some
// lane of the wave needs them.
__device__ __forceinline__ void lin_to_f(const unsigned char *lds, double lr, double lg, double lb, double &fx, double &fy, double &fz);
__device__ __forceinline__ void rgb_to_f(const unsigned char *lds, float r, float g, float b, double &fx, double &fy, double &fz) {
    lin_to_f(lds, expand(lds, r), expand(lds, g), expand(lds, b), fx, fy, fz);
}
// the same from linear (gamma-expanded) values: uint8 frames look their 256 possible expansions up (reinhard_persist.hip)
__device__ __forceinline__ void lin_to_f(const unsigned char *lds, double lr, double lg, double lb, double &fx, double &fy, double &fz) {
    const double x = fma(lb, CT_M02, fma(lg, CT_M01, lr * CT_M00));
    const double y = fma(lb, CT_M12, fma(lg, CT_M11, lr * CT_M10));
    const double z = fma(lb, CT_M22, fma(lg, CT_M21, lr * CT_M20));
    fx = cbrt_lut(lds, x);
    fy = cbrt_lut(lds, y);
    fz = cbrt_lut(lds, z);
    const int32_t m = min(min(__double2hiint(x), __double2hiint(y)), __double2hiint(z));
    if (__builtin_amdgcn_ballot_w64(m <= kToeHiFwd)) {
        asm volatile("; lab toe" : "+v"(fx));
        fx = (x > 0.008856) ? fx : fma(7.787, x, 16.0 / 116.0);
        fy = (y > 0.008856) ? fy : fma(7.787, y, 16.0 / 116.0);
        fz = (z > 0.008856) ? fz : fma(7.787, z, 16.0 / 116.0);
    }
}

// sRGB gamma compression + clip to [0,1], float32 result (u: linear value, any finite double)
__device__ __forceinline__ float compress_clip(const unsigned char *lds, double u) {
    const float uf = __builtin_amdgcn_fmed3f((float)u, 0.0f, 1.0f);
    const uint32_t bits = __float_as_uint(uf) + (1u << (22 - kCBits));
    const uint32_t off = (bits >> (19 - kCBits)) & (((16u << kCBits) - 1u) << 4);
    const float d = uf - __uint_as_float(bits & ~((1u << (23 - kCBits)) - 1u));
    const float4 e = *reinterpret_cast<const float4 *>(lds + (kLdsC - kCFirst * 16) + CT_LUT_OFF(off, 16));   // {a0, a1, a2, a3}
    const float p = fmaf(d, fmaf(d, fmaf(d, e.w, e.z), e.y), e.x);
    return (uf <= 0.0031308f) ? 12.92f * uf : p;
}

// one pixel: (fx, fy, fz), all finite -> clipped float32 sRGB
__device__ __forceinline__ void f_to_rgb_clip(const unsigned char *lds, double fx, double fy, double fz, float &r, float &g, float &b) {
    double x = (fx * fx) * fx, y = (fy * fy) * fy, z = (fz * fz) * fz;
    const int32_t m = min(min(__double2hiint(fx), __double2hiint(fy)), __double2hiint(fz));   // negative doubles: negative ints
    if (__builtin_amdgcn_ballot_w64(m <= kToeHiInv)) {
        asm volatile("; lab toe" : "+v"(x));
        fz = (fz < 0.0) ? 0.0 : fz;            // lab2xyz: z < 0 -> 0
        z = (fz * fz) * fz;
        x = (fx > 0.2068966) ? x : fma(fx, 1.0 / 7.787, -(16.0 / 116.0) / 7.787);
        y = (fy > 0.2068966) ? y : fma(fy, 1.0 / 7.787, -(16.0 / 116.0) / 7.787);
        z = (fz > 0.2068966) ? z : fma(fz, 1.0 / 7.787, -(16.0 / 116.0) / 7.787);
    }
    r = compress_clip(lds, fma(z, CT_I02, fma(y, CT_I01, x * CT_I00)));
    g = compress_clip(lds, fma(z, CT_I12, fma(y, CT_I11, x * CT_I10)));
    b = compress_clip(lds, fma(z, CT_I22, fma(y, CT_I21, x * CT_I20)));
}

// ---- float32 pieces (statistics sweep; tables from load_tables_f32) --------------------------------------------------
// Means and variances over ~2 M pixels only need UNBIASED per-pixel values: correctly rounded float32 arithmetic
// (errors ~6e-8, symmetric) changes the Lab statistics by ~1e-7, far below the float32 rounding of any output pixel,
// at half the VALU cost of float64.  (The apply sweep keeps float64: there a = 500 (fx - fy) must be right per pixel.)
__device__ __forceinline__ float expand32(const unsigned char *lds, float c) {
    const float y = fmaf(c, kAScale, kMagic);
    const uint32_t off = (__float_as_uint(y) << 4) - (kMagicBits << 4);
    const float4 e = *reinterpret_cast<const float4 *>(lds + kLdsA + CT_LUT_OFF(off, 16));      // {a0, a1, a2, node}
    const float d = c - e.w;
    return fmaf(d, fmaf(d, e.z, e.y), e.x);
}

// cube root of a float32 in [2^-7, 2): quadratic around the nearest of 128 nodes per octave
__device__ __forceinline__ float cbrt32(const unsigned char *lds, float v) {
#ifdef CT_STATS_CBRT_HW
    // no look-up: r ~ v^(-1/3) from the hardware log2 / exp2 (~5e-7 relative), y0 = v r^2, one correction step in the
    // residual g = y0 r = v r^3: y = y0 g^(-2/3) ~ y0 (1 + 2/3 (1 - g)); the residual comes out of one fma exactly, so what
    // is left is a third of y0's two roundings plus the final one (< 8e-8 relative, symmetric)
    const float r = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v) * (-1.0f / 3.0f));
    const float u = (v * r) * r;
    const float e = fmaf(-u, r, 1.0f);
    return fmaf(u * (2.0f / 3.0f), e, u);
#else
    const uint32_t bits = __float_as_uint(v) + (1u << (22 - kB32Bits));
    const uint32_t off = (bits >> (19 - kB32Bits)) & (((8u << kB32Bits) - 1u) << 4);
    const float4 e = *reinterpret_cast<const float4 *>(lds + kLdsB + CT_LUT_OFF(off, 16));      // {c, s1, s2, node}
    const float d = v - e.w;                                                    // exact
    return fmaf(d, fmaf(d, e.z, e.y), e.x);
#endif
}

// matrix row in float32 with two-piece constants: a constant rounded to float32 is off by up to 3e-8 relative for EVERY
// pixel -- a bias, not noise (measured: 2.5e-6 in the mean of a*) -- so each weight is hi + lo and the lo terms go first
#define CT_ROW32(lr, lg, lb, A, B, C)                                                                                                  \
    fmaf(lb, (float)(C), fmaf(lg, (float)(B), fmaf(lr, (float)(A),                                                                     \
         fmaf(lb, (float)((C) - (double)(float)(C)), fmaf(lg, (float)((B) - (double)(float)(B)), lr * (float)((A) - (double)(float)(A)))))))

// one pixel (float32, all three in [0,1]) -> (fx, fy, fz).  The linear toe of Lab's f() is rare for most images, so it sits
// behind one test per pixel: the selects run only when some lane of the wave needs them.  (Moving all four pixels of a
// lane through the tables in phases -- twelve look-ups in flight -- was measured 4 % slower: more live registers, and the
// LDS latency is already covered by the other waves.)
__device__ __forceinline__ void rgb_to_f32(const unsigned char *lds, float r, float g, float b, float &fx, float &fy, float &fz) {
    const float lr = expand32(lds, r), lg = expand32(lds, g), lb = expand32(lds, b);
    const float x = CT_ROW32(lr, lg, lb, CT_M00, CT_M01, CT_M02);
    const float y = CT_ROW32(lr, lg, lb, CT_M10, CT_M11, CT_M12);
    const float z = CT_ROW32(lr, lg, lb, CT_M20, CT_M21, CT_M22);
    fx = cbrt32(lds, x);
    fy = cbrt32(lds, y);
    fz = cbrt32(lds, z);
    if (__builtin_amdgcn_ballot_w64(fminf(fminf(x, y), z) <= 0.008856f)) {
        asm volatile("; lab toe" : "+v"(fx));
        fx = (x > 0.008856f) ? fx : fmaf(7.787f, x, (float)(16.0 / 116.0));
        fy = (y > 0.008856f) ? fy : fmaf(7.787f, y, (float)(16.0 / 116.0));
        fz = (z > 0.008856f) ? fz : fmaf(7.787f, z, (float)(16.0 / 116.0));
    }
}

// ---- float32 pieces on the float64-grade tables (reinhard_persist.hip: its LDS has no room for A32 / B32) -----------------------
// gamma expansion from table A: a0 rounded to float32 (to nearest: the error differs from node to node, no common bias), the
// node distance from the index exactly as expand() forms it
__device__ __forceinline__ float expand32_a(const unsigned char *lds, float c) {
    const float y = fmaf(c, kAScale, kMagic);
    const uint32_t off = (__float_as_uint(y) << 4) - (kMagicBits << 4);
    const float d = fmaf(y - kMagic, kANegInv, c);
    const uint4 e = *reinterpret_cast<const uint4 *>(lds + kLdsA + CT_LUT_OFF(off, 16));
    const float a0 = (float)__hiloint2double((int)e.y, (int)e.x);
    return fmaf(d, fmaf(d, __uint_as_float(e.w), __uint_as_float(e.z)), a0);
}
// cube root without a table: r ~ v^(-1/3) from the hardware log2 / exp2 (~5e-7 relative), u = v r^2, one correction step in the
// residual u r = v r^3 (exact out of one fma): < 8e-8 relative, symmetric (measured round 3: as fast as the B32 look-up)
__device__ __forceinline__ float cbrt32_hw(float v) {
    const float r = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(v) * (-1.0f / 3.0f));
    const float u = (v * r) * r;
    const float e = fmaf(-u, r, 1.0f);
    return fmaf(u * (2.0f / 3.0f), e, u);
}
__device__ __forceinline__ void lin32_to_f32(float lr, float lg, float lb, float &fx, float &fy, float &fz);
__device__ __forceinline__ void rgb_to_f32_a(const unsigned char *lds, float r, float g, float b, float &fx, float &fy, float &fz) {
    lin32_to_f32(expand32_a(lds, r), expand32_a(lds, g), expand32_a(lds, b), fx, fy, fz);
}
__device__ __forceinline__ void lin32_to_f32(float lr, float lg, float lb, float &fx, float &fy, float &fz) {
    const float x = CT_ROW32(lr, lg, lb, CT_M00, CT_M01, CT_M02);
    const float y = CT_ROW32(lr, lg, lb, CT_M10, CT_M11, CT_M12);
    const float z = CT_ROW32(lr, lg, lb, CT_M20, CT_M21, CT_M22);
    fx = cbrt32_hw(x);
    fy = cbrt32_hw(y);
    fz = cbrt32_hw(z);
    if (__builtin_amdgcn_ballot_w64(fminf(fminf(x, y), z) <= 0.008856f)) {
        asm volatile("; lab toe" : "+v"(fx));
        fx = (x > 0.008856f) ? fx : fmaf(7.787f, x, (float)(16.0 / 116.0));
        fy = (y > 0.008856f) ? fy : fmaf(7.787f, y, (float)(16.0 / 116.0));
        fz = (z > 0.008856f) ? fz : fmaf(7.787f, z, (float)(16.0 / 116.0));
    }
}

}  // namespace lut
}  // namespace ct
