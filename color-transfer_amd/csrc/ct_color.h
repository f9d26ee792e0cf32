// ct_color.h -- device-side sRGB <-> CIE-Lab arithmetic for gfx950, float64 with
// float32 hardware seeds.
//
// Semantics follow scikit-image 0.18.3 colorconv.py as called by the reference
// (methods/linear.py:25,26,40): rgb2xyz l.657-661, xyz2lab l.950-969, lab2xyz
// l.1010-1032, xyz2rgb l.612-619.  Accuracy target: <= 1e-11 relative on every
// intermediate, so that Lab agrees with the float64 reference to ~1e-9 (gate: 1e-4).
//
// Why float64: a* = 500 (fx - fy) amplifies a 1-ulp float32 error in f() to 6e-5, i.e.
// a float32 pipeline sits AT the 1e-4 gate.  CDNA4 issues
// v_fma_f64 at half the plain-f32 rate, which is far cheaper than any float-float
// emulation.  Each power function is one v_log_f32 + v_exp_f32 seed (rel. error
// < 5e-7) followed by ONE first-order correction in float64 built from the exact
// residual e = x^a * seed^b (== (1+eps)^b); the neglected term is O(eps^2) < 3e-12.
// No divisions, no f64 transcendentals.
#pragma once
#include <hip/hip_runtime.h>

namespace ct {

// ---- constants (printed with 17 significant digits from oracle/lab.py) -------------------
// xyz_from_rgb rows divided by the D65 white (0.95047, 1.0, 1.08883)
#define CT_M00 0.4339463633781182
#define CT_M01 0.3762138731364483
#define CT_M02 0.1898250339305817
#define CT_M10 0.212671
#define CT_M11 0.71516
#define CT_M12 0.072169
#define CT_M20 0.017756674595666908
#define CT_M21 0.10946887943939825
#define CT_M22 0.8727046462716862
// inv(xyz_from_rgb) columns multiplied by the D65 white
#define CT_I00 3.0799803022718044
#define CT_I01 -1.5371515162713185
#define CT_I02 -0.5428213080224701
#define CT_I10 -0.9212477523232383
#define CT_I11 1.8759900014898907
#define CT_I12 0.045247339514465995
#define CT_I20 0.05289046109881183
#define CT_I21 -0.20404133836651123
#define CT_I22 1.1512320119619401

__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// sRGB gamma expansion: c > 0.04045 ? ((c + 0.055) / 1.055) ** 2.4 : c / 12.92
__device__ __forceinline__ double srgb_to_linear(double c) {
    const double t = fma(c, 1.0 / 1.055, 0.055 / 1.055);
    // seed z0 ~ t^-0.6 ; exact: t^2.4 = t^3 * z with z^5 = t^-3
    const double z0 = (double)hw_exp2(-0.6f * hw_log2((float)t));
    const double t2 = t * t;
    const double z2 = z0 * z0;
    const double q = (t2 * t) * z0;        // t^3 z0
    const double e = q * (z2 * z2);        // t^3 z0^5 = (1+eps)^5
    const double pw = q * fma(-0.2, e, 1.2);
    const double lin = c * (1.0 / 12.92);
    return (c > 0.04045) ? pw : lin;
}

// Lab f(): v > 0.008856 ? cbrt(v) : 7.787 v + 16/116
__device__ __forceinline__ double lab_f_cbrt(double v) {
    // seed r0 ~ v^-1/3 ; cbrt(v) = v r^2 with r^3 = 1/v
    const double r0 = (double)hw_exp2((-1.0f / 3.0f) * hw_log2((float)v));
    const double b = (v * r0) * r0;        // v r0^2
    const double e = b * r0;               // v r0^3 = (1+eps)^3
    return b * fma(-2.0 / 3.0, e, 5.0 / 3.0);
}
__device__ __forceinline__ double lab_f(double v) {
    const double cb = lab_f_cbrt(v);
    const double lin = fma(7.787, v, 16.0 / 116.0);
    return (v > 0.008856) ? cb : lin;
}

// inverse of lab_f: t > 0.2068966 ? t^3 : (t - 16/116) / 7.787
__device__ __forceinline__ double lab_finv(double t) {
    const double cube = (t * t) * t;
    const double lin = fma(t, 1.0 / 7.787, -(16.0 / 116.0) / 7.787);
    return (t > 0.2068966) ? cube : lin;
}

// sRGB gamma compression: u > 0.0031308 ? 1.055 u^(1/2.4) - 0.055 : 12.92 u   (no clip)
__device__ __forceinline__ double linear_to_srgb(double u) {
    // seed s0 ~ u^-1/12 ; u^(5/12) = u s^7 with s^12 = 1/u
    const double s0 = (double)hw_exp2((-1.0f / 12.0f) * hw_log2((float)u));
    const double s2 = s0 * s0;
    const double s4 = s2 * s2;
    const double s5 = s4 * s0;
    const double p = u * (s5 * s2);        // u s0^7
    const double e = p * s5;               // u s0^12 = (1+eps)^12   (7 multiplications instead of 9)
    const double pw = p * fma(-7.0 / 12.0, e, 19.0 / 12.0);
    const double g = fma(1.055, pw, -0.055);
    const double lin = 12.92 * u;
    return (u > 0.0031308) ? g : lin;
}

// rgb (sRGB, [0,1]) -> (fx, fy, fz), the cube-root domain Lab is affine in
__device__ __forceinline__ void rgb_to_f(double r, double g, double b, double &fx, double &fy, double &fz) {
    const double lr = srgb_to_linear(r), lg = srgb_to_linear(g), lb = srgb_to_linear(b);
    const double x = fma(lb, CT_M02, fma(lg, CT_M01, lr * CT_M00));
    const double y = fma(lb, CT_M12, fma(lg, CT_M11, lr * CT_M10));
    const double z = fma(lb, CT_M22, fma(lg, CT_M21, lr * CT_M20));
    // The linear toe (v <= 0.008856, i.e. L* < 8) is rare and spatially coherent: evaluate it
    // only in waves where some lane needs it (wave-uniform branch, identical results).
    fx = lab_f_cbrt(x);
    fy = lab_f_cbrt(y);
    fz = lab_f_cbrt(z);
    const bool toe = !((x > 0.008856) & (y > 0.008856) & (z > 0.008856));   // also true for NaN
    if (__builtin_amdgcn_ballot_w64(toe)) {
        asm volatile("; lab toe" : "+v"(fx));   // keeps this a real branch (the compiler would flatten it into selects)
        fx = (x > 0.008856) ? fx : fma(7.787, x, 16.0 / 116.0);
        fy = (y > 0.008856) ? fy : fma(7.787, y, 16.0 / 116.0);
        fz = (z > 0.008856) ? fz : fma(7.787, z, 16.0 / 116.0);
    }
}

__device__ __forceinline__ void f_to_lab(double fx, double fy, double fz, double &L, double &a, double &b) {
    L = fma(116.0, fy, -16.0);
    a = 500.0 * (fx - fy);
    b = 200.0 * (fy - fz);
}

// (fx, fy, fz) -> sRGB, unclipped (clip happens after the cast to the output type)
__device__ __forceinline__ void f_to_rgb(double fx, double fy, double fz, double &r, double &g, double &b) {
    fz = (fz < 0.0) ? 0.0 : fz;            // lab2xyz: z < 0 -> 0 (NaN stays NaN)
    double x = (fx * fx) * fx, y = (fy * fy) * fy, z = (fz * fz) * fz;
    const bool toe = !((fx > 0.2068966) & (fy > 0.2068966) & (fz > 0.2068966));
    if (__builtin_amdgcn_ballot_w64(toe)) {  // wave-uniform: the linear toe is rare
        asm volatile("; lab toe" : "+v"(x));
        x = (fx > 0.2068966) ? x : fma(fx, 1.0 / 7.787, -(16.0 / 116.0) / 7.787);
        y = (fy > 0.2068966) ? y : fma(fy, 1.0 / 7.787, -(16.0 / 116.0) / 7.787);
        z = (fz > 0.2068966) ? z : fma(fz, 1.0 / 7.787, -(16.0 / 116.0) / 7.787);
    }
    const double lr = fma(z, CT_I02, fma(y, CT_I01, x * CT_I00));
    const double lg = fma(z, CT_I12, fma(y, CT_I11, x * CT_I10));
    const double lb = fma(z, CT_I22, fma(y, CT_I21, x * CT_I20));
    r = linear_to_srgb(lr);
    g = linear_to_srgb(lg);
    b = linear_to_srgb(lb);
}

template <typename T>
__device__ __forceinline__ T clip01(double v) {
    T o = (T)v;
    o = (o < (T)0) ? (T)0 : o;             // NaN compares false twice -> NaN propagates like np.clip
    o = (o > (T)1) ? (T)1 : o;
    return o;
}

}  // namespace ct
