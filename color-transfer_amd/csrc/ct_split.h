// ct_split.h -- float32 -> three bf16 pieces (hi + mid + lo) and the activation epilogue shared by the split-bf16 MFMA
// convolutions (conv_split.hip: input tile shared by a workgroup; conv_ws.hip: weights stationary in registers).
#pragma once
#include <hip/hip_runtime.h>

namespace ct {

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool GEN>
__device__ __forceinline__ float split_act(float v, int act) {
    if (!GEN) return v > 0.f ? v : 0.01f * v;
    switch (act) {
        case 1: return v > 0.f ? v : 0.01f * v;
        case 2: return v > 0.f ? v : 0.f;
        case 3: return 1.0f / (1.0f + expf(-v));
        case 4: return tanhf(v);
        case 5: return v / (1.0f + expf(-v));        // swish
        default: return v;
    }
}

// x -> (hi, mid, lo) bf16 bit patterns; hi + mid + lo == x up to 2^-24 relative.  NaN stays NaN; an infinity becomes
// (inf, NaN, NaN), i.e. an infinite activation yields NaN outputs where the f32 kernel yields +-inf / NaN.
__device__ __forceinline__ void split3(float x, unsigned int &h, unsigned int &m, unsigned int &l) {
    const __bf16 bh = (__bf16)x;
    const float r1 = x - (float)bh;
    const __bf16 bm = (__bf16)r1;
    const float r2 = r1 - (float)bm;
    const __bf16 bl = (__bf16)r2;
    h = __builtin_bit_cast(unsigned short, bh);
    m = __builtin_bit_cast(unsigned short, bm);
    l = __builtin_bit_cast(unsigned short, bl);
}

// two values at once, packed (x0 in the low half): one v_cvt_pk_bf16_f32 per piece, halves re-expanded by shift / mask
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split3x2(float x0, float x1, unsigned int &hw, unsigned int &mw, unsigned int &lw) {
    hw = pack_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(hw << 16), r1 = x1 - __uint_as_float(hw & 0xffff0000u);
    mw = pack_bf16(r0, r1);
    lw = pack_bf16(r0 - __uint_as_float(mw << 16), r1 - __uint_as_float(mw & 0xffff0000u));
}

// ---- two fp16 pieces (hi + lo, 11 + 11 mantissa bits) with power-of-two scales: conv_split.hip's F16 form ----
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2s __attribute__((ext_vector_type(2)));
// opaque to the compiler (it otherwise re-derives each half with v_fma_mixlo_f16 when the halves are converted back)
__device__ __forceinline__ unsigned int sp16_cvt_pk(float a, float b) {
    unsigned int r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void sp16_split2x2(float x0, float x1, unsigned int &hw, unsigned int &lw) {
    hw = sp16_cvt_pk(x0, x1);
    const f16x2s h = __builtin_bit_cast(f16x2s, hw);
    lw = sp16_cvt_pk(x0 - (float)h.x, x1 - (float)h.y);
}
// exponent e with 2^e * mx in [2^11, 2^12); `none` for mx == 0 / denormal; 0 for inf / NaN (they propagate)
__device__ __forceinline__ int sp16_scale_exp(float mx, int none) {
    const int fld = (int)(__float_as_uint(mx) >> 23);
    const int ex = fld == 0 ? none : fld == 255 ? 0 : 138 - fld;
    return min(max(ex, -100), 100);
}
__device__ __forceinline__ float sp16_pow2i(int e) { return __uint_as_float((unsigned int)(127 + e) << 23); }   // |e| <= 126
// max over the wave of a non-negative float; every lane returns it
__device__ __forceinline__ float sp16_wave_max(float v) {
    int x = __float_as_int(v);
#define CT_DPP_MAX(ctrl, rmask) x = max(x, __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false))
    CT_DPP_MAX(0x111, 0xf); CT_DPP_MAX(0x112, 0xf); CT_DPP_MAX(0x114, 0xf); CT_DPP_MAX(0x118, 0xf);
    CT_DPP_MAX(0x142, 0xa); CT_DPP_MAX(0x143, 0xc);
#undef CT_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

}  // namespace ct
