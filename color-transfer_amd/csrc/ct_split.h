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

}  // namespace ct
