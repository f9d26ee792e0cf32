// distort.hip -- the deterministic colour distortions of the reference's artificial test set (utils/data.py:12-22,120-125:
// identity + {brightness, contrast, saturation, hue, gamma} x 6 magnitudes applied to the uint8 ground-truth frame), on gfx950.
//
// The reference calls torchvision.transforms.functional.adjust_* on uint8 CHW tensors.  torchvision is third-party and
// absent offline: the arithmetic below restates its tensor backend (torchvision/transforms/_functional_tensor.py: _blend,
// rgb_to_grayscale, adjust_*, _rgb2hsv, _hsv2rgb, convert_image_dtype) operation by operation in float32, including the
// truncating float -> uint8 casts -- "parity unpinned" (oracle/distort.py is the same restatement in torch).
//
// in: uint8 [3][H][W] (what torchvision.io.read_image returns).  out_u8 (optional): the distorted uint8 frame; out_f32
// (optional): that frame / 255 as float32 [3][H][W] -- the `target / 255` the dataset hands to the model
// (utils/data.py:125).  One elementwise sweep; contrast needs the mean of the grey image first (exact integer sum).
#include "ct_common.h"

namespace ct {

enum { kDistIdentity = 0, kDistBrightness = 1, kDistContrast = 2, kDistSaturation = 3, kDistHue = 4, kDistGamma = 5 };

__device__ __forceinline__ float gray_u8(float r, float g, float b) {           // rgb_to_grayscale(...).to(uint8): truncation
    return truncf(0.2989f * r + 0.587f * g + 0.114f * b);
}
// _blend(...).clamp(0, 255).to(uint8); ratio and 1 - ratio are Python floats (float64) in torchvision, each rounded to
// float32 when it meets the tensor -- 1 - ratio is therefore formed in float64 on the host (one_minus), not as 1.0f - ratio
__device__ __forceinline__ float blend_u8(float a, float b, float ratio, float one_minus) {
    return truncf(fminf(fmaxf(ratio * a + one_minus * b, 0.f), 255.f));
}
__device__ __forceinline__ float to_u8(float x) { return truncf(x * 255.999f); }  // convert_image_dtype(float -> uint8): mul(255 + 1 - 1e-3)

__global__ __launch_bounds__(kBlock) void gray_sum_kernel(const uint8_t *__restrict__ in, int64_t n, unsigned long long *__restrict__ sum) {
    __shared__ unsigned long long red[4];
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        s += (unsigned long long)gray_u8((float)in[i], (float)in[n + i], (float)in[2 * n + i]);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, red[0] + red[1] + red[2] + red[3]);   // integer: order independent, exact
}

__global__ __launch_bounds__(kBlock) void distort_kernel(const uint8_t *__restrict__ in, int64_t n, int kind, float param, float one_minus,
                                                         const unsigned long long *__restrict__ gray_sum, uint8_t *__restrict__ out_u8,
                                                         float *__restrict__ out_f32) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float r = (float)in[i], g = (float)in[n + i], b = (float)in[2 * n + i];
    if (kind == kDistBrightness) {
        r = blend_u8(r, 0.f, param, one_minus); g = blend_u8(g, 0.f, param, one_minus); b = blend_u8(b, 0.f, param, one_minus);
    } else if (kind == kDistContrast) {
        const float mean = (float)((double)gray_sum[0] / (double)n);           // torch.mean of the uint8 grey image, in float32
        r = blend_u8(r, mean, param, one_minus); g = blend_u8(g, mean, param, one_minus); b = blend_u8(b, mean, param, one_minus);
    } else if (kind == kDistSaturation) {
        const float l = gray_u8(r, g, b);
        r = blend_u8(r, l, param, one_minus); g = blend_u8(g, l, param, one_minus); b = blend_u8(b, l, param, one_minus);
    } else if (kind == kDistGamma) {
        r = to_u8(fminf(fmaxf(powf(r / 255.f, param), 0.f), 1.f));
        g = to_u8(fminf(fmaxf(powf(g / 255.f, param), 0.f), 1.f));
        b = to_u8(fminf(fmaxf(powf(b / 255.f, param), 0.f), 1.f));
    } else if (kind == kDistHue) {
        r /= 255.f; g /= 255.f; b /= 255.f;
        // _rgb2hsv
        const float maxc = fmaxf(fmaxf(r, g), b), minc = fminf(fminf(r, g), b);
        const bool eqc = maxc == minc;
        const float cr = maxc - minc;
        const float s = cr / (eqc ? 1.f : maxc);
        const float div = eqc ? 1.f : cr;
        const float rc = (maxc - r) / div, gc = (maxc - g) / div, bc = (maxc - b) / div;
        const float hr = (maxc == r) ? (bc - gc) : 0.f;
        const float hg = ((maxc == g) && (maxc != r)) ? (2.0f + rc - bc) : 0.f;
        const float hb = ((maxc != g) && (maxc != r)) ? (4.0f + gc - rc) : 0.f;
        float h = fmodf((hr + hg + hb) / 6.0f + 1.0f, 1.0f);
        // h = (h + hue_factor) % 1.0  (python / torch remainder: result has the sign of the divisor)
        h = h + param;
        h = h - floorf(h);
        // _hsv2rgb
        const float v = maxc;
        const float h6 = h * 6.0f;
        const float fi = floorf(h6);
        const float f = h6 - fi;
        int idx = (int)fi % 6;
        idx = idx < 0 ? idx + 6 : idx;
        const float p = fminf(fmaxf(v * (1.0f - s), 0.f), 1.f);
        const float q = fminf(fmaxf(v * (1.0f - (s * f)), 0.f), 1.f);
        const float t = fminf(fmaxf(v * (1.0f - (s * (1.0f - f))), 0.f), 1.f);
        // rows of the reference's selection tensors: (v,q,p,p,t,v), (t,v,v,q,p,p), (p,p,t,v,v,q) indexed by idx
        r = to_u8(idx == 0 || idx == 5 ? v : idx == 1 ? q : idx == 4 ? t : p);
        g = to_u8(idx == 1 || idx == 2 ? v : idx == 0 ? t : idx == 3 ? q : p);
        b = to_u8(idx == 3 || idx == 4 ? v : idx == 2 ? t : idx == 5 ? q : p);
    }
    if (out_u8) { out_u8[i] = (uint8_t)r; out_u8[n + i] = (uint8_t)g; out_u8[2 * n + i] = (uint8_t)b; }
    if (out_f32) { out_f32[i] = r / 255.f; out_f32[n + i] = g / 255.f; out_f32[2 * n + i] = b / 255.f; }
}

}  // namespace ct

extern "C" {

// kind: 0 identity, 1 brightness (param = factor), 2 contrast, 3 saturation, 4 hue (param = hue_factor in [-0.5, 0.5]), 5 gamma.
// ws: >= 8 bytes, 8-byte aligned (the integer grey sum of the contrast distortion).
int ct_distort_u8(const uint8_t *in, int height, int width, int kind, double param, uint8_t *out_u8, float *out_f32, void *ws,
                  size_t ws_bytes, void *stream) {
    if (!in || height < 1 || width < 1 || kind < 0 || kind > 5 || (!out_u8 && !out_f32)) return CT_E_BADARG;
    if (kind == ct::kDistHue && !(param >= -0.5 && param <= 0.5)) return CT_E_BADARG;        // torchvision raises ValueError
    if ((kind == ct::kDistBrightness || kind == ct::kDistContrast || kind == ct::kDistSaturation || kind == ct::kDistGamma) && param < 0.0)
        return CT_E_BADARG;
    if (!ws || ws_bytes < 8 || (reinterpret_cast<uintptr_t>(ws) & 7)) return CT_E_WORKSPACE;
    const int64_t n = (int64_t)height * width;
    hipStream_t s = (hipStream_t)stream;
    if (kind == ct::kDistContrast) {
        { const int zr = ct::zero_async(ws, 8, s); if (zr) return zr; }
        const int blocks = (int)((n + ct::kBlock * 8 - 1) / (ct::kBlock * 8));
        hipLaunchKernelGGL(ct::gray_sum_kernel, dim3(blocks < 256 ? blocks : 256), dim3(ct::kBlock), 0, s, in, n, (unsigned long long *)ws);   // one same-address atomic per workgroup: keep them few
        CT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ct::distort_kernel, dim3((unsigned)((n + ct::kBlock - 1) / ct::kBlock)), dim3(ct::kBlock), 0, s, in, n, kind, (float)param,
                       (float)(1.0 - param), (const unsigned long long *)ws, out_u8, out_f32);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // extern "C"
