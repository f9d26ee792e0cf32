// metrics.hip -- per-frame quality metrics of Runner.test_step (reference methods/__init__.py:29-40) on gfx950:
//   SSIM  = piq.ssim(result, gt) with piq's defaults (11x11 Gaussian, sigma 1.5, k1 0.01, k2 0.03, data_range 1,
//           average-pool downsampling by max(1, round(min(H, W) / 256)), "valid" windows, mean over channels and positions)
//   iCID  = utils/icid.py:28-152 (perceptual intent, all seven maps, bilinear downsampling by the same factor, kornia's
//           rgb_to_lab, torchvision's 11x11 sigma-2 gaussian_blur with reflect padding)
// piq, kornia and torchvision are third-party and absent offline: their arithmetic is restated from their published
// sources (oracle/metrics.py says where) -- "parity unpinned" for those calls; SSIM's core is anchored on scikit-image's
// structural_similarity, iCID's own lines on a run of the reference's utils/icid.py (tests/golden/make_golden_*).
//
// Both are ONE fused pass per frame: a workgroup owns a 32x32 (SSIM) / 32x16 (iCID) tile of the metric map, brings the
// down-sampled inputs (plus a 5-pixel halo) into LDS straight from the full-resolution NCHW planes, runs the separable
// 11-tap Gaussian (rows into LDS, then columns) for all 5 / 11 moment maps at once, evaluates the per-pixel formula and
// reduces it to one float64 partial sum; a finishing kernel adds the partials in a fixed order (deterministic).  float32
// arithmetic like the reference's torch code, float64 sums.
#include "ct_common.h"

namespace ct {

constexpr int kTaps = 11, kHalo = kTaps - 1;

__device__ __forceinline__ void gauss_weights(float *w /* LDS [11] */, float sigma) {
    if (threadIdx.x < kTaps) {
        float s = 0.f;
        for (int k = 0; k < kTaps; ++k) { const float x = (float)(k - 5) / sigma; s += expf(-0.5f * x * x); }
        const float x = (float)((int)threadIdx.x - 5) / sigma;
        w[threadIdx.x] = expf(-0.5f * x * x) / s;
    }
}

__device__ __forceinline__ double block_reduce_1(double v, double *lds /* [4] */) {
    double a[1] = {v};
    block_sum<1>(a, lds);
    return a[0];
}

// -------------------------------------------------------------------------------------------------------------------
// SSIM.  grid = (tiles_x, tiles_y, batch * 3); a, b: [batch][3][H][W]; pooled size Hp x Wp = (H / f, W / f).
// -------------------------------------------------------------------------------------------------------------------
constexpr int kSsTW = 32, kSsTH = 32;

__global__ __launch_bounds__(kBlock) void ssim_tile_kernel(const float *__restrict__ a, const float *__restrict__ b, int H, int W, int f,
                                                           int Hp, int Wp, double *__restrict__ partials) {
    __shared__ float sx[kSsTH + kHalo][kSsTW + kHalo], sy[kSsTH + kHalo][kSsTW + kHalo];
    __shared__ float hb[5][kSsTH + kHalo][kSsTW];
    __shared__ float w[kTaps];
    __shared__ double red[4];
    const int plane = blockIdx.z;
    const float *pa = a + (size_t)plane * H * W, *pb = b + (size_t)plane * H * W;
    const int ox = blockIdx.x * kSsTW, oy = blockIdx.y * kSsTH;
    const int Ho = Hp - kHalo, Wo = Wp - kHalo;
    gauss_weights(w, 1.5f);
    const float inv = 1.0f / (float)(f * f);
    for (int i = threadIdx.x; i < (kSsTH + kHalo) * (kSsTW + kHalo); i += kBlock) {
        const int r = i / (kSsTW + kHalo), c = i % (kSsTW + kHalo);
        const int py = oy + r, px = ox + c;
        float vx = 0.f, vy = 0.f;
        if (py < Hp && px < Wp) {
            for (int dy = 0; dy < f; ++dy)
                for (int dx = 0; dx < f; ++dx) {
                    const size_t o = (size_t)(py * f + dy) * W + (px * f + dx);
                    vx += pa[o]; vy += pb[o];
                }
            vx *= inv; vy *= inv;                      // F.avg_pool2d(kernel_size = f)
        }
        sx[r][c] = vx; sy[r][c] = vy;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (kSsTH + kHalo) * kSsTW; i += kBlock) {
        const int r = i / kSsTW, c = i % kSsTW;
        float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
            const float x = sx[r][c + k], y = sy[r][c + k], g = w[k];
            m[0] = fmaf(g, x, m[0]); m[1] = fmaf(g, y, m[1]); m[2] = fmaf(g, x * x, m[2]); m[3] = fmaf(g, y * y, m[3]); m[4] = fmaf(g, x * y, m[4]);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) hb[q][r][c] = m[q];
    }
    __syncthreads();
    double acc = 0.0;
    for (int i = threadIdx.x; i < kSsTH * kSsTW; i += kBlock) {
        const int r = i / kSsTW, c = i % kSsTW;
        if (oy + r < Ho && ox + c < Wo) {
            float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < kTaps; ++k) {
#pragma unroll
                for (int q = 0; q < 5; ++q) m[q] = fmaf(w[k], hb[q][r + k][c], m[q]);
            }
            const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
            const float mxx = m[0] * m[0], myy = m[1] * m[1], mxy = m[0] * m[1];
            const float sxx = m[2] - mxx, syy = m[3] - myy, sxy = m[4] - mxy;
            const float cs = (2.0f * sxy + c2) / (sxx + syy + c2);
            acc += (double)((2.0f * mxy + c1) / (mxx + myy + c1) * cs);
        }
    }
    acc = block_reduce_1(acc, red);
    if (threadIdx.x == 0) partials[((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = acc;
}

// -------------------------------------------------------------------------------------------------------------------
// iCID.  grid = (tiles_x, tiles_y, batch); down-sampled size h x w; maps live on the whole h x w grid (reflect padding).
// -------------------------------------------------------------------------------------------------------------------
constexpr int kIcTW = 32, kIcTH = 16;

__device__ __forceinline__ float lab_f32(float v) {       // kornia rgb_to_lab: where(v > 0.008856, max(v, 0.008856)^(1/3), 7.787 v + 4/29)
    return v > 0.008856f ? cbrtf(v) : fmaf(7.787f, v, 4.0f / 29.0f);
}
__device__ __forceinline__ float srgb_lin_f32(float c) {  // kornia rgb_to_linear_rgb
    return c > 0.04045f ? powf((c + 0.055f) / 1.055f, 2.4f) : c / 12.92f;
}
__device__ __forceinline__ void rgb_to_lab_f32(float r, float g, float b, float &L, float &A, float &B) {
    r = srgb_lin_f32(r); g = srgb_lin_f32(g); b = srgb_lin_f32(b);
    const float x = (0.412453f * r + 0.357580f * g + 0.180423f * b) / 0.95047f;
    const float y = 0.212671f * r + 0.715160f * g + 0.072169f * b;
    const float z = (0.019334f * r + 0.119193f * g + 0.950227f * b) / 1.08883f;
    const float fx = lab_f32(x), fy = lab_f32(y), fz = lab_f32(z);
    L = 116.0f * fy - 16.0f; A = 500.0f * (fx - fy); B = 200.0f * (fy - fz);
}

// F.interpolate(scale_factor = 1/f, mode = "bilinear", align_corners = False) of one NCHW plane at output pixel (y, x)
__device__ __forceinline__ float bilinear_down(const float *__restrict__ p, int H, int W, int f, int y, int x) {
    if (f == 1) return p[(size_t)y * W + x];
    const float sy = fmaxf(((float)y + 0.5f) * (float)f - 0.5f, 0.f), sx = fmaxf(((float)x + 0.5f) * (float)f - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float wy = sy - (float)y0, wx = sx - (float)x0;
    const float v00 = p[(size_t)y0 * W + x0], v01 = p[(size_t)y0 * W + x1], v10 = p[(size_t)y1 * W + x0], v11 = p[(size_t)y1 * W + x1];
    return (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
}

__device__ __forceinline__ int reflect(int i, int n) {     // torch "reflect" padding (no edge repeat); n >= 6 here
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

__global__ __launch_bounds__(kBlock) void icid_tile_kernel(const float *__restrict__ a, const float *__restrict__ b, int H, int W, int f, int h,
                                                           int w_, double *__restrict__ partials) {
    // base planes: L1, C1, L2, C2, sqrt(H) of utils/icid.py:69-111
    __shared__ float base[5][kIcTH + kHalo][kIcTW + kHalo];
    __shared__ float hb[11][kIcTH + kHalo][kIcTW];
    __shared__ float w[kTaps];
    __shared__ double red[4];
    const int img = blockIdx.z;
    const float *pa = a + (size_t)img * 3 * H * W, *pb = b + (size_t)img * 3 * H * W;
    const size_t P = (size_t)H * W;
    const int ox = blockIdx.x * kIcTW, oy = blockIdx.y * kIcTH;
    gauss_weights(w, 2.0f);
    for (int i = threadIdx.x; i < (kIcTH + kHalo) * (kIcTW + kHalo); i += kBlock) {
        const int r = i / (kIcTW + kHalo), c = i % (kIcTW + kHalo);
        const int y = reflect(oy + r - 5, h), x = reflect(ox + c - 5, w_);
        float v[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < h && x >= 0 && x < w_) {       // tiles past the image edge: unused entries
            float L1, A1, B1, L2, A2, B2;
            rgb_to_lab_f32(bilinear_down(pa, H, W, f, y, x), bilinear_down(pa + P, H, W, f, y, x), bilinear_down(pa + 2 * P, H, W, f, y, x), L1, A1, B1);
            rgb_to_lab_f32(bilinear_down(pb, H, W, f, y, x), bilinear_down(pb + P, H, W, f, y, x), bilinear_down(pb + 2 * P, H, W, f, y, x), L2, A2, B2);
            const float C1 = sqrtf(A1 * A1 + B1 * B1), C2 = sqrtf(A2 * A2 + B2 * B2);
            const float dA = A1 - A2, dB = B1 - B2, dC = C1 - C2;
            const float Hd = dA * dA + dB * dB - dC * dC;
            v[0] = L1; v[1] = C1; v[2] = L2; v[3] = C2; v[4] = sqrtf(Hd < 0.f ? 0.f : Hd);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) base[q][r][c] = v[q];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (kIcTH + kHalo) * kIcTW; i += kBlock) {
        const int r = i / kIcTW, c = i % kIcTW;
        float m[11];
#pragma unroll
        for (int q = 0; q < 11; ++q) m[q] = 0.f;
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
            const float g = w[k], L1 = base[0][r][c + k], C1 = base[1][r][c + k], L2 = base[2][r][c + k], C2 = base[3][r][c + k];
            m[0] = fmaf(g, L1, m[0]); m[1] = fmaf(g, C1, m[1]); m[2] = fmaf(g, L2, m[2]); m[3] = fmaf(g, C2, m[3]);
            m[4] = fmaf(g, L1 * L1, m[4]); m[5] = fmaf(g, L2 * L2, m[5]); m[6] = fmaf(g, C1 * C1, m[6]); m[7] = fmaf(g, C2 * C2, m[7]);
            m[8] = fmaf(g, base[4][r][c + k], m[8]); m[9] = fmaf(g, L1 * L2, m[9]); m[10] = fmaf(g, C1 * C2, m[10]);
        }
#pragma unroll
        for (int q = 0; q < 11; ++q) hb[q][r][c] = m[q];
    }
    __syncthreads();
    double acc = 0.0;
    for (int i = threadIdx.x; i < kIcTH * kIcTW; i += kBlock) {
        const int r = i / kIcTW, c = i % kIcTW;
        if (oy + r < h && ox + c < w_) {
            float m[11];
#pragma unroll
            for (int q = 0; q < 11; ++q) m[q] = 0.f;
#pragma unroll
            for (int k = 0; k < kTaps; ++k) {
#pragma unroll
                for (int q = 0; q < 11; ++q) m[q] = fmaf(w[k], hb[q][r + k][c], m[q]);
            }
            const float muL1 = m[0], muC1 = m[1], muL2 = m[2], muC2 = m[3];
            const float sL1q = fmaxf(m[4] - muL1 * muL1, 0.f), sL2q = fmaxf(m[5] - muL2 * muL2, 0.f);
            const float sC1q = fmaxf(m[6] - muC1 * muC1, 0.f), sC2q = fmaxf(m[7] - muC2 * muC2, 0.f);
            const float sL1 = sqrtf(sL1q), sL2 = sqrtf(sL2q), sC1 = sqrtf(sC1q), sC2 = sqrtf(sC2q);
            const float dLq = (muL1 - muL2) * (muL1 - muL2), dCq = (muC1 - muC2) * (muC1 - muC2), dHq = m[8] * m[8];
            const float sL12 = m[9] - muL1 * muL2, sC12 = m[10] - muC1 * muC2;
            // perceptual intent: weights (0.002, 10, 10, 0.002, 0.002, 10, 10), exponents (1, 1, 3, 1, 1, 1, 1)
            const float m1 = 1.f / (0.002f * dLq + 1.f);
            const float m2 = (10.f + 2.f * sL1 * sL2) / (10.f + sL1q + sL2q);
            const float m3 = (10.f + fabsf(sL12)) / (10.f + sL1 * sL2);
            const float m4 = 1.f / (0.002f * dCq + 1.f);
            const float m5 = 1.f / (0.002f * dHq + 1.f);
            const float m6 = (10.f + 2.f * sC1 * sC2) / (10.f + sC1 * sC1 + sC2 * sC2);
            const float m7 = (10.f + fabsf(sC12)) / (10.f + sC1 * sC2);
            acc += (double)(m1 * m2 * (m3 * m3 * m3) * m4 * m5 * m6 * m7);
        }
    }
    acc = block_reduce_1(acc, red);
    if (threadIdx.x == 0) partials[((size_t)img * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = acc;
}

// out[i] = scale * (sum of the `n` partials of item i, per_item of them contiguous) / count + bias, added in a fixed order
__global__ __launch_bounds__(kBlock) void metric_finish_kernel(const double *__restrict__ partials, int per_item, double count, double scale,
                                                               double bias, double *__restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < per_item; i += kBlock) s += partials[(size_t)blockIdx.x * per_item + i];
    s = block_reduce_1(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = scale * (s / count) + bias;
}

static int metric_factor(int height, int width) {          // max(1, round(min(H, W) / 256)) with Python's round-half-even
    const int m = height < width ? height : width;
    const double q = (double)m / 256.0;
    double r = __builtin_floor(q + 0.5);
    if (q + 0.5 == r && ((long long)r % 2)) r -= 1.0;      // exact .5 -> even
    return r < 1.0 ? 1 : (int)r;
}

}  // namespace ct

extern "C" {

size_t ct_metric_workspace_bytes(int height, int width, int batch) {
    if (height < 1 || width < 1 || batch < 0) return 0;
    const int f = ct::metric_factor(height, width);
    const size_t t_ssim = (size_t)((width / f + ct::kSsTW - 1) / ct::kSsTW) * ((height / f + ct::kSsTH - 1) / ct::kSsTH) * 3;
    const size_t t_icid = (size_t)((width / f + ct::kIcTW - 1) / ct::kIcTW) * ((height / f + ct::kIcTH - 1) / ct::kIcTH);
    return (t_ssim > t_icid ? t_ssim : t_icid) * (size_t)batch * sizeof(double) + 64;
}

int ct_frame_ssim_f32(const float *a, const float *b, int height, int width, int batch, double *out, void *ws, size_t ws_bytes, void *stream) {
    if (!a || !b || !out || height < 1 || width < 1 || batch < 0) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    const int f = ct::metric_factor(height, width);
    const int Hp = height / f, Wp = width / f;
    if (Hp < ct::kTaps || Wp < ct::kTaps) return CT_E_BADARG;                 // piq asserts the kernel fits the image
    if (!ws || ws_bytes < ct_metric_workspace_bytes(height, width, batch)) return CT_E_WORKSPACE;
    const int Ho = Hp - ct::kHalo, Wo = Wp - ct::kHalo;
    const dim3 grid((Wo + ct::kSsTW - 1) / ct::kSsTW, (Ho + ct::kSsTH - 1) / ct::kSsTH, batch * 3);
    hipLaunchKernelGGL(ct::ssim_tile_kernel, grid, dim3(ct::kBlock), 0, (hipStream_t)stream, a, b, height, width, f, Hp, Wp, (double *)ws);
    CT_CHECK_LAUNCH();
    hipLaunchKernelGGL(ct::metric_finish_kernel, dim3(batch), dim3(ct::kBlock), 0, (hipStream_t)stream, (const double *)ws,
                       (int)(grid.x * grid.y * 3), (double)Ho * Wo * 3.0, 1.0, 0.0, out);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_frame_icid_f32(const float *a, const float *b, int height, int width, int batch, double *out, void *ws, size_t ws_bytes, void *stream) {
    if (!a || !b || !out || height < 1 || width < 1 || batch < 0) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    const int f = ct::metric_factor(height, width);
    const int h = (int)__builtin_floor((double)height * (1.0 / f)), w = (int)__builtin_floor((double)width * (1.0 / f));
    if (h < 6 || w < 6) return CT_E_BADARG;                                    // reflect padding of 5 needs more than 5 pixels
    if (!ws || ws_bytes < ct_metric_workspace_bytes(height, width, batch)) return CT_E_WORKSPACE;
    const dim3 grid((w + ct::kIcTW - 1) / ct::kIcTW, (h + ct::kIcTH - 1) / ct::kIcTH, batch);
    hipLaunchKernelGGL(ct::icid_tile_kernel, grid, dim3(ct::kBlock), 0, (hipStream_t)stream, a, b, height, width, f, h, w, (double *)ws);
    CT_CHECK_LAUNCH();
    hipLaunchKernelGGL(ct::metric_finish_kernel, dim3(batch), dim3(ct::kBlock), 0, (hipStream_t)stream, (const double *)ws,
                       (int)(grid.x * grid.y), (double)h * w, -1.0, 1.0, out);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // extern "C"
