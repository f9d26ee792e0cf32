// fsim.hip -- piq.fsim(result, gt) of Runner.test_step (reference methods/__init__.py:34; also methods/dmsct.py:129) on gfx950.
//
// FSIMc with piq's defaults: inputs x 255, average-pool by max(1, round(min(H, W) / 256)), RGB -> YIQ, Kovesi's phase
// congruency (PC_2) of the luminance through a bank of 4 orientations x 4 scales of log-Gabor filters applied in the
// frequency domain, Scharr gradient magnitude, similarity maps with the paper's constants, chromatic terms, PC-weighted
// mean.  piq is third-party and absent offline: the arithmetic is restated from its published source (oracle/metrics.py
// names the functions) -- "parity unpinned".
//
// Per call and image: one forward and 16 inverse 2-D FFTs of the pooled luminance (csrc/fft2d.hip: hand-written batched
// mixed-radix Stockham transforms in LDS, no plan, no state -- hipFFT until round 3), around them: pooling + YIQ, Scharr, spectrum x
// filter bank, an exact median (3-pass radix select on the float bit patterns; torch.median's lower median) of the
// smallest-scale energy per orientation for the noise threshold, the phase-congruency map, the similarity / score
// reduction (float64 partial sums, fixed order).  The filter bank and its three noise constants per orientation depend on
// the pooled size only: ct_fsim_setup_f32 builds them once on the device.  float32 arithmetic like the reference.
#include "ct_common.h"

namespace ct {

constexpr int kFsO = 4, kFsS = 4, kFsK = kFsO * kFsS;
constexpr double kPi = 3.14159265358979323846;

// csrc/fft2d.hip: batched in-place 2-D complex DFT (hand-written; hipFFT until round 3), sign -1 forward / +1 backward, unnormalised
int fft2d_c2c(float2 *data, int hp, int wp, int planes, int sign, hipStream_t s);
bool fft2d_supported(int hp, int wp);

struct FsimLayout {
    float2 *lum;          // [imgs][P]        luminance as complex -> its spectrum
    float2 *eo;           // [imgs][16][P]    filtered spectra -> even / odd responses (unnormalised inverse FFT)
    float *iq;            // [imgs][2][P]
    float *grad;          // [imgs][P]
    float *pc;            // [imgs][P]
    unsigned int *hist;   // [imgs * 4][2048]
    unsigned int *sel;    // [imgs * 4][2]   (prefix, remaining rank)
    float *thr;           // [imgs * 4]      noise threshold T per orientation
    double *part;         // [pairs][blocks][2]
    size_t total;
};
constexpr int kFsScoreBlocks = 64;

static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

static FsimLayout fsim_layout(void *ws, int imgs, int P) {
    FsimLayout l;
    char *p = reinterpret_cast<char *>(ws);
    size_t o = 0;
    l.lum = reinterpret_cast<float2 *>(p + o); o += up256((size_t)imgs * P * sizeof(float2));
    l.eo = reinterpret_cast<float2 *>(p + o); o += up256((size_t)imgs * kFsK * P * sizeof(float2));
    l.iq = reinterpret_cast<float *>(p + o); o += up256((size_t)imgs * 2 * P * sizeof(float));
    l.grad = reinterpret_cast<float *>(p + o); o += up256((size_t)imgs * P * sizeof(float));
    l.pc = reinterpret_cast<float *>(p + o); o += up256((size_t)imgs * P * sizeof(float));
    l.hist = reinterpret_cast<unsigned int *>(p + o); o += up256((size_t)imgs * kFsO * 2048 * sizeof(unsigned int));
    l.sel = reinterpret_cast<unsigned int *>(p + o); o += up256((size_t)imgs * kFsO * 2 * sizeof(unsigned int));
    l.thr = reinterpret_cast<float *>(p + o); o += up256((size_t)imgs * kFsO * sizeof(float));
    l.part = reinterpret_cast<double *>(p + o); o += up256((size_t)(imgs / 2 + 1) * kFsScoreBlocks * 2 * sizeof(double));
    l.total = o;
    return l;
}

// ---- setup: filter bank [o * 4 + s][P] (piq _construct_filters) -------------------------------------------------------
__device__ __forceinline__ double fs_axis(int i, int n) {             // piq get_meshgrid after ifftshift: index i of the shifted axis
    const int j = (i + n / 2) % n;                                     // un-shifted position
    return (n & 1) ? ((double)j - (double)(n - 1) / 2.0) / (double)(n - 1) : ((double)j - (double)n / 2.0) / (double)n;
}

__global__ __launch_bounds__(256) void fsim_filters_kernel(float *__restrict__ filt, int hp, int wp) {
    const int P = hp * wp;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int iy = p / wp, ix = p - iy * wp;
    const double gx = fs_axis(iy, hp), gy = fs_axis(ix, wp);            // piq names the row coordinate x
    double radius = sqrt(gx * gx + gy * gy);
    const double theta = atan2(-gy, gx);
    const double lp = 1.0 / (1.0 + pow(radius / 0.45, 30.0));
    if (p == 0) radius = 1.0;
    const double st = sin(theta), ct_ = cos(theta);
    const double theta_sigma = kPi / (kFsO * 1.2);
    const double lsf = log(0.55);
    double lg[kFsS];
#pragma unroll
    for (int s = 0; s < kFsS; ++s) {
        const double omega0 = 1.0 / (6.0 * (double)(1 << s));
        const double l = log(radius / omega0);
        lg[s] = p == 0 ? 0.0 : exp(-(l * l) / (2.0 * lsf * lsf)) * lp;
    }
#pragma unroll
    for (int o = 0; o < kFsO; ++o) {
        const double angl = o * kPi / kFsO;
        const double ds = st * cos(angl) - ct_ * sin(angl), dc = ct_ * cos(angl) + st * sin(angl);
        const double dth = fabs(atan2(ds, dc));
        const double spread = exp(-(dth * dth) / (2.0 * theta_sigma * theta_sigma));
#pragma unroll
        for (int s = 0; s < kFsS; ++s) filt[(size_t)(o * kFsS + s) * P + p] = (float)(spread * lg[s]);
    }
}

__global__ __launch_bounds__(256) void fsim_filters_to_complex_kernel(const float *__restrict__ filt, float2 *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = make_float2(filt[i], 0.f);
}

// consts[o][0..2] = em_n (sum of the scale-0 filter squared), sum_an2, sum_ai_aj of filters_ifft = real(ifft2(filters)) sqrt(P)
__global__ __launch_bounds__(256) void fsim_consts_kernel(const float *__restrict__ filt, const float2 *__restrict__ fi_unnorm, int P,
                                                          double *__restrict__ consts) {
    __shared__ double red[4 * 3];
    const int o = blockIdx.x;
    const double sc = 1.0 / sqrt((double)P);                            // (1 / P) of the inverse FFT x sqrt(P)
    double em = 0.0, an2 = 0.0, aij = 0.0;
    for (int p = threadIdx.x; p < P; p += 256) {
        const double f0 = filt[(size_t)(o * kFsS) * P + p];
        em += f0 * f0;
        double v[kFsS];
#pragma unroll
        for (int s = 0; s < kFsS; ++s) { v[s] = (double)fi_unnorm[(size_t)(o * kFsS + s) * P + p].x * sc; an2 += v[s] * v[s]; }
#pragma unroll
        for (int s = 0; s < kFsS - 1; ++s)
#pragma unroll
            for (int t = s + 1; t < kFsS; ++t) aij += v[s] * v[t];
    }
    double a[3] = {em, an2, aij};
    block_sum<3>(a, red);
    if (threadIdx.x == 0) { consts[o * 3 + 0] = a[0]; consts[o * 3 + 1] = a[1]; consts[o * 3 + 2] = a[2]; }
}

// ---- per call ------------------------------------------------------------------------------------------------------------
// average pooling + x255 + YIQ; image index = 2 * pair + (0: a, 1: b)
__global__ __launch_bounds__(256) void fsim_prep_kernel(const float *__restrict__ a, const float *__restrict__ b, int H, int W, int f, int hp,
                                                        int wp, float2 *__restrict__ lum, float *__restrict__ iq) {
    const int P = hp * wp;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int img = blockIdx.y;
    if (p >= P) return;
    const float *src = ((img & 1) ? b : a) + (size_t)(img >> 1) * 3 * H * W;
    const int py = p / wp, px = p - py * wp;
    float c[3];
    const float inv = 1.0f / (float)(f * f);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float s = 0.f;
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) s += src[(size_t)ch * H * W + (size_t)(py * f + dy) * W + px * f + dx] * 255.0f;
        c[ch] = s * inv;
    }
    const float y = 0.299f * c[0] + 0.587f * c[1] + 0.114f * c[2];
    const float i = 0.5959f * c[0] - 0.2746f * c[1] - 0.3213f * c[2];
    const float q = 0.2115f * c[0] - 0.5227f * c[1] + 0.3112f * c[2];
    lum[(size_t)img * P + p] = make_float2(y, 0.f);
    iq[((size_t)img * 2 + 0) * P + p] = i;
    iq[((size_t)img * 2 + 1) * P + p] = q;
}

// Scharr gradient magnitude of the luminance (zero padding), before the FFT overwrites it
__global__ __launch_bounds__(256) void fsim_grad_kernel(const float2 *__restrict__ lum, int hp, int wp, float *__restrict__ grad) {
    const int P = hp * wp;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int img = blockIdx.y;
    if (p >= P) return;
    const int y = p / wp, x = p - y * wp;
    const float2 *l = lum + (size_t)img * P;
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < hp && xx >= 0 && xx < wp) ? l[yy * wp + xx].x : 0.f; };
    const float a00 = at(y - 1, x - 1), a01 = at(y - 1, x), a02 = at(y - 1, x + 1), a10 = at(y, x - 1), a12 = at(y, x + 1);
    const float a20 = at(y + 1, x - 1), a21 = at(y + 1, x), a22 = at(y + 1, x + 1);
    const float gx = (-3.f * a00 + 3.f * a02 - 10.f * a10 + 10.f * a12 - 3.f * a20 + 3.f * a22) / 16.f;
    const float gy = (-3.f * a00 - 10.f * a01 - 3.f * a02 + 3.f * a20 + 10.f * a21 + 3.f * a22) / 16.f;
    grad[(size_t)img * P + p] = sqrtf(gx * gx + gy * gy);
}

__global__ __launch_bounds__(256) void fsim_apply_filters_kernel(const float2 *__restrict__ spec, const float *__restrict__ filt, int P,
                                                                 float2 *__restrict__ eo) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int img = blockIdx.y;
    if (p >= P) return;
    const float2 s = spec[(size_t)img * P + p];
#pragma unroll
    for (int k = 0; k < kFsK; ++k) {
        const float f = filt[(size_t)k * P + p];
        eo[((size_t)img * kFsK + k) * P + p] = make_float2(s.x * f, s.y * f);
    }
}

// squared amplitude of the scale-0 response of (image, orientation) = plane; the value whose median sets the noise level
__device__ __forceinline__ float fs_e2(const float2 *__restrict__ eo, int plane, int P, int p, float invP) {
    const int img = plane / kFsO, o = plane - img * kFsO;
    const float2 v = eo[((size_t)img * kFsK + o * kFsS) * P + p];
    const float an = sqrtf((v.x * invP) * (v.x * invP) + (v.y * invP) * (v.y * invP));
    return an * an;
}

// radix select, one pass: histogram of `bits` key bits at `shift` over the keys whose higher bits equal the prefix
__global__ __launch_bounds__(256) void fsim_select_hist_kernel(const float2 *__restrict__ eo, int P, float invP, int shift, int bits,
                                                               int first, const unsigned int *__restrict__ sel,
                                                               unsigned int *__restrict__ hist) {
    __shared__ unsigned int h[2048];
    const int plane = blockIdx.y;
    for (int i = threadIdx.x; i < 2048; i += 256) h[i] = 0;
    __syncthreads();
    const unsigned int prefix = first ? 0u : sel[plane * 2];
    const unsigned int mask = (1u << bits) - 1u;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const unsigned int key = __float_as_uint(fs_e2(eo, plane, P, p, invP));
        if (first || (key >> (shift + bits)) == prefix) atomicAdd(&h[(key >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (1 << bits); i += 256)
        if (h[i]) atomicAdd(&hist[(size_t)plane * 2048 + i], h[i]);
}

// pick the bin holding the wanted rank, extend the prefix, clear the histogram for the next pass; last pass: threshold T
__global__ __launch_bounds__(64) void fsim_select_pick_kernel(unsigned int *__restrict__ hist, unsigned int *__restrict__ sel, int bits, int first,
                                                              int last, unsigned int rank0, const double *__restrict__ consts,
                                                              float *__restrict__ thr, float k_noise) {
    const int plane = blockIdx.x;
    if (threadIdx.x == 0) {
        unsigned int rank = first ? rank0 : sel[plane * 2 + 1];
        const unsigned int prefix = first ? 0u : sel[plane * 2];
        unsigned int *h = hist + (size_t)plane * 2048;
        unsigned int cum = 0, bin = 0;
        for (unsigned int i = 0; i < (1u << bits); ++i) {
            if (cum + h[i] > rank) { bin = i; break; }
            cum += h[i];
        }
        const unsigned int key = (prefix << bits) | bin;
        sel[plane * 2] = key;
        sel[plane * 2 + 1] = rank - cum;
        if (last) {
            const int o = plane % kFsO;
            const float median = __uint_as_float(key);
            const float mean_e2n = -median / logf(0.5f);
            const float noise_power = mean_e2n / (float)consts[o * 3 + 0];
            const float ne2 = 2.f * noise_power * (float)consts[o * 3 + 1] + 4.f * noise_power * (float)consts[o * 3 + 2];
            const float tau = sqrtf(ne2 / 2.f);
            const float t = tau * sqrtf((float)(kPi / 2.0)) + k_noise * sqrtf((2.f - (float)(kPi / 2.0)) * tau * tau);
            thr[plane] = t / 1.7f;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) hist[(size_t)plane * 2048 + i] = 0;
}

// phase congruency map of one image (piq _phase_congruency after the filtering)
__global__ __launch_bounds__(256) void fsim_pc_kernel(const float2 *__restrict__ eo, int P, float invP, const float *__restrict__ thr,
                                                      float *__restrict__ pc) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int img = blockIdx.y;
    if (p >= P) return;
    const float eps = 1.1920929e-07f;
    float energy_all = 0.f, an_all = 0.f;
#pragma unroll
    for (int o = 0; o < kFsO; ++o) {
        float e[kFsS], od[kFsS], se = 0.f, so = 0.f;
#pragma unroll
        for (int s = 0; s < kFsS; ++s) {
            const float2 v = eo[((size_t)img * kFsK + o * kFsS + s) * P + p];
            e[s] = v.x * invP; od[s] = v.y * invP;
            se += e[s]; so += od[s];
            an_all += sqrtf(e[s] * e[s] + od[s] * od[s]);
        }
        const float xe = sqrtf(se * se + so * so) + eps;
        const float me = se / xe, mo = so / xe;
        float en = 0.f;
#pragma unroll
        for (int s = 0; s < kFsS; ++s) en += e[s] * me + od[s] * mo - fabsf(e[s] * mo - od[s] * me);
        energy_all += fmaxf(en - thr[img * kFsO + o], 0.f);
    }
    pc[(size_t)img * P + p] = (energy_all + eps) / (an_all + eps);
}

__device__ __forceinline__ float fs_sim(float a, float b, float c) { return (2.f * a * b + c) / (a * a + b * b + c); }

__global__ __launch_bounds__(256) void fsim_score_kernel(const float *__restrict__ pc, const float *__restrict__ grad, const float *__restrict__ iq,
                                                         int P, double *__restrict__ part) {
    __shared__ double red[4 * 2];
    const int pair = blockIdx.y;
    const float *pcx = pc + (size_t)(2 * pair) * P, *pcy = pcx + P;
    const float *gx = grad + (size_t)(2 * pair) * P, *gy = gx + P;
    const float *ix = iq + (size_t)(2 * pair) * 2 * P, *qx = ix + P, *iy = ix + 2 * P, *qy = ix + 3 * P;
    double s_score = 0.0, s_pc = 0.0;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const float a = pcx[p], b = pcy[p];
        const float pcm = a > b ? a : b;
        float score = fs_sim(gx[p], gy[p], 160.f) * fs_sim(a, b, 0.85f) * pcm;
        score *= powf(fabsf(fs_sim(ix[p], iy[p], 200.f) * fs_sim(qx[p], qy[p], 200.f)), 0.03f);
        s_score += (double)score;
        s_pc += (double)pcm;
    }
    double v[2] = {s_score, s_pc};
    block_sum<2>(v, red);
    if (threadIdx.x == 0) {
        part[((size_t)pair * gridDim.x + blockIdx.x) * 2 + 0] = v[0];
        part[((size_t)pair * gridDim.x + blockIdx.x) * 2 + 1] = v[1];
    }
}

__global__ __launch_bounds__(64) void fsim_finish_kernel(const double *__restrict__ part, int blocks, double *__restrict__ out) {
    if (threadIdx.x == 0) {
        double s = 0.0, w = 0.0;
        for (int i = 0; i < blocks; ++i) { s += part[((size_t)blockIdx.x * blocks + i) * 2]; w += part[((size_t)blockIdx.x * blocks + i) * 2 + 1]; }
        out[blockIdx.x] = s / w;
    }
}

static int fsim_factor(int h, int w) {
    const int m = h < w ? h : w;
    const double v = (double)m / 256.0;                    // Python round(): half to even
    double r = floor(v + 0.5);
    if (v + 0.5 == r && ((long long)r & 1)) r -= 1.0;
    return r < 1.0 ? 1 : (int)r;
}

}  // namespace ct

extern "C" {

// pooled size of piq.fsim's internal average pooling
int ct_fsim_pooled_size(int h, int w, int *hp, int *wp) {
    if (h < 1 || w < 1) return CT_E_BADARG;
    const int f = ct::fsim_factor(h, w);
    if (hp) *hp = h / f;
    if (wp) *wp = w / f;
    return CT_OK;
}

size_t ct_fsim_workspace_bytes(int batch, int h, int w) {
    if (batch < 1 || h < 1 || w < 1) return 0;
    const int f = ct::fsim_factor(h, w), hp = h / f, wp = w / f;
    if (hp < 2 || wp < 2 || !ct::fft2d_supported(hp, wp)) return 0;
    return ct::fsim_layout(nullptr, 2 * batch, hp * wp).total;
}

// filters: [16][hp*wp] float32 (orientation-major), consts: [4][3] float64; both on the device, valid for every frame of this size
int ct_fsim_setup_f32(int h, int w, float *filters, double *consts, void *ws, size_t ws_bytes, void *stream) {
    if (!filters || !consts || !ws || h < 1 || w < 1 || (reinterpret_cast<uintptr_t>(ws) & 255)) return CT_E_BADARG;
    const int f = ct::fsim_factor(h, w), hp = h / f, wp = w / f, P = hp * wp;
    if (hp < 2 || wp < 2) return CT_E_BADARG;
    if (!ct::fft2d_supported(hp, wp)) return CT_E_BADARG;
    const ct::FsimLayout l = ct::fsim_layout(ws, 2, P);
    if (ws_bytes < l.total) return CT_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ct::fsim_filters_kernel, dim3((P + 255) / 256), dim3(256), 0, s, filters, hp, wp);
    CT_CHECK_LAUNCH();
    const size_t n = (size_t)ct::kFsK * P;
    hipLaunchKernelGGL(ct::fsim_filters_to_complex_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float *)filters, l.eo, n);
    CT_CHECK_LAUNCH();
    { const int rc = ct::fft2d_c2c(l.eo, hp, wp, ct::kFsK, +1, s); if (rc) return rc; }
    hipLaunchKernelGGL(ct::fsim_consts_kernel, dim3(ct::kFsO), dim3(256), 0, s, (const float *)filters, (const float2 *)l.eo, P, consts);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

// out[b] = piq.fsim(a[b], b[b]) for batch frames [batch][3][h][w] float32 in [0, 1]
int ct_frame_fsim_f32(const float *a, const float *b, double *out, int batch, int h, int w, const float *filters, const double *consts,
                      void *ws, size_t ws_bytes, void *stream) {
    if (!a || !b || !out || !filters || !consts || !ws || batch < 0 || h < 1 || w < 1 || (reinterpret_cast<uintptr_t>(ws) & 255)) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    const int f = ct::fsim_factor(h, w), hp = h / f, wp = w / f, P = hp * wp, imgs = 2 * batch;
    if (hp < 2 || wp < 2 || imgs * ct::kFsO > 65535) return CT_E_BADARG;
    if (!ct::fft2d_supported(hp, wp)) return CT_E_BADARG;
    const ct::FsimLayout l = ct::fsim_layout(ws, imgs, P);
    if (ws_bytes < l.total) return CT_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 gp((P + 255) / 256, imgs);
    const float invP = 1.0f / (float)P;
    { const int zr = ct::zero_async(l.hist, (size_t)imgs * ct::kFsO * 2048 * sizeof(unsigned int), s); if (zr) return zr; }
    hipLaunchKernelGGL(ct::fsim_prep_kernel, gp, dim3(256), 0, s, a, b, h, w, f, hp, wp, l.lum, l.iq);
    CT_CHECK_LAUNCH();
    hipLaunchKernelGGL(ct::fsim_grad_kernel, gp, dim3(256), 0, s, (const float2 *)l.lum, hp, wp, l.grad);
    CT_CHECK_LAUNCH();
    { const int rc = ct::fft2d_c2c(l.lum, hp, wp, imgs, -1, s); if (rc) return rc; }
    hipLaunchKernelGGL(ct::fsim_apply_filters_kernel, gp, dim3(256), 0, s, (const float2 *)l.lum, filters, P, l.eo);
    CT_CHECK_LAUNCH();
    { const int rc = ct::fft2d_c2c(l.eo, hp, wp, imgs * ct::kFsK, +1, s); if (rc) return rc; }
    // exact lower median of the scale-0 energy per (image, orientation): 11 + 11 + 10 key bits
    const int planes = imgs * ct::kFsO;
    int hb = (P + 256 * 8 - 1) / (256 * 8);
    if (hb > 64) hb = 64;
    const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(ct::fsim_select_hist_kernel, dim3(hb, planes), dim3(256), 0, s, (const float2 *)l.eo, P, invP, shifts[pass], nbits[pass],
                           pass == 0 ? 1 : 0, (const unsigned int *)l.sel, l.hist);
        CT_CHECK_LAUNCH();
        hipLaunchKernelGGL(ct::fsim_select_pick_kernel, dim3(planes), dim3(64), 0, s, l.hist, l.sel, nbits[pass], pass == 0 ? 1 : 0, pass == 2 ? 1 : 0,
                           (unsigned int)((P - 1) / 2), consts, l.thr, 2.0f);
        CT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ct::fsim_pc_kernel, gp, dim3(256), 0, s, (const float2 *)l.eo, P, invP, (const float *)l.thr, l.pc);
    CT_CHECK_LAUNCH();
    int sb = (P + 255) / 256;
    if (sb > ct::kFsScoreBlocks) sb = ct::kFsScoreBlocks;
    hipLaunchKernelGGL(ct::fsim_score_kernel, dim3(sb, batch), dim3(256), 0, s, (const float *)l.pc, (const float *)l.grad, (const float *)l.iq, P,
                       l.part);
    CT_CHECK_LAUNCH();
    hipLaunchKernelGGL(ct::fsim_finish_kernel, dim3(batch), dim3(64), 0, s, (const double *)l.part, sb, out);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // extern "C"
