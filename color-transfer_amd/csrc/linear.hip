// linear.hip -- global statistical colour transfers on gfx950 (MI355X).
//
// Replaces the numpy/skimage sweeps of the reference's methods/linear.py:
//   A1  rgb2lab + np.mean/np.std          (linear.py:25-26,33-36)  -> moments_kernel<T, true>
//   A2  affine in Lab + lab2rgb           (linear.py:38-40)        -> reinhard_apply_kernel
//   A3  np.mean + np.cov                  (linear.py:64-67,103-106)-> moments_kernel<T, false>
//   A5  (x - mu_t) @ A + mu_r             (linear.py:80,122)       -> affine3x3_kernel
//
// All of them are single coalesced HBM sweeps: a lane owns 4 whole HWC pixels (48 B of
// f32) per iteration, Lab never leaves registers, moments are reduced wave (__shfl_down)
// -> LDS -> one partial per workgroup -> a tiny finishing kernel, in a fixed order, so
// results are bitwise reproducible run to run (no float atomics).
//
// Moments use the shifted-data form with a common pivot K = value of pixel 0 of the image:
// S1 = sum(x-K), S2 = sum((x-K)(x-K)^T) are plainly additive across lanes/workgroups, and
// mean = K + S1/n, M2 = S2 - S1 S1^T / n is stable in float64 even for near-constant images.
#include <atomic>

#include "ct_reinhard.h"
#include "ct_reinhard_persist.h"

namespace ct {
int conv_split_read_status(bool clear);      // conv_split.hip

constexpr int kPartialStride = 12;  // doubles per workgroup partial (6 used for Lab, 9 for RGB cov)
constexpr int kPivotStride = 4;

struct WsLayout {
    double *partials;  // [n_images][kMaxBlocksPerImage][kPartialStride]
    double *pivots;    // [n_images][kPivotStride]
    double *stats;     // [n_images][CT_RGB_STATS_STRIDE] (only the fused entries use it)
};

static size_t ws_bytes_for(int n_images) {
    return (size_t)n_images * ((size_t)kMaxBlocksPerImage * kPartialStride + kPivotStride + CT_RGB_STATS_STRIDE) *
           sizeof(double);
}

static WsLayout ws_carve(void *ws, int n_images) {
    WsLayout l;
    l.partials = reinterpret_cast<double *>(ws);
    l.pivots = l.partials + (size_t)n_images * kMaxBlocksPerImage * kPartialStride;
    l.stats = l.pivots + (size_t)n_images * kPivotStride;
    return l;
}

// -------------------------------------------------------------------------------------------
// A1 / A3: first and second moments of Lab (LAB=true, 6 sums) or RGB (LAB=false, 9 sums)
// -------------------------------------------------------------------------------------------

// grid = (G, n_images). Images [0, n_first) live at base0, the rest at base1 (so that the
// targets and references of a batch of pairs are swept by ONE launch).
// CT_WPE: optional occupancy attribute for tuning builds (tools/build_variant.sh); forcing 8 waves/SIMD spills and
// is slower, the kernels are VALU-bound and insensitive to the grid (measured r01).
#ifndef CT_WPE
#define CT_WPE
#endif
template <typename T, bool LAB>
__global__ __launch_bounds__(kBlock) CT_WPE void moments_kernel(const T *__restrict__ base0, const T *__restrict__ base1,
                                                         int n_first, int64_t n_pixels, double *__restrict__ partials,
                                                         double *__restrict__ pivots) {
    constexpr int NV = LAB ? 6 : 9;
    __shared__ double lds[4 * NV];
    const int img = blockIdx.y;
    const T *p = (img < n_first) ? base0 + (size_t)img * n_pixels * 3 : base1 + (size_t)(img - n_first) * n_pixels * 3;
    const bool vec = (reinterpret_cast<uintptr_t>(p) & 15) == 0;

    double k[3] = {0.0, 0.0, 0.0};
    if (n_pixels > 0) to_space<LAB>((double)p[0], (double)p[1], (double)p[2], k[0], k[1], k[2]);

    double s[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) s[i] = 0.0;

    const int64_t n_chunks = n_pixels >> 2;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    // register double-buffer: the next chunk's loads are in flight while this one is converted
    int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    Raw12<T> cur, nxt;
    if (c < n_chunks) load12_raw<T>(p + c * 12, vec, cur);
    for (; c < n_chunks; c += stride) {
        if (c + stride < n_chunks) load12_raw<T>(p + (c + stride) * 12, vec, nxt);
        double v[12];
        unpack12<T>(cur, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            double x, y, z;
            to_space<LAB>(v[3 * q], v[3 * q + 1], v[3 * q + 2], x, y, z);
            accumulate<LAB>(s, k, x, y, z);
        }
        cur = nxt;
    }
    // ragged tail (n_pixels % 4 pixels): lanes 0..2 of workgroup 0
    if (blockIdx.x == 0) {
        const int64_t px = (n_chunks << 2) + threadIdx.x;
        if (threadIdx.x < 3 && px < n_pixels) {
            double x, y, z;
            to_space<LAB>((double)p[px * 3], (double)p[px * 3 + 1], (double)p[px * 3 + 2], x, y, z);
            accumulate<LAB>(s, k, x, y, z);
        }
    }
    block_sum<NV>(s, lds);
    if (threadIdx.x == 0) {
        double *dst = partials + ((size_t)img * kMaxBlocksPerImage + blockIdx.x) * kPartialStride;
#pragma unroll
        for (int i = 0; i < NV; ++i) dst[i] = s[i];
        if (blockIdx.x == 0) {
            pivots[img * kPivotStride + 0] = k[0];
            pivots[img * kPivotStride + 1] = k[1];
            pivots[img * kPivotStride + 2] = k[2];
        }
    }
}


// grid = n_images, one workgroup each: adds the G partials in a fixed order and writes the record.
template <bool LAB>
__global__ __launch_bounds__(kBlock) void moments_finalize_kernel(const double *__restrict__ partials,
                                                                  const double *__restrict__ pivots, int n_blocks,
                                                                  int64_t n_pixels, double *__restrict__ stats, double var_floor) {
    constexpr int NV = LAB ? 6 : 9;
    __shared__ double lds[4 * NV];
    const int img = blockIdx.x;
    double s[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) s[i] = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += kBlock) {
        const double *src = partials + ((size_t)img * kMaxBlocksPerImage + b) * kPartialStride;
#pragma unroll
        for (int i = 0; i < NV; ++i) s[i] += src[i];
    }
    block_sum<NV>(s, lds);
    if (threadIdx.x == 0) {
        const double n = (double)n_pixels;
        const double *k = pivots + img * kPivotStride;
        const double m0 = s[0] / n, m1 = s[1] / n, m2 = s[2] / n;  // mean of (x - K)
        if (LAB) {
            lab_record(s, k, n, stats + (size_t)img * CT_LAB_STATS_STRIDE, var_floor);
        } else {
            double *o = stats + (size_t)img * CT_RGB_STATS_STRIDE;
            o[0] = k[0] + m0; o[1] = k[1] + m1; o[2] = k[2] + m2;
            const double d = n - 1.0;  // np.cov default ddof = 1
            const double cxx = fma(-s[0], m0, s[3]) / d, cxy = fma(-s[0], m1, s[4]) / d, cxz = fma(-s[0], m2, s[5]) / d;
            const double cyy = fma(-s[1], m1, s[6]) / d, cyz = fma(-s[1], m2, s[7]) / d, czz = fma(-s[2], m2, s[8]) / d;
            o[3] = cxx; o[4] = cxy; o[5] = cxz;
            o[6] = cxy; o[7] = cyy; o[8] = cyz;
            o[9] = cxz; o[10] = cyz; o[11] = czz;
            o[12] = n; o[13] = 0.0; o[14] = 0.0; o[15] = 0.0;
        }
    }
}

// -------------------------------------------------------------------------------------------

template <typename T, bool OUT_LAB>
__global__ __launch_bounds__(kBlock) CT_WPE void reinhard_apply_kernel(const T *__restrict__ target,
                                                                const double *__restrict__ stats_t,
                                                                const double *__restrict__ stats_r, T *__restrict__ out,
                                                                int64_t n_pixels) {
    const int img = blockIdx.y;
    const T *p = target + (size_t)img * n_pixels * 3;
    T *o = out + (size_t)img * n_pixels * 3;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(o)) & 15) == 0;
    const ReinhardCoef c = reinhard_coef(stats_t + (size_t)img * CT_LAB_STATS_STRIDE,
                                         stats_r + (size_t)img * CT_LAB_STATS_STRIDE);
    const int64_t n_chunks = n_pixels >> 2;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t ch = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    Raw12<T> cur, nxt;
    if (ch < n_chunks) load12_raw<T>(p + ch * 12, vec, cur);
    for (; ch < n_chunks; ch += stride) {
        if (ch + stride < n_chunks) load12_raw<T>(p + (ch + stride) * 12, vec, nxt);
        double v[12];
        T w[12];
        unpack12<T>(cur, v);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            reinhard_pixel<T, OUT_LAB>(c, v[3 * q], v[3 * q + 1], v[3 * q + 2], w[3 * q], w[3 * q + 1], w[3 * q + 2]);
        store12<T>(o + ch * 12, vec, w);
        cur = nxt;
    }
    if (blockIdx.x == 0) {
        const int64_t px = (n_chunks << 2) + threadIdx.x;
        if (threadIdx.x < 3 && px < n_pixels) {
            T a, b, d;
            reinhard_pixel<T, OUT_LAB>(c, (double)p[px * 3], (double)p[px * 3 + 1], (double)p[px * 3 + 2], a, b, d);
            o[px * 3] = a; o[px * 3 + 1] = b; o[px * 3 + 2] = d;
        }
    }
}

// -------------------------------------------------------------------------------------------
// Table-driven variants of A1 / A2 for float32 images (ct_color_lut.h): the power functions are LDS look-ups.
//
// Work layout: a wave owns TILES of 256 consecutive pixels; lane l holds pixels l, l+64, l+128, l+192 of the tile, each
// fetched / stored with ONE 12-byte access (global_load/store_dwordx3): consecutive lanes touch consecutive bytes, so
// every wave instruction covers 768 contiguous bytes whatever the alignment of the image (measured,
// tools/ubench/stream_patterns.hip: a copy runs at 5.5 TB/s this way against 4.6 TB/s with three 16-byte accesses per lane
// at a 48-byte lane stride).  512-thread workgroups share one 32 KB (statistics) / 37 KB (apply) table image and are
// persistent: one round of resident workgroups sweeps all tiles.  A tile whose wave holds any value outside [0,1] (or a
// NaN) is computed with the exact float64 code of ct_color.h, pixel by pixel.
// -------------------------------------------------------------------------------------------
// minimum waves per SIMD the table kernels are compiled for (register budget): tuning builds override
#ifndef CT_LUT_WPE_STATS
#define CT_LUT_WPE_STATS 4
#endif
#ifndef CT_LUT_WPE_APPLY
#define CT_LUT_WPE_APPLY 4
#endif
#ifndef CT_LUT_PREFETCH    // 1: register double buffer of the next tile, one 12-register copy per tile (default); 2: two tiles ahead; 0: rely on
                           // occupancy; 3: two register sets in alternating roles, loop unrolled by two, no copy -- measured round 4, same box and
                           // session: 37.0 k pairs/s against 37.5 k for 1 (the copies cost less than the unrolled loop's registers: 127 + 2 spilled)
#define CT_LUT_PREFETCH 1
#endif


// A1, float32 arithmetic on the table path (see ct_color_lut.h: statistics only need unbiased per-pixel values): per lane
// float32 shifted sums over its ~40 pixels, converted once to float64 for the fixed-shape reduction tree.  The exact
// fallback (out-of-range tiles, the ragged last tile) accumulates in float64 beside it.
__global__ __launch_bounds__(kLutBlock, CT_LUT_WPE_STATS) void lab_moments_lut_kernel(const float *__restrict__ base0,
                                                                                      const float *__restrict__ base1, int n_first,
                                                                                      int64_t n_pixels, double *__restrict__ partials,
                                                                                      double *__restrict__ pivots) {
    __shared__ __attribute__((aligned(16))) unsigned char tab[lut::kLdsBytesFwd];
    __shared__ double red[kLutWaves * 6];
    __shared__ float piv[3];
    const int img = blockIdx.y;
    const float *p = (img < n_first) ? base0 + (size_t)img * n_pixels * 3 : base1 + (size_t)(img - n_first) * n_pixels * 3;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t n_full = n_pixels / kTilePixels;                   // full tiles; the ragged rest is swept by workgroup 0
    const int64_t stride = (int64_t)gridDim.x * kLutWaves;
    int64_t t = (int64_t)blockIdx.x * kLutWaves + (threadIdx.x >> 6);
    float cur[12];
#if CT_LUT_PREFETCH
    float nxt[12];
#endif
#if CT_LUT_PREFETCH == 2
    float nx2[12];
#endif
#ifdef CT_DIAG_CLOCK      // diagnostic build: shader clock held during this kernel = d(s_memtime) / d(s_memrealtime) x 100 MHz
    const uint64_t clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (t < n_full) load_tile(p + t * (kTilePixels * 3), lane, cur);          // in flight while the tables are copied
#if CT_LUT_PREFETCH == 2
    if (t + stride < n_full) load_tile(p + (t + stride) * (kTilePixels * 3), lane, nxt);
#endif
    float p0[3] = {0.f, 0.f, 0.f};
    if (threadIdx.x == 0 && n_pixels > 0) { p0[0] = p[0]; p0[1] = p[1]; p0[2] = p[2]; }
    lut::load_tables<kLutBlock, false>(tab);
    __syncthreads();
    // Pivot of the shifted sums: the values of pixel 0 of the image (any point inside the data's range keeps the variance
    // well conditioned), put on a 2^-10 grid: (a float32 difference, on a 2^-26 grid) - (a pivot with finer bits) would round
    // the SAME way for every pixel -- a bias of half an ulp (measured: 4e-9, i.e. 2e-6 in mean a*).  Out-of-range pixel 0: 0.5.
    if (threadIdx.x == 0) {
        float fy = 0.5f, dxy = 0.0f, dyz = 0.0f;
        if (max(max(__float_as_uint(p0[0]), __float_as_uint(p0[1])), __float_as_uint(p0[2])) <= lut::kOneBits) {
            lut::rgb_to_f_stats(tab, p0[0], p0[1], p0[2], fy, dxy, dyz);
        }
        piv[0] = rintf(fy * 1024.0f) * (1.0f / 1024.0f);
        piv[1] = rintf(dxy * 1024.0f) * (1.0f / 1024.0f);
        piv[2] = rintf(dyz * 1024.0f) * (1.0f / 1024.0f);
    }
    __syncthreads();
    const float kf[3] = {__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(piv[0]))),
                         __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(piv[1]))),
                         __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(piv[2])))};
    const double kd[3] = {(double)kf[0], (double)kf[1], (double)kf[2]};
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    float sf[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto stats_tile = [&](float (&c)[12]) {
#ifdef CT_ABL_NOMATH
        if (true) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { sf[0] += c[3 * q]; sf[1] += c[3 * q + 1]; sf[2] += c[3 * q + 2]; }
        } else
#endif
#ifdef CT_ABL_NOSLOW
        if (false) {
#else
        if (__builtin_amdgcn_ballot_w64(max_bits12(c) > lut::kOneBits)) {
#endif
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                double x, y, z;
                to_space<true>((double)c[0], (double)c[1], (double)c[2], x, y, z);
                accumulate<true>(s, kd, x, y, z);
                rotate_pixels(c);
                asm volatile("" : "+v"(c[0]));     // keep the loop rolled
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float fy, dxy, dyz;
#ifdef CT_STATS_CBRT_LUT
                lut::rgb_to_f_stats(tab, c[3 * q], c[3 * q + 1], c[3 * q + 2], fy, dxy, dyz);
#else
                lut::rgb_to_f_stats_hw(tab, c[3 * q], c[3 * q + 1], c[3 * q + 2], fy, dxy, dyz);
#endif
                const float dx = fy - kf[0], dy = dxy - kf[1], dz = dyz - kf[2];
                sf[0] += dx; sf[1] += dy; sf[2] += dz;
                sf[3] = fmaf(dx, dx, sf[3]); sf[4] = fmaf(dy, dy, sf[4]); sf[5] = fmaf(dz, dz, sf[5]);
            }
        }
    };
#if CT_LUT_PREFETCH == 3
    // ping-pong: the two register sets swap roles and the loop is unrolled by two -- no 12-register copy per tile (3 of the 73
    // vector instructions per pixel)
    for (; t < n_full; t += 2 * stride) {
        const bool more = t + stride < n_full;
        if (more) load_tile(p + (t + stride) * (kTilePixels * 3), lane, nxt);
        stats_tile(cur);
        if (!more) break;
        if (t + 2 * stride < n_full) load_tile(p + (t + 2 * stride) * (kTilePixels * 3), lane, cur);
        stats_tile(nxt);
    }
#else
    for (; t < n_full; t += stride) {
#if CT_LUT_PREFETCH == 2
        if (t + 2 * stride < n_full) load_tile(p + (t + 2 * stride) * (kTilePixels * 3), lane, nx2);
#elif CT_LUT_PREFETCH
#ifdef CT_ABL_NOLOAD
        if (t == (int64_t)blockIdx.x * kLutWaves + (threadIdx.x >> 6))
#endif
        if (t + stride < n_full) load_tile(p + (t + stride) * (kTilePixels * 3), lane, nxt);
#endif
        stats_tile(cur);
#if CT_LUT_PREFETCH == 2
#pragma unroll
        for (int i = 0; i < 12; ++i) { cur[i] = nxt[i]; nxt[i] = nx2[i]; }
#elif CT_LUT_PREFETCH
#pragma unroll
        for (int i = 0; i < 12; ++i) cur[i] = nxt[i];
#else
        if (t + stride < n_full) load_tile(p + (t + stride) * (kTilePixels * 3), lane, cur);
#endif
    }
#endif
    if (blockIdx.x == 0) {                                           // ragged tail (n_pixels % 256), exact arithmetic
        const int64_t px = n_full * kTilePixels + threadIdx.x;
        if (threadIdx.x < kTilePixels && px < n_pixels) {
            double x, y, z;
            to_space<true>((double)p[px * 3], (double)p[px * 3 + 1], (double)p[px * 3 + 2], x, y, z);
            accumulate<true>(s, kd, x, y, z);
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) s[i] += (double)sf[i];
    block_sum_n<6, kLutWaves>(s, red);
    if (threadIdx.x == 0) {
        double *dst = partials + ((size_t)img * kMaxBlocksPerImage + blockIdx.x) * kPartialStride;
#pragma unroll
        for (int i = 0; i < 6; ++i) dst[i] = s[i];
        if (blockIdx.x == 0) {
            pivots[img * kPivotStride + 0] = kd[0];
            pivots[img * kPivotStride + 1] = kd[1];
            pivots[img * kPivotStride + 2] = kd[2];
#ifdef CT_DIAG_CLOCK
            const uint64_t clk1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime();
            pivots[img * kPivotStride + 3] = (double)(clk1 - clk0) / (double)(rt1 - rt0) * 0.1;     // GHz
#endif
        }
    }
}

// A2 on the table path: float32 difference forms (ct_color_lut.h: a* = 500 (fx - fy) must be right per pixel).  Affine scales
// above kFastScale (the forward error grows with them), pixels outside [0,1] and pixels within rounding of a kink of Lab's f()
// take the exact float64 code, tile by tile.
template <bool OUT_LAB>
__global__ __launch_bounds__(kLutBlock, CT_LUT_WPE_APPLY) void reinhard_apply_lut_kernel(const float *__restrict__ target,
                                                                                         const double *__restrict__ stats_t,
                                                                                         const double *__restrict__ stats_r,
                                                                                         float *__restrict__ out, int64_t n_pixels,
                                                                                         const double *__restrict__ partials,
                                                                                         const double *__restrict__ pivots, int n_partials,
                                                                                         int batch, double *__restrict__ stats_out,
                                                                                         const float *__restrict__ gt, double *__restrict__ sq_partials) {
    // gt != NULL: the per-frame squared error of the result against a ground-truth frame (the PSNR of Runner.test_step,
    // methods/__init__.py:32) is accumulated on the way out -- the corrected frame is not read back from HBM for it
    __shared__ __attribute__((aligned(16))) unsigned char tab[OUT_LAB ? lut::kLdsBytesFwd : lut::kLdsBytesAll];
    __shared__ double fin[12 * 8 + 2 * CT_LAB_STATS_STRIDE];
    const int img = blockIdx.y;
    const float *p = target + (size_t)img * n_pixels * 3;
    float *o = out + (size_t)img * n_pixels * 3;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t n_full = n_pixels / kTilePixels;
    const int64_t stride = (int64_t)gridDim.x * kLutWaves;
    int64_t t = (int64_t)blockIdx.x * kLutWaves + (threadIdx.x >> 6);
    float cur[12];
#if CT_LUT_PREFETCH
    float nxt[12];
#endif
    if (t < n_full) load_tile(p + t * (kTilePixels * 3), lane, cur);
    // Fused call (partials != NULL): the finishing step of the statistics sweep is done here, by every workgroup for its
    // own pair -- the partial sums of the target (image img) and the reference (image batch + img) are added in a fixed
    // order (8 interleaved chains per moment, then in chain order), identical in every workgroup -- instead of a
    // one-workgroup-per-image kernel of its own (5.5 us + a launch boundary per call).
    if (partials != nullptr && threadIdx.x < 96) {
        const int g = threadIdx.x >> 3, sub = threadIdx.x & 7;          // g = which * 6 + moment
        const int image = (g >= 6) ? batch + img : img;
        const double *src = partials + (size_t)image * kMaxBlocksPerImage * kPartialStride + (g % 6);
        double a = 0.0;
        for (int b = sub; b < n_partials; b += 8) a += src[(size_t)b * kPartialStride];
        fin[g * 8 + sub] = a;
    }
    lut::load_tables<kLutBlock, !OUT_LAB>(tab);
    const double *rec_t = stats_t + (size_t)img * CT_LAB_STATS_STRIDE, *rec_r = stats_r + (size_t)img * CT_LAB_STATS_STRIDE;
    if (partials != nullptr) {
        __syncthreads();
        if (threadIdx.x < 2) {
            const int image = threadIdx.x ? batch + img : img;
            double sum[6];
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                const double *f = fin + (threadIdx.x * 6 + m) * 8;
                sum[m] = ((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7];
            }
            double *rec = fin + 96 + threadIdx.x * CT_LAB_STATS_STRIDE;
            lab_record(sum, pivots + image * kPivotStride, (double)n_pixels, rec, kVarFloorF32);      // the fused call: float32 sweep
            if (blockIdx.x == 0 && stats_out != nullptr) {
#pragma unroll
                for (int m = 0; m < CT_LAB_STATS_STRIDE; ++m) stats_out[(size_t)image * CT_LAB_STATS_STRIDE + m] = rec[m];
            }
        }
        __syncthreads();
        rec_t = fin + 96;
        rec_r = fin + 96 + CT_LAB_STATS_STRIDE;
    }
    ReinhardCoef c = reinhard_coef(rec_t, rec_r);
    c.sL = uniform_f64(c.sL); c.sa = uniform_f64(c.sa); c.sb = uniform_f64(c.sb);
    c.cy = uniform_f64(c.cy); c.ca = uniform_f64(c.ca); c.cb = uniform_f64(c.cb);
    // the table path needs finite, moderate coefficients (every intermediate finite, the float32 error budget kept); anything
    // else -- a constant target gives inf / nan like the reference -- goes through the exact code
    const bool coef_bad = !reinhard_coef_fast(c);
    const float sLf = (float)c.sL, saf = (float)c.sa, sbf = (float)c.sb, cyf = (float)c.cy, caf = (float)c.ca, cbf = (float)c.cb;
    double sq = 0.0;
    __syncthreads();
    auto apply_one = [&](float (&cc)[12], int64_t tt) {
        float w[12];
#ifdef CT_APPLY_GT_EARLY
        float gv[12];
        if (gt != nullptr) load_tile(gt + ((size_t)img * n_pixels + (size_t)tt * kTilePixels) * 3, lane, gv);
#endif
        bool slow = false, skip = false;
#ifdef CT_ABL_NOMATH
        skip = true;
#pragma unroll
        for (int i = 0; i < 12; ++i) w[i] = cc[i] * sLf;
#endif
#ifndef CT_ABL_NOSLOW
        slow = coef_bad || __builtin_amdgcn_ballot_w64(max_bits12(cc) > lut::kOneBits);
#endif
        if (!slow && !skip) {
            bool near = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float fy, dxy, dyz;
                near |= lut::rgb_to_f(tab, cc[3 * q], cc[3 * q + 1], cc[3 * q + 2], fy, dxy, dyz);
                const float gy = fmaf(sLf, fy, cyf), dx = fmaf(saf, dxy, caf), dz = fmaf(sbf, dyz, cbf);
                if (OUT_LAB) {
                    w[3 * q] = fmaf(116.0f, gy, -16.0f); w[3 * q + 1] = 500.0f * dx; w[3 * q + 2] = 200.0f * dz;
                } else {
                    near |= lut::f_to_rgb_clip(tab, gy, dx, dz, w[3 * q], w[3 * q + 1], w[3 * q + 2]);
                }
            }
#ifndef CT_ABL_NOSLOW
            slow = __builtin_amdgcn_ballot_w64(near) != 0;        // a pixel within rounding of a kink of f(): the whole tile again, exactly
#endif
        }
        if (slow && !skip) {
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                rotate_pixels(w);                      // the result of pixel q lands in slot 3 and ends in slot q
                reinhard_pixel<float, OUT_LAB>(c, (double)cc[0], (double)cc[1], (double)cc[2], w[9], w[10], w[11]);
                rotate_pixels(cc);
                asm volatile("" : "+v"(cc[0]));       // keep the loop rolled
            }
        }
#ifdef CT_ABL_NOSTORE
        if (w[0] == 123.456f)
#endif
        store_tile(o + tt * (kTilePixels * 3), lane, w);
#ifdef CT_ABL_NOGT
        if (false) {
#else
        if (gt != nullptr) {
#endif
#ifndef CT_APPLY_GT_EARLY
            float gv[12];
            load_tile(gt + ((size_t)img * n_pixels + (size_t)tt * kTilePixels) * 3, lane, gv);
#endif
            float e = 0.f;
#pragma unroll
            for (int i = 0; i < 12; ++i) { const float d = w[i] - gv[i]; e = fmaf(d, d, e); }
            sq += (double)e;
        }
    };
#if CT_LUT_PREFETCH == 3
    for (; t < n_full; t += 2 * stride) {              // ping-pong register sets: see lab_moments_lut_kernel
        const bool more = t + stride < n_full;
        if (more) load_tile(p + (t + stride) * (kTilePixels * 3), lane, nxt);
        apply_one(cur, t);
        if (!more) break;
        if (t + 2 * stride < n_full) load_tile(p + (t + 2 * stride) * (kTilePixels * 3), lane, cur);
        apply_one(nxt, t + stride);
    }
#else
    for (; t < n_full; t += stride) {
#if CT_LUT_PREFETCH
#ifdef CT_ABL_NOLOAD
        if (t == (int64_t)blockIdx.x * kLutWaves + (threadIdx.x >> 6))
#endif
        if (t + stride < n_full) load_tile(p + (t + stride) * (kTilePixels * 3), lane, nxt);
#endif
        apply_one(cur, t);
#if CT_LUT_PREFETCH
#pragma unroll
        for (int i = 0; i < 12; ++i) cur[i] = nxt[i];
#else
        if (t + stride < n_full) load_tile(p + (t + stride) * (kTilePixels * 3), lane, cur);
#endif
    }
#endif
    if (blockIdx.x == 0) {                                           // ragged tail (n_pixels % 256), exact arithmetic
        const int64_t px = n_full * kTilePixels + threadIdx.x;
        if (threadIdx.x < kTilePixels && px < n_pixels) {
            float a, b, d;
            reinhard_pixel<float, OUT_LAB>(c, (double)p[px * 3], (double)p[px * 3 + 1], (double)p[px * 3 + 2], a, b, d);
            o[px * 3] = a; o[px * 3 + 1] = b; o[px * 3 + 2] = d;
            if (gt != nullptr) {
                const float *gp = gt + ((size_t)img * n_pixels + px) * 3;
                const double d0 = (double)a - gp[0], d1 = (double)b - gp[1], d2 = (double)d - gp[2];
                sq += (d0 * d0 + d1 * d1) + d2 * d2;
            }
        }
    }
    if (gt != nullptr) {                                             // uniform per launch
        double v[1] = {sq};
        __syncthreads();                                             // fin[] is free again
        block_sum_n<1, kLutWaves>(v, fin);
        if (threadIdx.x == 0) sq_partials[(size_t)img * kMaxBlocksPerImage + blockIdx.x] = v[0];
    }
}

// -------------------------------------------------------------------------------------------
// A5: out = (x - mu_t) @ A + mu_r, float64 arithmetic, unclipped
// -------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(kBlock) void affine3x3_kernel(const TI *__restrict__ in, const double *__restrict__ coef,
                                                           TO *__restrict__ out, int64_t n_pixels) {
    const int img = blockIdx.y;
    const TI *p = in + (size_t)img * n_pixels * 3;
    TO *o = out + (size_t)img * n_pixels * 3;
    const bool vin = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    const bool vout = (reinterpret_cast<uintptr_t>(o) & 15) == 0;
    const double *cf = coef + (size_t)img * 16;
    double A[9], mt[3], mr[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = cf[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { mt[i] = cf[9 + i]; mr[i] = cf[12 + i]; }

    auto px = [&](double r, double g, double b, TO &o0, TO &o1, TO &o2) {
        const double d0 = r - mt[0], d1 = g - mt[1], d2 = b - mt[2];
        // same order as a row-vector @ matrix product: sum over i of d_i A[i][j], then + mu_r
        o0 = (TO)(fma(d2, A[6], fma(d1, A[3], d0 * A[0])) + mr[0]);
        o1 = (TO)(fma(d2, A[7], fma(d1, A[4], d0 * A[1])) + mr[1]);
        o2 = (TO)(fma(d2, A[8], fma(d1, A[5], d0 * A[2])) + mr[2]);
    };
    const int64_t n_chunks = n_pixels >> 2;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    constexpr bool kF32IO = sizeof(TI) == 4 && sizeof(TO) == 4;
    __shared__ float xpose[kF32IO ? kBlock * 12 : 1];   // per-wave 3 KiB transpose buffers (float I/O only)
    for (int64_t ch0 = (int64_t)blockIdx.x * kBlock; ch0 < n_chunks; ch0 += stride) {
        const int64_t ch = ch0 + threadIdx.x;
        const int64_t wave_c0 = ch0 + (threadIdx.x & ~63);
        const bool full_wave = wave_c0 + 64 <= n_chunks;          // wave-uniform
        if (kF32IO && vin && vout && full_wave) {
            // fully coalesced 16-byte global accesses (lane i <-> base + 16 i, three times per wave); the HWC de-interleave
            // into "4 whole pixels per lane" and back happens in a per-wave LDS buffer (conflict-free b128 accesses)
            const int lane = threadIdx.x & 63;
            float *lw = xpose + (threadIdx.x >> 6) * (64 * 12);
            const float4 *g = reinterpret_cast<const float4 *>(p + wave_c0 * 12);
            float4 *l4 = reinterpret_cast<float4 *>(lw);
            l4[lane] = g[lane]; l4[64 + lane] = g[64 + lane]; l4[128 + lane] = g[128 + lane];
            __builtin_amdgcn_wave_barrier();
            const float4 *r4 = reinterpret_cast<const float4 *>(lw + lane * 12);
            const float4 a0 = r4[0], a1 = r4[1], a2 = r4[2];
            const float vi[12] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w};
            TO w[12];
#pragma unroll
            for (int q = 0; q < 4; ++q) px((double)vi[3 * q], (double)vi[3 * q + 1], (double)vi[3 * q + 2], w[3 * q], w[3 * q + 1], w[3 * q + 2]);
            __builtin_amdgcn_wave_barrier();
            float4 *w4 = reinterpret_cast<float4 *>(lw + lane * 12);
            w4[0] = make_float4((float)w[0], (float)w[1], (float)w[2], (float)w[3]);
            w4[1] = make_float4((float)w[4], (float)w[5], (float)w[6], (float)w[7]);
            w4[2] = make_float4((float)w[8], (float)w[9], (float)w[10], (float)w[11]);
            __builtin_amdgcn_wave_barrier();
            float4 *go = reinterpret_cast<float4 *>(reinterpret_cast<float *>(o) + wave_c0 * 12);
            go[lane] = l4[lane]; go[64 + lane] = l4[64 + lane]; go[128 + lane] = l4[128 + lane];
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (ch >= n_chunks) continue;
        double v[12];
        TO w[12];
        load12<TI>(p + ch * 12, vin, v);
#pragma unroll
        for (int q = 0; q < 4; ++q) px(v[3 * q], v[3 * q + 1], v[3 * q + 2], w[3 * q], w[3 * q + 1], w[3 * q + 2]);
        store12<TO>(o + ch * 12, vout, w);
    }
    if (blockIdx.x == 0) {
        const int64_t q = (n_chunks << 2) + threadIdx.x;
        if (threadIdx.x < 3 && q < n_pixels) {
            TO a, b, d;
            px((double)p[q * 3], (double)p[q * 3 + 1], (double)p[q * 3 + 2], a, b, d);
            o[q * 3] = a; o[q * 3 + 1] = b; o[q * 3 + 2] = d;
        }
    }
}

// -------------------------------------------------------------------------------------------
// A4 (sync-free variant): the 3x3 algebra of monge_kantorovitch_color_transfer on the device
// (methods/linear.py:108-118).  One thread per pair, float64.  The matrix square root of a symmetric
// positive (semi)definite 3x3 is V diag(sqrt(lambda)) V^T from a cyclic Jacobi eigen-decomposition;
// it is unique, so no LAPACK sign convention is involved (Xiao's SVD-based T is NOT sign invariant
// and stays on the host).  mode: 0 = "MK", 1 = "sqrt", 2 = "cholesky".
// coef[b] = { T (row-major, out = (x - mu_t) @ T + mu_r), mu_t, mu_r, 0 }.
// -------------------------------------------------------------------------------------------
struct M3 { double a[3][3]; };

__device__ inline M3 m3_mul(const M3 &x, const M3 &y) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.a[i][j] = fma(x.a[i][2], y.a[2][j], fma(x.a[i][1], y.a[1][j], x.a[i][0] * y.a[0][j]));
    return r;
}

__device__ inline void m3_eig_sym(M3 s, M3 &v, double (&lam)[3]) {   // s = v diag(lam) v^T, cyclic Jacobi
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v.a[i][j] = (i == j) ? 1.0 : 0.0;
    const double tiny = 1e-32 * (fabs(s.a[0][0]) + fabs(s.a[1][1]) + fabs(s.a[2][2]));
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(s.a[0][1]) + fabs(s.a[0][2]) + fabs(s.a[1][2]);
        if (off <= tiny) break;          // quadratic convergence: 4-5 sweeps for a 3x3
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = s.a[p][q];
                if (apq == 0.0) continue;
                const double theta = (s.a[q][q] - s.a[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; ++k) {   // columns p, q of s
                    const double skp = s.a[k][p], skq = s.a[k][q];
                    s.a[k][p] = c * skp - sn * skq;
                    s.a[k][q] = sn * skp + c * skq;
                }
                for (int k = 0; k < 3; ++k) {   // rows p, q of s
                    const double spk = s.a[p][k], sqk = s.a[q][k];
                    s.a[p][k] = c * spk - sn * sqk;
                    s.a[q][k] = sn * spk + c * sqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v.a[k][p], vkq = v.a[k][q];
                    v.a[k][p] = c * vkp - sn * vkq;
                    v.a[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; ++i) lam[i] = s.a[i][i];
}

__device__ inline M3 m3_from_eig(const M3 &v, const double (&lam)[3], int fn) {   // v f(diag(lam)) v^T
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) {
                const double f = fn == 0 ? sqrt(lam[k]) : 1.0 / sqrt(lam[k]);
                acc = fma(v.a[i][k] * f, v.a[j][k], acc);
            }
            r.a[i][j] = acc;
        }
    return r;
}

__device__ inline M3 m3_fun_sym(const M3 &s, int fn) {   // fn 0: sqrt, 1: inverse sqrt  (of a symmetric PSD matrix)
    M3 v;
    double lam[3];
    m3_eig_sym(s, v, lam);
    return m3_from_eig(v, lam, fn);
}

__device__ inline M3 m3_chol(const M3 &s) {   // lower L with L L^T = s
    M3 l;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) l.a[i][j] = 0.0;
    l.a[0][0] = sqrt(s.a[0][0]);
    l.a[1][0] = s.a[1][0] / l.a[0][0];
    l.a[2][0] = s.a[2][0] / l.a[0][0];
    l.a[1][1] = sqrt(s.a[1][1] - l.a[1][0] * l.a[1][0]);
    l.a[2][1] = (s.a[2][1] - l.a[2][0] * l.a[1][0]) / l.a[1][1];
    l.a[2][2] = sqrt(s.a[2][2] - l.a[2][0] * l.a[2][0] - l.a[2][1] * l.a[2][1]);
    return l;
}

__device__ inline M3 m3_inv_lower(const M3 &l) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.a[i][j] = 0.0;
    r.a[0][0] = 1.0 / l.a[0][0];
    r.a[1][1] = 1.0 / l.a[1][1];
    r.a[2][2] = 1.0 / l.a[2][2];
    r.a[1][0] = -l.a[1][0] * r.a[0][0] * r.a[1][1];
    r.a[2][1] = -l.a[2][1] * r.a[1][1] * r.a[2][2];
    r.a[2][0] = -(l.a[2][0] * r.a[0][0] + l.a[2][1] * r.a[1][0]) * r.a[2][2];
    return r;
}

__global__ void mk_coef_kernel(const double *__restrict__ stats_t, const double *__restrict__ stats_r, int mode, int batch,
                               double *__restrict__ coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const double *st = stats_t + (size_t)b * CT_RGB_STATS_STRIDE, *sr = stats_r + (size_t)b * CT_RGB_STATS_STRIDE;
    M3 ct_, cr;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) { ct_.a[i][j] = st[3 + 3 * i + j]; cr.a[i][j] = sr[3 + 3 * i + j]; }
    M3 T;
    if (mode == 0) {            // A = sqrtm(St); T = A^-1 sqrtm(A Sr A) A^-1          (linear.py:116-118)
        M3 vt;
        double lt[3];
        m3_eig_sym(ct_, vt, lt);
        const M3 A = m3_from_eig(vt, lt, 0), Ai = m3_from_eig(vt, lt, 1);
        M3 mid = m3_mul(m3_mul(A, cr), A);
        for (int i = 0; i < 3; ++i)   // symmetrise the rounding residue before the eigen-decomposition
            for (int j = i + 1; j < 3; ++j) { const double h = 0.5 * (mid.a[i][j] + mid.a[j][i]); mid.a[i][j] = h; mid.a[j][i] = h; }
        T = m3_mul(m3_mul(Ai, m3_fun_sym(mid, 0)), Ai);
    } else if (mode == 1) {     // T = sqrtm(Sr) sqrtm(St)^-1                         (linear.py:112-115)
        T = m3_mul(m3_fun_sym(cr, 0), m3_fun_sym(ct_, 1));
    } else {                    // T = chol(Sr) chol(St)^-1                           (linear.py:108-111)
        T = m3_mul(m3_chol(cr), m3_inv_lower(m3_chol(ct_)));
    }
    double *o = coef + (size_t)b * 16;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o[3 * i + j] = T.a[i][j];
    for (int i = 0; i < 3; ++i) { o[9 + i] = st[i]; o[12 + i] = sr[i]; }
    o[15] = 0.0;
}

// -------------------------------------------------------------------------------------------
// host-side launchers
// -------------------------------------------------------------------------------------------
template <typename T>
static int check_image_args(const T *p, int64_t n_pixels, int n_images) {
    if (n_pixels < 0 || n_images < 0) return CT_E_BADARG;
    if (n_images > 0 && n_pixels > 0 && p == nullptr) return CT_E_BADARG;
    if (reinterpret_cast<uintptr_t>(p) % sizeof(T)) return CT_E_ALIGN;
    return CT_OK;
}

static int check_ws(const void *ws, size_t ws_bytes, int n_images) {
    if (ws == nullptr || (reinterpret_cast<uintptr_t>(ws) & 15)) return CT_E_WORKSPACE;
    if (ws_bytes < ws_bytes_for(n_images)) return CT_E_WORKSPACE;
    return CT_OK;
}

// Optional HIP events bracketing the two streaming kernels of the Reinhard path (bench.py's roofline measurement):
// set with ct_profile_events(); NULL = off.  Recorded on the launch stream, so they time exactly one kernel.
static thread_local hipEvent_t g_prof_evt[4] = {nullptr, nullptr, nullptr, nullptr};     // per calling thread: the thread that set them launches the kernels they bracket

// Lab arithmetic of the float32 entries: 0 = table-driven (ct_color_lut.h, default), 1 = exact float64 (ct_color.h).
// float64 images always take the exact path.  Two levels: a process-wide default (ct_set_lab_mode / env CT_HIP_LAB; atomic) and a
// per-thread override (ct_set_lab_mode_thread; -1 = none), so that two host threads driving different streams with different
// modes do not race on one global -- every entry reads the mode once, through lab_mode(), on the calling thread.
static std::atomic<int> g_lab_mode_default{[] { const char *e = getenv("CT_HIP_LAB"); return (e && e[0] == 'e') ? 1 : 0; }()};
static thread_local int t_lab_mode = -1;
static inline int lab_mode() { return t_lab_mode >= 0 ? t_lab_mode : g_lab_mode_default.load(std::memory_order_relaxed); }

// Workgroups of a table kernel: ONE round of persistent workgroups -- as many as are resident at once (occupancy query,
// cached per kernel), each sweeping enough chunks to amortise its 32-37 KB table copy.  CT_HIP_LUT_BLOCKS overrides the
// total (tuning).
template <typename K>
static int resident_blocks(K kernel) {
    int dev = 0, per_cu = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kLutBlock, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    (void)hipGetLastError();
    return per_cu * cus;
}
static int lut_blocks_per_image(int resident, int64_t n_tiles, int n_images) {
    static int forced = [] { const char *e = getenv("CT_HIP_LUT_BLOCKS"); return e ? atoi(e) : 0; }();
    const int total = forced > 0 ? forced : resident;
    int64_t want = (n_tiles + kLutWaves - 1) / kLutWaves;
    int64_t cap = total / (n_images > 0 ? n_images : 1);
    if (cap < 4) cap = 4;
    if (cap > kMaxBlocksPerImage) cap = kMaxBlocksPerImage;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

template <typename T, bool LAB>
static int launch_moments(const T *base0, const T *base1, int n_first, int n_images, int64_t n_pixels,
                          const WsLayout &l, double *stats, hipStream_t s, int *deferred_partials = nullptr) {
    // deferred_partials: the caller's next kernel finishes the statistics itself (fused table-path Reinhard); receives the
    // number of partial sums per image, or stays 0 when this launch took a path that finishes here
    if (n_images == 0) return CT_OK;
    constexpr bool kLut = LAB && sizeof(T) == 4;
    const bool use_lut = kLut && lab_mode() == 0;
    int G = blocks_per_image(n_pixels >> 2, n_images);
    if constexpr (kLut) {
        static const int resident = resident_blocks(lab_moments_lut_kernel);
        if (use_lut) G = lut_blocks_per_image(resident, n_pixels / kTilePixels, n_images);
    }
    if (LAB && g_prof_evt[0]) (void)hipEventRecord(g_prof_evt[0], s);
    if constexpr (kLut) {
        if (use_lut)
            hipLaunchKernelGGL(lab_moments_lut_kernel, dim3(G, n_images), dim3(kLutBlock), 0, s, base0, base1, n_first, n_pixels,
                               l.partials, l.pivots);
    }
    if (!use_lut)
        hipLaunchKernelGGL((moments_kernel<T, LAB>), dim3(G, n_images), dim3(kBlock), 0, s, base0, base1, n_first,
                           n_pixels, l.partials, l.pivots);
    CT_CHECK_LAUNCH();
    if (LAB && g_prof_evt[1]) (void)hipEventRecord(g_prof_evt[1], s);
    if (use_lut && deferred_partials != nullptr && n_pixels > 0) {
        *deferred_partials = G;
        return CT_OK;
    }
    hipLaunchKernelGGL((moments_finalize_kernel<LAB>), dim3(n_images), dim3(kBlock), 0, s, l.partials, l.pivots, G,
                       n_pixels, stats, use_lut ? kVarFloorF32 : 0.0);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

template <typename T, bool OUT_LAB>
static int launch_reinhard_apply(const T *target, const double *st, const double *sr, T *out, int64_t n_pixels,
                                 int batch, hipStream_t s, const WsLayout *deferred = nullptr, int n_partials = 0,
                                 double *stats_out = nullptr, const float *gt = nullptr, double *sq_partials = nullptr,
                                 int *sq_blocks = nullptr) {
    if (batch == 0 || n_pixels == 0) return CT_OK;
    constexpr bool kLut = sizeof(T) == 4;
    const bool use_lut = kLut && lab_mode() == 0;
    int G = blocks_per_image(n_pixels >> 2, batch);
    if constexpr (kLut) {
        static const int resident = resident_blocks(reinhard_apply_lut_kernel<OUT_LAB>);
        if (use_lut) G = lut_blocks_per_image(resident, n_pixels / kTilePixels, batch);
    }
    if (g_prof_evt[2]) (void)hipEventRecord(g_prof_evt[2], s);
    if constexpr (kLut) {
        if (use_lut)
            hipLaunchKernelGGL((reinhard_apply_lut_kernel<OUT_LAB>), dim3(G, batch), dim3(kLutBlock), 0, s, target, st, sr, out,
                               n_pixels, (const double *)(deferred ? deferred->partials : nullptr),
                               (const double *)(deferred ? deferred->pivots : nullptr), n_partials, batch, stats_out, gt, sq_partials);
        if (use_lut && sq_blocks) *sq_blocks = G;
    }
    if (!use_lut)
        hipLaunchKernelGGL((reinhard_apply_kernel<T, OUT_LAB>), dim3(G, batch), dim3(kBlock), 0, s, target, st, sr, out,
                           n_pixels);
    CT_CHECK_LAUNCH();
    if (g_prof_evt[3]) (void)hipEventRecord(g_prof_evt[3], s);
    return CT_OK;
}

template <typename T>
static int lab_stats_impl(const T *rgb, int64_t n_pixels, int n_images, double *stats, void *ws, size_t ws_bytes,
                          void *stream) {
    int rc = check_image_args(rgb, n_pixels, n_images);
    if (rc) return rc;
    if (n_images > 0 && stats == nullptr) return CT_E_BADARG;
    if ((rc = check_ws(ws, ws_bytes, n_images))) return rc;
    return launch_moments<T, true>(rgb, rgb, n_images, n_images, n_pixels, ws_carve(ws, n_images), stats,
                                   (hipStream_t)stream);
}

template <typename T>
static int rgb_meancov_impl(const T *rgb, int64_t n_pixels, int n_images, double *stats, void *ws, size_t ws_bytes,
                            void *stream) {
    int rc = check_image_args(rgb, n_pixels, n_images);
    if (rc) return rc;
    if (n_images > 0 && stats == nullptr) return CT_E_BADARG;
    if ((rc = check_ws(ws, ws_bytes, n_images))) return rc;
    return launch_moments<T, false>(rgb, rgb, n_images, n_images, n_pixels, ws_carve(ws, n_images), stats,
                                    (hipStream_t)stream);
}

template <typename T>
static int reinhard_impl(const T *target, const T *reference, T *out, int64_t n_pixels, int batch,
                         double *stats_out, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_image_args(target, n_pixels, batch);
    if (rc) return rc;
    if ((rc = check_image_args(reference, n_pixels, batch))) return rc;
    if ((rc = check_image_args(out, n_pixels, batch))) return rc;
    if ((rc = check_ws(ws, ws_bytes, 2 * batch))) return rc;
    if (batch == 0) return CT_OK;
    if constexpr (sizeof(T) == 4) {
        // frames whose 1 / CUs share fits one CU's LDS: one persistent launch (reinhard_persist.hip) instead of the two sweeps
        if (lab_mode() == 0 && n_pixels > 0 && rp::eligible(n_pixels, false) && ws_bytes >= rp::ws_bytes(n_pixels, batch)) {
            if (g_prof_evt[0]) (void)hipEventRecord(g_prof_evt[0], (hipStream_t)stream);
            if (g_prof_evt[1]) (void)hipEventRecord(g_prof_evt[1], (hipStream_t)stream);
            return rp::launch<float>(target, reference, nullptr, out, nullptr, n_pixels, batch, stats_out, ws, ws_bytes, (hipStream_t)stream,
                                     g_prof_evt[2], g_prof_evt[3]);
        }
    }
    const WsLayout l = ws_carve(ws, 2 * batch);
    // one sweep over all 2*batch images; stats records [0,batch) = targets, [batch,2batch) = references
    double *stats = stats_out ? stats_out : l.stats;
    int deferred = 0;      // > 0: table path, the apply kernel finishes the statistics in its prologue (and writes stats_out)
    rc = launch_moments<T, true>(target, reference, batch, 2 * batch, n_pixels, l, stats, (hipStream_t)stream, &deferred);
    if (rc) return rc;
    return launch_reinhard_apply<T, false>(target, stats, stats + (size_t)batch * CT_LAB_STATS_STRIDE, out, n_pixels,
                                           batch, (hipStream_t)stream, deferred > 0 ? &l : nullptr, deferred, stats);
}

template <typename TI, typename TO>
static int affine_impl(const TI *in, const double *coef, TO *out, int64_t n_pixels, int batch, void *stream) {
    int rc = check_image_args(in, n_pixels, batch);
    if (rc) return rc;
    if ((rc = check_image_args(out, n_pixels, batch))) return rc;
    if (batch > 0 && coef == nullptr) return CT_E_BADARG;
    if (batch == 0 || n_pixels == 0) return CT_OK;
    const int G = blocks_per_image(n_pixels >> 2, batch);
    hipLaunchKernelGGL((affine3x3_kernel<TI, TO>), dim3(G, batch), dim3(kBlock), 0, (hipStream_t)stream, in, coef,
                       out, n_pixels);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

// -------------------------------------------------------------------------------------------
// Per-frame PSNR (the metric Runner.test_step logs, methods/__init__.py:32,37; piq.psnr semantics: inputs
// clamped by the caller, data_range 1, mean squared error over all elements of a frame, 10 log10(1/mse)).
// Deterministic: float64 partial sums per workgroup, fixed-order finish.  grid = (G, batch).
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sqerr_partial_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                               int64_t n, double *__restrict__ partials) {
    __shared__ double lds[4];
    const float *pa = a + (size_t)blockIdx.y * n, *pb = b + (size_t)blockIdx.y * n;
    double s[1] = {0.0};
    const bool vec = ((reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb)) & 15) == 0;
    const int64_t n4 = vec ? (n >> 2) : 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        const float4 x = reinterpret_cast<const float4 *>(pa)[i], y = reinterpret_cast<const float4 *>(pb)[i];
        const double d0 = (double)x.x - y.x, d1 = (double)x.y - y.y, d2 = (double)x.z - y.z, d3 = (double)x.w - y.w;
        s[0] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double d = (double)pa[i] - pb[i];
        s[0] += d * d;
    }
    block_sum<1>(s, lds);
    if (threadIdx.x == 0) partials[(size_t)blockIdx.y * kMaxBlocksPerImage + blockIdx.x] = s[0];
}

__global__ __launch_bounds__(kBlock) void psnr_finish_kernel(const double *__restrict__ partials, int n_blocks, int64_t n,
                                                             double *__restrict__ out) {
    __shared__ double lds[4];
    double s[1] = {0.0};
    for (int i = threadIdx.x; i < n_blocks; i += kBlock) s[0] += partials[(size_t)blockIdx.x * kMaxBlocksPerImage + i];
    block_sum<1>(s, lds);
    if (threadIdx.x == 0) {
        const double mse = s[0] / (double)n;
        out[blockIdx.x * 2] = mse;
        out[blockIdx.x * 2 + 1] = 10.0 * log10(1.0 / (mse > 1e-300 ? mse : 1e-300));
    }
}

// a1 fused with the per-frame PSNR of Runner.test_step (methods/__init__.py:30-32,37): color_transfer_between_images for
// `batch` pairs + PSNR(result, gt) per frame.  Table path: the squared error is accumulated by the apply sweep while it
// writes the result (one extra plane read, the result is never read back); exact path: the two steps one after the other.
// Workspace: the Reinhard layout followed by batch x kMaxBlocksPerImage doubles.
static size_t ws_bytes_reinhard_psnr(int batch) { return ws_bytes_for(2 * batch) + (size_t)batch * kMaxBlocksPerImage * sizeof(double); }

static int reinhard_psnr_impl(const float *target, const float *reference, const float *gt, float *out, double *psnr_out,
                              int64_t n_pixels, int batch, double *stats_out, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_image_args(target, n_pixels, batch);
    if (rc) return rc;
    if ((rc = check_image_args(reference, n_pixels, batch))) return rc;
    if ((rc = check_image_args(gt, n_pixels, batch))) return rc;
    if ((rc = check_image_args(out, n_pixels, batch))) return rc;
    if (batch > 0 && psnr_out == nullptr) return CT_E_BADARG;
    if (ws == nullptr || (reinterpret_cast<uintptr_t>(ws) & 15) || ws_bytes < ws_bytes_reinhard_psnr(batch)) return CT_E_WORKSPACE;
    if (batch == 0 || n_pixels == 0) return CT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (lab_mode() == 0 && rp::eligible(n_pixels, false) && ws_bytes >= rp::ws_bytes(n_pixels, batch)) {
        if (g_prof_evt[0]) (void)hipEventRecord(g_prof_evt[0], s);
        if (g_prof_evt[1]) (void)hipEventRecord(g_prof_evt[1], s);
        return rp::launch<float>(target, reference, gt, out, psnr_out, n_pixels, batch, stats_out, ws, ws_bytes, s, g_prof_evt[2], g_prof_evt[3]);
    }
    const WsLayout l = ws_carve(ws, 2 * batch);
    double *sq = reinterpret_cast<double *>(reinterpret_cast<char *>(ws) + ws_bytes_for(2 * batch));
    double *stats = stats_out ? stats_out : l.stats;
    int deferred = 0, sq_blocks = 0;
    rc = launch_moments<float, true>(target, reference, batch, 2 * batch, n_pixels, l, stats, s, &deferred);
    if (rc) return rc;
    rc = launch_reinhard_apply<float, false>(target, stats, stats + (size_t)batch * CT_LAB_STATS_STRIDE, out, n_pixels, batch, s,
                                             deferred > 0 ? &l : nullptr, deferred, stats, deferred > 0 ? gt : nullptr, sq, &sq_blocks);
    if (rc) return rc;
    if (sq_blocks == 0) {                       // exact path: a sweep of its own over (out, gt)
        sq_blocks = blocks_per_image((n_pixels * 3) >> 2, batch);
        hipLaunchKernelGGL(sqerr_partial_kernel, dim3(sq_blocks, batch), dim3(kBlock), 0, s, (const float *)out, gt, n_pixels * 3, sq);
        CT_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(psnr_finish_kernel, dim3(batch), dim3(kBlock), 0, s, (const double *)sq, sq_blocks, n_pixels * 3, psnr_out);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

// a3 fused: moments of all 2*batch images in one sweep, finishing kernel, 3x3 algebra, affine apply -- no host sync
template <typename T, typename TO>
static int mk_impl(const T *target, const T *reference, TO *out, int64_t n_pixels, int batch, int decomposition, void *ws,
                   size_t ws_bytes, void *stream) {
    int rc = check_image_args(target, n_pixels, batch);
    if (rc) return rc;
    if ((rc = check_image_args(reference, n_pixels, batch))) return rc;
    if ((rc = check_image_args(out, n_pixels, batch))) return rc;
    if (decomposition < 0 || decomposition > 2) return CT_E_BADARG;
    if ((rc = check_ws(ws, ws_bytes, 2 * batch))) return rc;
    if (batch == 0 || n_pixels == 0) return CT_OK;
    const WsLayout l = ws_carve(ws, 2 * batch);
    hipStream_t s = (hipStream_t)stream;
    rc = launch_moments<T, false>(target, reference, batch, 2 * batch, n_pixels, l, l.stats, s);
    if (rc) return rc;
    // coefficient records live behind the stats records (the workspace reserves CT_RGB_STATS_STRIDE doubles per image)
    double *coef = l.partials;   // the partial sums are dead once the finishing kernel has run
    hipLaunchKernelGGL(mk_coef_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, (const double *)l.stats,
                       (const double *)(l.stats + (size_t)batch * CT_RGB_STATS_STRIDE), decomposition, batch, coef);
    CT_CHECK_LAUNCH();
    return affine_impl<T, TO>(target, coef, out, n_pixels, batch, stream);
}

// the persistent launch by name (any frame size it supports; the automatic dispatch above only takes frames that fill every wave)
template <typename T>
static int reinhard_persist_entry(const T *target, const T *reference, const T *gt, float *out, double *psnr_out, int64_t n_pixels, int batch,
                                  double *stats_out, void *ws, size_t ws_bytes, void *stream) {
    int rc = check_image_args(target, n_pixels, batch);
    if (rc) return rc;
    if ((rc = check_image_args(reference, n_pixels, batch))) return rc;
    if ((rc = check_image_args(out, n_pixels, batch))) return rc;
    if (gt != nullptr && psnr_out == nullptr) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    if (!rp::eligible(n_pixels, true)) return CT_E_BADARG;
    return rp::launch<T>(target, reference, gt, out, psnr_out, n_pixels, batch, stats_out, ws, ws_bytes, (hipStream_t)stream,
                             g_prof_evt[2], g_prof_evt[3]);
}

}  // namespace ct

// -------------------------------------------------------------------------------------------
// C ABI (include/ct_hip.h)
// -------------------------------------------------------------------------------------------
extern "C" {

int ct_abi_version(void) { return CT_ABI_VERSION; }

int ct_device_status(int clear) {
    const int a = ct::rp::read_status(clear != 0), b = ct::conv_split_read_status(clear != 0);
    if (a < 0 || b < 0) return -1;
    return a | (b << 1);
}

int ct_set_lab_mode(int mode) {
    if (mode != CT_LAB_TABLE && mode != CT_LAB_EXACT) return CT_E_BADARG;
    ct::g_lab_mode_default.store(mode, std::memory_order_relaxed);
    return CT_OK;
}
int ct_get_lab_mode(void) { return ct::lab_mode(); }
int ct_set_lab_mode_thread(int mode) {
    if (mode != -1 && mode != CT_LAB_TABLE && mode != CT_LAB_EXACT) return CT_E_BADARG;
    ct::t_lab_mode = mode;
    return CT_OK;
}

void ct_profile_events(void *moments_start, void *moments_stop, void *apply_start, void *apply_stop) {
    ct::g_prof_evt[0] = (hipEvent_t)moments_start;
    ct::g_prof_evt[1] = (hipEvent_t)moments_stop;
    ct::g_prof_evt[2] = (hipEvent_t)apply_start;
    ct::g_prof_evt[3] = (hipEvent_t)apply_stop;
}

const char *ct_error_string(int code) {
    switch (code) {
        case CT_OK: return "ok";
        case CT_E_BADARG: return "bad argument (null pointer, negative size or unknown enum)";
        case CT_E_WORKSPACE: return "workspace missing, misaligned or smaller than ct_workspace_bytes()";
        case CT_E_ALIGN: return "image pointer not aligned to its element size";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown ct error";
    }
}

size_t ct_workspace_bytes(int kind, int64_t n_pixels, int n_images) {
    if (n_images < 0) return 0;
    switch (kind) {
        case CT_WS_LAB_STATS:
        case CT_WS_RGB_MEANCOV: return ct::ws_bytes_for(n_images);
        case CT_WS_REINHARD: { const size_t a = ct::ws_bytes_for(2 * n_images), b = ct::rp::ws_bytes(n_pixels, n_images); return a > b ? a : b; }
        case CT_WS_REINHARD_PSNR: { const size_t a = ct::ws_bytes_reinhard_psnr(n_images), b = ct::rp::ws_bytes(n_pixels, n_images); return a > b ? a : b; }
        case CT_WS_REINHARD_PERSIST: return ct::rp::ws_bytes(n_pixels, n_images);
        default: return 0;
    }
}

int ct_lab_stats_f32(const float *rgb, int64_t n_pixels, int n_images, double *stats, void *ws, size_t ws_bytes,
                     void *stream) {
    return ct::lab_stats_impl<float>(rgb, n_pixels, n_images, stats, ws, ws_bytes, stream);
}
int ct_lab_stats_f64(const double *rgb, int64_t n_pixels, int n_images, double *stats, void *ws, size_t ws_bytes,
                     void *stream) {
    return ct::lab_stats_impl<double>(rgb, n_pixels, n_images, stats, ws, ws_bytes, stream);
}

int ct_reinhard_apply_f32(const float *target, const double *stats_t, const double *stats_r, float *out,
                          int64_t n_pixels, int batch, void *stream) {
    int rc = ct::check_image_args(target, n_pixels, batch);
    if (rc) return rc;
    if ((rc = ct::check_image_args(out, n_pixels, batch))) return rc;
    if (batch > 0 && (!stats_t || !stats_r)) return CT_E_BADARG;
    return ct::launch_reinhard_apply<float, false>(target, stats_t, stats_r, out, n_pixels, batch, (hipStream_t)stream);
}
int ct_reinhard_apply_f64(const double *target, const double *stats_t, const double *stats_r, double *out,
                          int64_t n_pixels, int batch, void *stream) {
    int rc = ct::check_image_args(target, n_pixels, batch);
    if (rc) return rc;
    if ((rc = ct::check_image_args(out, n_pixels, batch))) return rc;
    if (batch > 0 && (!stats_t || !stats_r)) return CT_E_BADARG;
    return ct::launch_reinhard_apply<double, false>(target, stats_t, stats_r, out, n_pixels, batch,
                                                    (hipStream_t)stream);
}
int ct_reinhard_lab_f32(const float *target, const double *stats_t, const double *stats_r, float *out_lab,
                        int64_t n_pixels, int batch, void *stream) {
    int rc = ct::check_image_args(target, n_pixels, batch);
    if (rc) return rc;
    if ((rc = ct::check_image_args(out_lab, n_pixels, batch))) return rc;
    if (batch > 0 && (!stats_t || !stats_r)) return CT_E_BADARG;
    return ct::launch_reinhard_apply<float, true>(target, stats_t, stats_r, out_lab, n_pixels, batch,
                                                  (hipStream_t)stream);
}

int ct_reinhard_f32(const float *target, const float *reference, float *out, int64_t n_pixels, int batch,
                    double *stats_out, void *ws, size_t ws_bytes, void *stream) {
    return ct::reinhard_impl<float>(target, reference, out, n_pixels, batch, stats_out, ws, ws_bytes, stream);
}
int ct_reinhard_f64(const double *target, const double *reference, double *out, int64_t n_pixels, int batch,
                    double *stats_out, void *ws, size_t ws_bytes, void *stream) {
    return ct::reinhard_impl<double>(target, reference, out, n_pixels, batch, stats_out, ws, ws_bytes, stream);
}
int ct_reinhard_psnr_f32(const float *target, const float *reference, const float *gt, float *out, double *psnr_out, int64_t n_pixels,
                         int batch, double *stats_out, void *ws, size_t ws_bytes, void *stream) {
    return ct::reinhard_psnr_impl(target, reference, gt, out, psnr_out, n_pixels, batch, stats_out, ws, ws_bytes, stream);
}

int ct_reinhard_persist_supported(int64_t n_pixels) { return ct::rp::eligible(n_pixels, true) ? 1 : 0; }
int ct_reinhard_takes_persist(int64_t n_pixels) { return (ct::lab_mode() == 0 && ct::rp::eligible(n_pixels, false)) ? 1 : 0; }
int ct_reinhard_persist_f32(const float *target, const float *reference, const float *gt, float *out, double *psnr_out, int64_t n_pixels,
                            int batch, double *stats_out, void *ws, size_t ws_bytes, void *stream) {
    return ct::reinhard_persist_entry<float>(target, reference, gt, out, psnr_out, n_pixels, batch, stats_out, ws, ws_bytes, stream);
}
int ct_reinhard_psnr_u8(const uint8_t *target, const uint8_t *reference, const uint8_t *gt, float *out, double *psnr_out, int64_t n_pixels,
                        int batch, double *stats_out, void *ws, size_t ws_bytes, void *stream) {
    return ct::reinhard_persist_entry<uint8_t>(target, reference, gt, out, psnr_out, n_pixels, batch, stats_out, ws, ws_bytes, stream);
}

int ct_rgb_meancov_f32(const float *rgb, int64_t n_pixels, int n_images, double *stats, void *ws, size_t ws_bytes,
                       void *stream) {
    return ct::rgb_meancov_impl<float>(rgb, n_pixels, n_images, stats, ws, ws_bytes, stream);
}
int ct_rgb_meancov_f64(const double *rgb, int64_t n_pixels, int n_images, double *stats, void *ws, size_t ws_bytes,
                       void *stream) {
    return ct::rgb_meancov_impl<double>(rgb, n_pixels, n_images, stats, ws, ws_bytes, stream);
}

int ct_mk_f32_f32(const float *target, const float *reference, float *out, int64_t n_pixels, int batch, int decomposition, void *ws,
                  size_t ws_bytes, void *stream) {
    return ct::mk_impl<float, float>(target, reference, out, n_pixels, batch, decomposition, ws, ws_bytes, stream);
}
int ct_mk_f32_f64(const float *target, const float *reference, double *out, int64_t n_pixels, int batch, int decomposition, void *ws,
                  size_t ws_bytes, void *stream) {
    return ct::mk_impl<float, double>(target, reference, out, n_pixels, batch, decomposition, ws, ws_bytes, stream);
}
int ct_mk_f64_f64(const double *target, const double *reference, double *out, int64_t n_pixels, int batch, int decomposition, void *ws,
                  size_t ws_bytes, void *stream) {
    return ct::mk_impl<double, double>(target, reference, out, n_pixels, batch, decomposition, ws, ws_bytes, stream);
}

int ct_frame_psnr_f32(const float *a, const float *b, int64_t n_elems, int batch, double *out, void *ws, size_t ws_bytes, void *stream) {
    if (!a || !b || !out || n_elems < 1 || batch < 0) return CT_E_BADARG;
    if (!ws || ws_bytes < (size_t)batch * ct::kMaxBlocksPerImage * sizeof(double)) return CT_E_WORKSPACE;
    if (batch == 0) return CT_OK;
    const int G = ct::blocks_per_image(n_elems >> 2, batch);
    hipLaunchKernelGGL(ct::sqerr_partial_kernel, dim3(G, batch), dim3(ct::kBlock), 0, (hipStream_t)stream, a, b, n_elems, (double *)ws);
    CT_CHECK_LAUNCH();
    hipLaunchKernelGGL(ct::psnr_finish_kernel, dim3(batch), dim3(ct::kBlock), 0, (hipStream_t)stream, (const double *)ws, G, n_elems, out);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_mk_coef_f64(const double *stats_t, const double *stats_r, int decomposition, int batch, double *coef, void *stream) {
    if (!stats_t || !stats_r || !coef || batch < 0 || decomposition < 0 || decomposition > 2) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    hipLaunchKernelGGL(ct::mk_coef_kernel, dim3((batch + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats_t, stats_r, decomposition,
                       batch, coef);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

int ct_affine3x3_f32_f64(const float *in, const double *coef, double *out, int64_t n_pixels, int batch, void *stream) {
    return ct::affine_impl<float, double>(in, coef, out, n_pixels, batch, stream);
}
int ct_affine3x3_f64_f64(const double *in, const double *coef, double *out, int64_t n_pixels, int batch,
                         void *stream) {
    return ct::affine_impl<double, double>(in, coef, out, n_pixels, batch, stream);
}
int ct_affine3x3_f32_f32(const float *in, const double *coef, float *out, int64_t n_pixels, int batch, void *stream) {
    return ct::affine_impl<float, float>(in, coef, out, n_pixels, batch, stream);
}

}  // extern "C"
