// ct_reinhard.h -- device helpers shared by the Reinhard kernels of linear.hip (two sweeps) and reinhard_persist.hip
// (one persistent launch): Lab moment accumulation, the statistics record, the affine map in the cube-root domain and
// the 256-pixel tile I/O of the table path.  Replaces the numpy expressions of methods/linear.py:25-40.
#pragma once
#include "ct_color.h"
#include "ct_color_lut.h"
#include "ct_common.h"

namespace ct {

template <bool LAB>
__device__ __forceinline__ void to_space(double r, double g, double b, double &x, double &y, double &z) {
    if (LAB) {
        // moments are taken of (fy, fx - fy, fy - fz); L = 116 fy - 16, a = 500 (fx - fy),
        // b = 200 (fy - fz) are per-axis affine images of those, applied once in the finishing kernel
        double fx, fy, fz;
        rgb_to_f(r, g, b, fx, fy, fz);
        x = fy; y = fx - fy; z = fy - fz;
    } else {
        x = r; y = g; z = b;
    }
}

template <bool LAB>
__device__ __forceinline__ void accumulate(double (&s)[LAB ? 6 : 9], const double (&k)[3], double x, double y,
                                           double z) {
    const double dx = x - k[0], dy = y - k[1], dz = z - k[2];
    s[0] += dx; s[1] += dy; s[2] += dz;
    if (LAB) {
        s[3] = fma(dx, dx, s[3]); s[4] = fma(dy, dy, s[4]); s[5] = fma(dz, dz, s[5]);
    } else {
        s[3] = fma(dx, dx, s[3]); s[4] = fma(dx, dy, s[4]); s[5] = fma(dx, dz, s[5]);
        s[6] = fma(dy, dy, s[6]); s[7] = fma(dy, dz, s[7]); s[8] = fma(dz, dz, s[8]);
    }
}

// shifted sums of (fy, fx-fy, fy-fz) around pivot k over n pixels -> the Lab stats record {mean L,a,b ; std L,a,b ; n ; 0}
// floor: variances at or below it are zero.  The float32 sweeps pass kVarFloorF32: their shifted sums of a CONSTANT image leave a
// cancellation residue of either sign (<= (2^-11)^2 x 1e-7, the pivot sits on a 2^-10 grid), and the reference's own sigma of a
// constant image is 0 or 1e-17 by the luck of its pairwise sum (inf / nan or finite garbage): here it is exactly 0 -> inf / nan,
// deterministically.  (A frame with a true sigma below 2e-5 L* is constant to one part in a million.)
constexpr double kVarFloorF32 = 3e-14;
__device__ __forceinline__ double var_to_sd(double v, double floor) { return sqrt(v > floor ? v : (v == v ? 0.0 : v)); }
__device__ __forceinline__ void lab_record(const double *s, const double *k, double n, double *o, double floor = 0.0) {
    const double m0 = s[0] / n, m1 = s[1] / n, m2 = s[2] / n;  // mean of (x - K)
    // (fy, fx-fy, fy-fz) -> (L, a, b): scale 116/500/200, offset -16/0/0
    o[0] = fma(116.0, k[0] + m0, -16.0); o[1] = 500.0 * (k[1] + m1); o[2] = 200.0 * (k[2] + m2);
    // population variance (np.std, ddof 0); clamp the cancellation residue of constant images
    const double v0 = fma(-s[0], m0, s[3]) / n, v1 = fma(-s[1], m1, s[4]) / n, v2 = fma(-s[2], m2, s[5]) / n;
    o[3] = 116.0 * var_to_sd(v0, floor);
    o[4] = 500.0 * var_to_sd(v1, floor);
    o[5] = 200.0 * var_to_sd(v2, floor);
    o[6] = n; o[7] = 0.0;
}

// A2: Reinhard apply.  Lab is affine in (fx,fy,fz), so "to Lab, scale/shift, back from Lab"
// collapses into one affine map in the cube-root domain:
//   fy' = sL fy + cy ;  fx' = fy' + sa (fx - fy) + ca ;  fz' = fy' - sb (fy - fz) - cb
// -------------------------------------------------------------------------------------------
struct ReinhardCoef {
    double sL, sa, sb, cy, ca, cb;
};

__device__ __forceinline__ ReinhardCoef reinhard_coef(const double *st, const double *sr) {
    ReinhardCoef c;
    c.sL = sr[3] / st[3];  // sigma_r / sigma_t   (inf/nan on a constant target, like the reference)
    c.sa = sr[4] / st[4];
    c.sb = sr[5] / st[5];
    c.cy = (fma(-16.0 - st[0], c.sL, sr[0]) + 16.0) * (1.0 / 116.0);
    c.ca = fma(-st[1], c.sa, sr[1]) * (1.0 / 500.0);
    c.cb = fma(-st[2], c.sb, sr[2]) * (1.0 / 200.0);
    return c;
}

// The float32 table path (ct_color_lut.h) keeps its error budget (Lab <= 5e-5) for moderate maps only: its forward error is
// multiplied by the scales.  Everything else -- and inf / nan coefficients (a constant target, like the reference) -- takes the
// exact float64 code.
constexpr double kFastScale = 4.0, kFastOffset = 8.0;
__device__ __forceinline__ bool reinhard_coef_fast(const ReinhardCoef &c) {
    const double smax = fmax(fmax(fabs(c.sL), fabs(c.sa)), fabs(c.sb)), omax = fmax(fmax(fabs(c.cy), fabs(c.ca)), fabs(c.cb));
    return (smax <= kFastScale) & (omax <= kFastOffset);          // false for NaN
}

template <typename T, bool OUT_LAB>
__device__ __forceinline__ void reinhard_pixel(const ReinhardCoef &c, double r, double g, double b, T &o0, T &o1,
                                               T &o2) {
    double fx, fy, fz;
    rgb_to_f(r, g, b, fx, fy, fz);
    const double gy = fma(c.sL, fy, c.cy);
    const double gx = gy + fma(c.sa, fx - fy, c.ca);
    const double gz = gy - fma(c.sb, fy - fz, c.cb);
    if (OUT_LAB) {
        double L, A, B;
        f_to_lab(gx, gy, gz, L, A, B);
        o0 = (T)L; o1 = (T)A; o2 = (T)B;
    } else {
        double R, G, Bc;
        f_to_rgb(gx, gy, gz, R, G, Bc);
        o0 = clip01<T>(R); o1 = clip01<T>(G); o2 = clip01<T>(Bc);
    }
}

// ---- table path: a wave owns tiles of 256 consecutive pixels, lane l holds pixels l, l+64, l+128, l+192 (one 12-byte access each)
constexpr int kLutBlock = 512;
constexpr int kLutWaves = kLutBlock / kWave;
constexpr int kTilePixels = 4 * kWave;
typedef float float3v __attribute__((ext_vector_type(3)));
typedef float3v float3u __attribute__((aligned(4)));

// Non-temporal accesses (round 5): every frame of these sweeps is touched exactly once per launch, so its lines need not displace
// anything in L2 / MALL.  Measured on the two Reinhard sweeps, same box (tools/bench_sweeps.py): statistics 161.5 -> 157.8 us,
// apply + PSNR 238.6 -> 221.7 us, apply alone 159.0 -> 141.2 us (5.64 TB/s on its two planes).  -DCT_TEMPORAL restores plain accesses.
#ifndef CT_TEMPORAL
#define CT_NT_LOAD 1
#define CT_NT_STORE 1
#endif
__device__ __forceinline__ void load_tile(const float *tile, int lane, float (&e)[12]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef CT_NT_LOAD
        const float3v a = __builtin_nontemporal_load(reinterpret_cast<const float3u *>(tile + (j * kWave + lane) * 3));
#else
        const float3v a = *reinterpret_cast<const float3u *>(tile + (j * kWave + lane) * 3);
#endif
        e[3 * j] = a.x; e[3 * j + 1] = a.y; e[3 * j + 2] = a.z;
    }
}
__device__ __forceinline__ void store_tile(float *tile, int lane, const float (&e)[12]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef CT_NT_STORE
        __builtin_nontemporal_store(float3v{e[3 * j], e[3 * j + 1], e[3 * j + 2]}, reinterpret_cast<float3u *>(tile + (j * kWave + lane) * 3));
#else
        *reinterpret_cast<float3u *>(tile + (j * kWave + lane) * 3) = float3v{e[3 * j], e[3 * j + 1], e[3 * j + 2]};
#endif
    }
}

__device__ __forceinline__ uint32_t max3u(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }      // v_max3_u32
__device__ __forceinline__ uint32_t max_bits12(const float (&e)[12]) {
    const uint32_t m0 = max3u(__float_as_uint(e[0]), __float_as_uint(e[1]), __float_as_uint(e[2]));
    const uint32_t m1 = max3u(__float_as_uint(e[3]), __float_as_uint(e[4]), __float_as_uint(e[5]));
    const uint32_t m2 = max3u(__float_as_uint(e[6]), __float_as_uint(e[7]), __float_as_uint(e[8]));
    const uint32_t m3 = max3u(__float_as_uint(e[9]), __float_as_uint(e[10]), __float_as_uint(e[11]));
    return max(max3u(m0, m1, m2), m3);
}

// rotate the four pixels of a lane by one: the exact fallback stays a rolled loop over "pixel 0" (one copy of the code,
// few registers) without indexing the register array dynamically; four rotations restore the order
__device__ __forceinline__ void rotate_pixels(float (&e)[12]) {
    const float a = e[0], b = e[1], c = e[2];
#pragma unroll
    for (int i = 0; i < 9; ++i) e[i] = e[i + 3];
    e[9] = a; e[10] = b; e[11] = c;
}
__device__ __forceinline__ double uniform_f64(double v) {     // wave-uniform double -> SGPR pair
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

template <int NV, int NW>
__device__ __forceinline__ void block_sum_n(double (&v)[NV], double *lds /* [NW][NV] */) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += __shfl_down(v[i], off, kWave);
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) lds[wid * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double a = lds[i];
            for (int w = 1; w < NW; ++w) a += lds[w * NV + i];   // wave order: fixed
            v[i] = a;
        }
    }
}

}  // namespace ct
