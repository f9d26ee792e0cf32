// fft2d.hip -- batched in-place 2-D complex-to-complex DFT for the phase-congruency stage of FSIM (csrc/fsim.hip; the reference
// gets it from piq.fsim -> torch.fft.fft2 / ifft2, methods/__init__.py:34).  Hand-written for gfx950: it replaces the hipFFT
// calls this library made until round 3 (the one vendor library on the path), and with them the plan cache, its two mutexes
// and the per-plan work area.
//
// A plane [hp][wp] is transformed by two launches: every row (length wp), then every column (length hp).  A workgroup takes a
// few lines into LDS at once (rows: whole contiguous lines; columns: 16 neighbouring columns, read and written as 128-byte
// segments of the rows they cross) and runs a Stockham autosort transform on them -- mixed radix, the factors of the length in
// any order: 4, 2, 3, 5 as register butterflies, any other prime factor p as an O(p^2) butterfly straight out of LDS (so a
// prime length degenerates to a plain DFT: correct for every size, fast for the sizes frames have: 270 x 480 at 1080p, 256 x 256,
// 135 x 240, ...).  The n twiddles exp(+-2 pi i k / n) are computed once per workgroup with a float64 sincospi and kept in LDS;
// a radix-R stage multiplies input t of butterfly j by w^(k t n / (Ns R)), k = j mod Ns (no reduction needed: k t < Ns R), and the
// R-point DFT takes its own roots from the same table (index ((q t) mod R) n / R), so 4-point butterflies get exact 0 / +-1.
// No normalisation in either direction (like hipFFT / FFTW; fsim.hip scales where piq does).
#include "ct_common.h"

namespace ct {

constexpr int kFftMaxStages = 24;
constexpr int kFftThreads = 256;
constexpr int kFftLdsBudget = 96 * 1024;          // two line buffers + the twiddle table

struct FftArgs {
    float2 *data;
    int n;                 // line length
    int lines;             // lines per plane (rows pass: hp; columns pass: wp)
    int planes;
    int pitch;             // elements between consecutive rows of a plane (= wp)
    int plane_stride;      // elements per plane
    int lpw;               // lines per workgroup
    int n_stages;
    int sign;              // -1 forward, +1 backward
    int radix[kFftMaxStages];
};

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }

// one radix-R stage over `lpw` lines of length n in LDS: a -> b
template <int R>
__device__ __forceinline__ void fft_stage(const float2 *a, float2 *b, const float2 *tw, int n, int lpw, int Ns) {
    const int m = n / R, total = lpw * m, tstep = n / (Ns * R), rstep = n / R;
    for (int idx = threadIdx.x; idx < total; idx += kFftThreads) {
        const int line = idx / m, j = idx - line * m, k = j % Ns;
        const float2 *in = a + line * n;
        float2 v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            v[t] = in[j + t * m];
            if (t > 0) v[t] = cmul(v[t], tw[k * t * tstep]);
        }
        float2 *out = b + line * n + (j - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            float2 acc = v[0];
#pragma unroll
            for (int t = 1; t < R; ++t) {
                const float2 y = (q == 0) ? v[t] : cmul(v[t], tw[((q * t) % R) * rstep]);
                acc.x += y.x; acc.y += y.y;
            }
            out[q * Ns] = acc;
        }
    }
}

// any other (prime) factor: inputs re-read from LDS for every output
__device__ __forceinline__ void fft_stage_generic(const float2 *a, float2 *b, const float2 *tw, int n, int lpw, int Ns, int R) {
    const int m = n / R, total = lpw * m, tstep = n / (Ns * R), rstep = n / R;
    for (int idx = threadIdx.x; idx < total; idx += kFftThreads) {
        const int line = idx / m, j = idx - line * m, k = j % Ns;
        const float2 *in = a + line * n;
        float2 *out = b + line * n + (j - k) * R + k;
        for (int q = 0; q < R; ++q) {
            float2 acc = in[j];
            int qt = 0;
            for (int t = 1; t < R; ++t) {
                qt += q; if (qt >= R) qt -= R;                           // (q t) mod R
                const float2 x = cmul(in[j + t * m], tw[k * t * tstep]);
                const float2 y = cmul(x, tw[qt * rstep]);
                acc.x += y.x; acc.y += y.y;
            }
            out[q * Ns] = acc;
        }
    }
}

// COLS = false: lines are rows (element e of line l of plane p at p * plane_stride + l * pitch + e);
// COLS = true: lines are columns (at p * plane_stride + e * pitch + l)
template <bool COLS>
__global__ __launch_bounds__(kFftThreads) void fft_lines_kernel(const FftArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 fsm[];
    float2 *buf0 = fsm, *buf1 = fsm + (size_t)a.lpw * a.n, *tw = fsm + (size_t)2 * a.lpw * a.n;
    const int groups = (a.lines + a.lpw - 1) / a.lpw;
    const int plane = blockIdx.x / groups, l0 = (blockIdx.x - plane * groups) * a.lpw;
    const int nl = min(a.lpw, a.lines - l0);
    float2 *base = a.data + (size_t)plane * a.plane_stride;
    for (int k = threadIdx.x; k < a.n; k += kFftThreads) {
        double sn, cs;
        sincospi((double)a.sign * 2.0 * (double)k / (double)a.n, &sn, &cs);
        tw[k] = make_float2((float)cs, (float)sn);
    }
    if (!COLS) {
        for (int i = threadIdx.x; i < nl * a.n; i += kFftThreads) {
            const int l = i / a.n, e = i - l * a.n;
            buf0[l * a.n + e] = base[(size_t)(l0 + l) * a.pitch + e];
        }
    } else {
        for (int i = threadIdx.x; i < nl * a.n; i += kFftThreads) {          // consecutive threads: consecutive columns of one row
            const int e = i / nl, l = i - e * nl;
            buf0[l * a.n + e] = base[(size_t)e * a.pitch + l0 + l];
        }
    }
    for (int i = threadIdx.x + nl * a.n; i < a.lpw * a.n; i += kFftThreads) buf0[i] = make_float2(0.f, 0.f);     // idle lines of the last group
    __syncthreads();
    float2 *src = buf0, *dst = buf1;
    int Ns = 1;
    for (int st = 0; st < a.n_stages; ++st) {
        const int R = a.radix[st];
        switch (R) {
            case 2: fft_stage<2>(src, dst, tw, a.n, a.lpw, Ns); break;
            case 3: fft_stage<3>(src, dst, tw, a.n, a.lpw, Ns); break;
            case 4: fft_stage<4>(src, dst, tw, a.n, a.lpw, Ns); break;
            case 5: fft_stage<5>(src, dst, tw, a.n, a.lpw, Ns); break;
            default: fft_stage_generic(src, dst, tw, a.n, a.lpw, Ns, R); break;
        }
        __syncthreads();
        float2 *t = src; src = dst; dst = t;
        Ns *= R;
    }
    if (!COLS) {
        for (int i = threadIdx.x; i < nl * a.n; i += kFftThreads) {
            const int l = i / a.n, e = i - l * a.n;
            base[(size_t)(l0 + l) * a.pitch + e] = src[l * a.n + e];
        }
    } else {
        for (int i = threadIdx.x; i < nl * a.n; i += kFftThreads) {
            const int e = i / nl, l = i - e * nl;
            base[(size_t)e * a.pitch + l0 + l] = src[l * a.n + e];
        }
    }
}

// factors of n as radices: 4s first, then 2, 3, 5, then every other prime factor; false when there are too many stages
static bool fft_factor(int n, FftArgs &a) {
    int k = 0;
    auto push = [&](int r) { if (k < kFftMaxStages) a.radix[k] = r; ++k; };
    while (n % 4 == 0) { push(4); n /= 4; }
    while (n % 2 == 0) { push(2); n /= 2; }
    while (n % 3 == 0) { push(3); n /= 3; }
    while (n % 5 == 0) { push(5); n /= 5; }
    for (int p = 7; (long long)p * p <= n; p += 2)
        while (n % p == 0) { push(p); n /= p; }
    if (n > 1) push(n);
    a.n_stages = k;
    return k <= kFftMaxStages;
}

// true when fft2d_c2c can transform planes of this size (each axis: two line buffers + the twiddle table within the LDS budget)
bool fft2d_supported(int hp, int wp) {
    auto ok = [](int n) { return n >= 1 && (size_t)(2 + 1) * n * sizeof(float2) <= (size_t)kFftLdsBudget; };
    return ok(hp) && ok(wp);
}

template <bool COLS>
static int fft_pass(float2 *data, int hp, int wp, int planes, int sign, hipStream_t s) {
    FftArgs a;
    a.data = data; a.planes = planes; a.pitch = wp; a.plane_stride = hp * wp; a.sign = sign;
    a.n = COLS ? hp : wp;
    a.lines = COLS ? wp : hp;
    if (a.n == 1) return CT_OK;                                   // a 1-point transform is the identity
    if (!fft_factor(a.n, a)) return CT_E_BADARG;
    const int fit = (int)(((size_t)kFftLdsBudget / sizeof(float2) - a.n) / (2 * (size_t)a.n));      // lines whose two buffers fit beside the twiddles
    if (fit < 1) return CT_E_BADARG;
    int lpw = COLS ? 16 : (4096 / a.n < 1 ? 1 : 4096 / a.n);       // columns: 128-byte segments; rows: ~4 k points per workgroup
    if (lpw > fit) lpw = fit;
    if (lpw > a.lines) lpw = a.lines;
    a.lpw = lpw;
    const size_t lds = ((size_t)2 * lpw * a.n + a.n) * sizeof(float2);
    const long long groups = (a.lines + lpw - 1) / lpw;
    const long long grid = groups * planes;
    if (grid > 0x7fffffffLL) return CT_E_BADARG;
    auto kern = fft_lines_kernel<COLS>;
    static DynLdsAttr attr;             // per instantiation (COLS), per device
    {
        hipError_t e = attr.ensure(reinterpret_cast<const void *>(kern), kFftLdsBudget);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kFftThreads), lds, s, a);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

// in place; sign = -1: forward (exp(-2 pi i ..)), +1: backward, unnormalised
int fft2d_c2c(float2 *data, int hp, int wp, int planes, int sign, hipStream_t s) {
    if (!data || hp < 1 || wp < 1 || planes < 0 || (sign != 1 && sign != -1) || !fft2d_supported(hp, wp)) return CT_E_BADARG;
    if (planes == 0) return CT_OK;
    if ((long long)hp * wp > 0x7fffffffLL) return CT_E_BADARG;
    int rc = fft_pass<false>(data, hp, wp, planes, sign, s);
    if (rc) return rc;
    return fft_pass<true>(data, hp, wp, planes, sign, s);
}

}  // namespace ct

extern "C" {

// Batched in-place 2-D DFT of `planes` complex float32 planes [hp][wp] (interleaved re, im): what torch.fft.fft2 (inverse = 0)
// and torch.fft.ifft2 * hp * wp (inverse = 1: unnormalised) compute inside piq.fsim.  Any size whose longer axis is at most
// 4096 points; asynchronous on `stream`, no workspace, no plan, no state.
int ct_fft2d_c2c_f32(void *data, int hp, int wp, int planes, int inverse, void *stream) {
    return ct::fft2d_c2c(reinterpret_cast<float2 *>(data), hp, wp, planes, inverse ? 1 : -1, (hipStream_t)stream);
}

}  // extern "C"
