// idt.hip -- Pitie iterative distribution transfer on gfx950 (MI355X).
//
// Replaces the numpy sweeps of the reference's methods/iterative.py:31-55 (per iteration):
//   A6  d = r @ x.T, lo/hi over both images   (iterative.py:34-35,39-40) -> idt_minmax_kernel / fused in apply
//   A7  np.histogram x2 per axis               (iterative.py:42-43)       -> idt_hist_kernel (LDS-binned atomics)
//   A8  cumsum, normalise, np.interp (LUT f)   (iterative.py:45-51)       -> idt_lut_kernel
//   A9  np.interp(d0r, edges[1:], f, left=0), solve, add (iterative.py:53-55) -> idt_apply_kernel
//
// Exactness contract (mirrors oracle/idt_oracle.c operation for operation; the library is
// built with -ffp-contract=off, every fused op below is an explicit fma()):
//   * projection  fma(r2,x2, fma(r1,x1, r0*x0)) in float64;
//   * lo/hi are exact (order-independent integer atomics on an order-preserving key);
//   * bin of x = the unique k with edges[k] <= x < edges[k+1] (last bin closed), edges[i] =
//     i*step + lo, edges[bins] = hi -- numpy's estimate-then-correct rule lands on exactly this
//     k, so the estimate may use a reciprocal multiply instead of numpy's division;
//   * counts are integers (LDS + global integer atomics) -> order independent;
//   * LUT and interpolation use IEEE float64 division and separate multiply/add like numpy.
// Hence bin indices, counts, LUTs AND the float64 output are bitwise identical to the oracle.
#include "ct_common.h"

namespace ct {

constexpr int kIdtMaxBins = 1024;   // 3*bins*16 B of LUT in LDS (48 KiB) per workgroup
constexpr int kIdtBlock = 256;
constexpr int kIdtHistBlock = 1024;   // few workgroups (few contended atomic flushes) but many waves per CU

// ---- order-preserving key for float64 (atomicMax on u64; zero-initialised memory = "-inf") ----
__device__ __forceinline__ unsigned long long f64_key(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ double proj(const double *r, double x0, double x1, double x2) {
    return fma(r[2], x2, fma(r[1], x1, r[0] * x0));
}

// per (pair, iteration) workspace record, all zero-initialised by one zeroing launch per call (ct_common.h: zero_async)
// Same-address atomics from many workgroups serialise in the memory system (~20 ns each): the lo/hi keys and the histogram
// counts are therefore kept in SHARDS (workgroup g updates shard g % S; readers take the max / the sum over the shards).
// Integer max and integer add are order independent, so every bit of the result is unchanged; one pair's sweeps lose their
// 8-15 us atomic tails (round 3: 2.9 k -> 4+ k pairs/s for one pair per call).
constexpr int kMmShards = 16;
constexpr int kHistShards = 8;

struct IdtLayout {
    unsigned long long *mm;  // [batch][n_iter][kMmShards][3][4] keys: max(-t) , max(t), max(-r), max(r)  (min via negation)
    unsigned int *hist;      // [batch][n_iter][kHistShards][2][3][bins]
    unsigned int *hsum;      // [batch][n_iter][2][3][bins]: the shards added up by idt_lut_kernel (debug dump)
    double *lut;             // [batch][n_iter][3][bins][2]  (f[j], slope[j])
    double *par;             // [batch][n_iter][3][4]        (lo, hi, step, scale)
    size_t zero_bytes;       // leading bytes (mm + hist) that must be zero at call start
    size_t total_bytes;
};

static IdtLayout idt_layout(void *ws, int batch, int n_iter, int bins) {
    IdtLayout l;
    char *p = reinterpret_cast<char *>(ws);
    size_t off = 0;
    l.mm = reinterpret_cast<unsigned long long *>(p + off);
    off += (size_t)batch * n_iter * kMmShards * 3 * 4 * sizeof(unsigned long long);
    l.hist = reinterpret_cast<unsigned int *>(p + off);
    off += (size_t)batch * n_iter * kHistShards * 2 * 3 * bins * sizeof(unsigned int);
    off = (off + 15) & ~(size_t)15;
    l.zero_bytes = off;
    l.hsum = reinterpret_cast<unsigned int *>(p + off);
    off += (size_t)batch * n_iter * 2 * 3 * bins * sizeof(unsigned int);
    off = (off + 15) & ~(size_t)15;
    l.lut = reinterpret_cast<double *>(p + off);
    off += (size_t)batch * n_iter * 3 * bins * 2 * sizeof(double);
    l.par = reinterpret_cast<double *>(p + off);
    off += (size_t)batch * n_iter * 3 * 4 * sizeof(double);
    l.total_bytes = off;
    return l;
}

// -------------------------------------------------------------------------------------------
// A6 (initial): min/max of r_it @ x for it in [it0, it0 + n_rot) ; `which` = 0 target, 1 reference
// grid = (G, batch)
// -------------------------------------------------------------------------------------------
// One sweep serves up to four rotations: 4 pixels per lane come in as three 16-byte loads, the 3x3 rotations sit in
// scalar registers, and min/max are float64 v_min/v_max (a 64-bit integer max is a compare + two selects); the
// order-preserving keys that the integer atomicMax needs are built once per lane at the end.  A non-finite projection
// poisons the range (NaN), as np.histogram's range check does in the reference.
// grid = (G, batch, 2): blockIdx.z = 0 sweeps the target (iteration 0's rotation only: later iterations get their lo / hi from the
// apply sweep before them), blockIdx.z = 1 the reference (the rotations of all iterations) -- one launch instead of two.
template <typename T, int MAXROT>
__global__ __launch_bounds__(kIdtBlock) void idt_minmax_kernel(const T *__restrict__ tgt_img, int64_t n_tgt, const T *__restrict__ ref_img,
                                                               int64_t n_ref, const double *__restrict__ rot, int n_iter,
                                                               unsigned long long *__restrict__ mm) {
    constexpr int RG = 4;               // rotations per sweep
    __shared__ unsigned long long lds[4 * 6 * RG];
    const int b = blockIdx.y;
    const int which = blockIdx.z;
    const T *img = which ? ref_img : tgt_img;
    const int64_t n = which ? n_ref : n_tgt;
    const int it0 = 0, n_rot = which ? n_iter : 1;
    const T *p = img + (size_t)b * n * 3;
    const bool vec = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    const int64_t n_chunks = n >> 2;
    for (int q0 = 0; q0 < n_rot; q0 += RG) {
        const int nq = (n_rot - q0) < RG ? (n_rot - q0) : RG;
        double r[RG][9];
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
            for (int i = 0; i < 9; ++i) r[q][i] = rot[((size_t)b * n_iter + it0 + q0 + (q < nq ? q : 0)) * 9 + i];
        double mn[RG][3], mx[RG][3], bad = 0.0;
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) { mn[q][j] = INFINITY; mx[q][j] = -INFINITY; }
        auto pixel = [&](double x0, double x1, double x2) {
#pragma unroll
            for (int q = 0; q < RG; ++q)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double d = proj(r[q] + 3 * j, x0, x1, x2);
                    mn[q][j] = fmin(mn[q][j], d);
                    mx[q][j] = fmax(mx[q][j], d);
                    bad = fma(d, 0.0, bad);          // stays 0 unless d is NaN or infinite
                }
        };
        for (int64_t c = (int64_t)blockIdx.x * kIdtBlock + threadIdx.x; c < n_chunks; c += (int64_t)gridDim.x * kIdtBlock) {
            Raw12<T> raw;
            load12_raw<T>(p + c * 12, vec, raw);
            double v[12];
            unpack12<T>(raw, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) pixel(v[3 * e], v[3 * e + 1], v[3 * e + 2]);
        }
        if (blockIdx.x == 0) {          // ragged tail: n % 4 pixels
            const int64_t i = (n_chunks << 2) + threadIdx.x;
            if (threadIdx.x < 3 && i < n) pixel((double)p[3 * i], (double)p[3 * i + 1], (double)p[3 * i + 2]);
        }
        // keys: k[q][2j] = key(-min), k[q][2j+1] = key(max); a poisoned lane publishes NaN for all of them
        unsigned long long k[RG][6];
        const bool poisoned = !(bad == 0.0);
        const unsigned long long knan = f64_key(__longlong_as_double(0x7ff8000000000000ll));
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                k[q][2 * j] = poisoned ? knan : f64_key(-mn[q][j]);
                k[q][2 * j + 1] = poisoned ? knan : f64_key(mx[q][j]);
            }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
            for (int q = 0; q < RG; ++q)
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const unsigned long long o = __shfl_down(k[q][i], off, kWave);
                    k[q][i] = o > k[q][i] ? o : k[q][i];
                }
        }
        const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
        __syncthreads();                // the previous group's readers are done with lds
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < RG; ++q)
#pragma unroll
                for (int i = 0; i < 6; ++i) lds[(wid * RG + q) * 6 + i] = k[q][i];
        }
        __syncthreads();
        if (threadIdx.x < 6 * nq) {
            const int q = threadIdx.x / 6, i = threadIdx.x - 6 * q;
            unsigned long long m = lds[q * 6 + i];
            for (int w = 1; w < kIdtBlock / kWave; ++w) {
                const unsigned long long o = lds[(w * RG + q) * 6 + i];
                m = o > m ? o : m;
            }
            // mm[b][it][shard][j][which*2 + {0: max(-d), 1: max(d)}]
            unsigned long long *dst = mm + ((((size_t)b * n_iter + it0 + q0 + q) * kMmShards + (blockIdx.x % kMmShards)) * 3) * 4;
            const int j = i >> 1, mmx = i & 1;
            atomicMax(dst + j * 4 + which * 2 + mmx, m);
        }
    }
}

// decode lo/hi of (pair b, iteration it, axis j) from the key table; numpy widens lo == hi by 0.5
__device__ __forceinline__ void idt_range(const unsigned long long *mm, int j, double &lo, double &hi) {
    const unsigned long long *m = mm + j * 4;
    const double tmin = -key_f64(m[0]), tmax = key_f64(m[1]), rmin = -key_f64(m[2]), rmax = key_f64(m[3]);
    lo = tmin < rmin ? tmin : rmin;   // min(d0r[j].min(), d1r[j].min())
    hi = tmax > rmax ? tmax : rmax;
    if (lo == hi) { lo -= 0.5; hi += 0.5; }
}

__device__ __forceinline__ double idt_edge(int i, int bins, double lo, double hi, double step) {
    return (i == bins) ? hi : ((double)i * step + lo);
}

// unique k with edges[k] <= x < edges[k+1] (last bin closed) == numpy's histogram bin.
// The estimate (x - lo) * scale is right unless x sits within a rounding of an edge, so the common case is ONE check of the
// estimate against its two edges (same edge arithmetic, same comparisons as numpy's correction); only a wave in which some
// lane fails it -- an estimate that is off by one, x on the closed last edge, NaN -- runs the two correction rounds.  The
// result is the correction's result in every case (it would not move a k that passes the check): bit-identical bin indices,
// ~16 float64-rate instructions fewer per axis, pixel and sweep (round 4: the sweeps are bound by vector instruction issue).
__device__ __forceinline__ int idt_bin(double x, int bins, double lo, double hi, double step, double scale) {
    int k = (int)((x - lo) * scale);
    k = k < 0 ? 0 : (k > bins - 1 ? bins - 1 : k);
#ifndef CT_IDT_NO_FASTBIN
    // k <= bins - 1: edge k is never the special last edge, and edge k + 1 only matters when it is not (k < bins - 1)
    const bool settled = (x >= (double)k * step + lo) & ((k == bins - 1) | (x < (double)(k + 1) * step + lo));
    if (__builtin_amdgcn_ballot_w64(!settled) == 0) return k;
#endif
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        if (x < idt_edge(k, bins, lo, hi, step)) k -= (k > 0);
        else if (k != bins - 1 && x >= idt_edge(k + 1, bins, lo, hi, step)) k += 1;
    }
    return k;
}

// -------------------------------------------------------------------------------------------
// A8 part 1: (lo, hi, step, scale) of iteration `it`, axis j, from the min/max keys.  Every consumer evaluates this
// itself (same arithmetic => same bits) instead of waiting for a one-workgroup kernel to publish them.
// -------------------------------------------------------------------------------------------
__device__ __forceinline__ void idt_params(const unsigned long long *__restrict__ keys /* [3][4], shards reduced */, int j, int bins,
                                           double &lo, double &hi, double &step, double &scale) {
    idt_range(keys, j, lo, hi);
    step = (hi - lo) / (double)bins;
    scale = (double)bins / (hi - lo);
}

// the 12 keys of (pair b, iteration it): max over the shards, by the whole workgroup (>= 192 threads), into keys[12] in LDS;
// ends with a barrier
__device__ __forceinline__ void idt_reduce_keys(const unsigned long long *__restrict__ mm, int b, int n_iter, int it,
                                                unsigned long long *stage /* LDS [kMmShards * 12] */, unsigned long long *keys /* LDS [12] */) {
    const unsigned long long *src = mm + ((size_t)b * n_iter + it) * kMmShards * 12;
    if (threadIdx.x < kMmShards * 12) stage[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    if (threadIdx.x < 12) {
        unsigned long long m = stage[threadIdx.x];
#pragma unroll
        for (int sh = 1; sh < kMmShards; ++sh) {
            const unsigned long long o = stage[sh * 12 + threadIdx.x];
            m = o > m ? o : m;
        }
        keys[threadIdx.x] = m;
    }
    __syncthreads();
}

// -------------------------------------------------------------------------------------------
// A7: histograms of both images on the three rotated axes. grid = (G, batch), dynamic LDS = 6*bins*4 B
// -------------------------------------------------------------------------------------------
template <typename TT, typename TR>
__global__ __launch_bounds__(kIdtHistBlock) void idt_hist_kernel(const TT *__restrict__ tgt, int64_t n_t, const TR *__restrict__ ref,
                                                             int64_t n_r, const double *__restrict__ rot,
                                                             const unsigned long long *__restrict__ mm, int n_iter, int it, int bins,
                                                             unsigned int *__restrict__ hist, unsigned short *__restrict__ binidx) {
    extern __shared__ unsigned int lh[];  // [2][3][bins]
    __shared__ unsigned long long kstage[kMmShards * 12], keys[12];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 6 * bins; i += kIdtHistBlock) lh[i] = 0;
    double r[9], lo[3], hi[3], step[3], scale[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) r[i] = rot[((size_t)b * n_iter + it) * 9 + i];
    idt_reduce_keys(mm, b, n_iter, it, kstage, keys);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        idt_params(keys, j, bins, lo[j], hi[j], step[j], scale[j]);
    }
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * kIdtHistBlock;
    const TT *pt = tgt + (size_t)b * n_t * 3;
    for (int64_t i = (int64_t)blockIdx.x * kIdtHistBlock + threadIdx.x; i < n_t; i += stride) {
        const double x0 = (double)pt[3 * i], x1 = (double)pt[3 * i + 1], x2 = (double)pt[3 * i + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = idt_bin(proj(r + 3 * j, x0, x1, x2), bins, lo[j], hi[j], step[j], scale[j]);
            atomicAdd(&lh[j * bins + k], 1u);
            if (binidx) binidx[(((size_t)b * n_iter + it) * 3 + j) * n_t + i] = (unsigned short)k;
        }
    }
    const TR *pr = ref + (size_t)b * n_r * 3;
    for (int64_t i = (int64_t)blockIdx.x * kIdtHistBlock + threadIdx.x; i < n_r; i += stride) {
        const double x0 = (double)pr[3 * i], x1 = (double)pr[3 * i + 1], x2 = (double)pr[3 * i + 2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = idt_bin(proj(r + 3 * j, x0, x1, x2), bins, lo[j], hi[j], step[j], scale[j]);
            atomicAdd(&lh[(3 + j) * bins + k], 1u);
        }
    }
    __syncthreads();
    unsigned int *gh = hist + (((size_t)b * n_iter + it) * kHistShards + (blockIdx.x % kHistShards)) * 6 * bins;
    for (int i = threadIdx.x; i < 6 * bins; i += kIdtHistBlock) {
        const unsigned int c = lh[i];
        if (c) atomicAdd(gh + i, c);
    }
}

// -------------------------------------------------------------------------------------------
// A8 part 2: cumulative histograms -> LUT f and its slopes. grid = (3, batch); dynamic LDS
// -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kIdtBlock) void idt_lut_kernel(const unsigned int *__restrict__ hist, const unsigned long long *__restrict__ mm,
                                                            double *__restrict__ par, int n_iter, int it, int bins,
                                                            double *__restrict__ lut, unsigned int *__restrict__ hsum) {
    extern __shared__ double sm[];  // cp0[bins], cp1[bins], f[bins], then the two summed histograms as u32
    double *cp0 = sm, *cp1 = sm + bins, *f = sm + 2 * bins;
    unsigned int *h0 = reinterpret_cast<unsigned int *>(sm + 3 * bins), *h1 = h0 + bins;
    __shared__ unsigned long long kstage[kMmShards * 12], keys[12];
    const int j = blockIdx.x, b = blockIdx.y;
    // the shards of the two histograms of this axis, added up (exact integers) -> LDS, and published for the debug dump
    for (int i = threadIdx.x; i < bins; i += kIdtBlock) {
        unsigned int a0 = 0, a1 = 0;
#pragma unroll
        for (int sh = 0; sh < kHistShards; ++sh) {
            const unsigned int *g = hist + (((size_t)b * n_iter + it) * kHistShards + sh) * 6 * bins;
            a0 += g[j * bins + i];
            a1 += g[(3 + j) * bins + i];
        }
        h0[i] = a0; h1[i] = a1;
        hsum[(((size_t)b * n_iter + it) * 6 + j) * bins + i] = a0;
        hsum[(((size_t)b * n_iter + it) * 6 + 3 + j) * bins + i] = a1;
    }
    idt_reduce_keys(mm, b, n_iter, it, kstage, keys);        // ends with a barrier: h0 / h1 are complete too
    double lo, hi, step, scale;
    idt_params(keys, j, bins, lo, hi, step, scale);
    if (threadIdx.x == 0) {             // published for the debug dump (ct_idt_debug.par)
        double *q = par + (((size_t)b * n_iter + it) * 3 + j) * 4;
        q[0] = lo; q[1] = hi; q[2] = step; q[3] = scale;
    }
    // exact integer prefix sums of both histograms: each thread owns a contiguous segment (<= 4 bins for bins <= 1024); the 256
    // segment totals are scanned inside the waves with shuffles (no barrier) and the four wave totals through LDS (one barrier)
    // instead of a Hillis-Steele scan with 16 barriers -- this kernel is a chain of latencies (one pair per call: 4 x 7 us of 230)
    __shared__ unsigned long long wtot[2][kIdtBlock / kWave];
    const int per = (bins + kIdtBlock - 1) / kIdtBlock;
    const int b0 = threadIdx.x * per;
    unsigned long long s0 = 0, s1 = 0;
    for (int i = b0; i < b0 + per && i < bins; ++i) { s0 += h0[i]; s1 += h1[i]; }
    unsigned long long i0 = s0, i1 = s1;                  // inclusive scan over the lanes of the wave
    const int lane_ = threadIdx.x & (kWave - 1), wv_ = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned long long a0 = __shfl_up(i0, off, kWave), a1 = __shfl_up(i1, off, kWave);
        if (lane_ >= off) { i0 += a0; i1 += a1; }
    }
    if (lane_ == kWave - 1) { wtot[0][wv_] = i0; wtot[1][wv_] = i1; }
    __syncthreads();
    for (int ww = 0; ww < wv_; ++ww) { i0 += wtot[0][ww]; i1 += wtot[1][ww]; }
    {
        unsigned long long c0 = i0 - s0, c1 = i1 - s1;   // exclusive prefix of this segment
        for (int i = b0; i < b0 + per && i < bins; ++i) {
            c0 += h0[i]; cp0[i] = (double)c0;
            c1 += h1[i]; cp1[i] = (double)c1;
        }
    }
    __syncthreads();
    const double t0 = cp0[bins - 1], t1 = cp1[bins - 1];
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += kIdtBlock) { cp0[i] = cp0[i] / t0; cp1[i] = cp1[i] / t1; }
    __syncthreads();
    // f[i] = np.interp(cp0[i], cp1, edges[1:])
    for (int i = threadIdx.x; i < bins; i += kIdtBlock) {
        const double x = cp0[i];
        const double xp_first = cp1[0], xp_last = cp1[bins - 1];
        double res;
        if (x != x) res = x;
        else if (x > xp_last) res = idt_edge(bins, bins, lo, hi, step);      // right = fp[-1]
        else if (x < xp_first) res = idt_edge(1, bins, lo, hi, step);        // left  = fp[0]
        else {
            int a = 0, e = bins;  // cp1[a] <= x ; e = first index known > x (or bins)
            while (e - a > 1) {
                const int mid = (a + e) >> 1;
                if (cp1[mid] <= x) a = mid; else e = mid;
            }
            const double fa = idt_edge(a + 1, bins, lo, hi, step);
            if (a == bins - 1 || cp1[a] == x) res = fa;
            else {
                const double fb = idt_edge(a + 2, bins, lo, hi, step);
                const double slope = (fb - fa) / (cp1[a + 1] - cp1[a]);
                res = slope * (x - cp1[a]) + fa;
                if (res != res) {
                    res = slope * (x - cp1[a + 1]) + fb;
                    if (res != res && fa == fb) res = fa;
                }
            }
        }
        f[i] = res;
    }
    __syncthreads();
    double *o = lut + (((size_t)b * n_iter + it) * 3 + j) * bins * 2;
    for (int i = threadIdx.x; i < bins; i += kIdtBlock) {
        double slope = 0.0;
        if (i < bins - 1) {
            const double xa = idt_edge(i + 1, bins, lo, hi, step), xb = idt_edge(i + 2, bins, lo, hi, step);
            slope = (f[i + 1] - f[i]) / (xb - xa);
        }
        o[2 * i] = f[i];
        o[2 * i + 1] = slope;
    }
}

// -------------------------------------------------------------------------------------------
// A9 (+ next iteration's A6): d_r = interp(d0r, edges[1:], f, left=0); t += rinv @ (d_r - d0r)
// grid = (G, batch); dynamic LDS = 3*bins*16 B
// -------------------------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(kIdtBlock) void idt_apply_kernel(const TT *__restrict__ tgt, double *__restrict__ out, int64_t n_t,
                                                              const double *__restrict__ rot, const double *__restrict__ rinv,
                                                              const double *__restrict__ lut,
                                                              int n_iter, int it, int bins, int round_f32,
                                                              unsigned long long *__restrict__ mm) {
    extern __shared__ double2 sl[];  // [3][bins] (f, slope)
    __shared__ unsigned long long lds[4 * 6];
    __shared__ unsigned long long kstage[kMmShards * 12], keys[12];
    const int b = blockIdx.y;
    {
        const double2 *g = reinterpret_cast<const double2 *>(lut + ((size_t)b * n_iter + it) * 3 * bins * 2);
        for (int i = threadIdx.x; i < 3 * bins; i += kIdtBlock) sl[i] = g[i];
    }
    double r[9], ri[9], rn[9], lo[3], hi[3], step[3], scale[3];
    const bool has_next = (it + 1 < n_iter);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r[i] = rot[((size_t)b * n_iter + it) * 9 + i];
        ri[i] = rinv[((size_t)b * n_iter + it) * 9 + i];
        rn[i] = has_next ? rot[((size_t)b * n_iter + it + 1) * 9 + i] : 0.0;
    }
    idt_reduce_keys(mm, b, n_iter, it, kstage, keys);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        idt_params(keys, j, bins, lo[j], hi[j], step[j], scale[j]);
    }
    __syncthreads();
    // next iteration's lo/hi of the target: float64 v_min / v_max per pixel (a 64-bit integer max is a compare + two selects
    // and the order-preserving key four more integer ops: ~50 instructions per pixel in round 2); the keys the integer
    // atomicMax needs are built once per lane at the end, like in idt_minmax_kernel, non-finite projections poison the range
    double nmn[3] = {INFINITY, INFINITY, INFINITY}, nmx[3] = {-INFINITY, -INFINITY, -INFINITY}, bad = 0.0;
    const TT *p = tgt + (size_t)b * n_t * 3;
    double *o = out + (size_t)b * n_t * 3;
    for (int64_t i = (int64_t)blockIdx.x * kIdtBlock + threadIdx.x; i < n_t; i += (int64_t)gridDim.x * kIdtBlock) {
        const double x0 = (double)p[3 * i], x1 = (double)p[3 * i + 1], x2 = (double)p[3 * i + 2];
        double delta[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double x = proj(r + 3 * j, x0, x1, x2);
            double dr;
            if (x != x) dr = x;
            else if (x > hi[j]) dr = (double)bins;                         // right (cannot fire: x <= hi)
            else if (x < idt_edge(1, bins, lo[j], hi[j], step[j])) dr = 0.0;  // left = 0: whole first bin -> 0
            else {
                const int k = idt_bin(x, bins, lo[j], hi[j], step[j], scale[j]);
                const int jj = (x >= hi[j]) ? bins - 1 : k - 1;           // last j with edges[j+1] <= x
                const double2 fs = sl[j * bins + jj];
                const double xa = idt_edge(jj + 1, bins, lo[j], hi[j], step[j]);
                if (jj == bins - 1 || xa == x) dr = fs.x;
                else dr = fs.y * (x - xa) + fs.x;
            }
            if (round_f32) dr = (double)(float)dr;   // reference: d_r is float32 on iteration 0 for float32 input
            delta[j] = dr - x;
        }
        const double y0 = proj(ri + 0, delta[0], delta[1], delta[2]) + x0;
        const double y1 = proj(ri + 3, delta[0], delta[1], delta[2]) + x1;
        const double y2 = proj(ri + 6, delta[0], delta[1], delta[2]) + x2;
        o[3 * i] = y0; o[3 * i + 1] = y1; o[3 * i + 2] = y2;
        if (has_next) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double d = proj(rn + 3 * j, y0, y1, y2);
                nmn[j] = fmin(nmn[j], d);
                nmx[j] = fmax(nmx[j], d);
                bad = fma(d, 0.0, bad);              // stays 0 unless d is NaN or infinite
            }
        }
    }
    if (has_next) {
        unsigned long long key[6];
        const bool poisoned = !(bad == 0.0);
        const unsigned long long knan = f64_key(__longlong_as_double(0x7ff8000000000000ll));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            key[2 * j] = poisoned ? knan : f64_key(-nmn[j]);
            key[2 * j + 1] = poisoned ? knan : f64_key(nmx[j]);
        }
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const unsigned long long ov = __shfl_down(key[i], off, kWave);
                key[i] = ov > key[i] ? ov : key[i];
            }
        }
        const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) lds[wid * 6 + i] = key[i];
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            unsigned long long m = lds[threadIdx.x];
            for (int w = 1; w < kIdtBlock / kWave; ++w) {
                const unsigned long long ov = lds[w * 6 + threadIdx.x];
                m = ov > m ? ov : m;
            }
            const int j = threadIdx.x >> 1, mmx = threadIdx.x & 1;
            atomicMax(mm + ((((size_t)b * n_iter + it + 1) * kMmShards + (blockIdx.x % kMmShards)) * 3 + j) * 4 + mmx, m);   // target slots 0,1
        }
    }
}

static int idt_grid(int64_t n, int batch) {
    int64_t want = (n + kIdtBlock - 1) / kIdtBlock;
    int64_t cap = kTargetBlocks / (batch > 0 ? batch : 1);
    if (cap < 8) cap = 8;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

static int idt_minmax_grid(int64_t n, int batch) {
    static const int per_call = [] { const char *e = getenv("CT_IDT_MINMAX_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    int64_t want = (n / 4 + kIdtBlock - 1) / kIdtBlock;
    int64_t cap = per_call / (batch > 0 ? batch : 1);
    if (cap < 16) cap = 16;
    if (want > cap) want = cap;
    return want < 1 ? 1 : (int)want;
}
static int idt_apply_grid(int64_t n, int batch) {
    static const int per_call = [] { const char *e = getenv("CT_IDT_APPLY_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : kTargetBlocks; }();
    int64_t want = (n + kIdtBlock - 1) / kIdtBlock;
    int64_t cap = per_call / (batch > 0 ? batch : 1);
    if (cap > 1024) cap = 1024;          // its 6 same-address atomicMax per workgroup again: 1024 beat 2048 by 4 % (one pair per call)
    if (cap < 8) cap = 8;
    if (want > cap) want = cap;
    return want < 1 ? 1 : (int)want;
}

template <typename T>
static int idt_impl(const T *target, int64_t n_t, const T *reference, int64_t n_r, int batch, const double *rot,
                    const double *rinv, int n_iter, int bins, int round_dr_f32, double *out, void *ws, size_t ws_bytes,
                    const ct_idt_debug *dbg, void *stream_) {
    hipStream_t s = (hipStream_t)stream_;
    if (n_t < 0 || n_r < 0 || batch < 0 || n_iter < 0 || bins < 1 || bins > kIdtMaxBins) return CT_E_BADARG;
    if (batch == 0) return CT_OK;
    if ((n_t > 0 && (!target || !out)) || (n_r > 0 && !reference) || (n_iter > 0 && (!rot || !rinv))) return CT_E_BADARG;
    if (!ws || (reinterpret_cast<uintptr_t>(ws) & 15)) return CT_E_WORKSPACE;
    const IdtLayout l = idt_layout(ws, batch, n_iter > 0 ? n_iter : 1, bins);
    if (ws_bytes < l.total_bytes) return CT_E_WORKSPACE;
    hipError_t e;
    if (n_iter == 0) return CT_E_BADARG;   // the host returns the input untouched in that case
    if (n_t == 0) return CT_OK;
    { const int zr = zero_async(ws, l.zero_bytes, s); if (zr) return zr; }
    const int gt = idt_apply_grid(n_t, batch);
    // the lo/hi sweeps end in 6 * rotations same-address 64-bit atomicMax per workgroup, which serialise in L2 (~20 ns each):
    // with 2048 workgroups that tail was 40 of the sweep's 45 us -- one workgroup per CU keeps it under 5 us
    const int gmm_t = idt_minmax_grid(n_t, batch), gmm_r = idt_minmax_grid(n_r, batch);
    // reference: lo/hi of every iteration's projection in one go; target: iteration 0 only -- both in ONE launch (blockIdx.z)
    {
        const int gmm = gmm_t > gmm_r ? gmm_t : gmm_r;
        hipLaunchKernelGGL((idt_minmax_kernel<T, 8>), dim3(gmm, batch, n_r > 0 ? 2 : 1), dim3(kIdtBlock), 0, s, target, n_t, reference, n_r, rot,
                           n_iter, l.mm);
        CT_CHECK_LAUNCH();
    }
    // the histogram flush is 6*bins global integer atomics per workgroup onto the SAME addresses: keep the grid at
    // ~2 workgroups per CU (contended same-address atomics are an order of magnitude slower, MI355X_MICROARCH.md)
    int gh = idt_grid((n_t > n_r ? n_t : n_r) / 4 + 1, batch);
    {
        static const int hist_cap = [] { const char *e = getenv("CT_IDT_HIST_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
        int cap = 2 * hist_cap / batch;
        cap = cap > hist_cap ? hist_cap : (cap < 64 ? 64 : cap);
        if (gh > cap) gh = cap;                              // per pair
    }
    for (int it = 0; it < n_iter; ++it) {
        unsigned short *bi = (dbg && dbg->binidx) ? dbg->binidx : nullptr;
        if (it == 0) {
            hipLaunchKernelGGL((idt_hist_kernel<T, T>), dim3(gh, batch), dim3(kIdtHistBlock), 6 * bins * sizeof(unsigned int), s,
                               target, n_t, reference, n_r, rot, l.mm, n_iter, it, bins, l.hist, bi);
        } else {
            hipLaunchKernelGGL((idt_hist_kernel<double, T>), dim3(gh, batch), dim3(kIdtHistBlock),
                               6 * bins * sizeof(unsigned int), s, (const double *)out, n_t, reference, n_r, rot, l.mm,
                               n_iter, it, bins, l.hist, bi);
        }
        CT_CHECK_LAUNCH();
        hipLaunchKernelGGL(idt_lut_kernel, dim3(3, batch), dim3(kIdtBlock), 3 * bins * sizeof(double) + 2 * bins * sizeof(unsigned int), s,
                           l.hist, l.mm, l.par, n_iter, it, bins, l.lut, l.hsum);
        CT_CHECK_LAUNCH();
        const int rf = (round_dr_f32 && it == 0) ? 1 : 0;
        if (it == 0) {
            hipLaunchKernelGGL((idt_apply_kernel<T>), dim3(gt, batch), dim3(kIdtBlock), 3 * bins * sizeof(double2), s, target,
                               out, n_t, rot, rinv, l.lut, n_iter, it, bins, rf, l.mm);
        } else {
            hipLaunchKernelGGL((idt_apply_kernel<double>), dim3(gt, batch), dim3(kIdtBlock), 3 * bins * sizeof(double2), s,
                               (const double *)out, out, n_t, rot, rinv, l.lut, n_iter, it, bins, rf, l.mm);
        }
        CT_CHECK_LAUNCH();
    }
    if (dbg) {   // parity probes: device-to-device copies of the integer / LUT state
        if (dbg->hist &&
            (e = hipMemcpyAsync(dbg->hist, l.hsum, (size_t)batch * n_iter * 6 * bins * sizeof(unsigned int),
                                hipMemcpyDeviceToDevice, s)) != hipSuccess)
            return (int)e;
        if (dbg->lut && (e = hipMemcpyAsync(dbg->lut, l.lut, (size_t)batch * n_iter * 3 * bins * 2 * sizeof(double),
                                            hipMemcpyDeviceToDevice, s)) != hipSuccess)
            return (int)e;
        if (dbg->par && (e = hipMemcpyAsync(dbg->par, l.par, (size_t)batch * n_iter * 3 * 4 * sizeof(double),
                                            hipMemcpyDeviceToDevice, s)) != hipSuccess)
            return (int)e;
    }
    return CT_OK;
}

}  // namespace ct

extern "C" {

size_t ct_idt_workspace_bytes(int batch, int n_iter, int bins) {
    if (batch < 0 || n_iter < 0 || bins < 1 || bins > ct::kIdtMaxBins) return 0;
    return ct::idt_layout(nullptr, batch, n_iter > 0 ? n_iter : 1, bins).total_bytes;
}

int ct_idt_f32(const float *target, int64_t n_t, const float *reference, int64_t n_r, int batch, const double *rot,
               const double *rinv, int n_iter, int bins, int round_dr_f32, double *out, void *ws, size_t ws_bytes,
               const ct_idt_debug *dbg, void *stream) {
    return ct::idt_impl<float>(target, n_t, reference, n_r, batch, rot, rinv, n_iter, bins, round_dr_f32, out, ws, ws_bytes,
                               dbg, stream);
}

int ct_idt_f64(const double *target, int64_t n_t, const double *reference, int64_t n_r, int batch, const double *rot,
               const double *rinv, int n_iter, int bins, int round_dr_f32, double *out, void *ws, size_t ws_bytes,
               const ct_idt_debug *dbg, void *stream) {
    return ct::idt_impl<double>(target, n_t, reference, n_r, batch, rot, rinv, n_iter, bins, round_dr_f32, out, ws, ws_bytes,
                                dbg, stream);
}

}  // extern "C"
