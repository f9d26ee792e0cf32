// reinhard_persist.hip -- methods.linear.color_transfer_between_images (methods/linear.py:8-42) as ONE persistent launch.
//
// The two-sweep form (linear.hip: lab_moments_lut_kernel + reinhard_apply_lut_kernel) reads every target frame twice: once
// for its Lab statistics, once to apply the affine map -- 4 planes of HBM traffic per pair for 3 compulsory ones (5 for 4
// with the per-frame PSNR riding on the apply sweep), and it converts the target to the cube-root domain twice.  Here one
// workgroup per CU (1024 threads, the LDS to itself) owns 1 / CUs of every frame:
//
//   R(p)  streams its share of the reference of pair p        -> shifted moments of (fy, fx - fy, fy - fz)
//   T(p)  streams its share of the target, forward transform  -> moments, AND parks (fy, fx - fy, fy - fz) of every pixel in
//         LDS as a float32 triple (12 bytes per pixel, the size of the float32 pixel itself; the float32 difference forms of
//         ct_color_lut.h: no float64 instruction per pixel anywhere in this kernel); a tile with a value outside [0, 1], or with
//         a pixel within rounding of a kink of Lab's f(), is marked and applied with the exact code from memory
//   publish(p): the workgroup's 12 sums go out as ONE wave instruction of 64-bit integer atomic adds
//   A(p)  affine map + inverse transform + gamma + clip out of LDS, result written once; with a ground-truth frame the squared
//         error of the per-frame PSNR (methods/__init__.py:32) is taken from the registers that hold the result
//
// and the frames of a batch are software-pipelined as  R(p+1) | wait(p) | A(p) | T(p+1) | publish(p+1)  so that the grid-wide
// hand-off of pair p's statistics travels while every CU streams the next reference.
//
// The hand-off (cdna_hip_programming.md Guideline 16, R2 "the data is the flag").  A sum v is published as two 64-bit
// words  hi = floor(v 2^24) + 2^46,  lo = frac(v 2^24) 2^40  (|v| < 2^22: the moments of <= 2.6 M pixels with |x - K| <~ 1),
// each carrying (1 << 58) as an arrival count, added with agent-scope integer atomics into one of 8 shard records per pair.
// Integer addition is associative, so the totals -- and everything computed from them -- are bitwise independent of arrival
// order and timing; v is kept to 2^-64 absolute, finer than a float64 tree.  A reader knows a word is complete when its
// count field equals the number of workgroups of the shard: no flag, no fence, no ordering between words.  One lane of
// every workgroup also adds to a plain arrival word that the waiting wave polls (8 lanes, 8 lines) before it reads the 192
// data words, so a poll costs one line per shard.  Sums that do not fit the format (non-finite, huge) arrive with an
// "invalid" mark; such a pair is combined from float64 partials every workgroup also leaves in a slab (fixed order, the
// arithmetic of the two-sweep path), and its statistics are then NaN / inf / huge exactly like the reference's.
// Every spin is bounded (2 s of s_memrealtime); a workgroup that gives up sets the error word and writes NaN.
//
// Determinism: tile -> workgroup -> wave -> lane is a static map, the per-wave trees have a fixed shape, the cross-workgroup
// sum is exact integer arithmetic.  Results do not depend on the number of pairs per call.
#include <atomic>
#include <mutex>

#include "ct_reinhard.h"
#include "ct_reinhard_persist.h"

namespace ct {
namespace rp {

// register budget: the scheduler interleaves the four pixels of a lane as far as it can; a scheduling fence after every
// CT_RP_ILP pixels keeps the 16-wave form (128 registers) free of spills -- the other three waves of the SIMD cover the latencies
#ifndef CT_RP_ILP
#define CT_RP_ILP 1
#endif
#define CT_RP_PIXEL_FENCE(q) do { if (((q) + 1) % CT_RP_ILP == 0) __builtin_amdgcn_sched_barrier(0); } while (0)

#ifndef CT_RP_F32_WAVES
#define CT_RP_F32_WAVES 16
#endif
constexpr int kMaxWaves = 16;                     // waves per workgroup: 16 (4 per SIMD, 128 registers each) or 8 (256 registers)
constexpr int kTileBytes = kTilePixels * 12;      // a parked tile: 3 dwords per pixel
constexpr int kShards = 8;
constexpr int kRecWords = 32;                     // 64-bit words per shard record (two 128-byte lines): 24 data + 1 arrival
constexpr int kDataWords = 24;                    // (hi, lo) x 12 moments
constexpr int kSlabWord = 24;                     // arrivals of the float64 slab rows (read on the slow path only)
constexpr int kMaxPairs = 64;                     // pairs per launch (pivots of 2 x 64 images in LDS); longer batches: several launches
constexpr int kErrBytes = 64;                     // error word block in front of the records (zeroed with them)

constexpr uint64_t kValMask = (1ull << 52) - 1;
constexpr int kInvShift = 52, kCntShift = 58;
constexpr double kHiScale = 0x1p24, kLoScale = 0x1p40;
constexpr int64_t kHiBias = 1ll << 46;

// LDS layout behind the tables and the parked tiles
struct Scratch {
    double red[2][kMaxWaves][12];        // per-wave moment sums, by pair parity: [0,6) target, [6,12) reference
    double coef[2][8];                // sL, sa, sb, cy, ca, cb of the pair being applied, by pair parity
    double side[kMaxWaves][6];        // float64 moment sums of the exact code (out-of-range tiles, the ragged rest), per wave and phase
    double tot[12];
    double stat[2][CT_LAB_STATS_STRIDE];
    double fin[12 * 8];               // slow combine: 8 chains per moment
    unsigned long long rec[kShards * kDataWords];
    float piv[2 * kMaxPairs][4];      // pivot of the shifted sums per image (2^-10 grid)
    float lut255[256];                // uint8 frames: (float)k / 255
    float lin255f[256];               // uint8 frames: gamma expansion of lut255[k] (table E)
    unsigned cnt[2];
    unsigned ticket[3];               // arrivals of the waves at A(p), by pair % 3: the first one collects the statistics
    unsigned ready[2];                // p + 1 once the coefficients of pair p are in coef[p & 1]
    int flag[2];                      // bit 0: some partial sum did not fit the integer format, bit 1: timed out
};
constexpr int kLdsFixed = lut::kLdsBytesAll + (int)sizeof(Scratch);
constexpr int kLdsMax = 160 * 1024;
constexpr int kMaxParked = (kLdsMax - kLdsFixed) / (2 * kTileBytes);      // parked tiles per workgroup and pair (two pairs are in flight)
constexpr int kMaxSlots = 32 * kMaxWaves / 2;      // tiles per workgroup: 32 per wave of the 8-wave form (the raw-tile mask is 32 bits)
static_assert(lut::kLdsBytesAll % 16 == 0 && kTileBytes % 16 == 0, "16-byte aligned LDS regions");

struct Args {
    const void *target, *reference, *gt;
    float *out;
    int64_t n_pixels;
    int batch, slots, pslots;         // slots: tiles per workgroup (ceil(n_tiles / grid)); pslots: how many of them wait in LDS (x 2 pairs)
    unsigned long long *rec;          // [batch][kShards][kRecWords]
    double *slab;                     // [batch][grid][12]
    double *sq;                       // [batch][grid * kMaxWaves]
    unsigned *err;
    double *stats_t, *stats_r;        // NULL or the caller's Lab statistics records of the targets / references of this launch
    unsigned long long *stamps;       // diagnostic builds (-DCT_RP_STAMPS): [grid][2 waves][batch + 1][8] s_memrealtime values
};

__device__ __forceinline__ double wave_sum(double v) {          // fixed-shape tree; valid in lane 0
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}
// float32 sum over the wave with DPP row operations (fixed tree: shifts by 1, 2, 4, 8 inside the rows of 16 lanes, then the
// row totals broadcast into the next rows): six v_add_f32 and no LDS traffic; the total is returned wave-uniform.  Accuracy:
// six roundings of 6e-8 relative on a sum of 64 lanes x 8 pixels, unbiased -- 1e-10 on a mean over 2 M pixels.
__device__ __forceinline__ float wave_sum_f32(float v) {
#define CT_DPP_ADD(ctrl, rmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true))
    CT_DPP_ADD(0x111, 0xf);      // row_shr:1
    CT_DPP_ADD(0x112, 0xf);      // row_shr:2
    CT_DPP_ADD(0x114, 0xf);      // row_shr:4
    CT_DPP_ADD(0x118, 0xf);      // row_shr:8  -> lane 15 of every row holds its row's sum
    CT_DPP_ADD(0x142, 0xa);      // row_bcast:15 into rows 1 and 3
    CT_DPP_ADD(0x143, 0xc);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef CT_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ uint64_t realtime() { return __builtin_amdgcn_s_memrealtime(); }     // 100 MHz
constexpr uint64_t kSpinTicks = 200000000ull;                    // 2 s
// Sticky per-device status (one copy of this variable per device: the code object is loaded on each): set when a bounded spin
// gives up, never cleared by a launch; ct_device_status() reads (and optionally clears) it.  The per-call error word in the
// workspace turns that call's PSNR records into NaN without any synchronisation (psnr_finish_persist_kernel).
__device__ unsigned g_status;
__device__ __forceinline__ void give_up(unsigned *err) {
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)__hip_atomic_fetch_or(&g_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- tile I/O per input type -------------------------------------------------------------------------------------------------
// float32: lane l holds pixels l, l+64, l+128, l+192 of the tile (ct_reinhard.h: one 12-byte access per pixel, 768 contiguous bytes
// per wave instruction).  uint8: lane l holds pixels 4l .. 4l+3 = 12 contiguous bytes, ONE global_load_dwordx3 per tile -- with
// the float32 map it would be twelve byte loads, and a load instruction costs this chip hundreds of cycles whatever its width
// (measured: 23.4 -> 29.5 us per pair).  The price: the two kernels add their float32 moment terms in different lane orders, so a
// uint8 result equals the float32 kernel's on k / 255 to the last rounding of the statistics (<= 2e-7 at 1080p), not bit for bit.
template <typename T> struct Tile;
template <> struct Tile<float> {
    struct Raw { float e[12]; };
    static __device__ __forceinline__ void load(const void *p, int lane, Raw &r) { load_tile(reinterpret_cast<const float *>(p), lane, r.e); }
    static __device__ __forceinline__ bool in_range(const Raw &r) { return __builtin_amdgcn_ballot_w64(max_bits12(r.e) > lut::kOneBits) == 0; }
};
template <> struct Tile<uint8_t> {
    struct Raw { uint32_t d[3]; };
    static __device__ __forceinline__ void load(const void *p, int lane, Raw &r) {
        const uint8_t *q = reinterpret_cast<const uint8_t *>(p) + lane * 12;
        if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) {
            typedef uint32_t u3 __attribute__((ext_vector_type(3)));
            const u3 v = *reinterpret_cast<const u3 __attribute__((aligned(4))) *>(q);     // plain: non-temporal measured 2 % slower here
            r.d[0] = v.x; r.d[1] = v.y; r.d[2] = v.z;
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) r.d[i] = (uint32_t)q[4 * i] | ((uint32_t)q[4 * i + 1] << 8) | ((uint32_t)q[4 * i + 2] << 16) | ((uint32_t)q[4 * i + 3] << 24);
        }
    }
    static __device__ __forceinline__ bool in_range(const Raw &) { return true; }
};
__device__ __forceinline__ uint32_t byte_of(const uint32_t (&d)[3], int i) { return (d[i >> 2] >> (8 * (i & 3))) & 255u; }     // byte i of the lane's 12

// The exact float64 code (ct_color.h) for the rare tiles the tables do not cover -- a value outside [0, 1], a NaN, or affine
// coefficients that are not moderate finite numbers.  They work pixel by pixel from memory (the tile is in L2: it was just
// fetched) in rolled loops, so that they hold no register across the table path: inlined with register-resident tiles they cost
// the 16-wave kernel its 128-register budget (13 - 26 spilled registers).
template <typename T>
__device__ __forceinline__ void load_pixel(const T *p, double &r, double &g, double &b) {
    if constexpr (sizeof(T) == 4) { r = p[0]; g = p[1]; b = p[2]; }
    else { r = (float)p[0] / 255.0f; g = (float)p[1] / 255.0f; b = (float)p[2] / 255.0f; }        // IEEE division, like the tables
}
template <typename T> __device__ __forceinline__ int pixel_of(int lane, int q) { return sizeof(T) == 4 ? q * kWave + lane : lane * 4 + q; }

template <typename T>
__device__ __forceinline__ void exact_moments_tile(const T *tile, const float (&kf)[3], double *side, int lane) {
    const double kd[3] = {(double)kf[0], (double)kf[1], (double)kf[2]};
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        double r, g, b, x, y, z;
        load_pixel<T>(tile + pixel_of<T>(lane, q) * 3, r, g, b);
        to_space<true>(r, g, b, x, y, z);
        accumulate<true>(s, kd, x, y, z);
    }
#pragma unroll 1
    for (int m = 0; m < 6; ++m) {
        const double v = wave_sum(s[m]);
        if (lane == 0) side[m] += v;
    }
}

// ---- forward transform of one tile + its shifted moments; PARK: leave what the apply phase needs in the wave's LDS slot ----
// slot layout: dword (3 j + c) * 64 + lane = component c of the lane's pixel j (conflict-free ds_write_b32 / ds_read_b32, and
// nothing of the tile has to wait in registers for its neighbours); the parked value is the float32 triple (fy, fx - fy, fy - fz)
// STATS = false: the forward transform alone (a target tile fetched a second time during the apply stage)
// Returns true when the apply phase must take the exact code for this tile (from memory): a value outside [0, 1] / a NaN, or a
// pixel within rounding of the kink of Lab's f() (ct_color_lut.h).
template <typename T, bool PARK, bool STATS = true>
__device__ __forceinline__ bool fwd_tile(const unsigned char *tab, Scratch *sc, const void *tile, typename Tile<T>::Raw &cur, const float (&kf)[3],
                                         float (&sf)[6], uint32_t *slot, int w, int lane) {
#ifdef CT_RP_NOEXACT
    const bool raw = false;
#else
    const bool raw = !Tile<T>::in_range(cur);
#endif
    if constexpr (sizeof(T) == 4) {
        if (raw) {                       // some value outside [0, 1] or NaN: moments by the exact code
            if (STATS) exact_moments_tile<T>(reinterpret_cast<const T *>(tile), kf, &sc->side[w][0], lane);
            return true;
        }
    }
    bool near = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float fy, dxy, dyz;
        if constexpr (!PARK) {
            // the reference frame only feeds the statistics: plain float32 values of f(), unbiased per pixel
            // (cube roots from table F here, unlike the two-sweep statistics kernel: this launch is bound by vector issue -- 340
            // instructions per pixel pair at ~4 cycles each -- and the transcendental form costs three more per cube root)
            if constexpr (sizeof(T) == 4) {
                lut::rgb_to_f_stats(tab, cur.e[3 * q], cur.e[3 * q + 1], cur.e[3 * q + 2], fy, dxy, dyz);
            } else {
                const float l[3] = {sc->lin255f[byte_of(cur.d, 3 * q)], sc->lin255f[byte_of(cur.d, 3 * q + 1)], sc->lin255f[byte_of(cur.d, 3 * q + 2)]};
                lut::lin_to_f_stats(tab, l, fy, dxy, dyz);
            }
        } else {
            if constexpr (sizeof(T) == 4) {
                near |= lut::rgb_to_f(tab, cur.e[3 * q], cur.e[3 * q + 1], cur.e[3 * q + 2], fy, dxy, dyz);
            } else {
                const float l[3] = {sc->lin255f[byte_of(cur.d, 3 * q)], sc->lin255f[byte_of(cur.d, 3 * q + 1)], sc->lin255f[byte_of(cur.d, 3 * q + 2)]};
                near |= lut::lin_to_f(tab, l, fy, dxy, dyz);
            }
        }
        // moments in float32 around a pivot on a 2^-10 grid
        if (STATS) {
            const float e0 = fy - kf[0], e1 = dxy - kf[1], e2 = dyz - kf[2];
            sf[0] += e0; sf[1] += e1; sf[2] += e2;
            sf[3] = fmaf(e0, e0, sf[3]); sf[4] = fmaf(e1, e1, sf[4]); sf[5] = fmaf(e2, e2, sf[5]);
        }
        if (PARK) {
            slot[(3 * q) * kWave + lane] = __float_as_uint(fy);
            slot[(3 * q + 1) * kWave + lane] = __float_as_uint(dxy);
            slot[(3 * q + 2) * kWave + lane] = __float_as_uint(dyz);
        }
        CT_RP_PIXEL_FENCE(q);
    }
#ifdef CT_RP_NOEXACT
    return false;
#else
    // (the statistics keep the float32 value of a pixel next to the kink: 3.4e-7 on one pixel of two million)
    return PARK && __builtin_amdgcn_ballot_w64(near) != 0;
#endif
}

// ---- affine map + inverse transform of one parked tile, result stored, squared error against gv accumulated ----------------------
template <typename T> __device__ __forceinline__ void store_pixel(float *tile, int lane, int q, float r, float g, float b) {
    // non-temporal only in the float32 layout, where one store instruction of the wave covers 768 contiguous bytes; in the uint8
    // layout a lane writes its four pixels with four instructions (48-byte lane stride): those need L2 to merge them into lines
    // (non-temporal there: 48.5 -> 24.5 k pairs/s, round 5)
#ifdef CT_NT_STORE
    if constexpr (sizeof(T) == 4) {
        __builtin_nontemporal_store(float3v{r, g, b}, reinterpret_cast<float3u *>(tile + pixel_of<T>(lane, q) * 3));
        return;
    }
#endif
    *reinterpret_cast<float3u *>(tile + pixel_of<T>(lane, q) * 3) = float3v{r, g, b};
}

template <typename T, bool HAS_GT>
__device__ __forceinline__ void apply_tile(const unsigned char *tab, const Scratch *sc, const uint32_t *slot, bool raw, bool coef_bad,
                                           const ReinhardCoef &c, const float (&cf)[6], const void *tgt_tile, const void *gt_tile,
                                           float *out_tile, const typename Tile<T>::Raw &gv, double &sq, int lane) {
    float e = 0.f;
    auto gt_of = [&](int i) -> float {
        if constexpr (sizeof(T) == 4) return gv.e[i]; else return sc->lut255[byte_of(gv.d, i)];
    };
    bool slow = raw || coef_bad;
#ifdef CT_RP_NOEXACT
    slow = false;
#endif
    if (!slow) {
        bool near = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float fy = __uint_as_float(slot[(3 * q) * kWave + lane]), dxy = __uint_as_float(slot[(3 * q + 1) * kWave + lane]),
                        dyz = __uint_as_float(slot[(3 * q + 2) * kWave + lane]);
            // fy' = sL fy + cy ; fx' - fy' = sa (fx - fy) + ca ; fy' - fz' = sb (fy - fz) + cb
            const float gy = fmaf(cf[0], fy, cf[3]), dx = fmaf(cf[1], dxy, cf[4]), dz = fmaf(cf[2], dyz, cf[5]);
            float r, g, b;
            near |= lut::f_to_rgb_clip(tab, gy, dx, dz, r, g, b);
            store_pixel<T>(out_tile, lane, q, r, g, b);
            if (HAS_GT) {
                const float d0 = r - gt_of(3 * q), d1 = g - gt_of(3 * q + 1), d2 = b - gt_of(3 * q + 2);
                e = fmaf(d0, d0, e); e = fmaf(d1, d1, e); e = fmaf(d2, d2, e);
            }
            CT_RP_PIXEL_FENCE(q);
        }
#ifndef CT_RP_NOEXACT
        slow = __builtin_amdgcn_ballot_w64(near) != 0;      // a pixel within rounding of the kink of the inverse: the tile again, exactly
#endif
    }
    if (slow) {                          // exact float64 code from the pixels in memory (the tile was fetched a stage ago at most)
        e = 0.f;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const int px = pixel_of<T>(lane, q);
            double r, g, b;
            load_pixel<T>(reinterpret_cast<const T *>(tgt_tile) + px * 3, r, g, b);
            float o0, o1, o2;
            reinhard_pixel<float, false>(c, r, g, b, o0, o1, o2);
            *reinterpret_cast<float3u *>(out_tile + px * 3) = float3v{o0, o1, o2};
            if (HAS_GT) {                // the ground-truth pixel again from memory: no dynamic index into the prefetched registers
                double g0, g1, g2;
                load_pixel<T>(reinterpret_cast<const T *>(gt_tile) + px * 3, g0, g1, g2);
                const float d0 = o0 - (float)g0, d1 = o1 - (float)g1, d2 = o2 - (float)g2;
                e = fmaf(d0, d0, e); e = fmaf(d1, d1, e); e = fmaf(d2, d2, e);
            }
        }
    }
    if (HAS_GT) sq += (double)e;
}

// ---- the waiting wave: poll the arrival words of pair `rec`, read the 192 data words, leave the 12 totals in sc->tot ----------
// returns 0 ok, 1 some partial sum was marked invalid, 2 timed out
__device__ __forceinline__ int collect(const unsigned long long *rec, int nwg, int lane, Scratch *sc, unsigned *err) {
    const uint64_t t0 = realtime();
    for (;;) {                                                   // the data words carry their own arrival counts
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int idx = lane + kWave * j;                    // 0 .. 191
            const int shard = idx / kDataWords, q = idx - shard * kDataWords;
            const unsigned long long x = __hip_atomic_load(rec + shard * kRecWords + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok &= (x >> kCntShift) == (unsigned long long)((nwg - shard + kShards - 1) / kShards);
            sc->rec[idx] = x;
        }
        __threadfence_block();
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
        if (realtime() - t0 > kSpinTicks) { if (lane == 0) give_up(err); return 2; }
        __builtin_amdgcn_s_sleep(8);
    }
    unsigned long long val = 0, inv = 0;
    if (lane < kDataWords) {
#pragma unroll
        for (int s = 0; s < kShards; ++s) {
            const unsigned long long x = sc->rec[s * kDataWords + lane];
            val += x & kValMask;
            inv += (x >> kInvShift) & 63ull;
        }
    }
    // word 2m = hi, 2m+1 = lo of moment m
    double dv = (lane & 1) ? (double)val * 0x1p-64 : (double)((long long)val - (long long)nwg * kHiBias) * 0x1p-24;
    dv += __shfl_down(dv, 1, kWave);
    if (lane < kDataWords && !(lane & 1)) sc->tot[lane >> 1] = dv;
    __threadfence_block();
    return __builtin_amdgcn_ballot_w64(lane < kDataWords && inv != 0) ? 1 : 0;
}

// ---- statistics records and affine coefficients of pair `pair` from the 12 totals (lanes 0..5 of one wave) ---------------------
__device__ __forceinline__ void finish_stats(Scratch *sc, int par, int pair, int batch, int64_t n_pixels, int lane, double *stats_t, double *stats_r) {
    const double n = (double)n_pixels;
    if (lane < 6) {
        const int img = lane / 3, ch = lane - 3 * img;
        const double s1 = sc->tot[img * 6 + ch], s2 = sc->tot[img * 6 + 3 + ch];
        const double k = (double)sc->piv[img ? batch + pair : pair][ch];
        const double m0 = s1 / n;
        const double var = fma(-s1, m0, s2) / n;
        const double scale = ch == 0 ? 116.0 : (ch == 1 ? 500.0 : 200.0);
        const double mean = ch == 0 ? fma(116.0, k + m0, -16.0) : scale * (k + m0);
        const double sd = scale * var_to_sd(var, kVarFloorF32);
        sc->stat[img][ch] = mean;
        sc->stat[img][3 + ch] = sd;
        if (ch == 0) { sc->stat[img][6] = n; sc->stat[img][7] = 0.0; }
    }
    __threadfence_block();               // same wave: its LDS accesses execute in order
    if (lane < 3) {
        const double mt = sc->stat[0][lane], st = sc->stat[0][3 + lane], mr = sc->stat[1][lane], sr = sc->stat[1][3 + lane];
        const double s = sr / st;                    // sigma_r / sigma_t (inf / nan on a constant target, like the reference)
        double off;
        if (lane == 0) off = (fma(-16.0 - mt, s, mr) + 16.0) * (1.0 / 116.0);
        else off = fma(-mt, s, mr) * (lane == 1 ? 1.0 / 500.0 : 1.0 / 200.0);
        sc->coef[par][lane] = s;
        sc->coef[par][3 + lane] = off;
    }
    if (stats_t != nullptr && lane < 2 * CT_LAB_STATS_STRIDE) {
        const int img = lane / CT_LAB_STATS_STRIDE, m = lane - img * CT_LAB_STATS_STRIDE;
        (img ? stats_r : stats_t)[(size_t)pair * CT_LAB_STATS_STRIDE + m] = sc->stat[img][m];
    }
}

__device__ __forceinline__ float uniform_f32(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// moments of the ragged rest of a frame (thread t of the workgroup takes pixel t), exact float64 code, into the wave's side sums
template <typename T>
__device__ __forceinline__ void tail_moments(const T *p, int tail, const float (&kf)[3], Scratch *sc, int w, int lane) {
    if (w * kWave >= tail) return;                                    // wave-uniform
    double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int t = w * kWave + lane;
    if (t < tail) {
        const double kd[3] = {(double)kf[0], (double)kf[1], (double)kf[2]};
        double x, y, z, r0, r1, r2;
        if constexpr (sizeof(T) == 4) { r0 = p[t * 3]; r1 = p[t * 3 + 1]; r2 = p[t * 3 + 2]; }
        else { r0 = (float)p[t * 3] / 255.0f; r1 = (float)p[t * 3 + 1] / 255.0f; r2 = (float)p[t * 3 + 2] / 255.0f; }
        to_space<true>(r0, r1, r2, x, y, z);
        accumulate<true>(s, kd, x, y, z);
    }
#pragma unroll 1
    for (int m = 0; m < 6; ++m) {
        const double v = wave_sum(s[m]);
        if (lane == 0) sc->side[w][m] += v;
    }
}

template <typename T, bool HAS_GT, int kWaves>
__global__ __launch_bounds__(kWaves * kWave) void reinhard_persist_kernel(const Args a) {
    constexpr int kThreads = kWaves * kWave;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *tab = smem;
    uint32_t *park = reinterpret_cast<uint32_t *>(smem + lut::kLdsBytesAll);           // [2 pairs][pslots][768 dwords]
    Scratch *sc = reinterpret_cast<Scratch *>(smem + lut::kLdsBytesAll + (size_t)2 * a.pslots * kTileBytes);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = blockIdx.x, nwg = gridDim.x;
    const int B = a.batch;
    const int64_t n_full = a.n_pixels / kTilePixels;
    constexpr size_t kPx = 3 * sizeof(T);                        // bytes per pixel of the input frames
    // tiles of this wave: slots w, w + kWaves, ... of the workgroup; slot s holds tile s * nwg + g; slots below pslots are parked
    int Kw = 0;
    for (int s = w; s < a.slots && (int64_t)s * nwg + g < n_full; s += kWaves) ++Kw;
    const int tail = (g == nwg - 1) ? (int)(a.n_pixels - n_full * kTilePixels) : 0;       // ragged rest (< 256 pixels): threads [0, tail) of the last workgroup, exact code
    const int n_stages = 2 + 2 * B;          // S(0) S(1) A(0) S(2) A(1) S(3) ... : S(p) sweeps pair p (R then T, publish), A(p) applies it

    auto parked = [&](int k) -> bool { return w + kWaves * k < a.pslots; };
    auto slot_of = [&](int p, int k) -> uint32_t * { return park + ((size_t)(p & 1) * a.pslots + (w + kWaves * k)) * (kTileBytes / 4); };
    auto tile_off = [&](int k) -> size_t { return (size_t)((int64_t)(w + kWaves * k) * nwg + g) * kTilePixels; };     // in pixels
    auto tile_ptr = [&](const void *base, int pair, int k) -> const void * {
        return reinterpret_cast<const unsigned char *>(base) + ((size_t)pair * a.n_pixels + tile_off(k)) * kPx;
    };
    // The wave's loads, in the order it consumes them ("jobs"); every job fetches one tile one job ahead of its use.
    //   S stage (pair p):  j in [0, Kw): reference tile j;  j in [Kw, 2 Kw): target tile j - Kw
    //   A stage (pair p):  j = 2 k: target tile k again (tiles that are not parked);  j = 2 k + 1: ground-truth tile k (with a metric)
    auto stage_pair = [&](int st) -> int { return (st < 2 || (st & 1)) ? (st == 0 ? 0 : (st + 1) >> 1) : (st - 2) >> 1; };
    auto is_sweep = [&](int st) -> bool { return st < 2 || (st & 1); };
    auto job_exists = [&](int st, int j) -> bool {
        const int p = stage_pair(st);
        if (p >= B) return false;
        if (is_sweep(st)) return true;
        return (j & 1) ? HAS_GT : !parked(j >> 1);
    };
    auto job_ptr = [&](int st, int j) -> const void * {
        const int p = stage_pair(st);
        if (is_sweep(st)) return j < Kw ? tile_ptr(a.reference, p, j) : tile_ptr(a.target, p, j - Kw);
        return (j & 1) ? tile_ptr(a.gt, p, j >> 1) : tile_ptr(a.target, p, j >> 1);
    };
    auto next_load = [&](int st, int j) -> const void * {        // the job after (st, j), or null
        if (Kw == 0) return nullptr;
        do {
            if (++j >= 2 * Kw) { j = 0; ++st; }
        } while (st < n_stages && !job_exists(st, j));
        return st < n_stages ? job_ptr(st, j) : nullptr;
    };

    typename Tile<T>::Raw cur, nxt;
    if (Kw > 0 && B > 0) Tile<T>::load(job_ptr(0, 0), lane, cur);            // in flight while the tables are copied
    lut::load_tables<kThreads, true>(tab);
    float p0[3] = {0.5f, 0.5f, 0.5f};
    const bool piv_thread = threadIdx.x < 2 * B && a.n_pixels > 0;
    if (piv_thread) {
        const int img = threadIdx.x;
        const T *p = reinterpret_cast<const T *>(img < B ? a.target : a.reference) + (size_t)(img < B ? img : img - B) * a.n_pixels * 3;
        if constexpr (sizeof(T) == 4) { p0[0] = p[0]; p0[1] = p[1]; p0[2] = p[2]; }
        else { p0[0] = (float)p[0] / 255.0f; p0[1] = (float)p[1] / 255.0f; p0[2] = (float)p[2] / 255.0f; }
    }
    if (threadIdx.x < 2) { sc->cnt[threadIdx.x] = 0; sc->flag[threadIdx.x] = 0; sc->ready[threadIdx.x] = 0; }
    if (threadIdx.x < 3) sc->ticket[threadIdx.x] = 0;
    if (sizeof(T) == 1 && threadIdx.x < 256) sc->lut255[threadIdx.x] = (float)threadIdx.x / 255.0f;     // IEEE division: the reference's .float() / 255
    __syncthreads();
    if (sizeof(T) == 1 && threadIdx.x < 256) {                   // exactly what the float32 kernel computes for the value k / 255
        sc->lin255f[threadIdx.x] = lut::expand(tab, sc->lut255[threadIdx.x]);
    }
    if (threadIdx.x < 2 * B) {
        // pivot of the shifted sums: pixel 0 of the image in the cube-root domain, on a 2^-10 grid (ct_reinhard.h / linear.hip)
        float fy = 0.5f, dxy = 0.0f, dyz = 0.0f;
        if (piv_thread && max(max(__float_as_uint(p0[0]), __float_as_uint(p0[1])), __float_as_uint(p0[2])) <= lut::kOneBits) {
            lut::rgb_to_f_stats(tab, p0[0], p0[1], p0[2], fy, dxy, dyz);
        }
        sc->piv[threadIdx.x][0] = rintf(fy * 1024.0f) * (1.0f / 1024.0f);
        sc->piv[threadIdx.x][1] = rintf(dxy * 1024.0f) * (1.0f / 1024.0f);
        sc->piv[threadIdx.x][2] = rintf(dyz * 1024.0f) * (1.0f / 1024.0f);
    }
    __syncthreads();

    uint32_t rawmask0 = 0, rawmask1 = 0;     // per pair parity, bit k: tile k of this wave takes the exact code in A
#ifdef CT_RP_STAMPS          // diagnostic build only: where a stage's time goes (never timed, never shipped)
#define CT_RP_STAMP(slot) do { if (a.stamps && lane == 0 && (w == 0 || w == kWaves - 1)) \
        a.stamps[(((size_t)g * 2 + (w ? 1 : 0)) * n_stages + st) * 4 + (slot)] = realtime(); } while (0)
#else
#define CT_RP_STAMP(slot) do { } while (0)
#endif
    for (int st = 0; st < n_stages; ++st) {
        const int p = stage_pair(st), par = p & 1;
        if (p >= B) continue;
        CT_RP_STAMP(0);
        if (is_sweep(st)) {
            // ---------------- S(p): R -- moments of the reference share ----------------
            {
                float sf[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float kf[3] = {uniform_f32(sc->piv[B + p][0]), uniform_f32(sc->piv[B + p][1]), uniform_f32(sc->piv[B + p][2])};
                if (lane < 6) sc->side[w][lane] = 0.0;
                for (int k = 0; k < Kw; ++k) {
                    const void *np = next_load(st, k);
                    if (np) Tile<T>::load(np, lane, nxt);
                    fwd_tile<T, false>(tab, sc, tile_ptr(a.reference, p, k), cur, kf, sf, nullptr, w, lane);
                    cur = nxt;
                }
                if (tail > 0) tail_moments(reinterpret_cast<const T *>(a.reference) + ((size_t)p * a.n_pixels + n_full * kTilePixels) * 3, tail, kf, sc, w, lane);
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const float v = wave_sum_f32(sf[m]);
                    if (lane == 0) sc->red[par][w][6 + m] = (double)v + sc->side[w][m];
                }
            }
            CT_RP_STAMP(1);
            // ---------------- S(p): T -- moments of the target share; the first pslots tiles stay in LDS for A(p) ----------------
            {
                float sf[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const float kf[3] = {uniform_f32(sc->piv[p][0]), uniform_f32(sc->piv[p][1]), uniform_f32(sc->piv[p][2])};
                if (lane < 6) sc->side[w][lane] = 0.0;
                uint32_t rm = 0;
                for (int k = 0; k < Kw; ++k) {
                    const void *np = next_load(st, Kw + k);
                    if (np) Tile<T>::load(np, lane, nxt);
                    if (parked(k)) {
                        if (fwd_tile<T, true>(tab, sc, tile_ptr(a.target, p, k), cur, kf, sf, slot_of(p, k), w, lane)) rm |= 1u << k;
                    } else {
                        fwd_tile<T, false>(tab, sc, tile_ptr(a.target, p, k), cur, kf, sf, nullptr, w, lane);
                    }
                    cur = nxt;
                }
                if (par) rawmask1 = rm; else rawmask0 = rm;
                if (tail > 0) tail_moments(reinterpret_cast<const T *>(a.target) + ((size_t)p * a.n_pixels + n_full * kTilePixels) * 3, tail, kf, sc, w, lane);
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const float v = wave_sum_f32(sf[m]);
                    if (lane == 0) sc->red[par][w][m] = (double)v + sc->side[w][m];
                }
            }
            CT_RP_STAMP(2);
            // ---------------- publish(p): the wave whose arrival completes the workgroup adds the wave sums in wave order ----------------
            unsigned old = 0;
            if (lane == 0) {
                __threadfence_block();
                old = atomicAdd(&sc->cnt[par], 1u);
            }
            old = __builtin_amdgcn_readfirstlane(old);
            if (old == kWaves - 1) {
                __threadfence_block();
                double v = 0.0;
                if (lane < 12) {
                    v = sc->red[par][0][lane];
#pragma unroll
                    for (int ww = 1; ww < kWaves; ++ww) v += sc->red[par][ww][lane];
                }
                if (lane == 0) sc->cnt[par] = 0;
                // lanes 2m (hi) and 2m+1 (lo) carry moment m
                const double vm = __shfl(v, lane >> 1, kWave);
                const bool fits = fabs(vm) < 0x1p22;             // false for NaN too
                if (lane < kDataWords) {
                    const double sc24 = (fits ? vm : 0.0) * kHiScale;
                    const double fl = floor(sc24);
                    const unsigned long long hi = (unsigned long long)((long long)fl + kHiBias);
                    const unsigned long long lo = (unsigned long long)((sc24 - fl) * kLoScale);
                    const unsigned long long word = ((lane & 1) ? lo : hi) + (1ull << kCntShift) + (fits ? 0ull : (1ull << kInvShift));
                    (void)__hip_atomic_fetch_add(a.rec + ((size_t)p * kShards + (g % kShards)) * kRecWords + lane, word, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
                }
                // the float64 partials for the slow path (read only when some sum did not fit): write-through stores, drained,
                // then counted in the shard's slab word -- all of it behind the atomics, off the hand-off's critical path
                if (lane < 12)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.slab + ((size_t)p * nwg + g) * 12 + lane),
                                       (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0)
                    (void)__hip_atomic_fetch_add(a.rec + ((size_t)p * kShards + (g % kShards)) * kRecWords + kSlabWord, 1ull, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
            }
            CT_RP_STAMP(3);
            continue;
        }
        // ---------------- A(p): the statistics of pair p were published a whole stage ago ----------------
        // No workgroup barrier here: the first wave to arrive collects the totals and leaves the coefficients in LDS, the others
        // wait on an LDS word.  Waves of one workgroup so drift apart by up to a stage -- an old wave that would idle behind a
        // barrier (the SIMD arbitrates by age) starts the next sweep instead.  Safe without further guards: a wave can only pass
        // A(p) once EVERY wave of the grid has published S(p), so no per-parity LDS word of pair p is rewritten (by S(p+2) / A(p+2))
        // before all its readers are through.
        bool timed_out = false;
        {
            unsigned tk = 0;
            // pair % 3, not parity: a wave ARRIVES at A(p + 2) as soon as it has passed A(p + 1), i.e. once every wave of the grid has
            // published pair p + 1 -- which every wave does before it arrives at A(p).  The fastest wave can so be two apply phases
            // ahead of the slowest one's ARRIVAL (never of its passing: that needs pair p + 2 published, after A(p) in program order),
            // and a ticket shared by A(p) and A(p + 2) could be handed out twice.  (Found in round 5 with an experimental schedule whose
            // last apply phases follow each other without a sweep in between; this schedule has a sweep there, which made the race
            // a matter of tens of microseconds of skew rather than impossible.)
            if (lane == 0) tk = atomicAdd(&sc->ticket[p % 3], 1u);
            tk = __builtin_amdgcn_readfirstlane(tk);
            if (tk == 0) {
                const int rc0 = collect(a.rec + (size_t)p * kShards * kRecWords, nwg, lane, sc, a.err);
                if (rc0 == 0) finish_stats(sc, par, p, B, a.n_pixels, lane, g == 0 ? a.stats_t : nullptr, a.stats_r);
                if (lane == 0) {
                    sc->flag[par] = rc0;
                    __threadfence_block();
                    __hip_atomic_store(&sc->ready[par], (unsigned)(p + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                CT_RP_STAMP(1);
            }
            if (tk == kWaves - 1 && lane == 0) sc->ticket[p % 3] = 0;                // every wave is here: free for A(p + 3)
            // bounded like every other spin of this kernel (the collecting wave gives up after kSpinTicks itself and then publishes
            // rc = 2: this bound is a second line of defence, 2 x kSpinTicks)
            bool ready_ok = true;
            const uint64_t tw = realtime();
            while (__hip_atomic_load(&sc->ready[par], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)(p + 1)) {
                if (realtime() - tw > 2 * kSpinTicks) { ready_ok = false; if (lane == 0) give_up(a.err); break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ready_ok) timed_out = true;
        }
        CT_RP_STAMP(2);
        const int rc = timed_out ? 2 : sc->flag[par];
        if (rc == 1) {
            // some workgroup's sums did not fit the integer format (non-finite or huge values): combine the float64 partials
            // of the slab in a fixed order -- 8 interleaved chains per moment, then in chain order (linear.hip's prologue)
            if (threadIdx.x < 96) {
                // the slab rows are published behind the integer sums: wait until every shard has counted all of its rows
                const uint64_t t0 = realtime();
                for (;;) {
                    unsigned long long x = 0, want = 0;
                    if (lane < kShards) {
                        want = (unsigned long long)((nwg - lane + kShards - 1) / kShards);
                        x = __hip_atomic_load(a.rec + ((size_t)p * kShards + lane) * kRecWords + kSlabWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (__builtin_amdgcn_ballot_w64(x != want) == 0) break;
                    if (realtime() - t0 > kSpinTicks) { if (lane == 0) give_up(a.err); break; }
                    __builtin_amdgcn_s_sleep(8);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const int m = threadIdx.x >> 3, sub = threadIdx.x & 7;
                const double *src = a.slab + (size_t)p * nwg * 12 + m;
                double acc = 0.0;
                for (int b = sub; b < nwg; b += 8) acc += src[(size_t)b * 12];
                sc->fin[m * 8 + sub] = acc;
            }
            __syncthreads();
            if (threadIdx.x < 12) {
                const double *f = sc->fin + threadIdx.x * 8;
                sc->tot[threadIdx.x] = ((((((f[0] + f[1]) + f[2]) + f[3]) + f[4]) + f[5]) + f[6]) + f[7];
            }
            __syncthreads();
            if (w == 0) finish_stats(sc, par, p, B, a.n_pixels, lane, g == 0 ? a.stats_t : nullptr, a.stats_r);
            __syncthreads();
        }
        ReinhardCoef c;
        c.sL = uniform_f64(sc->coef[par][0]); c.sa = uniform_f64(sc->coef[par][1]); c.sb = uniform_f64(sc->coef[par][2]);
        c.cy = uniform_f64(sc->coef[par][3]); c.ca = uniform_f64(sc->coef[par][4]); c.cb = uniform_f64(sc->coef[par][5]);
        if (rc == 2) { c.sL = c.sa = c.sb = c.cy = c.ca = c.cb = __longlong_as_double(0x7ff8000000000000ll); }
        const bool coef_bad = !reinhard_coef_fast(c);
        const float cf[6] = {(float)c.sL, (float)c.sa, (float)c.sb, (float)c.cy, (float)c.ca, (float)c.cb};
        double sq = 0.0;
        const uint32_t rawmask = par ? rawmask1 : rawmask0;
        for (int k = 0; k < Kw; ++k) {
            const size_t off = (size_t)p * a.n_pixels + tile_off(k);
            const void *tgt_tile = reinterpret_cast<const unsigned char *>(a.target) + off * kPx;
            const void *gt_tile = HAS_GT ? reinterpret_cast<const unsigned char *>(a.gt) + off * kPx : nullptr;
            const uint32_t *slot = slot_of(p, k);
            bool raw = (rawmask >> k) & 1u;
            if (!parked(k)) {
                // job 2k: cur = the target tile, fetched a second time.  It goes through the wave's own slot 0 of this pair, free since
                // tile 0 was applied (every wave that has tiles has tile 0 parked: pslots >= min(slots, waves)): forward transform
                // into the slot, then the same apply as a parked tile
                const void *np = next_load(st, 2 * k);
                if (np) Tile<T>::load(np, lane, nxt);
                float none[6];
                const float kz[3] = {0.f, 0.f, 0.f};
                uint32_t *tmp = slot_of(p, 0);
                raw = fwd_tile<T, true, false>(tab, sc, tgt_tile, cur, kz, none, tmp, w, lane);
                slot = tmp;
                cur = nxt;
            }
            if (HAS_GT) {                    // job 2k+1: cur = ground-truth tile k
                const void *np = next_load(st, 2 * k + 1);
                if (np) Tile<T>::load(np, lane, nxt);
            }
            apply_tile<T, HAS_GT>(tab, sc, slot, raw, coef_bad, c, cf, tgt_tile, gt_tile, a.out + off * 3, cur, sq, lane);
            if (HAS_GT) cur = nxt;
        }
        if ((int)threadIdx.x < tail) {
            const size_t px = (size_t)p * a.n_pixels + n_full * kTilePixels + threadIdx.x;
            double r0, r1, r2;
            load_pixel<T>(reinterpret_cast<const T *>(a.target) + px * 3, r0, r1, r2);
            float o0, o1, o2;
            reinhard_pixel<float, false>(c, r0, r1, r2, o0, o1, o2);
            a.out[px * 3] = o0; a.out[px * 3 + 1] = o1; a.out[px * 3 + 2] = o2;
            if (HAS_GT) {
                double g0, g1, g2;
                load_pixel<T>(reinterpret_cast<const T *>(a.gt) + px * 3, g0, g1, g2);
                const double d0 = (double)o0 - g0, d1 = (double)o1 - g1, d2 = (double)o2 - g2;
                sq += (d0 * d0 + d1 * d1) + d2 * d2;
            }
        }
        if (HAS_GT) {
            const double v = wave_sum(sq);
            if (kWaves < kMaxWaves && lane == 0) a.sq[((size_t)p * nwg + g) * kMaxWaves + kWaves + w] = 0.0;
            if (lane == 0) a.sq[((size_t)p * nwg + g) * kMaxWaves + w] = v;
        }
        CT_RP_STAMP(3);
    }
}

// fixed-order finish of the per-wave squared-error partials -> {mse, PSNR} per frame (piq.psnr semantics, methods/__init__.py:32)
__global__ __launch_bounds__(256) void psnr_finish_persist_kernel(const double *__restrict__ sq, int n_parts, int64_t n_elems, double *__restrict__ out,
                                                                  const unsigned *__restrict__ err) {
    __shared__ double lds[4];
    double s[1] = {0.0};
    for (int i = threadIdx.x; i < n_parts; i += 256) s[0] += sq[(size_t)blockIdx.x * n_parts + i];
    block_sum<1>(s, lds);
    if (threadIdx.x == 0) {
        double mse = s[0] / (double)n_elems;
        if (*err != 0u) mse = __longlong_as_double(0x7ff8000000000000ll);       // a spin of the launch gave up: its frames are NaN, so is the metric
        out[blockIdx.x * 2] = mse;
        out[blockIdx.x * 2 + 1] = 10.0 * log10(1.0 / (mse > 1e-300 ? mse : (mse == mse ? 1e-300 : mse)));
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------------
constexpr int kMaxGrid = 63 * kShards;            // the arrival / invalid counts of a shard record are 6 bits wide
// workgroups of the launch = CUs of the CURRENT device (cached per device), or CT_HIP_PERSIST_WGS (tuning / a device with
// masked CUs); 0 = this device cannot take the launch (more workgroups than the hand-off's count fields hold)
static int grid_size() {
    static std::atomic<int> cache[kMaxDevices];
    const int dev = current_device();
    int n = cache[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipGetLastError();
        n = cus > 0 ? cus : 256;
        const char *e = getenv("CT_HIP_PERSIST_WGS");
        if (e && atoi(e) > 0 && atoi(e) <= n) n = atoi(e);      // never more workgroups than CUs: each needs a CU's LDS to itself
        if (n > kMaxGrid) n = -1;
        cache[dev].store(n, std::memory_order_relaxed);
    }
    return n > 0 ? n : 0;
}
static int slots_for(int64_t n_pixels) {
    const int64_t tiles = n_pixels / kTilePixels;
    const int g = grid_size();
    if (g <= 0) return kMaxSlots + 1;              // not eligible on this device
    return (int)((tiles + g - 1) / g);
}

bool eligible(int64_t n_pixels, bool any_size) {
    // The fused float32 entries (ct_reinhard_f32 / ct_reinhard_psnr_f32) keep the two sweeps unless CT_HIP_REINHARD_PERSIST=1: on
    // float32 frames both forms are bound by vector instruction issue, and this one executes 22 % more of them (DESIGN.md 4.1:
    // 33 - 34 k pairs/s against 36.5 k at 1080p); the uint8 entry and ct_reinhard_persist_f32 always come here.
    static const bool on = [] { const char *e = getenv("CT_HIP_REINHARD_PERSIST"); return e && e[0] == '1'; }();
    if (n_pixels < kTilePixels) return false;
    const int slots = slots_for(n_pixels);
    if (slots > kMaxSlots) return false;
    if (any_size) return true;
    return on && slots >= kMaxWaves && slots <= 2 * kMaxParked;   // every wave has a tile, and at least half of them wait in LDS
}

static size_t rec_bytes(int batch) { return kErrBytes + (size_t)batch * kShards * kRecWords * sizeof(unsigned long long); }    // a multiple of 16

size_t ws_bytes(int64_t n_pixels, int batch) {
    if (!eligible(n_pixels, true) || batch <= 0) return 0;
    const int b = batch < kMaxPairs ? batch : kMaxPairs;
    const size_t g = (size_t)grid_size();
    size_t n = rec_bytes(b) + (size_t)b * g * 12 * sizeof(double) + (size_t)b * g * kMaxWaves * sizeof(double);
#ifdef CT_RP_STAMPS
    n += g * 2 * (size_t)(2 + 2 * b) * 4 * sizeof(unsigned long long);
#endif
    return n;
}

// The launch chain of a device (see launch()): ONE for all instantiations -- a float32 launch on one stream and a uint8 launch on
// another starve each other exactly like two launches of the same type (round 5 kept these as function-local statics of the
// template, i.e. one chain per element type).
static std::mutex g_chain_mutex;
static hipEvent_t g_chain_event[kMaxDevices];
static hipStream_t g_chain_stream[kMaxDevices];

// waves per workgroup of the float32 instantiation: 16 (4 per SIMD, 128 registers: 54 of them spilled) or 8 (2 per SIMD, 256
// registers, nothing spilled); CT_HIP_PERSIST_WAVES presets it (tuning)
static int f32_waves() {
    static const int v = [] { const char *e = getenv("CT_HIP_PERSIST_WAVES"); const int x = e ? atoi(e) : 0; return x == 8 || x == 16 ? x : CT_RP_F32_WAVES; }();
    return v;
}

template <typename T>
int launch(const T *target, const T *reference, const T *gt, float *out, double *psnr_out, int64_t n_pixels, int batch, double *stats_out,
           void *ws, size_t ws_size, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (!eligible(n_pixels, true)) return CT_E_BADARG;
    if (ws == nullptr || (reinterpret_cast<uintptr_t>(ws) & 15) || ws_size < ws_bytes(n_pixels, batch)) return CT_E_WORKSPACE;
    if (gt != nullptr && psnr_out == nullptr) return CT_E_BADARG;
    {   // out must not overlap an input frame: the exact redo of a flagged tile re-reads the input after the fast result is stored
        const uintptr_t o0 = reinterpret_cast<uintptr_t>(out), o1 = o0 + (size_t)batch * n_pixels * 3 * sizeof(float);
        const size_t in_bytes = (size_t)batch * n_pixels * 3 * sizeof(T);
        for (const T *q : {target, reference, gt}) {
            const uintptr_t q0 = reinterpret_cast<uintptr_t>(q);
            if (q != nullptr && q0 < o1 && o0 < q0 + in_bytes) return CT_E_BADARG;
        }
    }
    const int g = grid_size();
    const int slots = slots_for(n_pixels);
    const int pslots = slots < kMaxParked ? slots : kMaxParked;
    const size_t lds = (size_t)kLdsFixed + (size_t)2 * pslots * kTileBytes;
    // 16 waves per workgroup (4 per SIMD, 128 registers).  The kernel is a template on the wave count: the 8-wave form (256 registers,
    // no spills) measured 28.5 - 35.9 k pairs/s against 34.2 - 34.5 k on float32 frames and is not instantiated.
    int waves = kMaxWaves;
    void (*kern)(const Args) = gt ? reinhard_persist_kernel<T, true, kMaxWaves> : reinhard_persist_kernel<T, false, kMaxWaves>;
    if constexpr (sizeof(T) == 4) {
        if (f32_waves() == 8) {
            waves = 8;
            kern = gt ? reinhard_persist_kernel<T, true, 8> : reinhard_persist_kernel<T, false, 8>;
        }
    }
    // function attributes are per device: cached per (device, instantiation of this template (T) x gt x wave count)
    static DynLdsAttr attr[4];
    const int dev = current_device();
    {
        hipError_t e = attr[(gt ? 1 : 0) + (waves == 8 ? 2 : 0)].ensure(reinterpret_cast<const void *>(kern), kLdsMax);
        if (e != hipSuccess) return (int)e;
    }
    // Two persistent launches must not share the GPU: each needs every one of its workgroups resident (one per CU, the CU's LDS to
    // itself), so a second one on another stream would leave both with workgroups that cannot start until the other ends -- a
    // deadlock only the bounded spins break (2 s, NaN results).  Launches of one device are therefore chained GPU-side through an
    // event: a launch on another stream waits for the previous one to finish.  (Not inside a stream capture: a captured graph
    // holds one launch chain of its own stream, and an event from outside the capture cannot be waited on there.)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    (void)hipGetLastError();
    const bool chain = cap == hipStreamCaptureStatusNone;
    hipEvent_t *const chain_event = g_chain_event;
    hipStream_t *const chain_stream = g_chain_stream;
    std::unique_lock<std::mutex> chain_lock(g_chain_mutex, std::defer_lock);
    if (chain) {
        chain_lock.lock();
        if (chain_event[dev] == nullptr) {
            if (hipEventCreateWithFlags(&chain_event[dev], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); chain_event[dev] = nullptr; }
        } else if (chain_stream[dev] != stream) {
            (void)hipStreamWaitEvent(stream, chain_event[dev], 0);
        }
    }
    for (int b0 = 0; b0 < batch; b0 += kMaxPairs) {
        const int b = batch - b0 < kMaxPairs ? batch - b0 : kMaxPairs;
        Args a;
        a.target = target + (size_t)b0 * n_pixels * 3;
        a.reference = reference + (size_t)b0 * n_pixels * 3;
        a.gt = gt ? gt + (size_t)b0 * n_pixels * 3 : nullptr;
        a.out = out + (size_t)b0 * n_pixels * 3;
        a.n_pixels = n_pixels; a.batch = b; a.slots = slots; a.pslots = pslots;
        unsigned char *base = reinterpret_cast<unsigned char *>(ws);
        a.err = reinterpret_cast<unsigned *>(base);
        a.rec = reinterpret_cast<unsigned long long *>(base + kErrBytes);
        a.slab = reinterpret_cast<double *>(base + rec_bytes(b));
        a.sq = a.slab + (size_t)b * g * 12;
#ifdef CT_RP_STAMPS
        a.stamps = reinterpret_cast<unsigned long long *>(a.sq + (size_t)b * g * kMaxWaves);
#else
        a.stamps = nullptr;
#endif
        // the caller's records: [targets of the whole batch][references of the whole batch]
        a.stats_t = stats_out ? stats_out + (size_t)b0 * CT_LAB_STATS_STRIDE : nullptr;
        a.stats_r = stats_out ? stats_out + (size_t)(batch + b0) * CT_LAB_STATS_STRIDE : nullptr;
        // zeroed by a kernel, not by hipMemsetAsync (ct_common.h: a captured memset node did not keep the stream order on replay)
        { const int zr = zero_async(base, rec_bytes(b), stream); if (zr) return zr; }
        if (ev_start && b0 == 0) (void)hipEventRecord(ev_start, stream);
        hipLaunchKernelGGL(kern, dim3(g), dim3(waves * kWave), lds, stream, a);
        CT_CHECK_LAUNCH();
        if (ev_stop && b0 + b >= batch) (void)hipEventRecord(ev_stop, stream);
        if (gt) {
            hipLaunchKernelGGL(psnr_finish_persist_kernel, dim3(b), dim3(256), 0, stream, (const double *)a.sq, g * kMaxWaves, n_pixels * 3,
                               psnr_out + (size_t)b0 * 2, (const unsigned *)a.err);
            CT_CHECK_LAUNCH();
        }
    }
    if (chain && chain_event[dev] != nullptr) {
        (void)hipEventRecord(chain_event[dev], stream);
        chain_stream[dev] = stream;
    }
    return CT_OK;
}

// sticky status of the current device: bit 0 = a bounded spin of a persistent launch gave up since the last clear
int read_status(bool clear) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_status), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (clear && v != 0) {
        const unsigned z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_status), &z, sizeof(z), 0, hipMemcpyHostToDevice);
    }
    return (int)(v & 1u);
}

template int launch<float>(const float *, const float *, const float *, float *, double *, int64_t, int, double *, void *, size_t, hipStream_t,
                           hipEvent_t, hipEvent_t);
template int launch<uint8_t>(const uint8_t *, const uint8_t *, const uint8_t *, float *, double *, int64_t, int, double *, void *, size_t,
                             hipStream_t, hipEvent_t, hipEvent_t);

}  // namespace rp
}  // namespace ct
