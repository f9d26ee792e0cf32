// conv_wino.hip -- 3x3 stride-1 "same" convolution with 32 < cin <= 64 input channels as Winograd F(2x2, 3x3) on the 16-bit matrix
// pipe, float32 operands as TWO fp16 pieces (three v_mfma_f32_16x16x32_f16 per product, float32 accumulation) like conv_ws.hip.
//
// Why: conv_ws.hip runs at the chip's power limit on real data (882 us on zeros, 1164 us on random data, same
// cycles), so what is left is fewer matrix instructions per output.  F(2x2, 3x3) needs 16 multiplications per 2x2 output tile and
// (cin, cout) pair instead of 36: 2.25x fewer MFMAs.  Error of the two-piece form against float64: 3.0e-7 of the output range,
// the direct two-piece form 1.9e-7, plain float32 1.9e-7 (tools/model_winograd_two_piece.py) -- float32-grade.
//
// Dataflow.  A persistent workgroup (one per CU, 8 waves) walks down a strip of 32 output columns, two output rows (= 16 tiles of
// 2x2) per step.  Wave w owns the transform positions 2w, 2w+1 of the 16 and keeps their transformed weights U_p = G g G^T
// (64 cout x 64 cin x two pieces = 64 VGPRs per position) in registers for its lifetime.  Per step:
//   T  every thread takes one (tile, channel pair): 4x4 patches of the raw float32 rows in the LDS ring -> V = B^T d B (32 adds
//      per channel) -> one power-of-two scale per tile row -> fp16 hi / lo -> the B-operand image of position p in LDS
//      (conflict-free 4-byte stores: a channel pair is one 32-bit word of a fragment);
//   C  a wave reads the 8 fragments of its two positions, [barrier: the image is consumed], 48 MFMAs
//      (M_p = U_p V_p: 4 cout blocks x 2 cin chunks x 3 products), unscales, writes M_p to LDS (the same buffer);
//   D  every thread takes (tile, 2 couts): Y = A^T M A (24 adds), bias, activation, skip, clamp, two 8-byte stores per cout -- a
//      wave instruction covers whole 128-byte lines of 4 output planes.
// The two input rows of the next step are requested at the start of a step and land in the ring (6 rows) at its end.
// Rounding: float32 sums per position in a fixed order; results are float32-grade, not bitwise those of conv_ws.
#include "ct_common.h"
#include "ct_conv.h"
#include "ct_split.h"
#include <type_traits>

namespace ct {

constexpr int kWnTiles = 16;           // 2x2 tiles per step = MFMA N: 32 output columns x 2 output rows
constexpr int kWnTW = 2 * kWnTiles;    // output columns of a strip
constexpr int kWnGroups = 10;          // staged input columns x0-4 .. x0+35 as aligned groups of four
constexpr int kWnRowStride = 44;       // floats per channel row of the ring; index = column - (x0 - 4) + 1: tile t reads 2t + 4 .. 2t + 7 as two aligned
                                       // 8-byte words (16-byte aligned staging with 4-byte reads instead: 925 -> 1115 us)
constexpr int kWnRing = 6;             // input rows resident: four in use, two being filled
constexpr int kWnMStride = 17;         // floats per (position, cout) row of the M image (16 tiles + 1)
constexpr int kWnThreads = 512;
constexpr size_t kWnRingBytes = (size_t)kWnRing * 64 * kWnRowStride * sizeof(float);
constexpr size_t kWnVBytes = (size_t)16 * 2 * 2 * 64 * 16;                       // [position][cin chunk][piece][lane] uint4
constexpr size_t kWnMBytes = (size_t)16 * 1024 * sizeof(float);                  // [position][cout block][i][k][tile]: the position's V bytes
constexpr size_t kWnLds = kWnRingBytes + (kWnMBytes > kWnVBytes ? kWnMBytes : kWnVBytes) + 64;

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned int wn_cvt_pk(float a, float b) {
    unsigned int r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float wn_wave_max(float v) {        // v >= 0; every lane returns the maximum
    int x = __float_as_int(v);
#define CT_DPP_MAX(ctrl, rmask) x = max(x, __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false))
    CT_DPP_MAX(0x111, 0xf); CT_DPP_MAX(0x112, 0xf); CT_DPP_MAX(0x114, 0xf); CT_DPP_MAX(0x118, 0xf);
    CT_DPP_MAX(0x142, 0xa); CT_DPP_MAX(0x143, 0xc);
#undef CT_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// ACTK: compile-time activation (0 none, 1 LeakyReLU(0.01), 2 ReLU, 3 = switch over a.act at run time), as in conv_ws.hip
// HAS_RES: a skip tensor is added after the activation; without one no skip row is requested at all (round 6: the no-skip launches
// used to read their own not-yet-written output as a dummy row -- 1.06 GB per 2-view 1080p launch for nothing)
template <int ACTK, bool HAS_RES>
__global__ __launch_bounds__(kWnThreads, 1) void conv_wino_kernel(ConvArgs a, int n_strips, int seg, int n_seg, int n_items) {
    extern __shared__ uint4 wn_smem[];
    float *ring = reinterpret_cast<float *>(wn_smem);                                         // [slot][channel][kWnRowStride]
    unsigned int *vimg = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(wn_smem) + kWnRingBytes);   // V image (32-bit words)
    // M image: position p's 4 KB are the SAME bytes as its V fragments -- only the wave that owns p reads the one and writes the
    // other, so no barrier separates them.  float index in a position: ((mb 4 + i) 4 + k) 16 + tile for cout 16 mb + 4 k + i
    // (tried and not kept, same-box A/B: swizzling k with i against the bank conflicts of the output transform's reads: no change;
    // the matrix phase in two cout halves with the first half's output transform in the shadow of the second: one more barrier
    // eats what the overlap wins, 1.215 -> 1.197 x conv_ws)
    float *mimg = reinterpret_cast<float *>(vimg);
    unsigned int *rowmax = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(wn_smem) + kWnLds - 64);   // [slot]: bits of max |x| of the row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t plane = (size_t)a.H * a.W;
    const unsigned int uplane = (unsigned int)plane;

    const int tt = tid & 15, cp = tid >> 4;          // T role: tile tt, channel pair cp
    const int dn = tid & 15, dc = tid >> 4;          // D role: tile dn, couts dc and dc + 32

#ifdef CT_WN_PROFILE
    // diagnostic build (tools/build_variant.sh, tools/prof_conv_wino.py): s_memtime ticks per phase and wave -> a.prof[block][wave][8]
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0;
#define WN_STAMP0() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt0) :: "memory")
#define WN_STAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); pt[i] += t__ - pt0; pt0 = t__; } while (0)
#else
#define WN_STAMP0() do { } while (0)
#define WN_STAMP(i) do { } while (0)
#endif
    uint4 wreg[2][4][2][2];              // [own position][cout block][cin chunk][piece]: A fragments
    int cur_grp = -1;
    const uint4 *wp16 = reinterpret_cast<const uint4 *>(a.wp);

    const int n_bands = n_items / n_strips;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const int bands_per_xcd = (n_bands + 7) >> 3;
    for (int li = wg_in_xcd; li < bands_per_xcd * n_strips; li += wgs_per_xcd) {         // XCD-aware order: see conv_ws.hip
        const int sx = li % n_strips;
        int t = (li / n_strips) * 8 + xcd;
        if (t >= n_bands) continue;
        const int sy = t % n_seg; t /= n_seg;
        const int nimg = t % a.n_images; const int grp = t / a.n_images;
        if (grp != cur_grp) {
            cur_grp = grp;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc)
                            wreg[q][mb][kc][pc] = wp16[(((((size_t)grp * 16 + (2 * wave + q)) * 4 + mb) * 2 + kc) * 2 + pc) * 64 + lane];
        }
        const int x0 = sx * kWnTW, y0 = sy * seg;
        const int rows = min(seg, a.H - y0);
        const float *in = a.in + (size_t)nimg * a.in_bstride;
        float *out = a.out + (size_t)nimg * a.out_bstride + (size_t)grp * 64 * plane;
        const float *res = HAS_RES ? a.residual + (size_t)nimg * a.res_bstride + (size_t)grp * 64 * plane : nullptr;
        const int cout_g = a.cout - grp * 64;
        const float bias0 = a.bias[grp * 64 + dc], bias1 = a.bias[grp * 64 + dc + 32];
        // stores through a buffer descriptor over this group's output planes: out-of-range offsets are dropped (conv_ws.hip)
        const __amdgpu_buffer_rsrc_t out_rs =
            __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((unsigned int)min(cout_g, 64) * uplane * 4u), 0x00020000);

        // ---- input rows: global -> registers -> ring.  Row rr (0 = image row y0 - 1) lives in slot rr % 6. ----
        // unit u < 640 of a row: channel u / 10, column group u % 10; a thread owns units tid and tid + 512 (tid < 128)
        const int u0c = tid / 10, u0g = tid - 10 * u0c;
        const int u1 = tid + 512, u1c = u1 / 10, u1g = u1 - 10 * u1c;
        const bool has_u1 = tid < 128;
        const int gx0 = x0 - 4 + 4 * u0g, gx1 = x0 - 4 + 4 * u1g;
        const bool col0 = gx0 >= 0 && gx0 < a.W && u0c < a.cin, col1 = has_u1 && gx1 >= 0 && gx1 < a.W && u1c < a.cin;
        // branch-free requests: clamped (always valid) addresses, masks when staged
        // (32-bit element offsets from the image's base: a 64-bit pointer per thread costs two registers of a full file)
        const unsigned int f0 = (unsigned int)min(u0c, a.cin - 1) * uplane + (unsigned int)min(max(gx0, 0), a.W - 4);
        const unsigned int f1 = has_u1 ? (unsigned int)min(u1c, a.cin - 1) * uplane + (unsigned int)min(max(gx1, 0), a.W - 4) : f0;
        float *const ring0 = ring + u0c * kWnRowStride + 4 * u0g + 1;
        float *const ring1 = ring + (has_u1 ? u1c : u0c) * kWnRowStride + 4 * (has_u1 ? u1g : u0g) + 1;
        auto fetch = [&](int rr, float4 &v0, float4 &v1) {
            const unsigned int yo = (unsigned int)(min(max(y0 - 1 + rr, 0), a.H - 1) * a.W);
            v0 = *reinterpret_cast<const float4 *>(in + (f0 + yo));
            v1 = *reinterpret_cast<const float4 *>(in + (f1 + yo));
        };
        // rows are staged in pairs (rr even, rr + 1): one maximum per pair in rowmax[slot / 2]
        auto stage = [&](int rr, int slot, float4 v0, float4 v1, float &m) {        // masks, maximum, registers -> ring slot
            const int y = y0 - 1 + rr;
            const bool yok = y >= 0 && y < a.H;
            if (!(yok && col0)) v0 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(yok && col1)) v1 = make_float4(0.f, 0.f, 0.f, 0.f);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w))));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w))));
            float *d0 = ring0 + slot * 64 * kWnRowStride;
            d0[0] = v0.x; d0[1] = v0.y; d0[2] = v0.z; d0[3] = v0.w;
            if (has_u1) {
                float *d1 = ring1 + slot * 64 * kWnRowStride;
                d1[0] = v1.x; d1[1] = v1.y; d1[2] = v1.z; d1[3] = v1.w;
            }
        };
        // prologue: rows rr = 0..3 -> slots 0..3
        __syncthreads();                                  // the previous item's last step is done with ring and images
        if (tid < kWnRing / 2) rowmax[tid] = 0u;
        __syncthreads();
        {
            float4 p0[4], p1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fetch(i, p0[i], p1[i]);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float m = 0.f;
                stage(2 * pr, 2 * pr, p0[2 * pr], p1[2 * pr], m);
                stage(2 * pr + 1, 2 * pr + 1, p0[2 * pr + 1], p1[2 * pr + 1], m);
                m = wn_wave_max(m);
                if (lane == 0) atomicMax(rowmax + pr, __float_as_uint(m));
            }
        }
        __syncthreads();
        // rows travel one step ahead of their staging: requested in the matrix phase of step s - 1 (here: rows 4, 5), staged in the
        // matrix phase of step s -- vector and LDS work in the shadow of the MFMAs -- and first read by the transform of step s + 1
        float4 na0, na1, nb0, nb1;
        fetch(4, na0, na1);
        fetch(5, nb0, nb1);

        // skip rows (D role): columns x0 + 2 dn, couts dc, dc + 32; clamped addresses
        const int ox = x0 + 2 * dn;
        const float *rb = res;
        const unsigned int rb0 = (unsigned int)min(dc, cout_g - 1) * uplane + (unsigned int)min(ox, a.W - 2);
        const unsigned int rb1 = (unsigned int)min(dc + 32, cout_g - 1) * uplane + (unsigned int)min(ox, a.W - 2);
        const int dco[2] = {dc, dc + 32};
        // M image read offsets of the two couts (floats inside a position)
        int mofs[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) mofs[j] = ((((dco[j] >> 4) * 4 + (dco[j] & 3)) * 4 + ((dco[j] >> 2) & 3)) * 16) + dn;
        // V image word of this thread: cin chunk cp / 16, lane 16 ((cp % 16) / 4) + tt, word cp % 4
        unsigned int *const vb = vimg + (((cp >> 4) * 2) * 64 + 16 * ((cp & 15) >> 2) + tt) * 4 + (cp & 3);
        const float *const tring = ring + (2 * cp) * kWnRowStride + 2 * tt + 4;

        const int steps = (rows + 1) >> 1;
        // one step; SB = (2 s) % 6 at compile time: the slot of row 2 s + i is (SB + i) % 6
        auto step = [&](int s, auto sb_c) {
            constexpr int SB = decltype(sb_c)::value;
            WN_STAMP0();
            const int oy = y0 + 2 * s;
            // the tile row's scale: 2^ex * max |x| in [2^9, 2^10): |V| <= 4 max |x| stays below 2^12 like conv_ws's staged rows
            const float mx = fmaxf(__uint_as_float(rowmax[SB / 2]), __uint_as_float(rowmax[((SB + 2) % kWnRing) / 2]));
            if (tid == 0) rowmax[((SB + 4) % kWnRing) / 2] = 0u;    // the pair staged in this step's matrix phase (a barrier from here); last read a step ago
            const int fld = (int)(__float_as_uint(mx) >> 23);
            int ex = (fld == 0 || fld == 255) ? 0 : 136 - fld;
            ex = __builtin_amdgcn_readfirstlane(min(max(ex, -100), 100));
            const float scale = __uint_as_float((unsigned int)(127 + ex) << 23);
            WN_STAMP(0);
            // ---------------- T: (tile tt, channels 2 cp, 2 cp + 1) -> V -> fp16 pieces -> B image ----------------
            {
                float v[2][4][4];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float d[4][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float *rp = tring + (((SB + i) % kWnRing) * 64 + e) * kWnRowStride;
                        const float2 lo = *reinterpret_cast<const float2 *>(rp), hi = *reinterpret_cast<const float2 *>(rp + 2);
                        d[i][0] = lo.x; d[i][1] = lo.y; d[i][2] = hi.x; d[i][3] = hi.y;
                    }
                    float w[4][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                       // B^T d: rows
                        w[0][j] = d[0][j] - d[2][j]; w[1][j] = d[1][j] + d[2][j]; w[2][j] = d[2][j] - d[1][j]; w[3][j] = d[1][j] - d[3][j];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {                       // (B^T d) B: columns
                        v[e][i][0] = w[i][0] - w[i][2]; v[e][i][1] = w[i][1] + w[i][2]; v[e][i][2] = w[i][2] - w[i][1]; v[e][i][3] = w[i][1] - w[i][3];
                    }
                }
                // hi = fp16(v 2^ex), lo = fp16(v 2^ex - hi): the scale rides in the converting fma (v_fma_mixlo / mixhi_f16 write one
                // half of the word each): four instructions per channel pair and position instead of six.  Issued in four sweeps over
                // the positions: a half-word write followed at once by its other half or by its reader costs wait states
                unsigned int hw[16], lw[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hw[p]) : "v"(v[0][p >> 2][p & 3]), "v"(scale));
#pragma unroll
                for (int p = 0; p < 16; ++p) asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hw[p]) : "v"(v[1][p >> 2][p & 3]), "v"(scale));
#pragma unroll
                for (int p = 0; p < 16; ++p)
                    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lw[p]) : "v"(v[0][p >> 2][p & 3]), "v"(scale), "v"(hw[p]));
#pragma unroll
                for (int p = 0; p < 16; ++p)
                    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lw[p]) : "v"(v[1][p >> 2][p & 3]), "v"(scale), "v"(hw[p]));
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    vb[p * (2 * 2 * 64 * 4)] = hw[p];                    // [p][kc][piece][lane][4 words]
                    vb[p * (2 * 2 * 64 * 4) + 64 * 4] = lw[p];
                }
            }
            WN_STAMP(1);
            // the skip rows of this step's outputs, requested after the transform (its live values and the 128 weight registers fill the file)
            float2 rq[2][2];
            if constexpr (HAS_RES) {
                const unsigned int yy0 = (unsigned int)(min(oy, a.H - 1) * a.W), yy1 = (unsigned int)(min(oy + 1, a.H - 1) * a.W);
                rq[0][0] = *reinterpret_cast<const float2 *>(rb + (rb0 + yy0)); rq[0][1] = *reinterpret_cast<const float2 *>(rb + (rb0 + yy1));
                rq[1][0] = *reinterpret_cast<const float2 *>(rb + (rb1 + yy0)); rq[1][1] = *reinterpret_cast<const float2 *>(rb + (rb1 + yy1));
            }
            __syncthreads();
            WN_STAMP(2);
            // ---------------- C: fragments of the own positions, MFMAs, M image (in place of the fragments) ----------------
            {
                // The matrix phase: six groups of eight independent MFMAs (one per accumulator); between the groups -- fenced, so that
                // the order survives the scheduler -- the vector work that depends on nothing here: rows 2s + 4, 2s + 5 (requested a
                // step ago) -> ring, their maximum, then the requests for rows 2s + 6, 2s + 7 into the same registers.  A group
                // issues in 32 cycles and keeps the matrix pipe busy for 128: the filler runs in its shadow.
                f32x4w acc[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) acc[q][mb] = f32x4w{0.f, 0.f, 0.f, 0.f};
                uint4 bq0[2][2], bq1[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) bq0[q][pc] = reinterpret_cast<const uint4 *>(vimg)[(((2 * wave + q) * 2 + 0) * 2 + pc) * 64 + lane];
                WN_STAMP(3);
                auto group = [&](int kc, int pr, const uint4 (&bq)[2][2]) {      // small terms first: lo x hi, hi x lo, hi x hi
                    const int pw = pr == 0 ? 1 : 0, pv = pr == 1 ? 1 : 0;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int mb = 0; mb < 4; ++mb)
                            acc[q][mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, wreg[q][mb][kc][pw]),
                                                                                __builtin_bit_cast(h16x8, bq[q][pv]), acc[q][mb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                };
                float m = 0.f;
                __builtin_amdgcn_sched_barrier(0);
                group(0, 0, bq0);
                stage(2 * s + 4, (SB + 4) % kWnRing, na0, na1, m);
                __builtin_amdgcn_sched_barrier(0);
                group(0, 1, bq0);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) bq1[q][pc] = reinterpret_cast<const uint4 *>(vimg)[(((2 * wave + q) * 2 + 1) * 2 + pc) * 64 + lane];
                stage(2 * s + 5, (SB + 5) % kWnRing, nb0, nb1, m);
                __builtin_amdgcn_sched_barrier(0);
                group(0, 2, bq0);
                m = wn_wave_max(m);
                if (lane == 0) atomicMax(rowmax + ((SB + 4) % kWnRing) / 2, __float_as_uint(m));
                __builtin_amdgcn_sched_barrier(0);
                group(1, 0, bq1);
                fetch(2 * s + 6, na0, na1);
                __builtin_amdgcn_sched_barrier(0);
                group(1, 1, bq1);
                fetch(2 * s + 7, nb0, nb1);
                __builtin_amdgcn_sched_barrier(0);
                group(1, 2, bq1);
                const float unscale = __uint_as_float((unsigned int)(127 + min(max(-ex - a.w_exp, -126), 127)) << 23);
                // lane holds M[cout 16 mb + 4 (lane / 16) + i][tile lane % 16]
                float *mw = mimg + (2 * wave) * 1024 + (lane >> 4) * 16 + (lane & 15);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                        for (int i = 0; i < 4; ++i) mw[q * 1024 + (mb * 4 + i) * 64] = acc[q][mb][i] * unscale;
            }
            WN_STAMP(4);
            __syncthreads();
            WN_STAMP(5);
            // ---------------- D: (tile dn, couts dc, dc + 32): Y = A^T M A, epilogue, stores ----------------
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int co = dco[j];
                float m[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) m[p] = mimg[p * 1024 + mofs[j]];
                float t0[4], t1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { t0[c] = (m[c] + m[4 + c]) + m[8 + c]; t1[c] = (m[4 + c] - m[8 + c]) - m[12 + c]; }
                float y[4] = {(t0[0] + t0[1]) + t0[2], (t0[1] - t0[2]) - t0[3], (t1[0] + t1[1]) + t1[2], (t1[1] - t1[2]) - t1[3]};
                const float bv = j ? bias1 : bias0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    y[i] += bv;
                    if constexpr (ACTK == 1) y[i] = fmaxf(y[i], 0.01f * y[i]);
                    else if constexpr (ACTK == 2) y[i] = fmaxf(y[i], 0.f);
                    else if constexpr (ACTK == 3) y[i] = split_act<true>(y[i], a.act);
                }
                if constexpr (HAS_RES) { y[0] += rq[j][0].x; y[1] += rq[j][0].y; y[2] += rq[j][1].x; y[3] += rq[j][1].y; }
                if (a.clamp) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = fminf(fmaxf(y[i], 0.f), 1.f);
                }
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const bool okc = co < cout_g && ox < a.W;
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const bool ok = okc && (2 * s + r) < rows;
                    const unsigned int off = ok ? ((unsigned int)co * uplane + (unsigned int)((oy + r) * a.W + ox)) * 4u : 0xffffffffu;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(y[2 * r]), __float_as_uint(y[2 * r + 1])}, out_rs, (int)off, 0, 0);
                }
            }
            WN_STAMP(6);
            __syncthreads();                              // M image consumed (the next transform overwrites it)
            WN_STAMP(7);
        };
#pragma unroll 1
        for (int s = 0; s < steps; s += 3) {
            step(s, std::integral_constant<int, 0>());
            if (s + 1 < steps) step(s + 1, std::integral_constant<int, 2>());
            if (s + 2 < steps) step(s + 2, std::integral_constant<int, 4>());
        }
    }
#ifdef CT_WN_PROFILE
    if (lane == 0 && a.prof) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.prof[((size_t)blockIdx.x * 8 + wave) * 8 + i] = pt[i];
    }
#endif
}

#ifdef CT_WN_PROFILE
static unsigned long long *g_wn_prof = nullptr;
#endif

// which Winograd kernel ct_conv3x3_wino16_f32 launches: 0 = conv_wino4.hip (four waves of 512 registers, pipelined; round 6, the
// default), 1 = the eight-wave kernel of this file (round 5).  Process-wide, atomic; env CT_HIP_WINO_FORM presets it.
static std::atomic<int> g_wino_form{[] { const char *e = getenv("CT_HIP_WINO_FORM"); return (e && e[0] == '1') ? 1 : 0; }()};

// 1 = not this kernel's geometry
int conv_wino(const ConvArgs &a, int N, hipStream_t s) {
    if (a.in2 != nullptr || a.cin <= 32 || a.cin > 64 || !a.f16 || (a.W & 3)) return 1;
    if ((unsigned long long)a.H * (unsigned long long)a.W * 64ull * 4ull >= (1ull << 32)) return 1;     // 32-bit byte offsets over 64 planes
    const int n_strips = (a.W + kWnTW - 1) / kWnTW;
    const long long imgs = (long long)N * a.groups;
    int n_seg = 1, seg = a.H;
    long long best = -1;
    for (int ns = 1; ns <= (a.H + 15) / 16; ++ns) {          // even row segments: the split with the fewest steps of the busiest workgroup
        int sg = (a.H + ns - 1) / ns;
        sg += sg & 1;
        const int ns_eff = (a.H + sg - 1) / sg;
        const long long bands_per_xcd = (imgs * ns_eff + 7) / 8;
        const long long rounds = (bands_per_xcd * n_strips + 31) / 32;
        const long long cost = rounds * (sg / 2 + 2);
        if (best < 0 || cost < best) { best = cost; n_seg = ns_eff; seg = sg; }
    }
    const long long n_items = imgs * n_seg * n_strips;
    if (n_items > 0x7fffffffLL) return CT_E_BADARG;
    typedef void (*kern_t)(ConvArgs, int, int, int, int);
    static const kern_t kerns[8] = {conv_wino_kernel<0, false>, conv_wino_kernel<1, false>, conv_wino_kernel<2, false>, conv_wino_kernel<3, false>,
                                    conv_wino_kernel<0, true>,  conv_wino_kernel<1, true>,  conv_wino_kernel<2, true>,  conv_wino_kernel<3, true>};
    const int actk = ((a.act >= 0 && a.act <= 2) ? a.act : 3) + (a.residual ? 4 : 0);
    static DynLdsAttr attr[8];
    if (attr[actk].ensure(reinterpret_cast<const void *>(kerns[actk]), kWnLds) != hipSuccess) return CT_E_BADARG;
    ConvArgs b = a;
    b.n_images = N;
#ifdef CT_WN_PROFILE
    b.prof = g_wn_prof;
#endif
    hipLaunchKernelGGL(kerns[actk], dim3(256), dim3(kWnThreads), kWnLds, s, b, n_strips, seg, n_seg, (int)n_items);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // namespace ct

extern "C" {

// Winograd F(2x2, 3x3) form of ct_conv3x3_ws16_f32 (same arguments; 3x3, stride 1, padding 1, 32 < cin <= 64, w % 4 == 0).
// wq16: fp16 bit patterns [ceil(cout/64)][16 positions][4 cout blocks][2 cin chunks][piece hi/lo][64 lanes][8] of
// (G g G^T) * 2^w_exp in the A-fragment order of v_mfma_f32_16x16x32_f16 (ct_hip.pack_conv_weight_wino16).
int ct_conv3x3_wino16_f32(const float *in, const void *wq16, int w_exp, const float *bias, const float *residual, float *out, int n, int cin,
                          int cout, int h, int w, long long in_bstride, long long out_bstride, long long res_bstride, int act, int clamp,
                          void *stream) {
    if (!in || !wq16 || !bias || !out || n < 0 || cin <= 32 || cin > 64 || cout < 1 || h < 0 || w < 0 || act < 0 || act > 5) return CT_E_BADARG;
    if ((w % 4) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(wq16) & 15) ||
        (in_bstride % 4) || (out_bstride % 4) || (residual && ((reinterpret_cast<uintptr_t>(residual) & 15) || (res_bstride % 4))))
        return CT_E_ALIGN;
    if (w_exp < -100 || w_exp > 100) return CT_E_BADARG;
    if (n == 0 || h == 0 || w == 0) return CT_OK;
    ct::ConvArgs a;
    a.in = in; a.in2 = nullptr; a.cin1 = cin; a.in2_bstride = 0;
    a.wp = reinterpret_cast<const float *>(wq16); a.bias = bias; a.residual = residual; a.out = out;
    a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.in_bstride = in_bstride; a.out_bstride = out_bstride; a.res_bstride = res_bstride;
    a.act = act; a.clamp = clamp; a.groups = (cout + 63) / 64; a.prof = nullptr;
    a.f16 = 1; a.w_exp = w_exp;
    int rc = 1;
    if (ct::g_wino_form.load(std::memory_order_relaxed) == 0) rc = ct::conv_wino4(a, n, (hipStream_t)stream);   // 1: geometry beyond its 30-bit offsets
    if (rc == 1) rc = ct::conv_wino(a, n, (hipStream_t)stream);
    return rc == 1 ? CT_E_BADARG : rc;
}

int ct_set_conv_wino_form(int form) {
    if (form != 0 && form != 1) return CT_E_BADARG;
    ct::g_wino_form.store(form, std::memory_order_relaxed);
    return CT_OK;
}

#ifdef CT_WN_PROFILE
void ct_conv_wino_set_prof(unsigned long long *p) { ct::g_wn_prof = p; }
#endif

}  // extern "C"
