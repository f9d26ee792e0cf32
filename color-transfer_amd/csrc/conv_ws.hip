// conv_ws.hip -- 3x3 stride-1 "same" convolution with 32 < cin <= 64 input channels on the 16-bit matrix pipe, WEIGHTS STATIONARY.
//
// Same idea as conv_split.hip -- float32 operands as short-mantissa pieces, float32 accumulation -- in two forms: three bf16
// pieces / six v_mfma_f32_32x32x16_bf16 per 16-channel product (conv_split's arithmetic), or TWO fp16 pieces / three
// v_mfma_f32_32x32x16_f16 with power-of-two scales per staged input row and per layer (the default, ct_conv3x3_ws16_f32; see the
// comment at the kernel).  Different dataflow: conv_split shares one input tile between the four waves of a workgroup and
// streams the weights of every tap through an LDS ring -- one barrier per tap, a conversion phase between the stages, two
// workgroups per CU competing for the matrix pipe: 47 % MFMA-busy, 22 % of a wave's time in the per-tap barriers, 13 % in the
// conversion phase (tools/prof_conv_split.py).  Here the contraction is split ACROSS the waves: wave (c, m) of the eight owns
// input channels [16 c, 16 c + 16) and output channels [32 m, 32 m + 32) and keeps the weights of all nine taps of that block
// in registers (18 or 27 fragments = 72 / 108 VGPRs) for the lifetime of a persistent workgroup (one per CU; the two waves of a
// chunk share a SIMD).  A workgroup walks down a 32-pixel wide strip of a row segment; per output row (a "step") a wave runs
//   X: the reduce of the PREVIOUS row (the four chunk sums of its 8 output channels in a fixed order, bias, activation, ResB
//      skip, clamp, one 16-byte store per lane) issued between the MFMAs of this row (9 taps x 3 or 6), whose B fragments are
//      conflict-free 16-byte LDS reads a tap row ahead; then its 32 x 32 partial sums -> LDS;
//   Y: requests (skip row of the next step, input row four steps ahead, branch-free), the maximum of the row staged next (fp16
//      form), the float32 -> pieces split of its half of the row requested two steps ago -> the chunk's four-row LDS ring;
// the two waves of a SIMD run X and Y in opposite order and ONE barrier ends the step.  No weight traffic after the prologue,
// no per-tap barriers, a value is split once, by one wave; strips are ordered so that neighbours meet in the same XCD's L2.
// The measured steps that led here (one wave per SIMD: 39 % MFMA-busy; power limit of the bf16 form) are in DESIGN.md 4.4.
// Rounding: float32 sums per chunk, then ((p0 + p1) + p2) + p3 + bias -- float32-grade like conv_split (not bitwise equal).
#include "ct_common.h"
#include "ct_conv.h"
#include "ct_split.h"
#include <type_traits>

namespace ct {

constexpr int kWsTW = 32;                    // output columns of a strip (= MFMA N)
constexpr int kWsCols = 40;                  // staged columns x0-4 .. x0+35 (ten aligned groups of four)
constexpr int kWsSlotMax = 3 * 2 * kWsCols;  // 16-byte entries of one input row of one chunk: [piece][k-half][column] (3 bf16 or 2 fp16 pieces)
constexpr int kWsRing = 4;                   // input rows resident per wave (three in use, one being filled)
constexpr int kWsPS = 36;                    // floats per row of a partial-sum tile (16-byte aligned; 40 / 44 and s_setprio around X: no change)
constexpr int kWsChunks = 4;                 // 16-channel chunks = SIMDs
constexpr int kWsWaves = 8;                  // (chunk, 32-channel half of the outputs)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// max over the wave of a non-negative float (DPP row shifts + row broadcasts; every lane returns the maximum)
__device__ __forceinline__ float wave_max_nonneg(float v) {
    int x = __float_as_int(v);      // non-negative floats order like their bit patterns
#define CT_DPP_MAX(ctrl, rmask) x = max(x, __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false))
    CT_DPP_MAX(0x111, 0xf);         // row_shr:1
    CT_DPP_MAX(0x112, 0xf);         // row_shr:2
    CT_DPP_MAX(0x114, 0xf);         // row_shr:4
    CT_DPP_MAX(0x118, 0xf);         // row_shr:8   -> lane 15 of every row holds the row maximum
    CT_DPP_MAX(0x142, 0xa);         // row_bcast:15 into rows 1 and 3
    CT_DPP_MAX(0x143, 0xc);         // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave maximum
#undef CT_DPP_MAX
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// two float32 -> packed fp16 (round to nearest even), opaque to the compiler (it otherwise recomputes each half with
// v_fma_mixlo_f16 when the halves are converted back for the residual)
__device__ __forceinline__ f16x2 cvt_pk_f16(float a, float b) {
    unsigned int r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return __builtin_bit_cast(f16x2, r);
}

// F16: the operands are split into TWO fp16 pieces (11 + 11 mantissa bits) and a product is three MFMAs
// (a_hi b_hi + a_hi b_lo + a_lo b_hi; what is dropped is 2^-22 relative) instead of six bf16 ones: half the matrix work, which
// is what counts on real data -- the bf16 form runs into the chip's power limit (1.65 ms per 1080p ResB conv on random
// data against 1.23 ms on zeros).  fp16 has 5 exponent bits, so every staged input row carries a power-of-two scale of its
// own (its maximum is brought to [2^11, 2^12); smaller values keep an absolute error below 2^-25 of the row maximum), the
// weights one per layer (host), and the accumulators are rescaled when the scale changes from one row to the next.
// ACTK: the activation is a compile-time choice for the common cases (0 none, 1 LeakyReLU(0.01), 2 ReLU; 3 = switch over a.act
// at run time): the generic switch costs ~10 scalar branches per output value, between the MFMAs of phase X
template <int ACTK, bool F16>
__global__ __launch_bounds__(512, 1) void conv_ws_kernel(ConvArgs a, int n_strips, int seg, int n_seg, int n_items) {
    constexpr int PIECES = F16 ? 2 : 3;
    constexpr int kWsSlot = PIECES * 2 * kWsCols;
    extern __shared__ uint4 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, nl = lane & 31, hl = lane >> 5;
    // the wave number in a scalar register: the X / Y order below is then a UNIFORM branch around two separate row loops.  As a
    // lane-dependent select both orders shared one loop body, and the compiler -- which must assume that either side's loads can
    // be in flight when the other side starts -- drained the vector memory counter at the head of every step: the wave that
    // starts with X waited for the requests it had issued just before the barrier (a full memory latency per row step),
    // the one that starts with Y for the store of the previous X (round 5, tools/prof_conv_ws.py)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = wave & 3, mt = wave >> 2;         // waves c and c + 4 run on SIMD c
    uint4 *ring = smem16 + chunk * (kWsRing * kWsSlot);                                   // this chunk's input rows
    float *part = reinterpret_cast<float *>(smem16 + kWsChunks * kWsRing * kWsSlot);      // [2][chunk][64][kWsPS]
    float *rowmax = part + 2 * kWsChunks * 64 * kWsPS;                                    // F16: [chunk][half][4]: max |x| of a row half
    int *slot_exp = reinterpret_cast<int *>(rowmax + kWsChunks * 2 * 4);                  // F16: [chunk][ring slot]: scale exponent
    const size_t plane = (size_t)a.H * a.W;
    const unsigned int uplane = (unsigned int)plane;
    const int n_chunks = (a.cin + 15) / 16;
    const bool has_chunk = chunk < n_chunks;
    const int c_base = chunk * 16;

    // staging role: unit u = lane + 40 mt (lanes 0..39) -> channel pair cp = u / 10, column group grp = u % 10
    const int u_id = lane + 40 * mt;
    const int u_cp = u_id / 10, u_grp = u_id - 10 * u_cp;
    const bool u_lane = lane < 40;
    const bool u_on = u_lane && (c_base + 2 * u_cp < a.cin);

#ifdef CT_CONV_PROFILE
    // diagnostic build (make prof, tools/prof_conv_ws.py): s_memtime ticks per phase, per wave -> a.prof[block][wave][8]
    // 0 X (reduce of the previous row, MFMAs, partial sums), 1 the step's barrier, 2 Y (requests, row maximum, staging), 4 everything else
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0;
#define WS_STAMP0() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt0) :: "memory")
#define WS_STAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); pt[i] += t__ - pt0; pt0 = t__; } while (0)
    WS_STAMP0();
#else
#define WS_STAMP(i) do { } while (0)
#endif
    uint4 wreg[9][PIECES];
    int cur_grp = -1;
    const uint4 *wp16 = reinterpret_cast<const uint4 *>(a.wp);

    // Work items = (band, strip): a band is one row segment of one image and output group.  The halo columns of a strip are
    // the edge columns of its neighbours (a strip row is exactly one 128-byte line per channel), so neighbouring strips must
    // meet in the SAME L2: workgroup b runs on XCD b % 8 (dispatch is round robin; an assumption that only costs speed if
    // wrong), XCD p owns the bands p, p + 8, .. and its 32 workgroups sweep them side by side, strip after strip.  Without
    // this every strip row pulled three lines through the fabric for one line of new data (measured: 38 M L2 misses per
    // 1080p convolution, the kernel without its MFMAs still took 60 % of the full time).
    const int n_bands = n_items / n_strips;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const int bands_per_xcd = (n_bands + 7) >> 3;
    for (int li = wg_in_xcd; li < bands_per_xcd * n_strips; li += wgs_per_xcd) {
        const int sx = li % n_strips;
        int t = (li / n_strips) * 8 + xcd;                 // band
        if (t >= n_bands) continue;                         // uniform per workgroup
        const int sy = t % n_seg; t /= n_seg;
        const int nimg = t % a.n_images; const int grp = t / a.n_images;
        if (grp != cur_grp) {
            cur_grp = grp;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int p = 0; p < PIECES; ++p)
                    wreg[tap][p] = has_chunk ? wp16[((((size_t)grp * n_chunks + chunk) * 9 + tap) * PIECES + p) * 128 + mt * 64 + lane]
                                             : make_uint4(0u, 0u, 0u, 0u);
        }
        const int x0 = sx * kWsTW, y0 = sy * seg;
        const int rows = min(seg, a.H - y0);
        float *out = a.out + (size_t)nimg * a.out_bstride + (size_t)grp * 64 * plane;
        const float *res = a.residual ? a.residual + (size_t)nimg * a.res_bstride + (size_t)grp * 64 * plane : nullptr;
        const int cout_g = a.cout - grp * 64;
        const int e_row = lane >> 3, e_g = lane & 7;      // epilogue role: output channel 8 wave + e_row, columns 4 e_g ..
        const int e_co = 8 * wave + e_row;
        const float bias_l = a.bias[grp * 64 + e_co];     // zero padded to 64
        const bool e_ok = (x0 + 4 * e_g) < a.W && e_co < cout_g;
        // output rows leave through a buffer descriptor over this wave's eight output planes: a store that must not happen
        // (columns past the image, rows of padding steps, channels past cout -- zero records) gets an out-of-range offset and is
        // dropped by the hardware.  No branch around the store: the compiler's count of the vector memory operations in flight
        // stays exact (a store under a branch made every later wait for an input row wait for the store as well)
        const int e_planes = min(max(cout_g - 8 * wave, 0), 8);
        const __amdgpu_buffer_rsrc_t out_rs =
            __builtin_amdgcn_make_buffer_rsrc(out + (size_t)(8 * wave) * plane, 0, (int)((unsigned int)e_planes * uplane * 4u), 0x00020000);

        // input rows travel through three register sets: the row staged at the end of step r was requested two steps earlier
        // (one step of distance left the HBM latency exposed: the kernel without its MFMAs took 60 % of the full time)
        float4 qa[3], qb[3], rq[3];
        auto fetch_row = [&](int y, float4 &la, float4 &lb) {   // input row y of this chunk, this wave's half of the units -> registers
            // branch-free on purpose: every lane loads from a clamped (always valid) address and zero_row() blanks what lies
            // outside the image afterwards -- loads under a branch make the compiler wait for ALL outstanding loads
            // (vmcnt(0)) at the first use, which would put the request of two steps ahead on the critical path
            const int yc = min(max(y, 0), a.H - 1);
            const int gx = min(max(x0 - 4 + 4 * u_grp, 0), a.W - 4);
            const int c0 = min(c_base + 2 * u_cp, a.cin - 1), c1 = min(c_base + 2 * u_cp + 1, a.cin - 1);
            const float *base = a.in + (size_t)nimg * a.in_bstride + (unsigned int)(yc * a.W + gx);
            la = *reinterpret_cast<const float4 *>(base + (unsigned int)c0 * uplane);
            lb = *reinterpret_cast<const float4 *>(base + (unsigned int)c1 * uplane);
        };
        auto row_mask = [&](int y, float &ma, float &mb) {      // 1 / 0: the unit's pixels and channels exist
            const int gx = x0 - 4 + 4 * u_grp;
#ifdef CT_WS_ABL_NOLOAD
            const bool ok = false;
#else
            const bool ok = u_on && y >= 0 && y < a.H && gx >= 0 && gx < a.W;
#endif
            ma = ok ? 1.f : 0.f;
            mb = (ok && c_base + 2 * u_cp + 1 < a.cin) ? 1.f : 0.f;
        };
        // interior strips / rows / full chunks need no masks at all (uniform per workgroup and row)
        const bool strip_inside = x0 >= 4 && x0 + 36 <= a.W && c_base + 16 <= a.cin;
        // F16: max |x| of this wave's half of input row y (zeros outside the image) -> rowmax[chunk][mt][idx]
        // (MFMA and VALU cycles add up on this chip -- DESIGN.md 4.5 -- so every vector instruction of the staging phase costs the
        // matrix pipe its slot: interior rows of interior strips, the common case, take a mask-free form; round 5)
        auto note_row_max = [&](int idx, int y, const float4 &la, const float4 &lb) {
            if constexpr (F16) {
                float m;
                if (strip_inside && y >= 0 && y < a.H) {            // uniform per workgroup and row
                    m = fmaxf(fmaxf(fmaxf(fabsf(la.x), fabsf(la.y)), fabsf(la.z)), fmaxf(fmaxf(fabsf(la.w), fabsf(lb.x)), fabsf(lb.y)));
                    m = fmaxf(fmaxf(m, fabsf(lb.z)), fabsf(lb.w));
                } else {
                    float ma = 1.f, mb = 1.f;
                    row_mask(y, ma, mb);
                    m = 0.f;
                    if (ma != 0.f) m = fmaxf(fmaxf(fabsf(la.x), fabsf(la.y)), fmaxf(fabsf(la.z), fabsf(la.w)));
                    if (mb != 0.f) m = fmaxf(m, fmaxf(fmaxf(fabsf(lb.x), fabsf(lb.y)), fmaxf(fabsf(lb.z), fabsf(lb.w))));
                }
                m = wave_max_nonneg(m);
                if (lane == 0) rowmax[(chunk * 2 + mt) * 4 + idx] = m;
            }
        };
        auto stage_row = [&](int slot, int y, const float4 &la, const float4 &lb, int max_idx) {   // registers -> pieces -> LDS slot
            const bool plain = strip_inside && y >= 0 && y < a.H;  // uniform: no masks at all
            float ma = 1.f, mb = 1.f;
            if (!plain) row_mask(y, ma, mb);
            float scale = 1.f;
            if constexpr (F16) {
                // the row's scale: both halves' maxima were written at least one barrier ago.  2^ex * max in [2^11, 2^12).
                const float mx = fmaxf(rowmax[(chunk * 2 + 0) * 4 + max_idx], rowmax[(chunk * 2 + 1) * 4 + max_idx]);
                const int fld = (int)(__float_as_uint(mx) >> 23);                  // biased exponent (mx >= 0; inf / nan: 255)
                int ex = (fld == 0 || fld == 255) ? 0 : 138 - fld;                   // 12 - (floor(log2 mx) + 1)
                ex = __builtin_amdgcn_readfirstlane(min(max(ex, -100), 100));
                scale = __uint_as_float((unsigned int)(127 + ex) << 23);
                if (lane == 0) slot_exp[chunk * kWsRing + slot] = ex;                // both waves of the chunk write the same value
            }
#ifdef CT_WS_ABL_NOSTAGE
            if (la.x == 123.456f) {
#else
            if (u_lane) {
#endif
                const int kh2 = u_cp >> 2, wsel = u_cp & 3;
                unsigned int *d = reinterpret_cast<unsigned int *>(ring + slot * kWsSlot) + ((kh2 * kWsCols + 4 * u_grp) * 4 + wsel);
                auto pieces = [&](const float (&xa)[4], const float (&xb)[4]) {
#pragma unroll
                    for (int px = 0; px < 4; ++px) {
                        if constexpr (F16) {
                            const float sa = xa[px] * scale, sb = xb[px] * scale;
                            const f16x2 h = cvt_pk_f16(sa, sb);
                            // lo = s - float(hi), exactly, in ONE instruction per value: v_fma_mix_f32 reads the fp16 half directly
                            // (convert + subtract were two; the compiler does not form it across the opaque v_cvt_pk_f16_f32)
                            float da, db;
                            const unsigned int hu = __builtin_bit_cast(unsigned int, h);
                            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(da) : "v"(hu), "v"(sa));
                            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(db) : "v"(hu), "v"(sb));
                            const f16x2 l = cvt_pk_f16(da, db);
                            d[px * 4] = hu;
                            d[(2 * kWsCols + px) * 4] = __builtin_bit_cast(unsigned int, l);
                        } else {
                            unsigned int hw, mw, lw;
                            split3x2(xa[px], xb[px], hw, mw, lw);
                            d[px * 4] = hw;
                            d[(2 * kWsCols + px) * 4] = mw;
                            d[(4 * kWsCols + px) * 4] = lw;
                        }
                    }
                };
                if (plain) {
                    const float xa[4] = {la.x, la.y, la.z, la.w}, xb[4] = {lb.x, lb.y, lb.z, lb.w};
                    pieces(xa, xb);
                } else {
                    const float xa[4] = {ma != 0.f ? la.x : 0.f, ma != 0.f ? la.y : 0.f, ma != 0.f ? la.z : 0.f, ma != 0.f ? la.w : 0.f};
                    const float xb[4] = {mb != 0.f ? lb.x : 0.f, mb != 0.f ? lb.y : 0.f, mb != 0.f ? lb.z : 0.f, mb != 0.f ? lb.w : 0.f};
                    pieces(xa, xb);
                }
            }
        };

        auto fetch_skip = [&](int r, float4 &rv) {             // ResB skip of output row y0 + r: branch-free like the row fetch
            const int yr = min(max(y0 + r, 0), a.H - 1);        // (without a skip tensor: one cached line of the bias, ignored -- the
            const unsigned int so = (unsigned int)min(e_co, cout_g - 1) * uplane + (unsigned int)(yr * a.W + min(x0 + 4 * e_g, a.W - 4));
            rv = *reinterpret_cast<const float4 *>(res ? res + so : a.bias);   // request count per step stays what the exact waits assume; round 5 read the output row here: 1.06 GB per launch)
        };
        // prologue: input rows y0-1, y0, y0+1 -> slots 0, 1, 2 (the barrier publishes them to the chunk's other wave); rows
        // y0+2 and y0+3 are requested now and staged at the end of steps 0 and 1
        {
            float4 pa[3], pb[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) fetch_row(y0 - 1 + i, pa[i], pb[i]);
            fetch_skip(0, rq[0]);
            rq[2] = make_float4(0.f, 0.f, 0.f, 0.f);
            fetch_row(y0 + 2, qa[0], qb[0]);
            fetch_row(y0 + 3, qa[1], qb[1]);
            if constexpr (F16) {
#pragma unroll
                for (int i = 0; i < 3; ++i) note_row_max(i, y0 - 1 + i, pa[i], pb[i]);
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) stage_row(i, y0 - 1 + i, pa[i], pb[i], i);
            if constexpr (F16) {
                __syncthreads();                          // the maxima of the three rows are consumed
                note_row_max(0, y0 + 2, qa[0], qb[0]);    // row y0+2 is staged in Y(0): index = step parity
            }
        }
        __syncthreads();

        int exps[3] = {0, 0, 0};                                  // F16: scale exponents of the rows of the current step
        if constexpr (F16) {                                       // X(0) shifts them once more: preload rows y0-1, y0 as [1], [2]
            exps[1] = __builtin_amdgcn_readfirstlane(slot_exp[chunk * kWsRing + 0]);
            exps[2] = __builtin_amdgcn_readfirstlane(slot_exp[chunk * kWsRing + 1]);
        }
        // ---- the two phases of a row step ----------------------------------------------------------------------------
        // reduce(r): the four chunk sums of output row y0 + r -> bias, activation, skip, clamp -> 16-byte store; in three parts
        // that phase X places between the MFMAs of its three tap rows (their issue slots are free while the pipe is busy)
        float4 rp[4];
        float rvv[4];
        auto reduce_issue = [&](int r) {
            const float *q = part + ((r & 1) * kWsChunks * 64 + e_co) * kWsPS + 4 * e_g;
#pragma unroll
            for (int c = 0; c < 4; ++c) rp[c] = *reinterpret_cast<const float4 *>(q + c * 64 * kWsPS);
        };
        auto reduce_compute = [&](const float4 &rv) {
            const float bv = bias_l;
            rvv[0] = (((rp[0].x + rp[1].x) + rp[2].x) + rp[3].x) + bv; rvv[1] = (((rp[0].y + rp[1].y) + rp[2].y) + rp[3].y) + bv;
            rvv[2] = (((rp[0].z + rp[1].z) + rp[2].z) + rp[3].z) + bv; rvv[3] = (((rp[0].w + rp[1].w) + rp[2].w) + rp[3].w) + bv;
            if constexpr (ACTK == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rvv[i] = fmaxf(rvv[i], 0.01f * rvv[i]);       // LeakyReLU(0.01): max(v, 0.01 v)
            } else if constexpr (ACTK == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rvv[i] = fmaxf(rvv[i], 0.f);
            } else if constexpr (ACTK == 3) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rvv[i] = split_act<true>(rvv[i], a.act);
            }
            if (res != nullptr) { rvv[0] += rv.x; rvv[1] += rv.y; rvv[2] += rv.z; rvv[3] += rv.w; }   // uniform select
            if (a.clamp) {
#pragma unroll
                for (int i = 0; i < 4; ++i) rvv[i] = fminf(fmaxf(rvv[i], 0.f), 1.f);
            }
        };
        auto reduce_store = [&](int r) {
#ifdef CT_WS_ABL_NOSTORE
            const bool ok = e_ok && r >= 0 && r < rows && rvv[0] == 123.456f;
#else
            const bool ok = e_ok && r >= 0 && r < rows;      // the first X has no row to finish; the last steps may be padding
#endif
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = {__float_as_uint(rvv[0]), __float_as_uint(rvv[1]), __float_as_uint(rvv[2]), __float_as_uint(rvv[3])};
            const unsigned int off = ok ? ((unsigned int)e_row * uplane + (unsigned int)((y0 + r) * a.W + x0 + 4 * e_g)) * 4u : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b128(v, out_rs, (int)off, 0, 0);
        };
        auto reduce_row = [&](int r, const float4 &rv) { reduce_issue(r); reduce_compute(rv); reduce_store(r); };
#ifdef CT_WS_ABL_NOREDUCE
#define reduce_issue(r) do { } while (0)
#define reduce_compute(rv) do { rvv[0] = rv.x; } while (0)
#define reduce_store(r) do { } while (0)
#endif
        // X(r): finish output row r-1, the MFMAs of output row y0 + r on this wave's (chunk, half) block, partial sums -> LDS.
        auto phase_x = [&](int r, const float4 &rv) {
            reduce_issue(r - 1);
            // two accumulators, alternating: a chain of MFMAs on ONE accumulator waits out the result latency of every link
            // while the partner wave is in its Y phase (measured: X = 2780 cycles for 1728 cycles of MFMA issue)
            f32x16s acc, acc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
            int e_cur = 0;
            if constexpr (F16) {                                   // scales of the three rows: two are known from the last step
                exps[0] = exps[1]; exps[1] = exps[2];
                exps[2] = __builtin_amdgcn_readfirstlane(slot_exp[chunk * kWsRing + ((r + 2) & (kWsRing - 1))]);
            }
            // B fragments of a tap row are read one row ahead of the MFMAs that consume them
            // (fp16 form; the bf16 form has no registers to spare for it and reads per tap)
            uint4 bq[F16 ? 3 : 1][3][PIECES];
            auto read_b = [&](int ky) {
                const uint4 *rowp = ring + ((r + ky) & (kWsRing - 1)) * kWsSlot + hl * kWsCols + 3 + nl;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int p = 0; p < PIECES; ++p) bq[F16 ? ky : 0][kx][p] = rowp[2 * p * kWsCols + kx];
            };
            if constexpr (F16) read_b(0);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if constexpr (F16) { if (ky < 2) read_b(ky + 1); }
                else read_b(ky);
                if constexpr (F16) __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs they do not feed
                if constexpr (F16) {
                    const int ek = exps[ky];
                    if (ky > 0 && ek != e_cur) {                          // rare: neighbouring rows mostly share their scale
                        const float f = __uint_as_float((unsigned int)(127 + min(max(ek - e_cur, -126), 127)) << 23);
#pragma unroll
                        for (int i = 0; i < 16; ++i) { acc[i] *= f; acc2[i] *= f; }
                    }
                    e_cur = ek;
                }
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int tap = ky * 3 + kx;
                    if constexpr (F16) {
                        const f16x8 bh = __builtin_bit_cast(f16x8, bq[ky][kx][0]), bl = __builtin_bit_cast(f16x8, bq[ky][kx][1]);
                        const f16x8 ah = __builtin_bit_cast(f16x8, wreg[tap][0]), al = __builtin_bit_cast(f16x8, wreg[tap][1]);
#ifdef CT_WS_ABL_NOMFMA
                        if (tap == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                        else { acc[0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, bl).y) + __builtin_bit_cast(float, __builtin_bit_cast(uint4, ah).x); }
                        if (false) {
#else
                        if (tap & 1) {
#endif
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc2, 0, 0, 0);
                        } else {
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
                        }
                    } else {
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, bq[0][kx][0]);
                        const bf16x8 bm = __builtin_bit_cast(bf16x8, bq[0][kx][1]);
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, bq[0][kx][PIECES - 1]);
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, wreg[tap][0]), am = __builtin_bit_cast(bf16x8, wreg[tap][1]),
                                     al = __builtin_bit_cast(bf16x8, wreg[tap][PIECES - 1]);
#ifdef CT_WS_ABL_NOMFMA
                        if (tap == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                        else { acc[0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, bm).x) + __builtin_bit_cast(float, __builtin_bit_cast(uint4, bl).y) +
                                         __builtin_bit_cast(float, __builtin_bit_cast(uint4, ah).x) + __builtin_bit_cast(float, __builtin_bit_cast(uint4, am).x); }
#else
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);       // small terms first (as in conv_split)
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc2, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc2, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc2, 0, 0, 0);
#endif
                    }
                }
                // the previous row's epilogue in the issue slots the matrix pipe leaves free
                if (ky == 0) reduce_compute(rv);
                if (ky == 1) reduce_store(r - 1);
                if constexpr (F16) __builtin_amdgcn_sched_barrier(0);
            }
            WS_STAMP(6);
            // lane owns column nl, output channels 32 mt + (i & 3) + 8 (i >> 2) + 4 hl
            float *pw = part + (((r & 1) * kWsChunks + chunk) * 64 + 32 * mt) * kWsPS + nl;
            float unscale = 1.f;
            if constexpr (F16) unscale = __uint_as_float((unsigned int)(127 + min(max(-e_cur - a.w_exp, -126), 127)) << 23);
#pragma unroll
#ifdef CT_WS_ABL_NOPART
            if (acc[0] == 123.456f)
#endif
            for (int i = 0; i < 16; ++i) pw[((i & 3) + 8 * (i >> 2) + 4 * hl) * kWsPS] = F16 ? (acc[i] + acc2[i]) * unscale : acc[i] + acc2[i];
        };
        auto phase_x_full = [&](int r, auto set_c) {
            constexpr int SET = decltype(set_c)::value;
            phase_x(r, rq[(SET + 2) % 3]);
        };
        // Y(r): request the skip row of output row r+1 and input row r+4, stage input row r+2 (requested two Y phases ago; F16:
        // with the scale its maximum, noted one Y phase ago, asks for).  Loads are issued oldest-needed first (vmcnt is in order).
        auto phase_y = [&](int r, auto set_c) {
            constexpr int SET = decltype(set_c)::value;      // r % 3
            fetch_skip(r + 1, rq[(SET + 1) % 3]);                  // consumed by reduce(r+1) inside X(r+2): more than a step from now
            fetch_row(y0 + r + 4, qa[(SET + 2) % 3], qb[(SET + 2) % 3]);
            WS_STAMP(3);
            note_row_max((r + 1) & 1, y0 + r + 3, qa[(SET + 1) % 3], qb[(SET + 1) % 3]);   // F16: the row staged in Y(r+1)
            WS_STAMP(5);
            stage_row((r + 3) & (kWsRing - 1), y0 + r + 2, qa[SET], qb[SET], r & 1);   // the slot of input row y0+r-1 is free since X(r-1)
        };
        // The two waves of a SIMD run the phases of a step in opposite order -- while one of them is in X (27 or 54 MFMAs) the
        // other one does its Y; with the same order in both, the matrix pipe sat idle through every Y (measured: full kernel =
        // kernel without MFMAs + MFMA time) -- and ONE barrier ends the step.  Nothing inside a step depends on the other
        // phase of the same step: X(r) reads ring slots r .. r+2 and the partial sums of step r-1 and writes the partial sums
        // r & 1 (last read in step r-1); Y(r) writes ring slot r+3 (last read in step r-1), its exponent, and the row
        // maximum (r+1) & 1, and reads the maxima r & 1 (written in step r-1).
        auto step = [&](int r, auto set_c, auto mt_c) {
            WS_STAMP(4);
            if constexpr (decltype(mt_c)::value == 0) { phase_x_full(r, set_c); WS_STAMP(0); phase_y(r, set_c); WS_STAMP(2); }
            else { phase_y(r, set_c); WS_STAMP(2); phase_x_full(r, set_c); WS_STAMP(0); }
            __syncthreads();
            WS_STAMP(1);
        };
        const int steps = (rows + 2) / 3 * 3;
        auto row_loop = [&](auto mt_c) {
#pragma unroll 1
            for (int r = 0; r < steps; r += 3) {
                step(r, std::integral_constant<int, 0>(), mt_c);
                step(r + 1, std::integral_constant<int, 1>(), mt_c);
                step(r + 2, std::integral_constant<int, 2>(), mt_c);
            }
        };
        if (mt == 0) row_loop(std::integral_constant<int, 0>());       // uniform: mt lives in a scalar register
        else row_loop(std::integral_constant<int, 1>());
        reduce_row(steps - 1, rq[2]);                        // steps % 3 == 0: the skip row of the last step sits in set 2
        __syncthreads();      // the partial-sum tiles of the last rows are read before the next item overwrites them
    }
#ifdef CT_CONV_PROFILE
    WS_STAMP(4);
    if (lane == 0 && a.prof) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.prof[((size_t)blockIdx.x * kWsWaves + wave) * 8 + i] = pt[i];
    }
#endif
}

#ifdef CT_CONV_PROFILE
static unsigned long long *g_ws_prof = nullptr;     // diagnostic build: every conv_ws launch stamps its phases here
#endif

// 1 = not this kernel's geometry (the caller falls back to conv_split_kernel)
int conv_ws(const ConvArgs &a, int N, bool gen, hipStream_t s) {
    static const int enabled = [] { const char *e = getenv("CT_HIP_CONV_WS"); return e ? atoi(e) : 1; }();
#ifndef CT_CONV_PROFILE
    if (a.prof != nullptr) return 1;
#endif
    if ((!enabled && !a.f16) || a.in2 != nullptr || a.cin <= 32 || a.cin > 64) return 1;
    if ((unsigned long long)a.H * (unsigned long long)a.W * 64ull >= (1ull << 32)) return 1;     // the kernel indexes a 64-channel image with 32 bits (and eight output planes with 32-bit BYTE offsets): the tile kernel takes larger ones
    const int n_strips = (a.W + kWsTW - 1) / kWsTW;
    // row segments: the split that minimises the row steps of the busiest workgroup (32 workgroups per XCD sweep the bands
    // of that XCD; a segment costs its rows + 2 halo rows of staging)
    const long long cols = (long long)N * a.groups * n_strips;
    const long long imgs = (long long)N * a.groups;
    int n_seg = 1, seg = a.H;
    long long best = -1;
    for (int ns = 1; ns <= (a.H + 15) / 16; ++ns) {
        const int sg = (a.H + ns - 1) / ns;
        const int ns_eff = (a.H + sg - 1) / sg;
        const long long bands_per_xcd = (imgs * ns_eff + 7) / 8;
        const long long rounds = (bands_per_xcd * n_strips + 31) / 32;
        const long long cost = rounds * ((sg + 2) / 3 * 3 + 2);      // steps run in threes; two halo rows of prologue
        if (best < 0 || cost < best) { best = cost; n_seg = ns_eff; seg = sg; }
    }
    static const int forced_seg = [] { const char *e = getenv("CT_HIP_WS_SEG"); return e ? atoi(e) : 0; }();   // tuning only
    if (forced_seg > 0) { seg = forced_seg < a.H ? forced_seg : a.H; n_seg = (a.H + seg - 1) / seg; }
    const long long n_items = cols * n_seg;
    if (n_items > 0x7fffffffLL) return CT_E_BADARG;
    const int pieces = a.f16 ? 2 : 3;
    const size_t lds = (size_t)kWsChunks * kWsRing * pieces * 2 * kWsCols * 16 + (size_t)2 * kWsChunks * 64 * kWsPS * sizeof(float) + 256;
    const int grid = 256;       // a multiple of 8 (the XCD-aware item order); idle workgroups leave at once
    ConvArgs b = a;
    b.n_images = N;
#ifdef CT_CONV_PROFILE
    if (g_ws_prof) b.prof = g_ws_prof;
#endif
    const int actk = !gen ? 1 : (a.act >= 0 && a.act <= 2) ? a.act : 3;      // gen == false: the DCMCS3DI entry (LeakyReLU or none)
    const int actk_eff = (!gen && a.act == 0) ? 0 : actk;
    const int variant = (a.f16 ? 4 : 0) + actk_eff;
    typedef void (*kern_t)(ConvArgs, int, int, int, int);
    static const kern_t kerns[8] = {conv_ws_kernel<0, false>, conv_ws_kernel<1, false>, conv_ws_kernel<2, false>, conv_ws_kernel<3, false>,
                                    conv_ws_kernel<0, true>,  conv_ws_kernel<1, true>,  conv_ws_kernel<2, true>,  conv_ws_kernel<3, true>};
    const kern_t kern = kerns[variant];
    static DynLdsAttr attr[8];          // per kernel variant, per device
    {
        hipError_t e = attr[variant].ensure(reinterpret_cast<const void *>(kern), lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kWsWaves), lds, s, b, n_strips, seg, n_seg, (int)n_items);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // namespace ct

extern "C" {

// The fp16 two-piece form of the weight-stationary kernel (conv_ws_kernel<.., true>): 3x3, stride 1, padding 1, 32 < cin <= 64.
// wp16: fp16 bit patterns [ceil(cout/64)][ceil(cin/16)][9][piece hi/lo][m][k-half][cout%32][8 channels] of weight * 2^w_exp
// (ct_hip.pack_conv_weight_split16); bias zero-padded to 64 * ceil(cout/64).
int ct_conv3x3_ws16_f32(const float *in, const void *wp16, int w_exp, const float *bias, const float *residual, float *out, int n, int cin,
                        int cout, int h, int w, long long in_bstride, long long out_bstride, long long res_bstride, int act, int clamp,
                        void *stream) {
    if (!in || !wp16 || !bias || !out || n < 0 || cin <= 32 || cin > 64 || cout < 1 || h < 0 || w < 0 || act < 0 || act > 5) return CT_E_BADARG;
    if ((w % 4) || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (reinterpret_cast<uintptr_t>(wp16) & 15) ||
        (in_bstride % 4) || (out_bstride % 4) || (residual && ((reinterpret_cast<uintptr_t>(residual) & 15) || (res_bstride % 4))))
        return CT_E_ALIGN;
    if (w_exp < -100 || w_exp > 100) return CT_E_BADARG;
    if (n == 0 || h == 0 || w == 0) return CT_OK;
    ct::ConvArgs a;
    a.in = in; a.in2 = nullptr; a.cin1 = cin; a.in2_bstride = 0;
    a.wp = reinterpret_cast<const float *>(wp16); a.bias = bias; a.residual = residual; a.out = out;
    a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.in_bstride = in_bstride; a.out_bstride = out_bstride; a.res_bstride = res_bstride;
    a.act = act; a.clamp = clamp; a.groups = (cout + 63) / 64; a.prof = nullptr;
    a.f16 = 1; a.w_exp = w_exp;
    const int rc = ct::conv_ws(a, n, true, (hipStream_t)stream);
    return rc == 1 ? CT_E_BADARG : rc;
}

#ifdef CT_CONV_PROFILE
void ct_conv_ws_set_prof(unsigned long long *p) { ct::g_ws_prof = p; }
#endif

}  // extern "C"
