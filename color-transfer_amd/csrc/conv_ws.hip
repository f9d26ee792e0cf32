// conv_ws.hip -- 3x3 stride-1 "same" convolution with <= 64 input channels on the bf16 matrix pipe, WEIGHTS STATIONARY.
//
// Same arithmetic as conv_split.hip (float32 operands as three bf16 pieces, six v_mfma_f32_32x32x16_bf16 per 16-channel
// product, float32 accumulation), different dataflow.  conv_split shares one input tile between the four waves of a
// workgroup and streams the weights of every tap through an LDS ring: one barrier per tap, a conversion phase between the
// stages, two workgroups per CU competing for the matrix pipe -- measured 47 % MFMA-busy, 22 % of a wave's time in the
// per-tap barriers, 13 % in the conversion phase (tools/prof_conv_split.py).  Here the contraction is split ACROSS the
// waves instead: wave (c, m) of the eight owns input channels [16 c, 16 c + 16) and output channels [32 m, 32 m + 32) and
// keeps the weights of all nine taps x three pieces of that block in registers (27 fragments = 108 VGPRs) for the lifetime
// of a persistent workgroup (one per CU; the two waves of a chunk share a SIMD, so while one of them stages, writes or
// reduces, the other one's MFMAs keep the matrix pipe busy).  A workgroup walks down a 32-pixel wide strip; per output row
//   * the two waves of a chunk fetch the next input row of ITS 16 channels (float32 NCHW, 16-byte loads, half the columns
//     each), split it in registers and write it into the chunk's four-row LDS ring [piece][k-half][column][8 channels];
//   * every wave runs 9 taps x 6 MFMAs (B fragments = one conflict-free 16-byte LDS read each);
//   * writes its 32 x 32 partial sums to LDS; after ONE barrier per row every wave adds the four partial sums of 8 output
//     channels in a fixed order (chunk 0..3), applies bias / activation / ResB skip / clamp and stores 16-byte rows.
// No weight traffic after the prologue, no per-tap barriers, the float32 -> bf16 split of a value happens once, by one wave.
// (First version, one wave per SIMD with all 64 output channels: 39 % MFMA-busy -- the ~500 non-MFMA instructions of a row
// ran beside nothing; profiles/r03_conv_ws_notes.)
// Rounding: float32 sums per chunk, then ((p0 + p1) + p2) + p3 + bias -- float32-grade like conv_split (not bitwise equal).
#include "ct_common.h"
#include "ct_conv.h"
#include "ct_split.h"
#include <type_traits>

namespace ct {

constexpr int kWsTW = 32;                    // output columns of a strip (= MFMA N)
constexpr int kWsCols = 40;                  // staged columns x0-4 .. x0+35 (ten aligned groups of four)
constexpr int kWsSlot = 3 * 2 * kWsCols;     // 16-byte entries of one input row of one wave: [piece][k-half][column]
constexpr int kWsRing = 4;                   // input rows resident per wave (three in use, one being filled)
constexpr int kWsPS = 36;                    // floats per row of a partial-sum tile (16-byte aligned, k-halves on disjoint banks)
constexpr int kWsChunks = 4;                 // 16-channel chunks = SIMDs
constexpr int kWsWaves = 8;                  // (chunk, 32-channel half of the outputs)

template <bool GEN>
__global__ __launch_bounds__(512, 1) void conv_ws_kernel(ConvArgs a, int n_strips, int seg, int n_seg, int n_items) {
    extern __shared__ uint4 smem16[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const int chunk = wave & 3, mt = wave >> 2;         // waves c and c + 4 run on SIMD c
    uint4 *ring = smem16 + chunk * (kWsRing * kWsSlot);                                   // this chunk's input rows
    float *part = reinterpret_cast<float *>(smem16 + kWsChunks * kWsRing * kWsSlot);      // [2][chunk][64][kWsPS]
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
    // 0 X (MFMAs + partial sums), 1 barrier after X, 2 Y (reduce, requests, staging), 3 barrier after Y, 4 everything else
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0;
#define WS_STAMP0() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pt0) :: "memory")
#define WS_STAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); pt[i] += t__ - pt0; pt0 = t__; } while (0)
    WS_STAMP0();
#else
#define WS_STAMP(i) do { } while (0)
#endif
    uint4 wreg[9][3];
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
                for (int p = 0; p < 3; ++p)
                    wreg[tap][p] = has_chunk ? wp16[((((size_t)grp * n_chunks + chunk) * 9 + tap) * 3 + p) * 128 + mt * 64 + lane]
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
        auto stage_row = [&](int slot, int y, const float4 &la, const float4 &lb) {       // registers -> three bf16 pieces -> LDS slot
            float ma, mb;
            row_mask(y, ma, mb);
#ifdef CT_WS_ABL_NOSTAGE
            if (la.x == 123.456f) {
#else
            if (u_lane) {
#endif
                const int kh2 = u_cp >> 2, wsel = u_cp & 3;
                unsigned int *d = reinterpret_cast<unsigned int *>(ring + slot * kWsSlot) + ((kh2 * kWsCols + 4 * u_grp) * 4 + wsel);
                const float xa[4] = {ma != 0.f ? la.x : 0.f, ma != 0.f ? la.y : 0.f, ma != 0.f ? la.z : 0.f, ma != 0.f ? la.w : 0.f};
                const float xb[4] = {mb != 0.f ? lb.x : 0.f, mb != 0.f ? lb.y : 0.f, mb != 0.f ? lb.z : 0.f, mb != 0.f ? lb.w : 0.f};
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    unsigned int hw, mw, lw;
                    split3x2(xa[px], xb[px], hw, mw, lw);
                    d[px * 4] = hw;
                    d[(2 * kWsCols + px) * 4] = mw;
                    d[(4 * kWsCols + px) * 4] = lw;
                }
            }
        };

        // prologue: input rows y0-1, y0, y0+1 -> slots 0, 1, 2 (the barrier publishes them to the chunk's other wave); rows
        // y0+2 and y0+3 are requested now and staged at the end of steps 0 and 1
        {
            float4 pa[3], pb[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) fetch_row(y0 - 1 + i, pa[i], pb[i]);
            rq[0] = make_float4(0.f, 0.f, 0.f, 0.f);
            fetch_row(y0 + 2, qa[0], qb[0]);
            fetch_row(y0 + 3, qa[1], qb[1]);
#pragma unroll
            for (int i = 0; i < 3; ++i) stage_row(i, y0 - 1 + i, pa[i], pb[i]);
        }
        __syncthreads();

        // ---- the two phases of a row step ----------------------------------------------------------------------------
        // X(r): the MFMAs of output row y0 + r on this wave's (chunk, half) block, partial sums -> LDS.
        auto phase_x = [&](int r) {
            // two accumulators, alternating: a chain of 54 MFMAs on ONE accumulator waits out the result latency of every link
            // while the partner wave is in its Y phase (measured: X = 2780 cycles for 1728 cycles of MFMA issue)
            f32x16s acc, acc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const uint4 *rowp = ring + ((r + ky) & (kWsRing - 1)) * kWsSlot + hl * kWsCols + 3 + nl;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int tap = ky * 3 + kx;
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, rowp[kx]);
                    const bf16x8 bm = __builtin_bit_cast(bf16x8, rowp[2 * kWsCols + kx]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, rowp[4 * kWsCols + kx]);
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, wreg[tap][0]), am = __builtin_bit_cast(bf16x8, wreg[tap][1]),
                                 al = __builtin_bit_cast(bf16x8, wreg[tap][2]);
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
            // lane owns column nl, output channels 32 mt + (i & 3) + 8 (i >> 2) + 4 hl
            float *pw = part + (((r & 1) * kWsChunks + chunk) * 64 + 32 * mt) * kWsPS + nl;
#pragma unroll
            for (int i = 0; i < 16; ++i) pw[((i & 3) + 8 * (i >> 2) + 4 * hl) * kWsPS] = acc[i] + acc2[i];
        };
        // reduce(r): the four chunk sums of output row y0 + r -> bias, activation, skip, clamp -> 16-byte store
        auto reduce_row = [&](int r, const float4 &rv) {
            const int y = y0 + r;
            const float *q = part + ((r & 1) * kWsChunks * 64 + e_co) * kWsPS + 4 * e_g;
            const float4 p0 = *reinterpret_cast<const float4 *>(q), p1 = *reinterpret_cast<const float4 *>(q + 64 * kWsPS),
                         p2 = *reinterpret_cast<const float4 *>(q + 2 * 64 * kWsPS), p3 = *reinterpret_cast<const float4 *>(q + 3 * 64 * kWsPS);
            const float bv = bias_l;
            float v[4] = {(((p0.x + p1.x) + p2.x) + p3.x) + bv, (((p0.y + p1.y) + p2.y) + p3.y) + bv,
                          (((p0.z + p1.z) + p2.z) + p3.z) + bv, (((p0.w + p1.w) + p2.w) + p3.w) + bv};
            if (a.act) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = split_act<GEN>(v[i], a.act);
            }
            if (res != nullptr) { v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w; }   // uniform select
            if (a.clamp) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = fminf(fmaxf(v[i], 0.f), 1.f);
            }
#ifdef CT_WS_ABL_NOSTORE
            if (e_ok && r >= 0 && r < rows && v[0] == 123.456f)
#else
            if (e_ok && r >= 0 && r < rows)                  // the first Y has no row to finish; the last steps may be padding
#endif
                *reinterpret_cast<float4 *>(out + (unsigned int)e_co * uplane + (unsigned int)(y * a.W + x0 + 4 * e_g)) = make_float4(v[0], v[1], v[2], v[3]);
        };
        auto fetch_skip = [&](int r, float4 &rv) {             // ResB skip of output row y0 + r: branch-free like the row fetch
            const int yr = min(max(y0 + r, 0), a.H - 1);        // (without a skip tensor the output row is read and ignored)
            rv = *reinterpret_cast<const float4 *>((res ? res : out) + (unsigned int)min(e_co, cout_g - 1) * uplane +
                                                   (unsigned int)(yr * a.W + min(x0 + 4 * e_g, a.W - 4)));
        };
        // Y(r): everything that is not matrix work -- finish output row r-1, request the skip row of row r and input row
        // r+4, stage input row r+2 (requested two Y phases ago).  Loads are issued oldest-needed first: vmcnt retires in order.
        auto phase_y = [&](int r, auto set_c) {
            constexpr int SET = decltype(set_c)::value;      // r % 3
            fetch_skip(r, rq[(SET + 1) % 3]);
            fetch_row(y0 + r + 4, qa[(SET + 2) % 3], qb[(SET + 2) % 3]);
            reduce_row(r - 1, rq[SET]);
            stage_row((r + 3) & (kWsRing - 1), y0 + r + 2, qa[SET], qb[SET]);   // the slot of input row y0+r-1 is free since X(r-1)
        };
        // The two waves of a SIMD run the phases in opposite order, with a barrier after every half step: while one of them is
        // in X (54 MFMAs) the other one does its Y -- with the same order in both, the matrix pipe sat idle through every Y
        // (measured: full kernel = kernel without MFMAs + MFMA time).  Half step h: waves with mt == 0 run X(h/2) when h is
        // even and Y(h/2) when it is odd; waves with mt == 1 the other way round.  Hazards: reduce(r-1) inside Y(r) runs at half
        // steps 2r and 2r+1, after the last X(r-1) (2r-1); the row staged in Y(r) is read from X(r+1) on (>= 2r+2); its slot
        // was last read in X(r-1); the partial-sum buffer r & 1 is rewritten in X(r+2) (>= 2r+4), after reduce(r) (<= 2r+3).
        auto half_step = [&](int r, auto set_c, bool do_x) {
            WS_STAMP(4);
            if (do_x) { phase_x(r); WS_STAMP(0); }
            else { phase_y(r, set_c); WS_STAMP(2); }
            __syncthreads();
            if (do_x) WS_STAMP(1); else WS_STAMP(3);
        };
        const int steps = (rows + 2) / 3 * 3;
#pragma unroll 1
        for (int r = 0; r < steps; r += 3) {
            half_step(r, std::integral_constant<int, 0>(), mt == 0);
            half_step(r, std::integral_constant<int, 0>(), mt == 1);
            half_step(r + 1, std::integral_constant<int, 1>(), mt == 0);
            half_step(r + 1, std::integral_constant<int, 1>(), mt == 1);
            half_step(r + 2, std::integral_constant<int, 2>(), mt == 0);
            half_step(r + 2, std::integral_constant<int, 2>(), mt == 1);
        }
        reduce_row(steps - 1, rq[0]);                        // steps % 3 == 0: the skip row of the last step sits in set 0
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

// 1 = not this kernel's geometry (the caller falls back to conv_split_kernel)
int conv_ws(const ConvArgs &a, int N, bool gen, hipStream_t s) {
    static const int enabled = [] { const char *e = getenv("CT_HIP_CONV_WS"); return e ? atoi(e) : 1; }();
#ifndef CT_CONV_PROFILE
    if (a.prof != nullptr) return 1;
#endif
    if (!enabled || a.in2 != nullptr || a.cin <= 32 || a.cin > 64) return 1;
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
    const size_t lds = (size_t)kWsChunks * kWsRing * kWsSlot * 16 + (size_t)2 * kWsChunks * 64 * kWsPS * sizeof(float);
    const int grid = 256;       // a multiple of 8 (the XCD-aware item order); idle workgroups leave at once
    ConvArgs b = a;
    b.n_images = N;
    auto kern = gen ? conv_ws_kernel<true> : conv_ws_kernel<false>;
    static bool attr_set[2] = {false, false};
    if (!attr_set[gen]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[gen] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kWsWaves), lds, s, b, n_strips, seg, n_seg, (int)n_items);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // namespace ct
