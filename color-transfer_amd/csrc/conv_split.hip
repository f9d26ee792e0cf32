// conv_split.hip -- the stride-1 convolutions of cnn.hip on the bf16 matrix pipe with float32-grade accuracy.
//
// v_mfma_f32_32x32x2_f32 (cnn.hip) runs at 1/16 of the bf16 MFMA rate on gfx950.  Here every float32 operand is split
// into three bf16 pieces, x = hi + mid + lo (8 + 8 + 8 mantissa bits, each difference exact in float32), and a product
// is evaluated as the six partial products whose weight is >= 2^-16 of the leading one
//     a*b ~= a_lo*b_hi + a_hi*b_lo + a_mid*b_mid + a_mid*b_hi + a_hi*b_mid + a_hi*b_hi        (float32 accumulation)
// -- what is dropped (mid*lo, lo*mid, lo*lo) is <= 2^-23 relative, the size of one float32 rounding.  Six
// v_mfma_f32_32x32x16_bf16 (32 cycles each, K = 16) replace eight 32x32x2 f32 MFMAs (64 cycles each): 2.67x the
// throughput at the same accuracy class (NOT bitwise the fmaf chain of the f32 kernel; parity bars are unchanged).
//
// Same implicit GEMM as conv_mfma_kernel: M = 64 output channels (A = weights, pre-split on the host, streamed by one wave
// from L2 through a two-tap LDS ring), N = 32 consecutive pixels of a row (B from an LDS halo tile), persistent workgroups walking
// 8x32-pixel tiles, 16 input channels per stage (= one bf16 K step per tap).  The float32 NCHW input is split while it
// is staged: a thread fetches 8 channels x 4 pixels with 16-byte loads, converts in registers and writes the LDS tile
// [piece][row][k-half][column][8 channels] so that an MFMA B fragment (8 consecutive channels of one pixel) is one
// conflict-free 16-byte read.  Epilogue (bias, activation, ResB skip, clamp, 16-byte stores through a per-wave LDS
// transpose) as in cnn.hip.  Requires W % 4 == 0 and 16-byte aligned tensors; other cases stay on the f32 kernel.
#include "ct_common.h"
#include "ct_conv.h"
#include "ct_split.h"

namespace ct {

constexpr int kSpTH = 8;      // output rows per workgroup (two per wave)
constexpr int kSpTW = 32;     // output columns per workgroup (= MFMA N)
constexpr int kSpKC = 16;     // input channels per stage (= MFMA K)
constexpr int kSpCUs = 256;
constexpr int kSkErrWord = 1000;   // stream-K scratch: 32-bit word (of the 1024 flag words) counting consumers that gave up waiting
// sticky per-device status (ct_device_status bit 1): some stream-K consumer gave up since the last clear
__device__ unsigned g_sk_status;

// LDS-DMA: 64 lanes x 16 bytes from global straight into LDS at lds_addr + 16 * lane (no registers).  Issued through
// inline asm on purpose: behind the builtin the compiler puts an s_waitcnt vmcnt(0) in front of every later LDS read
// (it cannot tell which ones alias), i.e. the streaming wave would wait out the full L2 latency of the loads it has
// just issued; here the landing is awaited explicitly, with a counted vmcnt, right before the publishing barrier.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned int lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_addr) : "memory");
    // m0 is written above.  It cannot be declared: hipcc rejects "m0" in a clobber list ("reserved register", ignored
    // with a warning) -- the compiler never allocates m0 and only sets it itself right in front of the few instructions
    // that read it (LDS-DMA builtins, s_movrel, s_sendmsg), none of which this kernel uses (checked in the ISA:
    // `make asm`, then `grep -n m0 build/conv_split*.s` shows only this s_mov_b32).
}

// wp : bf16 bit patterns [group][chunk16][tap][piece][m][lane = 32*khalf + cout%32][8 channels]   (host: pack_conv_weight_split)
// F16: two fp16 pieces per operand and three v_mfma_f32_32x32x16_f16 per product instead of three bf16 pieces and six MFMAs (half
// the matrix work, 2/3 of the LDS operand bytes; 2^-22 relative is dropped).  fp16 has five exponent bits: the weights carry one
// power of two per layer (a.w_exp, host), every staged 16-channel input tile a RUNNING one per output tile (its maximum joins the
// minimum exponent of the channels seen so far; the waves leave their maxima in LDS before the barrier that already separates "tile
// consumed" from "next tile staged"); when it drops the accumulators are rescaled at the top of the stage.  Bias, pre-activation
// addend, activation, skip and clamp all run in the epilogue on the unscaled value.
template <int KH, int KW, bool GEN, bool F16>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(ConvArgs a, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int PADY = KH / 2, PADX = KW / 2;
    constexpr bool HALO = (KW > 1);
    constexpr int ROWS = kSpTH + KH - 1;
    constexpr int TWP = HALO ? kSpTW + 8 : kSpTW;   // LDS row = image columns [x0-4, x0+36)
    constexpr int COL0 = HALO ? 4 : 0;
    constexpr int PSZ = ROWS * 2 * TWP;                // 16-byte entries per piece
    constexpr int NU = 2 * ROWS * 8;                   // (k-half, row, 4-column group) units of 8 channels x 4 pixels
    constexpr int HC = KW - 1;
    constexpr int NE = kSpKC * ROWS * HC;              // halo scalars per stage
    constexpr int NF = 192;                            // threads that fetch the input tile (waves 0..2; wave 3 streams weights)
    constexpr int PFE = (NE + NF - 1) / NF;
    constexpr int MT = 2, COUTP = 64, RPW = kSpTH / 4, TAPS = KH * KW;
    constexpr int NP = F16 ? 2 : 3;                    // pieces per operand
    static_assert(NU <= NF, "tile geometry");
    extern __shared__ uint4 smem16[];
    constexpr int WSLOT = NP * MT * 64;                 // 16-byte entries of one tap's weight fragments
    uint4 *tin = smem16;                                // [NP][ROWS][2][TWP]
    constexpr int WRING = F16 ? 8 : 4;                  // taps of weights resident in LDS (4 KiB each as two pieces, 6 KiB as three)
    constexpr int WAHEAD = WRING - 1;                   // a tap's fragments are requested WAHEAD taps before their use
    uint4 *wl = smem16 + NP * PSZ;                      // [WRING][piece][m][lane]
    float *stg = reinterpret_cast<float *>(smem16 + NP * PSZ + WRING * WSLOT) + (threadIdx.x >> 6) * (32 * 32);
    float *mxw = reinterpret_cast<float *>(smem16 + NP * PSZ + WRING * WSLOT) + 4 * (32 * 32);   // F16: the waves' tile maxima

    const int tid = threadIdx.x, lane = tid & 63, nl = lane & 31, hl = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // in an SGPR: the loader wave's bookkeeping is a scalar branch away
    const size_t plane = (size_t)a.H * a.W;
    const unsigned int uplane = (unsigned int)plane;
    const int n_chunks = (a.cin + kSpKC - 1) / kSpKC;
    // XCD-aware tile order (round 3): workgroup b runs on XCD b % 8 (round-robin dispatch; an assumption that only costs
    // speed if wrong); XCD p walks the contiguous range [p T, (p + 1) T) of the row-major (tile, group) list with its
    // workgroups side by side, so the halo columns / rows of a tile -- the edge lines of its neighbours -- are found in the
    // same L2 instead of being pulled through the fabric once per tile (conv_ws.hip measured 3 lines per line of new data).
    const bool xcd_order = (gridDim.x & 7) == 0 && n_tiles >= (int)gridDim.x;
    const int tiles_per_xcd = (n_tiles + 7) >> 3, wgs_per_xcd = (int)gridDim.x >> 3;
    const int xcd_first = (int)(blockIdx.x & 7) * tiles_per_xcd;
    const int xcd_count = max(0, min(tiles_per_xcd, n_tiles - xcd_first));
    const int my_tiles = xcd_order ? (xcd_count - (int)(blockIdx.x >> 3) + wgs_per_xcd - 1) / wgs_per_xcd
                                   : (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto tile_id = [&](int k) -> int {            // this workgroup's k-th (tile, group) index
        return xcd_order ? xcd_first + (int)(blockIdx.x >> 3) + k * wgs_per_xcd : (int)blockIdx.x + k * (int)gridDim.x;
    };
    // ---- stream-K (F16 form; the launcher sets a.sk_ws when it pays): the launch's time is a staircase in units / resident
    // workgroups (tools/bench_conv_quant.py: 512 units 100 us, 544 units 170 us), so instead of whole units a workgroup takes an
    // equal share of its XCD's STAGES: positions [ra, rb) of the list (unit, chunk) of that XCD's contiguous unit range.  A share
    // is at least one unit long, so a unit is cut at most once: into a head [0, c) and a tail [c, n_chunks) that belong to
    // neighbouring workgroups.  A workgroup runs its head piece FIRST and leaves the unscaled float32 partial sums in its slot of
    // a.sk_ws (write-through stores, then a flag); it runs its tail piece LAST, adds the neighbour's partial in the epilogue and
    // clears the flag -- by then the neighbour, which started with that head, has long published it (workgroups are dispatched
    // in index order, so the producer blockIdx - 8 is never behind a resident consumer).  Sums are added in a fixed order.
    const bool sk = F16 && a.sk_ws != nullptr;
    int sk_ufirst = 0, sk_ca = 0, sk_ulast = 0, sk_cbl = 0, sk_has_t = 0, sk_has_h = 0, sk_nf = 0, sk_f0 = 0, sk_first = 0, sk_stages = 0;
    int sk_col_len = 1, sk_col_rem = 0;
    if (sk) {
        const int x = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
        sk_first = (int)((long long)n_tiles * x / 8);
        const int count = (int)((long long)n_tiles * (x + 1) / 8) - sk_first;
        sk_col_len = count / wgs_per_xcd; sk_col_rem = count - sk_col_len * wgs_per_xcd;
        const long long sx = (long long)count * n_chunks;
        const int ra = (int)(sx * j / wgs_per_xcd), rb = (int)(sx * (j + 1) / wgs_per_xcd);
        sk_stages = rb - ra;
        sk_ufirst = ra / n_chunks; sk_ca = ra - sk_ufirst * n_chunks;
        sk_ulast = (rb - 1) / n_chunks; sk_cbl = rb - sk_ulast * n_chunks;
        sk_has_t = sk_ca > 0; sk_has_h = sk_cbl < n_chunks;
        sk_nf = sk_ulast - sk_ufirst + 1 - sk_has_t - sk_has_h; sk_f0 = sk_ufirst + sk_has_t;
    }
    // a workgroup's work: segments = (unit, chunks [cb, ce)); kind 0: a whole unit, 1: a head piece (handed on), 2: a tail piece
    const int n_seg = sk ? sk_nf + sk_has_t + sk_has_h : my_tiles;
    const int n_stages = sk ? sk_stages : my_tiles * n_chunks;
    if (n_stages == 0) return;
    struct SegCur { int seg, chunk, ce, unit; };
    auto seg_kind = [&](int i) -> int { return !sk ? 0 : (sk_has_h && i == 0) ? 1 : (i - sk_has_h < sk_nf) ? 0 : 2; };
    // position in the XCD's list -> unit: the list is the column-major reading of the row-major [.][wgs_per_xcd] unit matrix, so
    // that workgroups j, j + 1, ... -- whose shares start about one column apart -- work on NEIGHBOURING units at the same time
    // (the groups of one tile, adjacent tiles: shared input tiles and halos in the XCD's L2), as the strided order of the plain
    // mode has it
    auto sk_unit = [&](int p) -> int {
        const int big = sk_col_rem * (sk_col_len + 1);
        const int col = p < big ? p / (sk_col_len + 1) : sk_col_rem + (p - big) / sk_col_len;
        const int row = p < big ? p - col * (sk_col_len + 1) : (p - big) - (col - sk_col_rem) * sk_col_len;
        return sk_first + row * wgs_per_xcd + col;
    };
    auto seg_set = [&](SegCur &c, int i) {
        const int kind = seg_kind(i);
        c.seg = i;
        c.ce = kind == 1 ? sk_cbl : n_chunks;
        c.unit = !sk ? tile_id(i) : sk_unit(kind == 1 ? sk_ulast : kind == 2 ? sk_ufirst : sk_f0 + i - sk_has_h);
        c.chunk = kind == 2 ? sk_ca : 0;
    };
    auto seg_next = [&](SegCur &c) {               // one stage on (stays on the last stage's successor at the very end)
        if (++c.chunk == c.ce && c.seg + 1 < n_seg) seg_set(c, c.seg + 1);
    };

    // ---- staging roles (fixed per thread) ----
    const bool unit = tid < NU;
    const int u_h = tid / (ROWS * 8), u_rem = tid - u_h * (ROWS * 8), u_row = u_rem >> 3, u_g = u_rem & 7;
    float4 pf4[8];
    float pfe[PFE > 0 ? PFE : 1];
    auto fetch_tile = [&](const SegCur &c) {
        const int chunk = c.chunk;
        const int t = c.unit / a.groups;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int x0 = tx * kSpTW, y0 = ty * kSpTH, c0 = chunk * kSpKC;
        // two-source input: a 16-channel chunk lies entirely in one of the tensors (cin1 % 16 == 0)
        const bool third = a.in3 && c0 >= a.cin2;
        const bool second = !third && a.in2 && c0 >= a.cin1;
        const float *in = third ? a.in3 + (size_t)n * a.in3_bstride + (size_t)(c0 - a.cin2) * plane
                        : second ? a.in2 + (size_t)n * a.in2_bstride + (size_t)(c0 - a.cin1) * plane
                                 : a.in + (size_t)n * a.in_bstride + (size_t)c0 * plane;
        if (unit) {
            const int gy = y0 + u_row - PADY, gx = x0 + 4 * u_g;
            const bool ok = gy >= 0 && gy < a.H && gx < a.W;          // W % 4 == 0: all four columns or none
            const unsigned int off = (unsigned int)(gy * a.W + gx);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 8 * u_h + j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok && c0 + c < a.cin) v = *reinterpret_cast<const float4 *>(in + (unsigned int)c * uplane + off);
                pf4[j] = v;
            }
        }
        if constexpr (HALO) {
#pragma unroll
            for (int j = 0; j < PFE; ++j) {
                const int e = (NF - 1 - tid) + j * NF;            // the fetch threads without a unit take the halo first
                const int c = e / (ROWS * HC), rem = e - c * (ROWS * HC);
                const int yy = rem / HC, side = rem - yy * HC;
                const int gy = y0 + yy - PADY, gx = side < PADX ? x0 - PADX + side : x0 + kSpTW + side - PADX;
                float v = 0.f;
                if (tid < NF && e < NE && c0 + c < a.cin && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = in[(unsigned int)c * uplane + (unsigned int)(gy * a.W + gx)];
                pfe[j] = v;
            }
        }
    };
    auto store_tile = [&](float sc) {
        if (unit) {
            uint4 *dst = tin + (u_row * 2 + u_h) * TWP + COL0 + 4 * u_g;
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                unsigned int hw[4], mw[4], lw[4];
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    const float4 v0 = pf4[2 * jp], v1 = pf4[2 * jp + 1];
                    const float x0 = px == 0 ? v0.x : px == 1 ? v0.y : px == 2 ? v0.z : v0.w;
                    const float x1 = px == 0 ? v1.x : px == 1 ? v1.y : px == 2 ? v1.z : v1.w;
                    if constexpr (F16) {
                        sp16_split2x2(x0 * sc, x1 * sc, hw[jp], lw[jp]);
                        continue;
                    }
#ifdef CT_SPLIT_SCALAR
                    unsigned int h0, m0, l0, h1, m1, l1;
                    split3(x0, h0, m0, l0);
                    split3(x1, h1, m1, l1);
                    hw[jp] = h0 | (h1 << 16); mw[jp] = m0 | (m1 << 16); lw[jp] = l0 | (l1 << 16);
#else
                    split3x2(x0, x1, hw[jp], mw[jp], lw[jp]);
#endif
                }
                dst[px] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
                if constexpr (F16) {
                    dst[PSZ + px] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                } else {
                    dst[PSZ + px] = make_uint4(mw[0], mw[1], mw[2], mw[3]);
                    dst[2 * PSZ + px] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
                }
            }
        }
        if constexpr (HALO) {
            unsigned short *t16 = reinterpret_cast<unsigned short *>(tin);
#pragma unroll
            for (int j = 0; j < PFE; ++j) {
                const int e = (NF - 1 - tid) + j * NF;
                const int c = e / (ROWS * HC), rem = e - c * (ROWS * HC);
                const int yy = rem / HC, side = rem - yy * HC;
                const int col = side < PADX ? COL0 - PADX + side : COL0 + kSpTW + side - PADX;
                if (tid < NF && e < NE) {
                    const int idx = ((yy * 2 + (c >> 3)) * TWP + col) * 8 + (c & 7);
                    if constexpr (F16) {
                        unsigned int hw, lw;
                        sp16_split2x2(pfe[j] * sc, 0.f, hw, lw);
                        t16[idx] = (unsigned short)hw;
                        t16[PSZ * 8 + idx] = (unsigned short)lw;
                    } else {
                        unsigned int h, m, l;
                        split3(pfe[j], h, m, l);
                        t16[idx] = (unsigned short)h;
                        t16[PSZ * 8 + idx] = (unsigned short)m;
                        t16[2 * PSZ * 8 + idx] = (unsigned short)l;
                    }
                }
            }
        }
    };
    // ---- A operand: wave 3 streams the weight fragments of one tap ([piece][m], 1 KiB each) from L2 straight into a
    // four-slot LDS ring with LDS-DMA loads (global_load_lds_dwordx4: no registers), three taps ahead of their use;
    // every wave reads its fragments from there.  (Per-wave register loads would tie the weights to the input-tile
    // fetch through the in-order vmcnt counter: each weight wait would also wait for the HBM latency of the tile loads
    // issued before it; and an L2 round trip is several taps long.)  g = stage * TAPS + tap counts this workgroup's taps.
    const bool wloader = (wave == 3);
    const unsigned int wl_addr = (unsigned int)reinterpret_cast<uintptr_t>(wl);   // LDS byte address (low half of the flat address)
    const int n_gtaps = n_stages * TAPS;
    // issue cursor of the loader wave (taps are issued strictly in order): segment / chunk, tap within the stage, ring slot, and the
    // byte offset of the tap's fragments -- (chunk, tap) are contiguous within a group, so it only steps, except at a new unit
    SegCur wcur;
    seg_set(wcur, 0);
    int wi_tap = 0, wi_slot = 0;
    auto w_unit_off = [&](const SegCur &c) -> unsigned long long {
        return (((unsigned long long)(c.unit % a.groups) * n_chunks + c.chunk) * TAPS) * (unsigned long long)(NP * MT * 1024);
    };
    unsigned long long w_off = w_unit_off(wcur);
    const char *wlane = reinterpret_cast<const char *>(a.wp) + lane * 16;
#ifdef CT_SPLIT_ABL_NOW
    int wi_n = 0;
#endif
    auto w_issue = [&]() {                                 // the loader wave only
        const char *src = wlane + w_off;
        const unsigned int dst = wl_addr + (unsigned int)(wi_slot * WSLOT * 16);   // + 16 * lane is implied by the instruction
#ifdef CT_SPLIT_ABL_NOW
        if (wi_n++ < TAPS)                                 // diagnostic: only the first stage's weights are ever loaded
#endif
#pragma unroll
        for (int f = 0; f < NP * MT; ++f) glds16(src + f * 1024, __builtin_amdgcn_readfirstlane(dst + f * 1024));
        w_off += NP * MT * 1024;
        wi_slot = (wi_slot + 1 == WRING) ? 0 : wi_slot + 1;
        if (++wi_tap == TAPS) {
            wi_tap = 0;
            const int seg0 = wcur.seg;
            seg_next(wcur);
            if (wcur.seg != seg0) w_off = w_unit_off(wcur);
        }
    };
    // before the barrier that ends tap g: the fragments of tap g+1 have landed (loads of later taps may be in flight)
    auto w_landed = [&](int g) {
        // vmcnt(ahead * NP * MT): the counter has six bits, [3:0] and [15:14] of the immediate
#define CT_VMCNT(n) __builtin_amdgcn_s_waitcnt(0x0F70 | ((n) & 15) | (((n) >> 4) << 14))
        if (g + WAHEAD < n_gtaps) {                            // all but the workgroup's last taps: WAHEAD - 1 taps were issued after tap g+1
            CT_VMCNT((WAHEAD - 1) * NP * MT);
            return;
        }
        const int ahead = n_gtaps - g - 2;                    // taps issued after tap g+1 (their loads may still be in flight)
        if constexpr (WAHEAD == 3) {
            if (ahead == 1) CT_VMCNT(NP * MT);
            else CT_VMCNT(0);
        } else {
            switch (ahead) {
                case 5: CT_VMCNT(5 * NP * MT); break;
                case 4: CT_VMCNT(4 * NP * MT); break;
                case 3: CT_VMCNT(3 * NP * MT); break;
                case 2: CT_VMCNT(2 * NP * MT); break;
                case 1: CT_VMCNT(NP * MT); break;
                default: CT_VMCNT(0); break;
            }
        }
#undef CT_VMCNT
    };

#ifdef CT_CONV_PROFILE
    // diagnostic build (make prof, tools/prof_conv_split.py): per-phase cycle totals of every wave 0 -> a.prof[block][8]
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0;
#define CT_STAMP(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define CT_PHASE(i) do { unsigned long long t__; CT_STAMP(t__); pt[i] += t__ - pt0; pt0 = t__; } while (0)
    CT_STAMP(pt0);
#else
#define CT_PHASE(i) do { } while (0)
#endif
    f32x16s acc[RPW][MT];
    // phase stagger of the two co-resident workgroups (see conv_mfma_kernel)
    if (n_stages >= 2 * n_chunks) {
        const unsigned int hw_wave_slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((4 - 1) << 11));   // HW_ID[3:0]
        if (hw_wave_slot & 1) {
            const int half_tile_cycles = n_chunks * TAPS * RPW * MT * 6 * 32 / 2;
            for (int c = 0; c < half_tile_cycles; c += 64 * 64) __builtin_amdgcn_s_sleep(64);
        }
    }
    const bool res_in_acc = !F16 && (a.residual != nullptr) && (a.act == 0 || a.res_pre);
    // F16: exponent of the tile in LDS / of the accumulators' domain; the maxima of the fetched tile (waves 0..2 fetch)
    int e_stage = 100, e_cur = 100;
    auto note_max = [&]() {
        float mx = 0.f;
        if (unit) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                mx = fmaxf(mx, fmaxf(fmaxf(fabsf(pf4[j].x), fabsf(pf4[j].y)), fmaxf(fabsf(pf4[j].z), fabsf(pf4[j].w))));
        }
        if constexpr (HALO) {
#pragma unroll
            for (int j = 0; j < PFE; ++j) mx = fmaxf(mx, fabsf(pfe[j]));
        }
        mx = sp16_wave_max(mx);
        if (lane == 0) mxw[wave] = mx;
    };
    auto tile_exp = [&]() {
        return __builtin_amdgcn_readfirstlane(sp16_scale_exp(fmaxf(fmaxf(mxw[0], mxw[1]), fmaxf(mxw[2], mxw[3])), 100));
    };
    auto init_acc = [&](int tg) {  // raw float4 rows of the skip tensor (or zeros) of unit tg; finish_acc() re-lays them out
        const int t = tg / a.groups, grp = tg - t * a.groups;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const float *__restrict__ res = res_in_acc ? a.residual + (size_t)n * a.res_bstride + (size_t)grp * COUTP * plane : nullptr;
        const int cout_g = a.cout - grp * COUTP;
        const int x4 = tx * kSpTW + 4 * (lane & 7);
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const int y = ty * kSpTH + wave * RPW + q;
            const bool inb = res_in_acc && (y < a.H) && (x4 < a.W);
            const unsigned int pix = (unsigned int)(y * a.W + x4);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int co = m * 32 + (lane >> 3) + 8 * j;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (inb && co < cout_g) v = *reinterpret_cast<const float4 *>(res + (unsigned int)co * uplane + pix);
                    acc[q][m][4 * j + 0] = v.x; acc[q][m][4 * j + 1] = v.y;
                    acc[q][m][4 * j + 2] = v.z; acc[q][m][4 * j + 3] = v.w;
                }
        }
    };
    auto finish_acc = [&]() {
        if (!res_in_acc) return;      // zeros need no re-layout
#pragma unroll
        for (int q = 0; q < RPW; ++q)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4 *>(stg + ((lane >> 3) + 8 * j) * 32 + 4 * (lane & 7)) =
                        make_float4(acc[q][m][4 * j], acc[q][m][4 * j + 1], acc[q][m][4 * j + 2], acc[q][m][4 * j + 3]);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[q][m][r] = stg[((r & 3) + 8 * (r >> 2) + 4 * hl) * 32 + nl];
                __builtin_amdgcn_wave_barrier();
            }
    };

    if (wloader) {
        for (int g0 = 0; g0 < WAHEAD && g0 < n_gtaps; ++g0) w_issue();
        w_landed(-1);
    }
    SegCur cc;                     // the stage being computed (its successor's tile is fetched under its MFMAs)
    seg_set(cc, 0);
    fetch_tile(cc);
    if constexpr (F16) {
        note_max();
        __syncthreads();
        e_stage = tile_exp();
        store_tile(sp16_pow2i(e_stage));
    } else {
        init_acc(cc.unit);
        store_tile(1.f);
    }
    for (int stage = 0; stage < n_stages; ++stage, seg_next(cc)) {
        const int seg_k = seg_kind(cc.seg);
        const bool seg_first = (cc.chunk == (seg_k == 2 ? sk_ca : 0)), seg_last = (cc.chunk + 1 == cc.ce);
        const int grp = cc.unit % a.groups;
        if constexpr (F16) {
            if (seg_first) {
#pragma unroll
                for (int q = 0; q < RPW; ++q)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;
            } else if (e_stage != e_cur) {             // the running scale dropped: move the accumulators to the new domain
                const float rs = __builtin_amdgcn_ldexpf(1.0f, e_stage - e_cur);
#pragma unroll
                for (int q = 0; q < RPW; ++q)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[q][m][r] *= rs;
            }
            e_cur = e_stage;
        } else if (seg_first) {
            finish_acc();
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float bv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bv[r] = a.bias[grp * COUTP + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl];   // padded
#pragma unroll
                for (int q = 0; q < RPW; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][m][r] += bv[r];
            }
        }
        CT_PHASE(0);                                       // store_tile / accumulator init
        __syncthreads();                                   // this stage's tile is visible
        CT_PHASE(1);
        const bool next_stage = (stage + 1 < n_stages);
        // B fragments of tap t: [q][piece]; read one tap ahead of the MFMAs that consume them
        uint4 bc[RPW][NP], bn[RPW][NP];
        auto read_b = [&](int tap, uint4 (&b)[RPW][NP]) {
            const int ky = tap / KW, kx = tap - ky * KW;
            const uint4 *bp = tin + ((wave * RPW + ky) * 2 + hl) * TWP + (COL0 - PADX) + kx + nl;
#pragma unroll
            for (int q = 0; q < RPW; ++q)
#pragma unroll
                for (int p = 0; p < NP; ++p) b[q][p] = bp[p * PSZ + q * 2 * TWP];
        };
        auto mfma6 = [&](const uint4 *ws, const uint4 (&b)[RPW][NP]) {
            uint4 w[NP][MT];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int m = 0; m < MT; ++m) w[p][m] = ws[(p * MT + m) * 64];
            if constexpr (F16) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f16x8s ah = __builtin_bit_cast(f16x8s, w[0][m]), al = __builtin_bit_cast(f16x8s, w[1][m]);
#pragma unroll
                    for (int q = 0; q < RPW; ++q) {        // small terms first
                        const f16x8s bh = __builtin_bit_cast(f16x8s, b[q][0]), bl = __builtin_bit_cast(f16x8s, b[q][1]);
                        acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[q][m], 0, 0, 0);
                        acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[q][m], 0, 0, 0);
                        acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[q][m], 0, 0, 0);
                    }
                }
            } else {
            bf16x8 bf[RPW][3];
#pragma unroll
            for (int q = 0; q < RPW; ++q)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[q][p] = __builtin_bit_cast(bf16x8, b[q][p]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, w[0][m]), am = __builtin_bit_cast(bf16x8, w[1][m]),
                             al = __builtin_bit_cast(bf16x8, w[2][m]);
#pragma unroll
                for (int q = 0; q < RPW; ++q) {            // small terms first
                    acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bf[q][0], acc[q][m], 0, 0, 0);
                    acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[q][2], acc[q][m], 0, 0, 0);
                    acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[q][1], acc[q][m], 0, 0, 0);
                    acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bf[q][0], acc[q][m], 0, 0, 0);
                    acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[q][1], acc[q][m], 0, 0, 0);
                    acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bf[q][0], acc[q][m], 0, 0, 0);
                }
            }
            }
        };
        read_b(0, bc);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const bool last_tap = (tap + 1 == TAPS);
            const int g = stage * TAPS + tap;
            if (wloader && g + WAHEAD < n_gtaps) w_issue();  // tap g + WAHEAD; its slot was last read in tap g-1: free since that barrier
            if (tap == 0 && next_stage) {                    // next halo tile: in flight under this stage's MFMAs
                SegCur fc = cc;
                seg_next(fc);
                fetch_tile(fc);
            }
            if (!last_tap) read_b(tap + 1, bn);
            mfma6(wl + (g % WRING) * WSLOT + lane, bc);
#pragma unroll
            for (int q = 0; q < RPW; ++q)
#pragma unroll
                for (int p = 0; p < NP; ++p) bc[q][p] = bn[q][p];
            CT_PHASE(2);                                     // tap work: LDS operand reads + MFMAs (+ prefetch issue)
            if (wloader) { w_landed(g); CT_PHASE(6); }
            if (!last_tap) __syncthreads();                  // weight slot g%4 is free, slot (g+1)%4 is published
            CT_PHASE(5);
        }
        if (F16 && seg_last && seg_k == 1) {
            // ---- stream-K hand-off: the head piece's partial sums, unscaled, [(q, m, r / 2)][thread] float2 into this workgroup's
            // slot; write-through (agent-scope) stores, drained, then the flag
            const int un = -(e_cur + a.w_exp);
            unsigned long long *slot = reinterpret_cast<unsigned long long *>(a.sk_ws) + (size_t)blockIdx.x * (RPW * MT * 8 * 256) + tid;
            static_assert(RPW == 2 && MT == 2, "block select below");
#pragma unroll 1
            for (int qm = 0; qm < RPW * MT; ++qm) {            // rolled, the block picked with selects (register pressure, code size)
#pragma unroll
                for (int r2 = 0; r2 < 8; ++r2) {
                    const float v0 = qm == 0 ? acc[0][0][2 * r2] : qm == 1 ? acc[0][1][2 * r2] : qm == 2 ? acc[1][0][2 * r2] : acc[1][1][2 * r2];
                    const float v1 = qm == 0 ? acc[0][0][2 * r2 + 1] : qm == 1 ? acc[0][1][2 * r2 + 1] : qm == 2 ? acc[1][0][2 * r2 + 1] : acc[1][1][2 * r2 + 1];
                    const unsigned long long lo = __float_as_uint(__builtin_amdgcn_ldexpf(v0, un));
                    const unsigned long long hi = __float_as_uint(__builtin_amdgcn_ldexpf(v1, un));
                    __hip_atomic_store(slot + (qm * 8 + r2) * 256, lo | (hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) (void)__hip_atomic_exchange(a.sk_flags + blockIdx.x, a.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (seg_last) {
            // ---- epilogue: lane owns pixels (y0 + wave*RPW + q, x0+nl), channels (r&3)+8(r>>2)+4hl of each 32-tile ----
            const bool take = F16 && seg_k == 2;      // stream-K tail piece: the neighbour's head partial is added below
            const unsigned long long *part = nullptr;
            if (take) {
                if (tid == 0) {
                    // bounded (1 s of s_memrealtime at 100 MHz): a producer that never shows up -- it cannot happen while all
                    // workgroups of the launch are resident -- costs a wrong tile and a count in the error word, not a hung queue
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    bool seen;
                    while (!(seen = __hip_atomic_load(a.sk_flags + (blockIdx.x - 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.sk_epoch) &&
                           __builtin_amdgcn_s_memrealtime() - t0 < 100000000ull)
                        __builtin_amdgcn_s_sleep(4);
                    if (!seen) {
                        (void)__hip_atomic_fetch_add(a.sk_flags + kSkErrWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        (void)__hip_atomic_fetch_or(&g_sk_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    (void)__hip_atomic_exchange(a.sk_flags + (blockIdx.x - 8), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next launch
                }
                __syncthreads();
                // read below with agent-scope loads (they bypass this XCD's L2 for these addresses only); an acquire fence would
                // invalidate the whole L2 under the other workgroups' input tiles and weights
                part = reinterpret_cast<const unsigned long long *>(a.sk_ws) + (size_t)(blockIdx.x - 8) * (RPW * MT * 8 * 256) + tid;
            }
            const int t = cc.unit / a.groups;
            const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
            float *__restrict__ out = a.out + (size_t)n * a.out_bstride + (size_t)grp * COUTP * plane;
            const float *__restrict__ res = a.residual ? a.residual + (size_t)n * a.res_bstride + (size_t)grp * COUTP * plane : nullptr;
            const bool late_res = (res != nullptr) && !res_in_acc;
            const int cout_g = a.cout - grp * COUTP;
            const bool full = (cout_g >= COUTP);
            const int x4 = tx * kSpTW + 4 * (lane & 7);
            if constexpr (F16) {
                // unscale, bias, pre-activation addend, activation, skip, clamp -- all on the float32 value
                const int un = -(e_cur + a.w_exp);
                const float *__restrict__ bias_g = a.bias + grp * COUTP;
                if (a.rows_channels > 0) {
                    float *__restrict__ orow = a.out + (size_t)n * plane * a.rows_channels + a.rows_c0 + grp * COUTP;
                    const int x = tx * kSpTW + nl;
                    static_assert(RPW == 2 && MT == 2, "block select below");
#pragma unroll 1
                    for (int qm = 0; qm < RPW * MT; ++qm) {            // rolled like the NCHW form below (code size)
                        const int q = qm >> 1, m = qm & 1;
                        const int y = ty * kSpTH + wave * RPW + q;
                        f32x16s blk;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            blk[r] = qm == 0 ? acc[0][0][r] : qm == 1 ? acc[0][1][r] : qm == 2 ? acc[1][0][r] : acc[1][1][r];
                        if (y < a.H && x < a.W) {
                            float *__restrict__ op = orow + (size_t)(y * a.W + x) * a.rows_channels;
#pragma unroll 1
                            for (int jj = 0; jj < 4; ++jj) {
                                const int co = m * 32 + 8 * jj + 4 * hl;
                                const float4 b4 = *reinterpret_cast<const float4 *>(bias_g + co);
                                const float r0 = jj == 0 ? blk[0] : jj == 1 ? blk[4] : jj == 2 ? blk[8] : blk[12];
                                const float r1 = jj == 0 ? blk[1] : jj == 1 ? blk[5] : jj == 2 ? blk[9] : blk[13];
                                const float r2 = jj == 0 ? blk[2] : jj == 1 ? blk[6] : jj == 2 ? blk[10] : blk[14];
                                const float r3 = jj == 0 ? blk[3] : jj == 1 ? blk[7] : jj == 2 ? blk[11] : blk[15];
                                float4 v = make_float4(__builtin_amdgcn_ldexpf(r0, un) + b4.x, __builtin_amdgcn_ldexpf(r1, un) + b4.y,
                                                       __builtin_amdgcn_ldexpf(r2, un) + b4.z, __builtin_amdgcn_ldexpf(r3, un) + b4.w);
                                if (a.act) {
                                    v.x = split_act<GEN>(v.x, a.act); v.y = split_act<GEN>(v.y, a.act);
                                    v.z = split_act<GEN>(v.z, a.act); v.w = split_act<GEN>(v.w, a.act);
                                }
                                if (full || co < cout_g) *reinterpret_cast<float4 *>(op + co) = v;
                            }
                        }
                    }
                } else {
                    // ROLLED over the (row, 32-channel tile) blocks and over the four 8-channel groups of a block: the generic
                    // activation (expf / tanhf inline) was instantiated 64 times here -- 100 KB of code streaming through the 64 KB
                    // instruction cache once per output tile, evicting the tap loop of this and the neighbouring workgroups.  The
                    // block's accumulator is picked with selects (the registers cannot be indexed dynamically).
                    static_assert(RPW == 2 && MT == 2, "block select below");
#pragma unroll 1
                    for (int qm = 0; qm < RPW * MT; ++qm) {
                        const int q = qm >> 1, m = qm & 1;
                        const int y = ty * kSpTH + wave * RPW + q;
                        f32x16s blk;
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            blk[r] = qm == 0 ? acc[0][0][r] : qm == 1 ? acc[0][1][r] : qm == 2 ? acc[1][0][r] : acc[1][1][r];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ch = (r & 3) + 8 * (r >> 2) + 4 * hl;
                            stg[ch * 32 + nl] = __builtin_amdgcn_ldexpf(blk[r], un) + bias_g[m * 32 + ch];
                        }
                        if (take) {
#pragma unroll
                            for (int r2 = 0; r2 < 8; ++r2) {
                                const unsigned long long pv = __hip_atomic_load(part + (qm * 8 + r2) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const int r = 2 * r2, ch = (r & 3) + 8 * (r >> 2) + 4 * hl;      // r even: r + 1 is the next channel
                                stg[ch * 32 + nl] += __uint_as_float((unsigned int)pv);
                                stg[(ch + 1) * 32 + nl] += __uint_as_float((unsigned int)(pv >> 32));
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
                        for (int j = 0; j < 4; ++j) {
                            const int co = m * 32 + (lane >> 3) + 8 * j;
                            float4 v = *reinterpret_cast<const float4 *>(stg + ((lane >> 3) + 8 * j) * 32 + 4 * (lane & 7));
                            if (y < a.H && x4 < a.W && (full || co < cout_g)) {
                                const unsigned int o = (unsigned int)co * uplane + (unsigned int)(y * a.W + x4);
                                float4 rr = make_float4(0.f, 0.f, 0.f, 0.f);
                                if (res) rr = *reinterpret_cast<const float4 *>(res + o);
                                if (a.res_pre) { v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
                                if (a.act) {
                                    v.x = split_act<GEN>(v.x, a.act); v.y = split_act<GEN>(v.y, a.act);
                                    v.z = split_act<GEN>(v.z, a.act); v.w = split_act<GEN>(v.w, a.act);
                                }
                                if (!a.res_pre) { v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
                                if (a.post_op) {
                                    const size_t po = (size_t)n * a.out_bstride + (size_t)grp * COUTP * plane + o;
                                    const float4 q1 = *reinterpret_cast<const float4 *>(a.p1 + po);
                                    if (a.post_op == 1) {
                                        v.x *= q1.x; v.y *= q1.y; v.z *= q1.z; v.w *= q1.w;
                                    } else {
                                        const float4 q2 = *reinterpret_cast<const float4 *>(a.p2 + po);
                                        v.x = (1.0f - q1.x) * q2.x + q1.x * v.x; v.y = (1.0f - q1.y) * q2.y + q1.y * v.y;
                                        v.z = (1.0f - q1.z) * q2.z + q1.z * v.z; v.w = (1.0f - q1.w) * q2.w + q1.w * v.w;
                                    }
                                }
                                if (a.clamp) {
                                    v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
                                    v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
                                }
                                *reinterpret_cast<float4 *>(out + o) = v;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            } else if (a.rows_channels > 0) {
                // token-rows output [N*H, W, rows_channels]: the lane's four consecutive channels of each 8-group are one 16-byte
                // store at its pixel (no staging); the attention kernels read this layout, so the NCHW tensor and its transpose
                // never exist
                float *__restrict__ orow = a.out + (size_t)n * plane * a.rows_channels + a.rows_c0 + grp * COUTP;
                const int x = tx * kSpTW + nl;
#pragma unroll
                for (int q = 0; q < RPW; ++q) {
                    const int y = ty * kSpTH + wave * RPW + q;
                    if (y < a.H && x < a.W) {
                        float *__restrict__ op = orow + (size_t)(y * a.W + x) * a.rows_channels;
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int co = m * 32 + 8 * j + 4 * hl;
                                float4 v = make_float4(acc[q][m][4 * j], acc[q][m][4 * j + 1], acc[q][m][4 * j + 2], acc[q][m][4 * j + 3]);
                                if (a.act) {
                                    v.x = split_act<GEN>(v.x, a.act); v.y = split_act<GEN>(v.y, a.act);
                                    v.z = split_act<GEN>(v.z, a.act); v.w = split_act<GEN>(v.w, a.act);
                                }
                                if (full || co < cout_g) *reinterpret_cast<float4 *>(op + co) = v;
                            }
                        }
                    }
                }
            } else
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const int y = ty * kSpTH + wave * RPW + q;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[q][m][r];
                        if (a.act) v = split_act<GEN>(v, a.act);
                        if (a.clamp && !late_res) v = fminf(fmaxf(v, 0.f), 1.f);
                        stg[((r & 3) + 8 * (r >> 2) + 4 * hl) * 32 + nl] = v;
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = m * 32 + (lane >> 3) + 8 * j;
                        float4 v = *reinterpret_cast<const float4 *>(stg + ((lane >> 3) + 8 * j) * 32 + 4 * (lane & 7));
                        if (y < a.H && x4 < a.W && (full || co < cout_g)) {
                            const unsigned int o = (unsigned int)co * uplane + (unsigned int)(y * a.W + x4);
                            if (late_res) {
                                const float4 rr = *reinterpret_cast<const float4 *>(res + o);
                                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                                if (a.clamp) {
                                    v.x = fminf(fmaxf(v.x, 0.f), 1.f); v.y = fminf(fmaxf(v.y, 0.f), 1.f);
                                    v.z = fminf(fmaxf(v.z, 0.f), 1.f); v.w = fminf(fmaxf(v.w, 0.f), 1.f);
                                }
                            }
                            *reinterpret_cast<float4 *>(out + o) = v;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if constexpr (!F16) {
                if (next_stage) {
                    SegCur fc = cc;
                    seg_next(fc);
                    init_acc(fc.unit);
                }
            }
        }
        if constexpr (F16) { if (next_stage) note_max(); }   // waits for the tile loads issued at tap 0
        CT_PHASE(3);                       // epilogue
        __syncthreads();                   // every wave is done reading this stage's tile (F16: the next tile's maxima are visible)
        CT_PHASE(4);
        if constexpr (F16) {
            if (next_stage) {
                const int et = tile_exp();
                e_stage = seg_last ? et : min(e_stage, et);                    // a new output tile (or piece) starts its own running scale
                store_tile(sp16_pow2i(e_stage));
            }
        } else {
            if (next_stage) store_tile(1.f);
        }
    }
#ifdef CT_CONV_PROFILE
    CT_PHASE(0);
    if (lane == 0 && (wave == 0 || wave == 3) && a.prof) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.prof[((size_t)(wave ? 512 : 0) + blockIdx.x) * 8 + i] = pt[i];
    }
#endif
}

// stream-K scratch: flags (one 32-bit word per workgroup, 4 KiB reserved), then one partial tile (256 threads x 64 floats) per workgroup
constexpr size_t kSkFlagBytes = 4096;
constexpr size_t kSkSlotBytes = 256 * 64 * sizeof(float);
constexpr size_t kSkBytes = kSkFlagBytes + (size_t)2 * kSpCUs * kSkSlotBytes;

template <int KH, int KW, bool GEN, bool F16 = false>
static int launch_split(const ConvArgs &a_in, int N, hipStream_t s) {
    ConvArgs a = a_in;
    constexpr int ROWS = kSpTH + KH - 1, TWP = (KW > 1) ? kSpTW + 8 : kSpTW, NP = F16 ? 2 : 3;
    const size_t lds = (size_t)NP * ROWS * 2 * TWP * 16 + (size_t)(F16 ? 8 : 4) * NP * 2 * 64 * 16 + (size_t)4 * 32 * 32 * sizeof(float) + 16;
    const int tiles_x = (a.W + kSpTW - 1) / kSpTW, tiles_y = (a.H + kSpTH - 1) / kSpTH;
    const long long n_tiles = (long long)tiles_x * tiles_y * N * a.groups;
    if (n_tiles > 0x7fffffffLL) return CT_E_BADARG;
    static const int wgs_per_cu = [] { const char *e = getenv("CT_HIP_SPLIT_WGS"); int v = e ? atoi(e) : 0; return v > 0 ? v : 2; }();
    const int grid = n_tiles < wgs_per_cu * kSpCUs ? (int)n_tiles : wgs_per_cu * kSpCUs;
    // stream-K only where whole units quantise badly: more units than resident workgroups and a last round that is mostly idle
    static const bool sk_on = [] { const char *e = getenv("CT_HIP_SPLIT_SK"); return !(e && atoi(e) == 0); }();
    const double rounds = (double)n_tiles / (double)grid;
    if (!(F16 && sk_on && a.sk_ws && a.rows_channels == 0 && wgs_per_cu == 2 && grid == 2 * kSpCUs && n_tiles > grid &&
          (double)((n_tiles + grid - 1) / grid) > 1.08 * rounds)) {
        a.sk_ws = nullptr;
        a.sk_flags = nullptr;
    } else {
        // flags carry a per-launch epoch (process-wide counter, never 0).  A replayed graph repeats its epoch; there the
        // consumer's clearing store is what separates the replays, as before
        static std::atomic<unsigned int> epoch{0};
        unsigned int e = epoch.fetch_add(1, std::memory_order_relaxed) + 1u;
        if (e == 0u) e = epoch.fetch_add(1, std::memory_order_relaxed) + 1u;
        a.sk_epoch = e;
    }
    hipLaunchKernelGGL((conv_split_kernel<KH, KW, GEN, F16>), dim3(grid), dim3(256), lds, s, a, tiles_x, tiles_y, (int)n_tiles);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

// 1 = this geometry / alignment has no split-bf16 kernel (the caller uses the exact-f32 one)
int conv_split(const ConvArgs &a, int N, int kh, int kw, bool gen, hipStream_t s) {
    const bool vec = (a.W % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.in) & 15) == 0) && (a.in_bstride % 4 == 0) &&
                     (!a.in2 || (((reinterpret_cast<uintptr_t>(a.in2) & 15) == 0) && (a.in2_bstride % 4 == 0) && (a.cin1 % 16 == 0))) &&
                     (!a.in3 || (((reinterpret_cast<uintptr_t>(a.in3) & 15) == 0) && (a.in3_bstride % 4 == 0) && (a.cin2 % 16 == 0))) &&
                     ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0) && (a.out_bstride % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.wp) & 15) == 0) &&
                     (!a.residual || (((reinterpret_cast<uintptr_t>(a.residual) & 15) == 0) && (a.res_bstride % 4 == 0)));
    if (!vec) return 1;
    if (a.rows_channels > 0 && ((a.rows_channels & 3) || (a.rows_c0 & 3) || (a.cout & 3) || a.rows_c0 + a.cout > a.rows_channels || a.residual || a.clamp))
        return CT_E_BADARG;
    if (kh == 3 && kw == 3 && a.rows_channels == 0 && !a.res_pre && !a.post_op) {
        const int rc = conv_ws(a, N, gen, s);
        if (rc != 1) return rc;
    }
    if (a.f16) {
        if (kh == 3 && kw == 3) return launch_split<3, 3, true, true>(a, N, s);
        if (kh == 1 && kw == 1) return launch_split<1, 1, true, true>(a, N, s);
        if (kh == 1 && kw == 5) return launch_split<1, 5, true, true>(a, N, s);
        if (kh == 5 && kw == 1) return launch_split<5, 1, true, true>(a, N, s);
        if (kh == 2 && kw == 2) return launch_split<2, 2, true, true>(a, N, s);     // taps at offsets -1 / 0: a stride-2 3x3 after space-to-depth
        return 1;
    }
    if (gen) {
        if (kh == 3 && kw == 3) return launch_split<3, 3, true>(a, N, s);
        if (kh == 1 && kw == 1) return launch_split<1, 1, true>(a, N, s);
        if (kh == 1 && kw == 5) return launch_split<1, 5, true>(a, N, s);
        if (kh == 5 && kw == 1) return launch_split<5, 1, true>(a, N, s);
    } else {
        if (kh == 3 && kw == 3) return launch_split<3, 3, false>(a, N, s);
        if (kh == 1 && kw == 1) return launch_split<1, 1, false>(a, N, s);
    }
    return 1;
}

// sticky status of the current device: 1 = a stream-K consumer gave up waiting for its producer since the last clear (-1: unreadable)
int conv_split_read_status(bool clear) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_sk_status), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (clear && v != 0) {
        const unsigned z = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sk_status), &z, sizeof(z), 0, hipMemcpyHostToDevice);
    }
    return (int)(v & 1u);
}

}  // namespace ct

extern "C" {

size_t ct_conv_split_scratch_bytes(void) { return ct::kSkBytes; }

int ct_conv2d_split_f32(const float *in, const float *in2, int cin1, const float *in3, int cin2, const void *wp_split, const float *bias,
                        const float *residual, float *out, int n, int cin, int cout, int h, int w, int kh, int kw, long long in_bstride,
                        long long in2_bstride, long long in3_bstride, long long out_bstride, long long res_bstride, int act, int clamp,
                        int res_pre_act, int f16, int w_exp, int post_op, const float *p1, const float *p2, void *scratch,
                        long long scratch_bytes, void *stream) {
    if (!in || !wp_split || !bias || !out || n < 0 || cin < 1 || cout < 1 || h < 0 || w < 0 || act < 0 || act > 5) return CT_E_BADARG;
    if (post_op < 0 || post_op > 2 || (post_op && (!f16 || !p1 || (post_op == 2 && !p2)))) return CT_E_BADARG;
    if (post_op && ((reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 15)) return CT_E_ALIGN;
    if (in2 && (cin1 < 16 || cin1 >= cin || (cin1 % 16))) return CT_E_BADARG;
    if (in3 && (!in2 || cin2 <= cin1 || cin2 >= cin || (cin2 % 16))) return CT_E_BADARG;
    if (scratch && ((reinterpret_cast<uintptr_t>(scratch) & 15) || scratch_bytes < (long long)ct::kSkBytes)) return CT_E_WORKSPACE;
    if (n == 0 || h == 0 || w == 0) return CT_OK;
    ct::ConvArgs a;
    a.in = in; a.in2 = in2; a.cin1 = in2 ? cin1 : cin; a.in2_bstride = in2_bstride;
    a.in3 = in3; a.cin2 = in3 ? cin2 : cin; a.in3_bstride = in3_bstride;
    a.wp = reinterpret_cast<const float *>(wp_split); a.bias = bias; a.residual = residual; a.out = out;
    a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.in_bstride = in_bstride; a.out_bstride = out_bstride; a.res_bstride = res_bstride;
    a.act = act; a.clamp = clamp; a.groups = (cout + 63) / 64; a.prof = nullptr;
    a.res_pre = (residual && res_pre_act) ? 1 : 0;
    a.f16 = f16 ? 1 : 0; a.w_exp = f16 ? w_exp : 0;
    a.post_op = post_op; a.p1 = p1; a.p2 = p2;
    if (scratch && f16) {
        a.sk_flags = reinterpret_cast<unsigned int *>(scratch);
        a.sk_ws = reinterpret_cast<float *>(reinterpret_cast<char *>(scratch) + ct::kSkFlagBytes);
    }
    const int rc = ct::conv_split(a, n, kh, kw, true, (hipStream_t)stream);
    return rc == 1 ? CT_E_BADARG : rc;
}

int ct_conv2d_split_rows_f32(const float *in, const void *wp_split, const float *bias, float *out_rows, int n, int cin, int cout, int h,
                             int w, int kh, int kw, long long in_bstride, int rows_channels, int rows_c0, int act, int f16, int w_exp,
                             void *stream) {
    if (!in || !wp_split || !bias || !out_rows || n < 0 || cin < 1 || cout < 1 || h < 0 || w < 0 || act < 0 || act > 5) return CT_E_BADARG;
    if (rows_channels < 4 || rows_c0 < 0) return CT_E_BADARG;
    if (n == 0 || h == 0 || w == 0) return CT_OK;
    ct::ConvArgs a;
    a.in = in; a.in2 = nullptr; a.cin1 = cin; a.in2_bstride = 0;
    a.wp = reinterpret_cast<const float *>(wp_split); a.bias = bias; a.residual = nullptr; a.out = out_rows;
    a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.in_bstride = in_bstride; a.out_bstride = 0; a.res_bstride = 0;
    a.act = act; a.clamp = 0; a.groups = (cout + 63) / 64; a.prof = nullptr;
    a.rows_channels = rows_channels; a.rows_c0 = rows_c0;
    a.f16 = f16 ? 1 : 0; a.w_exp = f16 ? w_exp : 0;
    const int rc = ct::conv_split(a, n, kh, kw, true, (hipStream_t)stream);
    return rc == 1 ? CT_E_BADARG : rc;
}

#ifdef CT_CONV_PROFILE
int ct_conv2d_split_prof_f32(const float *in, const void *wp_split, const float *bias, const float *residual, float *out, int n,
                             int cin, int cout, int h, int w, unsigned long long *prof, int f16, int w_exp, int kh, int kw, void *stream) {
    ct::ConvArgs a;
    a.in = in; a.in2 = nullptr; a.cin1 = cin; a.in2_bstride = 0;
    a.wp = reinterpret_cast<const float *>(wp_split); a.bias = bias; a.residual = residual; a.out = out;
    a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.in_bstride = (long long)cin * h * w; a.out_bstride = (long long)cout * h * w; a.res_bstride = a.out_bstride;
    a.act = 0; a.clamp = 0; a.groups = (cout + 63) / 64; a.prof = prof;
    a.f16 = f16; a.w_exp = w_exp;
    return ct::conv_split(a, n, kh, kw, f16 != 0, (hipStream_t)stream);
}
#endif

}  // extern "C"
