// conv_wino4.hip -- the ResB convolution (3x3, stride 1, "same", 32 < cin <= 64) as Winograd F(2x2, 3x3) on two fp16 pieces, round 6:
// FOUR waves of 512 registers instead of conv_wino.hip's eight of 256, no input ring, two V images, one software pipeline.
//
// Why: conv_wino.hip runs transform -> matrix -> output transform as three barrier-separated phases
// of the same eight waves; half of every step is latency, and with 128 of a wave's 256 registers holding weights nothing can be
// carried across a phase (both round-5 variants that tried died of spills).  Here a wave has the whole file of its SIMD:
//   * the transformed weights of its FOUR positions (one row i of the 4x4 position grid: 4 x 64 cout x 64 cin x two pieces = 256
//     registers) live in the accumulation registers a[0:255] and feed v_mfma_f32_16x16x32_f16 directly as its A operand
//     (inline asm with "a" constraints: through the builtin the allocator parks them in AGPRs and copies them back, 4 moves per use);
//   * the 256 architectural registers carry the pipeline: the 4x4 input patches of the NEXT step straight from global memory (two
//     row pairs resident, the third in flight: no LDS ring, so LDS holds TWO 64 KB V images), 32 accumulators, fragments;
//   * EVERY matrix instruction is followed by a slice of vector work that does not depend on it (one wave per SIMD has nobody else
//     to fill the 16 cycles; tools/ubench/mfma_agpr_valu.hip: an MFMA + k vector instructions issue in 11 + 4 k cycles):
//       phase X (step s)   the 48 MFMAs of cout blocks 2, 3 of step s, interleaved with M A of blocks 0, 1 (4 adds per element in
//                          the accumulators' lanes -- the wave owns a ROW of positions, which halves what crosses LDS) and the whole
//                          input transform of step s + 1 (V = B^T d B -> scale -> fp16 hi / lo -> the other V image);
//                          then M A of blocks 2, 3 over the wave's own, consumed, fragments          [barrier]
//       phase Y            the 48 MFMAs of cout blocks 0, 1 of step s + 1 (its V image is complete), interleaved with the output side
//                          of step s: every thread takes (two tiles, 2 couts): A^T (M A), unscale + bias in one fma, activation,
//                          skip, 16-byte stores (whole 128-byte lines); then the maximum of the freshly landed row pair  [barrier]
// Same arithmetic contract as conv_wino.hip (float32-grade, tools/model_winograd_two_piece.py); the sums are taken in a different
// order (row of positions first), so results are not bitwise those of conv_wino.hip.  Weight image: ct_hip.pack_conv_weight_wino16, unchanged.
#include "ct_common.h"
#include "ct_conv.h"
#include "ct_split.h"
#include <type_traits>
#include <utility>

namespace ct {
namespace w4 {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));     // a 4-byte aligned 16-byte global load (one global_load_dwordx4)

constexpr int kThreads = 256;
constexpr int kTW = 32;                                   // output columns of a strip = 16 tiles of 2x2
constexpr int kVWords = 16384;                            // 32-bit words of one V image: [position 16][cin chunk 2][piece 2][lane 64][4]
constexpr size_t kLds = 2 * (size_t)kVWords * 4 + 16384 + 128;    // two images + M A of cout blocks 0, 1 + the row-pair maxima [4 slots][4 waves]

// ---- the vector work of the two phases as flat lists of small operations; a phase is cut into 48 slices of about equal weight
// (weight ~ instructions), one slice behind each MFMA -------------------------------------------------------------------------------
enum : int { OP_W, OP_V, OP_H0, OP_H1, OP_L0, OP_L1, OP_SH, OP_SL, OP_CP, OP_LD, OP_SK,   // input transform of step s + 1; OP_SK: skip-row request
             OP_TAIL,                                                              // M A of cout blocks 0, 1 (accumulators of phase Y)
             OP_RD, OP_YA, OP_EP, OP_ST,                                           // output side of step s
             OP_MX, OP_MXF };                                                      // maximum of the landed row pair (4 chains), its reduction
struct Op { int kind, r, a, b, c, wt; };
constexpr int kMaxOps = 400;
struct Prog { Op op[kMaxOps]; int n; int wsum; };

constexpr void push(Prog &p, int kind, int r, int a, int b, int c, int wt = 1) {
    p.op[p.n].kind = kind; p.op[p.n].r = r; p.op[p.n].a = a; p.op[p.n].b = b; p.op[p.n].c = c; p.op[p.n].wt = wt;
    ++p.n; p.wsum += wt;
}
// Spacing of the load instructions: the four waves of a workgroup run in lockstep, a 1 KB load occupies the CU's address path for 16
// cycles, and a wave that issues into a busy path waits (tools/ubench/vmem_issue.hip: back to back, every load costs every wave ~64
// cycles of issue) -- they are dealt into the arithmetic ONE AT A TIME, at least kGap weight units apart.
constexpr int kGap = 16;
// m: arithmetic in order; q: loads in order, each with the first position of m it may take (in .wt)
constexpr Prog merge_spaced(const Prog &m, const Prog &q) {
    Prog p{};
    int qi = 0, since = kGap;                   // weight since the last load
    for (int i = 0; i < m.n; ++i) {
        if (qi < q.n && i >= q.op[qi].wt && since >= kGap) {
            push(p, q.op[qi].kind, q.op[qi].r, q.op[qi].a, q.op[qi].b, q.op[qi].c, 1);
            ++qi; since = 0;
        }
        push(p, m.op[i].kind, m.op[i].r, m.op[i].a, m.op[i].b, m.op[i].c, m.op[i].wt);
        since += m.op[i].wt;
    }
    for (; qi < q.n; ++qi) push(p, q.op[qi].kind, q.op[qi].r, q.op[qi].a, q.op[qi].b, q.op[qi].c, 1);
    return p;
}
// The input transform of a step in two parts.
// Part W (runs in phase Y of the step BEFORE the one that finishes it): what frees the registers of patch rows 0, 1, so that the
// requests for the pair after next go out a whole step before they are needed: role 0: all of B^T d; role 1: row 0 of B^T d and a
// COPY of patch row 1 (8 moves; rows 1-3 of B^T d read the copy in part R: 16 registers held across the phases instead of 32).
constexpr void push_part_w(Prog &m, Prog &q) {
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 2; ++e)
            for (int c = 0; c < 4; ++c) push(m, OP_W, 0, e, i, c);
    for (int e = 0; e < 2; ++e)
        for (int row = 0; row < 2; ++row) push(q, OP_LD, 0, e, row, 0, m.n);
    for (int e = 0; e < 2; ++e)
        for (int c = 0; c < 4; ++c) { push(m, OP_W, 1, e, 0, c); push(m, OP_CP, 1, e, c, 0); }
    for (int e = 0; e < 2; ++e)
        for (int row = 0; row < 2; ++row) push(q, OP_LD, 1, e, row, 0, m.n);
}
// Part R: V row by V row: (B^T d) B, hi, lo, stores
constexpr void push_rows(Prog &p, int r) {
    for (int i = 0; i < 4; ++i) {
        for (int e = 0; e < 2; ++e)
            for (int c = 0; c < 4; ++c) push(p, OP_V, r, e, i, c);
        for (int c = 0; c < 4; ++c) push(p, OP_H0, r, 4 * i + c, 0, 0);
        for (int c = 0; c < 4; ++c) push(p, OP_H1, r, 4 * i + c, 0, 0);
        for (int c = 0; c < 4; ++c) push(p, OP_L0, r, 4 * i + c, 0, 0);
        for (int c = 0; c < 4; ++c) push(p, OP_L1, r, 4 * i + c, 0, 0);
        for (int c = 0; c < 4; ++c) { push(p, OP_SH, r, 4 * i + c, 0, 0); push(p, OP_SL, r, 4 * i + c, 0, 0); }
    }
}
constexpr void push_part_r(Prog &m) {
    push_rows(m, 0);
    for (int i = 1; i < 4; ++i)
        for (int e = 0; e < 2; ++e)
            for (int c = 0; c < 4; ++c) push(m, OP_W, 1, e, i, c);
    push_rows(m, 1);
}
// part W with its loads right where they may go (prologue: nothing to interleave with); with_r: the whole transform
constexpr Prog make_prog_t(bool with_r) {
    Prog m{}, q{};
    push_part_w(m, q);
    Prog p{};
    for (int i = 0; i <= m.n; ++i) {
        for (int j = 0; j < q.n; ++j)
            if (q.op[j].wt == i) push(p, OP_LD, q.op[j].r, q.op[j].a, q.op[j].b, 0, 1);
        if (i < m.n) push(p, m.op[i].kind, m.op[i].r, m.op[i].a, m.op[i].b, m.op[i].c, 1);
    }
    if (with_r) push_part_r(p);
    return p;
}
// phase X: M A of blocks 0, 1 first (their MFMAs ran in phase Y of the previous step), then the input transform of step s + 1 with
// the step's twelve load instructions dealt in: the requests for the pair after next as early as part W allows (they have to be back
// by the end of this step), then those for the skip rows of this step's outputs (used in phase Y).
// (Tried: part W and its requests in phase Y of the step before, a whole step of lead for the loads: 7100 -> 8700 cycles per step --
// twelve memory instructions in the short slices of phase Y; profiles/r06_conv_wino4_ablations.txt.)
constexpr Prog make_prog_x(bool has_res) {
    Prog m{}, q{};
    for (int ml = 0; ml < 2; ++ml)
        for (int i = 0; i < 4; ++i) push(m, OP_TAIL, 0, ml, i, 0, 6);
    push_part_w(m, q);
    if (has_res)
        for (int i = 0; i < 4; ++i) push(q, OP_SK, i >> 1, i & 1, 0, 0, m.n);
    push_part_r(m);
    return merge_spaced(m, q);
}
// phase Y: k = 0, 1: the thread's cout of blocks (mh, mh + 2); all LDS reads first, then per cout: A^T (M A), epilogue, stores
constexpr Prog make_prog_y() {
    Prog p{};
    for (int k = 0; k < 2; ++k)
        for (int w = 0; w < 4; ++w)
            for (int b = 0; b < 2; ++b) push(p, OP_RD, k, w, b, 0, 1);
    // the output side, per cout: A^T (M A), epilogue, stores
    for (int k = 0; k < 2; ++k) {
        for (int a = 0; a < 2; ++a)
            for (int c = 0; c < 4; ++c) push(p, OP_YA, k, a, c, 0, 2);
        for (int a = 0; a < 2; ++a) {
            for (int c = 0; c < 4; ++c) push(p, OP_EP, k, a, c, 0, 4);
            push(p, OP_ST, k, a, 0, 0, 2);
        }
    }
    // the maximum of the row pair that landed during phase X, LAST (its loads went out in the first half of phase X and are the
    // one thing of a step that comes from HBM with nothing to hide behind): four independent chains inside the last MFMA slices
    // instead of round 6's first form, a serial block of 16 v_max3 + 6 DPP steps in front of the barrier
    for (int i = 0; i < 16; ++i) push(p, OP_MX, i >> 3, (i >> 2) & 1, (i >> 1) & 1, i & 1, 1);        // (role, channel, row, half)
    push(p, OP_MXF, 0, 0, 0, 0, 10);
    return p;
}
constexpr Prog kProgX0 = make_prog_x(false), kProgX1 = make_prog_x(true), kProgY = make_prog_y(), kProgT0 = make_prog_t(true);
// operations [lo, hi) of slice C of NCH: cut where the running weight crosses C / NCH of the total
constexpr int slice_lo(const Prog &p, int C, int NCH) {
    int acc = 0;
    for (int i = 0; i < p.n; ++i) {
        if (acc * NCH >= C * p.wsum) return i;
        acc += p.op[i].wt;
    }
    return p.n;
}

struct St {
    u32x4 wr[4][4][2][2];          // [own position j][cout block][cin chunk][piece]: A fragments, accumulation registers
    float P[2][2][2][2][4];        // input patches [set = pair % 2][role][channel of the pair][row of the pair][column]
    f32x4 acc[4][2];               // [j][cout block of the pass]
    u32x4 bq[2][2][2];             // B fragments [position % 2][cin chunk][piece]: one position in use, the next one landing
    float w[2][4][4], v[2][4][4];  // B^T d, (B^T d) B of the role in flight
    float d1c[2][4], w0b[2][4];    // role 1: copy of patch row 1, row 0 of B^T d taken early (see make_prog_x)
    unsigned int hw[16], lw[16];
    float scale;
    f32x2 tq[2][4][2];             // phase Y: (M A)[w][b] of two tiles, for the thread's two couts
    float y[2][2][4];              // [k][output row][4 columns]
    f32x4 rq[2][2];                // skip rows [k][output row]
    float mx4[4];                  // four chains of the landed pair's maximum
};

#ifndef W4_ABL_NO_MFMA
#define W4_MFMA0(acc, a, b) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "v"(b))
#define W4_MFMA(acc, a, b) asm("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b))
#else       // timing ablation: no matrix instruction (tools/bench_conv_ws.py on libct_tune_*; results are wrong)
#define W4_MFMA0(acc, a, b) asm("v_mov_b32 %0, %2" : "=v"(acc[0]) : "a"(a), "v"(b[0]))
#define W4_MFMA(acc, a, b) asm("" : "+v"(acc) : "a"(a), "v"(b))
#endif

// everything an operation needs beside the registers in St (one instance per item; all of it is loop invariant)
struct Ctx {
    unsigned int *vb0, *vb1;        // V image word of this thread's T role: + ((p 2 + kc) 2 + piece) 256
    float *mxw;                     // M A of blocks 0, 1, this wave's part: + ((b 2 + ml) 4 + i) 64
    const float *mxr, *mr0;         // phase Y reads: blocks 0, 1: + w 1024 + b 512; blocks 2, 3 (in place): + PAR kVWords + w 4096 + b 512
    __amdgpu_buffer_rsrc_t out_rs, res_rs;
    unsigned int sto[2];            // byte offset of (cout k, first column) in this group's planes, or 2^31 (out of range)
    float bias[2];
    int act;
    float *rowmax_w;                // where this wave leaves the landed pair's maximum (set per step)
    bool lane63;
};

// one operation of a program.  TP = (index of the transformed step) % 2: patch rows 0, 1 are pair t (set TP), rows 2, 3 pair t + 1
// (set TP ^ 1), the image built is TP; PAR = parity of the step whose output side runs (phase Y)
template <int ACTK, bool HAS_RES, int TP, int PAR, const Prog &PG, int N, typename LoadF>
__device__ __forceinline__ void do_op(St &st, const Ctx &cx, const LoadF &load_row, float unscale, unsigned int ro0, unsigned int ro1) {
    constexpr Op o = PG.op[N];
    constexpr int r = o.r;
#ifdef W4_ABL_NO_T          // timing ablations (results are wrong): no transform / no patch loads
    if constexpr (o.kind < OP_LD) return;
#endif
#ifdef W4_ABL_NO_LD
    if constexpr (o.kind == OP_LD) return;
#endif
    auto d = [&](int e, int i, int c) -> float { return i < 2 ? ((r == 1 && i == 1) ? st.d1c[e][c] : st.P[TP][r][e][i][c]) : st.P[TP ^ 1][r][e][i - 2][c]; };
    if constexpr (o.kind == OP_W) {
        constexpr int e = o.a, i = o.b, c = o.c;
        if constexpr (i == 0 && r == 1) st.w0b[e][c] = d(e, 0, c) - d(e, 2, c);
        else if constexpr (i == 0) st.w[e][0][c] = d(e, 0, c) - d(e, 2, c);
        else if constexpr (i == 1) st.w[e][1][c] = d(e, 1, c) + d(e, 2, c);
        else if constexpr (i == 2) st.w[e][2][c] = d(e, 2, c) - d(e, 1, c);
        else st.w[e][3][c] = d(e, 1, c) - d(e, 3, c);
    } else if constexpr (o.kind == OP_V) {
        constexpr int e = o.a, i = o.b, c = o.c;
        auto w = [&](int cc) -> float { return (r == 1 && i == 0) ? st.w0b[e][cc] : st.w[e][i][cc]; };
        if constexpr (c == 0) st.v[e][i][0] = w(0) - w(2);
        else if constexpr (c == 1) st.v[e][i][1] = w(1) + w(2);
        else if constexpr (c == 2) st.v[e][i][2] = w(2) - w(1);
        else st.v[e][i][3] = w(1) - w(3);
    } else if constexpr (o.kind == OP_H0) {     // hi = fp16(v 2^ex): the scale rides in the converting fma; one half of the word each
        constexpr int p = o.a;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(st.hw[p]) : "v"(st.v[0][p >> 2][p & 3]), "v"(st.scale));
    } else if constexpr (o.kind == OP_H1) {
        constexpr int p = o.a;
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(st.hw[p]) : "v"(st.v[1][p >> 2][p & 3]), "v"(st.scale));
    } else if constexpr (o.kind == OP_L0) {     // lo = fp16(v 2^ex - hi)
        constexpr int p = o.a;
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(st.lw[p]) : "v"(st.v[0][p >> 2][p & 3]), "v"(st.scale), "v"(st.hw[p]));
    } else if constexpr (o.kind == OP_L1) {
        constexpr int p = o.a;
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(st.lw[p]) : "v"(st.v[1][p >> 2][p & 3]), "v"(st.scale), "v"(st.hw[p]));
    } else if constexpr (o.kind == OP_SH) {     // word (position p, cin chunk r, piece 0) of the image being built
        constexpr int p = o.a;
#ifndef W4_ABL_NO_VSTORE
        (TP ? cx.vb1 : cx.vb0)[((p * 2 + r) * 2 + 0) * 256] = st.hw[p];
#else
        asm volatile("" :: "v"(st.hw[p]));
#endif
    } else if constexpr (o.kind == OP_SL) {
        constexpr int p = o.a;
#ifndef W4_ABL_NO_VSTORE
        (TP ? cx.vb1 : cx.vb0)[((p * 2 + r) * 2 + 1) * 256] = st.lw[p];
#else
        asm volatile("" :: "v"(st.lw[p]));
#endif
    } else if constexpr (o.kind == OP_CP) {
        constexpr int e = o.a, c = o.b;
        st.d1c[e][c] = st.P[TP][1][e][1][c];
    } else if constexpr (o.kind == OP_LD) {     // the row pair after next into the registers of pair t
        constexpr int e = o.a, row = o.b;
        load_row(r, e, row, st.P[TP][r][e][row]);
    } else if constexpr (o.kind == OP_SK) {     // one skip row (ro0 / ro1: the byte offsets of this step's two output rows, clamped into the image)
        constexpr int k = o.r, a = o.a;
        st.rq[k][a] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cx.res_rs, (int)(cx.sto[k] + (a ? ro1 : ro0)), 0, 0));
    } else if constexpr (o.kind == OP_TAIL) {   // M A in the accumulators' lanes: b = 0: m0 + m1 + m2, b = 1: m1 - m2 - m3 (unscaled in phase Y)
        constexpr int ml = o.a, i = o.b;
#ifndef W4_ABL_NO_TAIL
        const float m0 = st.acc[0][ml][i], m1 = st.acc[1][ml][i], m2 = st.acc[2][ml][i], m3 = st.acc[3][ml][i];
        cx.mxw[((0 * 2 + ml) * 4 + i) * 64] = (m0 + m1) + m2;
        cx.mxw[((1 * 2 + ml) * 4 + i) * 64] = (m1 - m2) - m3;
#endif
    } else if constexpr (o.kind == OP_RD) {
        constexpr int k = o.r, w = o.a, b = o.b;
#ifdef W4_ABL_NO_DLDS
        st.tq[k][w][b] = f32x2{unscale, unscale};
#else
        if constexpr (k == 0) st.tq[0][w][b] = *reinterpret_cast<const f32x2 *>(cx.mxr + w * 1024 + b * 512);
        else st.tq[1][w][b] = *reinterpret_cast<const f32x2 *>(cx.mr0 + PAR * kVWords + w * 4096 + b * 512);
#endif
    } else if constexpr (o.kind == OP_YA) {     // column c = 2 tile + b of output row a
        constexpr int k = o.r, a = o.a, c = o.b, t = c >> 1, b = c & 1;
        if constexpr (a == 0) st.y[k][0][c] = (st.tq[k][0][b][t] + st.tq[k][1][b][t]) + st.tq[k][2][b][t];
        else st.y[k][1][c] = (st.tq[k][1][b][t] - st.tq[k][2][b][t]) - st.tq[k][3][b][t];
    } else if constexpr (o.kind == OP_EP) {
        constexpr int k = o.r, a = o.a, c = o.b;
        float v = fmaf(st.y[k][a][c], unscale, cx.bias[k]);
        if constexpr (ACTK == 1) v = fmaxf(v, 0.01f * v);
        else if constexpr (ACTK == 2) v = fmaxf(v, 0.f);
        else if constexpr (ACTK == 3) v = split_act<true>(v, cx.act);
        if constexpr (HAS_RES) v += st.rq[k][a][c];
        st.y[k][a][c] = v;
    } else if constexpr (o.kind == OP_MX) {     // pair landed in set TP ^ 1 ... of the step two ahead: the caller passes its set as TP
        constexpr int e = o.a, row = o.b, half = o.c, idx = ((r * 2 + e) * 2 + row) * 2 + half;
        const float(&q)[4] = st.P[TP][r][e][row];
        if constexpr (idx < 4) asm("v_max_f32 %0, |%1|, |%2|" : "=v"(st.mx4[idx & 3]) : "v"(q[2 * half]), "v"(q[2 * half + 1]));
        else asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(st.mx4[idx & 3]) : "v"(q[2 * half]), "v"(q[2 * half + 1]));
    } else if constexpr (o.kind == OP_MXF) {
        float m = fmaxf(fmaxf(st.mx4[0], st.mx4[1]), fmaxf(st.mx4[2], st.mx4[3]));
        int x = __float_as_int(m);
#define CT_DPP_MAX(ctrl, rmask) x = max(x, __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false))
        CT_DPP_MAX(0x111, 0xf); CT_DPP_MAX(0x112, 0xf); CT_DPP_MAX(0x114, 0xf); CT_DPP_MAX(0x118, 0xf);
        CT_DPP_MAX(0x142, 0xa); CT_DPP_MAX(0x143, 0xc);
#undef CT_DPP_MAX
        if (cx.lane63) *cx.rowmax_w = __int_as_float(x);
    } else {                                    // OP_ST: four columns of one output row of one cout
        constexpr int k = o.r, a = o.a;
#ifdef W4_ABL_NO_ST
        if (st.y[k][a][0] == 12345.678f)
#endif
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(st.y[k][a][0]), __float_as_uint(st.y[k][a][1]), __float_as_uint(st.y[k][a][2]),
                                                     __float_as_uint(st.y[k][a][3])}, cx.out_rs, (int)(cx.sto[k] + (a ? ro1 : ro0)), 0, 0);
    }
}
template <int ACTK, bool HAS_RES, int TP, int PAR, const Prog &PG, int LO, typename LoadF, int... I>
__device__ __forceinline__ void do_ops(St &st, const Ctx &cx, const LoadF &load_row, float unscale, unsigned int ro0, unsigned int ro1, std::integer_sequence<int, I...>) {
    (do_op<ACTK, HAS_RES, TP, PAR, PG, LO + I>(st, cx, load_row, unscale, ro0, ro1), ...);
}

template <typename F, int... C>
__device__ __forceinline__ void for_each_c(const F &f, std::integer_sequence<int, C...>) { (f(std::integral_constant<int, C>()), ...); }

// ACTK: compile-time activation (0 none, 1 LeakyReLU(0.01), 2 ReLU, 3 = switch over a.act at run time); HAS_RES: a skip tensor is added
template <int ACTK, bool HAS_RES>
__global__ __launch_bounds__(kThreads, 1) void conv_wino4_kernel(ConvArgs a, int n_strips, int seg, int n_seg, int n_items) {
    extern __shared__ uint4 w4_smem[];
    unsigned int *const vimg = reinterpret_cast<unsigned int *>(w4_smem);                 // [2 images][kVWords]
    float *const mx = reinterpret_cast<float *>(vimg + 2 * kVWords);                       // M A of cout blocks 0, 1: [wave][b][ml][i][lane]
    float *const rowmax = mx + 4096;                                                       // [pair % 4][wave]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned int uplane = (unsigned int)(a.H * a.W);
    const size_t plane = (size_t)a.H * a.W;

#ifdef CT_W4_PROFILE
    // diagnostic build (tools/prof_conv_wino4.py): s_memtime ticks per phase and wave -> a.prof[block][wave][10].  The stamps are
    // only ISSUED in place (a scalar memory instruction each); their results are awaited once, at the end of the step, so that no
    // stamp drains the LDS queue in the middle of a phase.
    unsigned long long pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ts[10];
#define W4_STAMP0() asm volatile("s_memtime %0" : "=s"(ts[9]) :: "memory")
#define W4_STAMP(i) asm volatile("s_memtime %0" : "=s"(ts[i]) :: "memory")
#define W4_STAMPS_END() do { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ts[0]), "+s"(ts[1]), "+s"(ts[2]), "+s"(ts[3]), "+s"(ts[4]), "+s"(ts[5]), "+s"(ts[9]) :: "memory"); \
        pt[0] += ts[0] - ts[9]; _Pragma("unroll") for (int i__ = 1; i__ < 6; ++i__) pt[i__] += ts[i__] - ts[i__ - 1]; } while (0)
#else
#define W4_STAMP0() do { } while (0)
#define W4_STAMP(i) do { } while (0)
#define W4_STAMPS_END() do { } while (0)
#endif
    St st;
    Ctx cx;
    const u32x4 *wp16 = reinterpret_cast<const u32x4 *>(a.wp);
    cx.act = a.act;
    cx.lane63 = (threadIdx.x & 63) == 63;
    cx.rowmax_w = nullptr;

    // T role r of a thread: tile tt, channel pair 16 r + 4 wave + lane / 16  (the wave is the k-group of its words in the B fragment)
    const int tt = lane & 15, cq = lane >> 4;
    cx.vb0 = vimg + 64 * wave + 4 * tt + cq;
    int vofs1 = kVWords;
    asm("" : "+v"(vofs1));                                                // opaque: the second image gets a base register of its own, so that
    cx.vb1 = cx.vb0 + vofs1;                                              // its stores keep 16-bit immediate offsets (no add per store)
    // M A: wave w writes element ((b 2 + ml) 4 + i) 64 + lane (accumulator lane = cout 16 mb + 4 (lane / 16) + i, tile lane % 16) of its
    // 4 KB -- blocks 0, 1 to mx, blocks 2, 3 over position 4 w of the consumed V image.
    // D role: tiles 2 tp, 2 tp + 1 (tp = lane % 8), couts 16 mb + 4 cqd + wave for mb = mh (k = 0) and mh + 2 (k = 1): element i = wave
    // of the accumulator lanes 16 cqd + 2 tp, + 1 -- one 8-byte LDS read per (writer wave, b), conflict-free
    const int tp = lane & 7, cqd = (lane >> 3) & 3, mh = lane >> 5;
    cx.mxw = mx + wave * 1024 + lane;
    float *const mw0 = reinterpret_cast<float *>(vimg) + (4 * wave) * 1024 + lane;
    cx.mxr = mx + (mh * 4 + wave) * 64 + 16 * cqd + 2 * tp;
    cx.mr0 = reinterpret_cast<const float *>(vimg) + (mh * 4 + wave) * 64 + 16 * cqd + 2 * tp;
    unsigned int touch = 0;             // destination of the prefetch loads (never read; kept live so that nothing else gets the register)

    // Work split: equal row segments per (output group, image), strips of a segment dealt to the workgroups of ONE XCD (blockIdx % 8):
    // neighbouring strips and their halo columns meet in one L2, and -- what matters more -- the workgroups of the chip walk down the
    // SAME rows at the same time, so HBM sees long runs of each channel row.  (Tried, round 6: one sequence of all steps cut into 256
    // equal ranges, every workgroup 253 steps instead of 270 for most: the launches without a skip tensor gained 3 %, those with one
    // went from 7400 to 10050 cycles per step -- three streams at 256 unrelated rows; profiles/r06_conv_wino4_ablations.txt.)
    const int n_bands = n_items / n_strips;
    const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const int bands_per_xcd = (n_bands + 7) >> 3;
    int cur_grp = -1;
    for (int li = wg_in_xcd; li < bands_per_xcd * n_strips; li += wgs_per_xcd) {         // XCD-aware order: see conv_ws.hip
        const int sx = li % n_strips;
        int t = (li / n_strips) * 8 + xcd;
        if (t >= n_bands) continue;
        const int sy = t % n_seg; t /= n_seg;
        const int nimg = t % a.n_images; const int grp = t / a.n_images;
        // the wave's 64 A fragments, straight into the accumulation registers (global_load writes a[] on gfx950; the compiler's own
        // route is 256 VGPRs of loads and as many moves).  64 KB per workgroup from L2, when the output group changes (once per launch
        // for cout <= 64).
        if (grp != cur_grp) {
            cur_grp = grp;
            const u32x4 *wg = wp16 + ((size_t)grp * 16 + 4 * wave) * 1024 + lane;          // + ((j 4 + mb) 2 + kc) 2 + pc) 64
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc)
                            asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(st.wr[j][mb][kc][pc]) : "v"(wg + (((j * 4 + mb) * 2 + kc) * 2 + pc) * 64) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int x0 = sx * kTW, y0 = sy * seg;
        const int rows = min(seg, a.H - y0);
        const float *in = a.in + (size_t)nimg * a.in_bstride;
        float *out = a.out + (size_t)nimg * a.out_bstride + (size_t)grp * 64 * plane;
        const int cout_g = min(a.cout - grp * 64, 64);
        cx.out_rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((unsigned int)cout_g * uplane * 4u), 0x00020000);
        cx.res_rs = cx.out_rs;
        if constexpr (HAS_RES)
            cx.res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.residual + (size_t)nimg * a.res_bstride + (size_t)grp * 64 * plane), 0,
                                                       (int)((unsigned int)cout_g * uplane * 4u), 0x00020000);

        // ---- input patches: role r, channel 2 (16 r + 4 wave + cq) + e, columns x0 + 2 tt - 1 .. + 2, always an in-bounds 16-byte load ----
        const int px = x0 + 2 * tt - 1;
        const int pxs = min(max(px, 0), a.W - 4);
        // fix-up code of a lane (edge strips only): 0 as loaded, 1 = left image edge (loaded one column to the right), 2 = right edge
        // (loaded one to the left), 3 = dead tile; bit 2 + 2 r + e: channel >= cin (zero)
        int fix = x0 + 2 * tt >= a.W ? 3 : (px < 0 ? 1 : (px > pxs ? 2 : 0));
        unsigned int voff[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int ch = 2 * (16 * r + 4 * wave + cq) + e;
                if (ch >= a.cin) fix |= 4 << (2 * r + e);
                voff[r][e] = (unsigned int)min(ch, a.cin - 1) * uplane + (unsigned int)pxs;
            }
        const bool edge = __builtin_amdgcn_readfirstlane((x0 == 0 || x0 + kTW >= a.W || a.cin < 64) ? 1 : 0) != 0;
        // pair k = input rows y0 - 1 + 2 k, + 1
        auto row_of = [&](int k, int row) { return y0 - 1 + 2 * k + row; };
        auto load_pair_row = [&](int k, int r, int e, int row, float (&dst)[4]) {
            const unsigned int yo = (unsigned int)(min(max(row_of(k, row), 0), a.H - 1) * a.W);
            const f32x4 v = *reinterpret_cast<const f32x4u *>(in + yo + voff[r][e]);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        };
        auto load_pair = [&](int k, float (&dst)[2][2][2][4]) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int row = 0; row < 2; ++row) load_pair_row(k, r, e, row, dst[r][e][row]);
        };
        // (tried, -DW4_TOUCH: pair k's lines into L2 a step before the 16-byte loads ask for them, one dword per lane and row into a
        // register that nothing reads -- gfx950 has no prefetch instruction.  The wait for the landed pair shrinks, but every vector
        // memory instruction costs a wave ~100 cycles of issue here, and eight more per step cost 500-1000: 8386 -> 7360 cycles per
        // step without them, profiles/r06_conv_wino4_ablations.txt)
        auto touch_pair = [&](int k) {
#ifdef W4_TOUCH
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const float *rp = in + (unsigned int)(min(max(row_of(k, row), 0), a.H - 1) * a.W);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int e = 0; e < 2; ++e)
                        asm volatile("global_load_dword %0, %1, %2" : "+v"(touch) : "v"(voff[r][e] * 4u), "s"(rp) : "memory");
            }
#endif
        };
        // what the landed pair k still needs: zero rows outside the image (uniform), edge columns / missing channels (edge strips), then
        // the maximum of |x| over the pair, per wave, into rowmax[k % 4][wave]
        auto fix_pair = [&](int k, float (&p)[2][2][2][4]) {
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const int y = row_of(k, row);
                if (y < 0 || y >= a.H) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int e = 0; e < 2; ++e)
#pragma unroll
                            for (int c = 0; c < 4; ++c) p[r][e][row][c] = 0.f;
                }
            }
            if (edge) {
                const int fx = fix & 3;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const bool dead = fx == 3 || ((fix >> (2 + 2 * r + e)) & 1);
#pragma unroll
                        for (int row = 0; row < 2; ++row) {
                            float(&q)[4] = p[r][e][row];
                            const float l0 = q[0], l1 = q[1], l2 = q[2], l3 = q[3];
                            q[0] = dead ? 0.f : (fx == 1 ? 0.f : (fx == 2 ? l1 : l0));
                            q[1] = dead ? 0.f : (fx == 1 ? l0 : (fx == 2 ? l2 : l1));
                            q[2] = dead ? 0.f : (fx == 1 ? l1 : (fx == 2 ? l3 : l2));
                            q[3] = dead ? 0.f : (fx == 1 ? l2 : (fx == 2 ? 0.f : l3));
                        }
                    }
            }
        };
        auto finish_pair = [&](int k, float (&p)[2][2][2][4]) {       // prologue form: fix-ups, then the maximum as one block
            fix_pair(k, p);
            float m = 0.f;
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int row = 0; row < 2; ++row) {
                        const float(&q)[4] = p[r][e][row];
                        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(q[0]), "v"(q[1]));
                        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(q[2]), "v"(q[3]));
                    }
            int x = __float_as_int(m);
#define CT_DPP_MAX(ctrl, rmask) x = max(x, __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false))
            CT_DPP_MAX(0x111, 0xf); CT_DPP_MAX(0x112, 0xf); CT_DPP_MAX(0x114, 0xf); CT_DPP_MAX(0x118, 0xf);
            CT_DPP_MAX(0x142, 0xa); CT_DPP_MAX(0x143, 0xc);
#undef CT_DPP_MAX
            if (lane == 63) rowmax[(k & 3) * 4 + wave] = __int_as_float(x);
        };
        // the scale of step t (pairs t, t + 1): 2^ex max |x| in [2^9, 2^10), so |V| <= 4 max |x| stays below 2^12 (conv_wino.hip).
        // In two halves: the LDS reads at the head of a step, the arithmetic a few slices later (nothing waits for the reads)
        auto ex_of = [&](f32x4 ma, f32x4 mb) {
            const float mxv = fmaxf(fmaxf(fmaxf(ma.x, ma.y), fmaxf(ma.z, ma.w)), fmaxf(fmaxf(mb.x, mb.y), fmaxf(mb.z, mb.w)));
            const int fld = (int)(__float_as_uint(mxv) >> 23);
            const int ex = (fld == 0 || fld == 255) ? 0 : 136 - fld;
            return __builtin_amdgcn_readfirstlane(min(max(ex, -100), 100));
        };
        auto max_of = [&](int k) { return *reinterpret_cast<const f32x4 *>(rowmax + (k & 3) * 4); };

        // D role addressing: byte offset of (cout, column x0 + 4 tp) in this group's planes, or out of range (2^31; sizes are below 2^30)
        const int ox = x0 + 4 * tp;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int co = 16 * (mh + 2 * k) + 4 * cqd + wave;
            cx.sto[k] = (co < cout_g && ox < a.W) ? ((unsigned int)co * uplane + (unsigned int)ox) * 4u : 0x80000000u;
            cx.bias[k] = a.bias[grp * 64 + co];
        }
        // the B fragments of position j = 0 of V image `img`, both cin chunks
        auto first_frags = [&](int img) {
            const u32x4 *frag = reinterpret_cast<const u32x4 *>(vimg + img * kVWords) + (4 * wave) * 256 + lane;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) st.bq[0][kc][pc] = frag[(kc * 2 + pc) * 64];
        };
        // MFMA number Q of a pass (cout blocks 2 PASS, 2 PASS + 1) over V image IMG, with the fragment reads that keep it fed
        auto mfma = [&](auto pass_c, auto img_c, auto q_c) {
            constexpr int PASS = decltype(pass_c)::value, IMG = decltype(img_c)::value, Q = decltype(q_c)::value;
            constexpr int j = Q / 12, q = Q % 12, kc = q / 6, pr = (q % 6) / 2, ml = q % 2;
            constexpr int pw = pr == 0 ? 1 : 0, pv = pr == 1 ? 1 : 0;            // small terms first: lo x hi, hi x lo, hi x hi
            const u32x4 *frag = reinterpret_cast<const u32x4 *>(vimg + IMG * kVWords) + (4 * wave) * 256 + lane;   // + ((j 2 + kc) 2 + piece) 64
            if constexpr (q < 2) W4_MFMA0(st.acc[j][ml], st.wr[j][2 * PASS + ml][kc][pw], st.bq[j & 1][kc][pv]);
            else W4_MFMA(st.acc[j][ml], st.wr[j][2 * PASS + ml][kc][pw], st.bq[j & 1][kc][pv]);
            // the next position's four fragments into the other register set, one read per slice, a whole position (12 slices) ahead of
            // their first use: the slices of phase Y are shorter than an LDS round trip / 6 (round 6: 16 registers more, fewer waits)
            if constexpr (j < 3 && q < 4) st.bq[(j + 1) & 1][q >> 1][q & 1] = frag[(((j + 1) * 2 + (q >> 1)) * 2 + (q & 1)) * 64];
        };

        const int steps = (rows + 1) >> 1;
        // ---- prologue: pairs 0, 1 -> maxima -> T(0) -> V image 0 -> the first pass of step 0; pair 2 in flight meanwhile ----
        __syncthreads();                                  // the previous item's last step is done with the images and the maxima
        load_pair(0, st.P[0]);
        load_pair(1, st.P[1]);
        touch_pair(2); touch_pair(3);
        finish_pair(0, st.P[0]);
        finish_pair(1, st.P[1]);
        __syncthreads();
        int ex_next = ex_of(max_of(0), max_of(1));
        st.scale = __uint_as_float((unsigned int)(127 + ex_next) << 23);
        {
            auto ld = [&](int r, int e, int row, float (&dst)[4]) { load_pair_row(2, r, e, row, dst); };
            do_ops<ACTK, HAS_RES, 0, 0, kProgT0, 0>(st, cx, ld, 0.f, 0u, 0u, std::make_integer_sequence<int, kProgT0.n>());
        }
        finish_pair(2, st.P[0]);
        __syncthreads();
        first_frags(0);
        for_each_c([&](auto q_c) { mfma(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), q_c); }, std::make_integer_sequence<int, 48>());

        // one step; PAR = s % 2: V image PAR is consumed, image PAR ^ 1 built (step s + 1: pairs s + 1 in set PAR ^ 1, s + 2 in set PAR)
        auto step = [&](int s, auto par_c) {
            constexpr int PAR = decltype(par_c)::value;
            W4_STAMP0();
            const int oy = y0 + 2 * s;
            const int ex = ex_next;
            const f32x4 mxa = max_of(s + 1), mxb = max_of(s + 2);
            touch_pair(s + 4);
            const unsigned int sk0 = (unsigned int)(min(oy, a.H - 1) * a.W) * 4u, sk1 = (unsigned int)(min(oy + 1, a.H - 1) * a.W) * 4u;
            first_frags(PAR);          // position 0 again: the second pass re-reads the fragments (its accumulators are the first pass's)
            auto ld = [&](int r, int e, int row, float (&dst)[4]) { load_pair_row(s + 3, r, e, row, dst); };
            // ---------------- phase X: cout blocks 2, 3 of step s; M A of blocks 0, 1; T(s + 1) ----------------
            __builtin_amdgcn_sched_barrier(0);
            W4_STAMP(0);
            for_each_c([&](auto c_c) {
                constexpr int C = decltype(c_c)::value;
                constexpr const Prog &PX = HAS_RES ? kProgX1 : kProgX0;
                constexpr int lo = slice_lo(PX, C, 48), hi = slice_lo(PX, C + 1, 48);
                // the first pass's accumulators are read by the OP_TAIL operations, which lead the program: the second pass's MFMAs
                // start behind them (slices that still hold an OP_TAIL get no MFMA; they are made up for at the end)
                constexpr int n_tail_slices = [] { int c = 0; while (slice_lo(HAS_RES ? kProgX1 : kProgX0, c, 48) < 8) ++c; return c; }();
                if constexpr (C >= n_tail_slices) mfma(std::integral_constant<int, 1>(), par_c, std::integral_constant<int, C - n_tail_slices>());
                if constexpr (C == 6) {
                    ex_next = ex_of(mxa, mxb);
                    st.scale = __uint_as_float((unsigned int)(127 + ex_next) << 23);
                }
                do_ops<ACTK, HAS_RES, PAR ^ 1, PAR, PX, lo>(st, cx, ld, 0.f, sk0, sk1, std::make_integer_sequence<int, hi - lo>());
                __builtin_amdgcn_sched_barrier(0);
            }, std::make_integer_sequence<int, 48>());
            {
                constexpr int n_tail_slices = [] { int c = 0; while (slice_lo(HAS_RES ? kProgX1 : kProgX0, c, 48) < 8) ++c; return c; }();
                for_each_c([&](auto c_c) { mfma(std::integral_constant<int, 1>(), par_c, std::integral_constant<int, 48 - n_tail_slices + decltype(c_c)::value>()); },
                           std::make_integer_sequence<int, n_tail_slices>());
            }
            W4_STAMP(1);
            // M A of blocks 2, 3 over the wave's own, consumed, fragments (position 4 w of image PAR).  The last MFMAs wrote
            // acc[3][*] a few cycles ago and the compiler does not know an MFMA when it sees one in asm
            asm volatile("s_nop 15" : "+v"(st.acc[3][0]), "+v"(st.acc[3][1]));
#ifndef W4_ABL_NO_TAIL
            {
                float *mw = mw0 + PAR * kVWords;
#pragma unroll
                for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float m0 = st.acc[0][ml][i], m1 = st.acc[1][ml][i], m2 = st.acc[2][ml][i], m3 = st.acc[3][ml][i];
                        mw[((0 * 2 + ml) * 4 + i) * 64] = (m0 + m1) + m2;
                        mw[((1 * 2 + ml) * 4 + i) * 64] = (m1 - m2) - m3;
                    }
            }
#endif
            W4_STAMP(2);
#ifndef W4_ABL_NO_BAR1
            __syncthreads();
#endif
            W4_STAMP(3);
            // ---------------- phase Y: cout blocks 0, 1 of step s + 1; the output side of step s ----------------
            {
                const float unscale = __uint_as_float((unsigned int)(127 + min(max(-ex - a.w_exp, -126), 127)) << 23);
                // a row past the segment: + 2^30 puts the offset out of range (sizes are below 2^30 bytes, invalid lanes at 2^31)
                const unsigned int ro0 = (2 * s < rows) ? (unsigned int)(oy * a.W) * 4u : 0x40000000u;
                const unsigned int ro1 = (2 * s + 1 < rows) ? (unsigned int)((oy + 1) * a.W) * 4u : 0x40000000u;
                first_frags(PAR ^ 1);
                fix_pair(s + 3, st.P[PAR ^ 1]);            // rows outside the image, edge strips: uniform branches, normally nothing
                cx.rowmax_w = rowmax + ((s + 3) & 3) * 4 + wave;
                __builtin_amdgcn_sched_barrier(0);
                for_each_c([&](auto c_c) {
                    constexpr int C = decltype(c_c)::value;
                    constexpr int lo = slice_lo(kProgY, C, 48), hi = slice_lo(kProgY, C + 1, 48);
                    mfma(std::integral_constant<int, 0>(), std::integral_constant<int, PAR ^ 1>(), c_c);
                    do_ops<ACTK, HAS_RES, PAR ^ 1, PAR, kProgY, lo>(st, cx, ld, unscale, ro0, ro1, std::make_integer_sequence<int, hi - lo>());
                    __builtin_amdgcn_sched_barrier(0);
                }, std::make_integer_sequence<int, 48>());
            }
            W4_STAMP(4);
#ifndef W4_ABL_NO_BAR2
            __syncthreads();
#endif
            W4_STAMP(5);
            W4_STAMPS_END();
        };
#pragma unroll 1
        for (int s = 0; s < steps; s += 2) {
            step(s, std::integral_constant<int, 0>());
            if (s + 1 < steps) step(s + 1, std::integral_constant<int, 1>());
        }
    }
    asm volatile("" :: "v"(touch));
#ifdef CT_W4_PROFILE
    if (lane == 0 && a.prof) {
#pragma unroll
        for (int i = 0; i < 10; ++i) a.prof[((size_t)blockIdx.x * 4 + wave) * 10 + i] = pt[i];
    }
#endif
}

}  // namespace w4

#ifdef CT_W4_PROFILE
static unsigned long long *g_w4_prof = nullptr;
#endif

// 1 = not this kernel's geometry
int conv_wino4(const ConvArgs &a, int N, hipStream_t s) {
    if (a.in2 != nullptr || a.cin <= 32 || a.cin > 64 || !a.f16 || (a.W & 3) || a.W < 4) return 1;
    if (a.clamp) return 1;                  // the one clamped convolution of a network stays on conv_wino.hip
    if ((unsigned long long)a.H * (unsigned long long)a.W * 64ull * 4ull >= (1ull << 30)) return 1;     // 30-bit byte offsets over 64 planes (the kernel's out-of-range marks)
    const int n_strips = (a.W + w4::kTW - 1) / w4::kTW;
    const long long imgs = (long long)N * a.groups;
    int n_seg = 1, seg = a.H;
    long long best = -1;
    for (int ns = 1; ns <= (a.H + 15) / 16; ++ns) {          // even row segments: the split with the fewest steps of the busiest workgroup
        int sg = (a.H + ns - 1) / ns;
        sg += sg & 1;
        const int ns_eff = (a.H + sg - 1) / sg;
        const long long bands_per_xcd = (imgs * ns_eff + 7) / 8;
        const long long rounds = (bands_per_xcd * n_strips + 31) / 32;
        const long long cost = rounds * (sg / 2 + 3);
        if (best < 0 || cost < best) { best = cost; n_seg = ns_eff; seg = sg; }
    }
    const long long n_items = imgs * n_seg * n_strips;
    if (n_items > 0x7fffffffLL) return CT_E_BADARG;
    typedef void (*kern_t)(ConvArgs, int, int, int, int);
    static const kern_t kerns[8] = {w4::conv_wino4_kernel<0, false>, w4::conv_wino4_kernel<1, false>, w4::conv_wino4_kernel<2, false>, w4::conv_wino4_kernel<3, false>,
                                    w4::conv_wino4_kernel<0, true>,  w4::conv_wino4_kernel<1, true>,  w4::conv_wino4_kernel<2, true>,  w4::conv_wino4_kernel<3, true>};
    const int k = ((a.act >= 0 && a.act <= 2) ? a.act : 3) + (a.residual ? 4 : 0);
    static DynLdsAttr attr[8];
    if (attr[k].ensure(reinterpret_cast<const void *>(kerns[k]), w4::kLds) != hipSuccess) return CT_E_BADARG;
    ConvArgs b = a;
    b.n_images = N;
#ifdef CT_W4_PROFILE
    b.prof = g_w4_prof;
#endif
    hipLaunchKernelGGL(kerns[k], dim3(256), dim3(w4::kThreads), w4::kLds, s, b, n_strips, seg, n_seg, (int)n_items);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // namespace ct

#ifdef CT_W4_PROFILE
extern "C" void ct_conv_wino4_set_prof(unsigned long long *p) { ct::g_w4_prof = p; }
#endif
