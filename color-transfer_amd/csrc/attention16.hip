// attention16.hip -- the streaming (online-softmax) attention of gmflow.hip with float32 operands as TWO fp16 pieces
// (11 + 11 mantissa bits) and three v_mfma_f32_32x32x16_f16 per 16-wide product (a_hi b_hi + a_hi b_lo + a_lo b_hi, float32
// accumulation; what is dropped is 2^-22 relative) instead of three bf16 pieces and six MFMAs: half the matrix work, which is what
// counts on real data (the 16-bit pipe runs into the chip's power limit: conv_ws.hip).  Replaces, same arguments and results to
// float32 grade:
//   attention16_tokens_kernel<C, CV, MAP>  <- attention_tokens_kernel<C, CV, MAP, true>  (pasmnet/attention.py:39-41 +
//        utils.py:30,123-125 for C = 64; unimatch/attention.py:48-107,199-216 and matching.py:10-39 for C = 128)
//   attention16_colsum_kernel<C>           <- attention_colsum_kernel<C>                 (pasmnet/utils.py:31,34)
// fp16 has five exponent bits, so every operand carries a power-of-two scale that brings its maximum to [2^11, 2^12) (anything
// smaller keeps an absolute error below 2^-36 of that maximum):
//   * a query / key row held by a lane as the B operand: one scale per row (the lane's own);
//   * a 32-row tile staged through LDS as the A operand (K, V, or Q in the column-sum kernel): one scale per tile -- every wave
//     leaves the maximum of its part in LDS before the barrier that already separates "tile consumed" from "next tile staged";
//   * scores: s = acc * (2^-e_tile * 2^-e_row), folded into the FMA that forms the exp2 argument (the running maximum is taken on
//     the raw accumulators: the factor is positive);
//   * P = exp2(s - m + 15) <= 2^15 feeds the P.V product; the sum l runs in the same scaled domain, so the factor cancels.  A
//     probability keeps 22 bits down to 2^-29 of the row's running maximum and an absolute error of 2^-40 of it below (fp16
//     subnormals): the result carries 2^-22 sum_j p_j |v_j| + 2^-40 sum_j |v_j| -- the second term shows only when the values span
//     some ten decades along the keys (tests/test_gmflow_kernels_gpu.py::assert_attention_bound);
//   * V tiles use a RUNNING scale (the minimum exponent so far): when it changes, the accumulators are rescaled by the same
//     multiply that applies the online-softmax correction.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cmath>
#include "ct_attention16.h"

namespace ct {

typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8h __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2h __attribute__((ext_vector_type(2)));

static constexpr float kLog2eH = 1.4426950408889634f, kLn2H = 0.6931471805599453f;

// max over the wave of a non-negative float; every lane returns it
__device__ __forceinline__ float wave_max_nonneg_h(float v) {
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

// opaque to the compiler (it otherwise re-derives each half with v_fma_mixlo_f16 when the halves are converted back)
__device__ __forceinline__ unsigned int cvt_pk_f16h(float a, float b) {
    unsigned int r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split2x2h(float x0, float x1, unsigned int &hw, unsigned int &lw) {
    hw = cvt_pk_f16h(x0, x1);
    const f16x2h h = __builtin_bit_cast(f16x2h, hw);
    lw = cvt_pk_f16h(x0 - (float)h.x, x1 - (float)h.y);
}
__device__ __forceinline__ void split2x8h(const float (&x)[8], uint4 &h, uint4 &l) {
    unsigned int hw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) split2x2h(x[2 * i], x[2 * i + 1], hw[i], lw[i]);
    h = make_uint4(hw[0], hw[1], hw[2], hw[3]); l = make_uint4(lw[0], lw[1], lw[2], lw[3]);
}
// s += A . B with A, B given as (hi, lo) fragments; small terms first
__device__ __forceinline__ void mfma_split3h(f32x16h &s, const uint4 (&a)[2], const uint4 (&b)[2]) {
    const f16x8h ah = __builtin_bit_cast(f16x8h, a[0]), al = __builtin_bit_cast(f16x8h, a[1]);
    const f16x8h bh = __builtin_bit_cast(f16x8h, b[0]), bl = __builtin_bit_cast(f16x8h, b[1]);
    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, s, 0, 0, 0);
}
// exponent e with 2^e * mx in [2^11, 2^12); `none` for mx == 0 / denormal (no constraint); 0 for inf / NaN (they propagate)
__device__ __forceinline__ int scale_exp_h(float mx, int none) {
    const int fld = (int)(__float_as_uint(mx) >> 23);                  // biased exponent (mx >= 0)
    const int ex = fld == 0 ? none : fld == 255 ? 0 : 138 - fld;       // 12 - (floor(log2 mx) + 1)
    return min(max(ex, -100), 100);
}
__device__ __forceinline__ float pow2i_h(int e) { return __uint_as_float((unsigned int)(127 + e) << 23); }   // |e| <= 126

constexpr int kSsRowH(int C) { return 2 * C + 16; }   // bytes per LDS row of a tile image: 16-byte fragment reads are conflict free

// XCD-aware workgroup order (round 6).  Workgroups are dealt to the 8 XCDs round-robin by their linear id, and an XCD's L2 is its own:
// with the plain (x, y) order the ~15 query tiles of one token row land on all eight XCDs and each of them fetches that row's keys and
// values from HBM -- profiles/r06_dcmcs3di_1080p_traffic.json: 11.4 GB read per launch of the 1080p parallax attention whose inputs
// are 2.4 GB.  Remapped, the tiles an XCD receives one after the other are the tiles of consecutive rows: lin -> (lin % 8) * (n / 8) +
// lin / 8 on the first n = 8 * (total / 8) workgroups (a bijection), identity on the rest.  Returns the (x, y) this workgroup acts as.
__device__ __forceinline__ void xcd_order(int &bx, int &by) {
    bx = blockIdx.x; by = blockIdx.y;
    if (gridDim.z != 1) return;
    const unsigned int gx = gridDim.x, total = gx * gridDim.y, lin = by * gx + bx, full = total & ~7u;
    if (lin >= full) return;
    const unsigned int vid = (lin & 7u) * (full >> 3) + (lin >> 3);
    bx = (int)(vid % gx); by = (int)(vid / gx);
}

template <int C, int CV, bool MAP>
__global__ __launch_bounds__(256, 2) void attention16_tokens_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                                    const float *__restrict__ v, const int *__restrict__ region,
                                                                    const int *__restrict__ rowmap, float *__restrict__ out,
                                                                    float *__restrict__ stats, int L, float scale,
                                                                    float *__restrict__ part, long long kv_shift, long long kv_total) {
    constexpr bool PVS = CV >= 32;                  // values on the matrix pipe
    constexpr int NVT = PVS ? CV / 32 : 1;
    constexpr int KV4 = (32 * C / 4) / 256;         // float4 per thread of one K tile
    constexpr int VV4 = PVS ? (32 * CV / 4) / 256 : 1;
    constexpr int SROW = kSsRowH(C);                // K tile as [piece][key][C] fp16, rows padded to SROW bytes
    // V as [piece][key][CV] fp16, rows of VROWB bytes with (VROWB / 4) % 64 == 16 or 48: the four rows a transposed read gathers
    // lie in four disjoint 16-bank windows (cdna_hip_programming.md T10)
    constexpr int VROWB = (CV * 2) % 256 == 0 ? CV * 2 + 64 : CV * 2;
    static_assert(!PVS || ((VROWB / 4) % 64 == 16 || (VROWB / 4) % 64 == 48), "V image rows must not share banks");
    constexpr float POFF = PVS ? 15.0f : 0.0f;      // P = exp2(s - m + POFF): at most 2^15 in fp16, hi/lo pieces never subnormal where it matters
    __shared__ __attribute__((aligned(16))) unsigned char Ks[2 * 32 * SROW];
    __shared__ __attribute__((aligned(16))) unsigned char Vs[PVS ? 2 * 32 * VROWB : 256];
    __shared__ int Rs[32];
    __shared__ __attribute__((aligned(16))) float Mx[8];    // [wave][K, V] tile maxima of the tile about to be staged
    __shared__ float Gs[2];                                 // {2^-e, 2^e} of the staged K tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    int bxo, b;
    xcd_order(bxo, b);
    const int q0 = (bxo * 4 + wave) * 32;
    const size_t tb = (size_t)b * L;
    const int qi = q0 + nl;
    const bool qlive = qi < L;
    const int qclamp = qlive ? qi : L - 1;
    auto row = [&](int i) -> size_t {
        if constexpr (MAP) return (size_t)rowmap[tb + i];
        else return tb + i;
    };
    auto kvrow = [&](int i) -> size_t {
        size_t r = row(i);
        if constexpr (MAP) {
            r += (size_t)kv_shift;
            if (r >= (size_t)kv_total && kv_total > 0) r -= (size_t)kv_total;
        }
        return r;
    };
    // B operand of S^T = K Q^T: lane (query nl, half hl) holds channels 16 st + 8 hl + j of its query, times scale * log2(e)
    // (scores live in the log2 domain) times 2^eq
    uint4 qf[C / 16][2];
    float gq, sq;                                   // 2^-eq, 2^eq
    {
        const float qs = scale * kLog2eH;
        const float *qp = q + row(qclamp) * C + 8 * hl;
        float x[C / 16][8];
        float amax = 0.f;
#pragma unroll
        for (int st = 0; st < C / 16; ++st) {
            const float4 t0 = *reinterpret_cast<const float4 *>(qp + 16 * st), t1 = *reinterpret_cast<const float4 *>(qp + 16 * st + 4);
            x[st][0] = t0.x * qs; x[st][1] = t0.y * qs; x[st][2] = t0.z * qs; x[st][3] = t0.w * qs;
            x[st][4] = t1.x * qs; x[st][5] = t1.y * qs; x[st][6] = t1.z * qs; x[st][7] = t1.w * qs;
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(x[st][j]));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const int eq = scale_exp_h(amax, 0);
        sq = pow2i_h(eq); gq = pow2i_h(-eq);
#pragma unroll
        for (int st = 0; st < C / 16; ++st) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[st][j] *= sq;
            split2x8h(x[st], qf[st][0], qf[st][1]);
        }
    }
    const int qreg = region ? region[tb + qclamp] : 0;

    float4 kpre[KV4], vpre[VV4];
    int rpre = 0;
    size_t krow[KV4], vrow[PVS ? VV4 : 1];
    auto fetch_rows = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KV4; ++i) {
            const int key = (tid + i * 256) / (C / 4);
            krow[i] = kvrow(j0 + key < L ? j0 + key : L - 1);
        }
        if constexpr (PVS) {
#pragma unroll
            for (int i = 0; i < VV4; ++i) {
                const int key = (tid + i * 256) / (CV / 4);
                vrow[i] = kvrow(j0 + key < L ? j0 + key : L - 1);
            }
        }
    };
    auto fetch = [&](int j0) {
#pragma unroll
        for (int i = 0; i < KV4; ++i) {
            const int f = tid + i * 256, key = f / (C / 4), c4 = f - key * (C / 4);
            kpre[i] = (j0 + key < L) ? *reinterpret_cast<const float4 *>(k + krow[i] * C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if constexpr (PVS) {
#pragma unroll
            for (int i = 0; i < VV4; ++i) {
                const int f = tid + i * 256, key = f / (CV / 4), c4 = f - key * (CV / 4);
                vpre[i] = (j0 + key < L) ? *reinterpret_cast<const float4 *>(v + vrow[i] * CV + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if constexpr (CV == 2) {
            if (tid < 16) {   // 32 keys x 2 channels = 16 float4
                const int key = 2 * tid;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (MAP) {
                    if (j0 + key < L) { const float2 u = *reinterpret_cast<const float2 *>(v + kvrow(j0 + key) * 2); t.x = u.x; t.y = u.y; }
                    if (j0 + key + 1 < L) { const float2 u = *reinterpret_cast<const float2 *>(v + kvrow(j0 + key + 1) * 2); t.z = u.x; t.w = u.y; }
                } else if (j0 + key + 1 < L) t = *reinterpret_cast<const float4 *>(v + (tb + j0 + key) * 2);
                else if (j0 + key < L) { const float2 u = *reinterpret_cast<const float2 *>(v + (tb + j0 + key) * 2); t.x = u.x; t.y = u.y; }
                vpre[0] = t;
            }
        }
        if (region && tid < 32) rpre = (j0 + tid < L) ? region[tb + j0 + tid] : 0;
    };
    auto amax4 = [](const float4 &t) { return fmaxf(fmaxf(fabsf(t.x), fabsf(t.y)), fmaxf(fabsf(t.z), fabsf(t.w))); };
    // the maxima of this wave's part of the fetched tile -> Mx; the caller's next barrier publishes them
    auto note_max = [&]() {
        float mk = 0.f, mv = 0.f;
#pragma unroll
        for (int i = 0; i < KV4; ++i) mk = fmaxf(mk, amax4(kpre[i]));
        if constexpr (PVS) {
#pragma unroll
            for (int i = 0; i < VV4; ++i) mv = fmaxf(mv, amax4(vpre[i]));
        }
        mk = wave_max_nonneg_h(mk);
        if constexpr (PVS) mv = wave_max_nonneg_h(mv);
        if (lane == 0) { Mx[2 * wave] = mk; Mx[2 * wave + 1] = mv; }
    };
    int e_stage = 100, e_cur = 100;                 // running V exponent: of the staged tile / of the accumulators' domain
    auto stage = [&]() {
        const float4 m0 = *reinterpret_cast<const float4 *>(Mx), m1 = *reinterpret_cast<const float4 *>(Mx + 4);
        const int ek = __builtin_amdgcn_readfirstlane(scale_exp_h(fmaxf(fmaxf(m0.x, m0.z), fmaxf(m1.x, m1.z)), 0));
        const float sk = pow2i_h(ek);
        if (tid == 0) { Gs[0] = pow2i_h(-ek); Gs[1] = sk; }
#pragma unroll
        for (int i = 0; i < KV4; ++i) {
            const int f = tid + i * 256, key = f / (C / 4), c4 = f - key * (C / 4);
            unsigned int h0, l0, h1, l1;
            split2x2h(kpre[i].x * sk, kpre[i].y * sk, h0, l0);
            split2x2h(kpre[i].z * sk, kpre[i].w * sk, h1, l1);
            unsigned char *kd = Ks + key * SROW + 8 * c4;
            *reinterpret_cast<uint2 *>(kd) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(kd + 32 * SROW) = make_uint2(l0, l1);
        }
        if constexpr (PVS) {
            const int ev = scale_exp_h(fmaxf(fmaxf(m0.y, m0.w), fmaxf(m1.y, m1.w)), 100);
            e_stage = __builtin_amdgcn_readfirstlane(min(e_stage, ev));
            const float sv = pow2i_h(e_stage);
#pragma unroll
            for (int i = 0; i < VV4; ++i) {
                const int f = tid + i * 256, key = f / (CV / 4), c4 = f - key * (CV / 4);
                unsigned int h0, l0, h1, l1;
                split2x2h(vpre[i].x * sv, vpre[i].y * sv, h0, l0);
                split2x2h(vpre[i].z * sv, vpre[i].w * sv, h1, l1);
                unsigned char *vd = Vs + key * VROWB + 8 * c4;
                *reinterpret_cast<uint2 *>(vd) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(vd + 32 * VROWB) = make_uint2(l0, l1);
            }
        } else if constexpr (CV == 2) {
            if (tid < 16) *reinterpret_cast<float4 *>(Vs + 16 * tid) = vpre[0];   // float Vs[key*2 + ch]
        }
        if (region && tid < 32) Rs[tid] = rpre;
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x16h o[NVT];
    float o2x = 0.f, o2y = 0.f;
    if constexpr (PVS) {
#pragma unroll
        for (int j = 0; j < NVT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[j][r] = 0.f;
    }
    // key split (gridDim.z > 1): keys [jb, je) only, unnormalised partial (o, max, sum) to `part` (attention_combine_kernel merges)
    const int nsplit = gridDim.z, split = blockIdx.z;
    const int kchunk = ((L + nsplit - 1) / nsplit + 31) & ~31;
    const int jb = split * kchunk, je = (jb + kchunk < L) ? jb + kchunk : L;
    fetch_rows(jb);
    fetch(jb);
    fetch_rows(jb + 32);
    note_max();
    __syncthreads();
    stage();
    e_cur = e_stage;
    __syncthreads();
    for (int j0 = jb; j0 < je; j0 += 32) {
        const bool more = j0 + 32 < je;
        if (more) {
            fetch(j0 + 32);
            fetch_rows(j0 + 64);
        }
        // ---- S^T tile: A = K rows (key nl) from LDS, B = Q; LDS operand reads run one group ahead of the MFMAs ----
        f32x16h s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        {
            const unsigned char *kp = Ks + nl * SROW + 16 * hl;
            uint4 ac[2], an[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) ac[p] = *reinterpret_cast<const uint4 *>(kp + p * 32 * SROW);
#pragma unroll
            for (int st = 0; st < C / 16; ++st) {
                if (st + 1 < C / 16) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) an[p] = *reinterpret_cast<const uint4 *>(kp + p * 32 * SROW + 32 * (st + 1));
                }
                mfma_split3h(s, ac, qf[st]);
#pragma unroll
                for (int p = 0; p < 2; ++p) ac[p] = an[p];
            }
        }
        // lane: query nl; s[r] * f = log2-domain score of key j0 + (r&3)+8(r>>2)+4hl
        const float f = Gs[0] * gq;
        if (region) {                                    // shifted-window mask, one uniform branch per tile
            const float maskv = fmaxf((-100.0f * kLog2eH) * (Gs[1] * sq), -3.0e38f);   // -100 in the accumulators' domain, finite
            int rk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rk[r] = Rs[(r & 3) + 8 * (r >> 2) + 4 * hl];
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] += (rk[r] != qreg) ? maskv : 0.0f;
        }
        if (j0 + 32 > L) {                               // ragged last tile
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = (j0 + (r & 3) + 8 * (r >> 2) + 4 * hl < L) ? s[r] : -INFINITY;
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          // the other half of the keys of this query
        const float m_new = fmaxf(m_run, mx * f);        // finite: key j0 exists and a mask only subtracts 100
        const float corr = __builtin_amdgcn_exp2f(m_run - m_new);   // exp2(-inf) = 0 on the first tile
        const float off = POFF - m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(s[r], f, off));
            s[r] = p;
            psum += p;
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * corr + psum;
        m_run = m_new;
        if constexpr (PVS) {
            // the accumulators live in the domain 2^(e_cur + 15); a new running V exponent rides on the softmax correction
            const float ce = __builtin_amdgcn_ldexpf(corr, e_stage - e_cur);
            e_cur = e_stage;
#pragma unroll
            for (int j = 0; j < NVT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[j][r] *= ce;
            // O^T[c][query] += sum_key V[key][c] P[key][query].  B = P: the registers 8t .. 8t+7 of a lane are its B fragment of
            // K step t (keys 16t + 8(j>>2) + 4hl + (j&3)); A = V^T: two 4-key column gathers from the row-major [key][channel]
            // image with ds_read_b64_tr_b16.
            typedef short s16x4h __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4h *lds_s16x4;
            const unsigned char *vb = Vs + (4 * hl + ((lane & 15) >> 2)) * VROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float x[8] = {s[8 * t], s[8 * t + 1], s[8 * t + 2], s[8 * t + 3], s[8 * t + 4], s[8 * t + 5], s[8 * t + 6], s[8 * t + 7]};
                uint4 pf[2];
                split2x8h(x, pf[0], pf[1]);
#pragma unroll
                for (int j = 0; j < NVT; ++j) {
                    uint4 vf[2];
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const unsigned char *a = vb + (p * 32 + 16 * t) * VROWB + 64 * j;
                        const s16x4h lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a));
                        const s16x4h hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a + 8 * VROWB));
                        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                        vf[p] = make_uint4(l2.x, l2.y, h2.x, h2.y);
                    }
                    mfma_split3h(o[j], vf, pf);
                }
            }
        } else if constexpr (CV == 2) {
            const float *vs2 = reinterpret_cast<const float *>(Vs);
            float ax = 0.f, ay = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float2 vv = *reinterpret_cast<const float2 *>(vs2 + ((r & 3) + 8 * (r >> 2) + 4 * hl) * 2);
                ax += s[r] * vv.x;
                ay += s[r] * vv.y;
            }
            ax += __shfl_xor(ax, 32, 64);
            ay += __shfl_xor(ay, 32, 64);
            o2x = o2x * corr + ax;
            o2y = o2y * corr + ay;
        }
        if (more) note_max();            // waits for the loads issued above this tile's arithmetic
        __syncthreads();                 // every wave is done with this tile; the next tile's maxima are visible
        if (more) {
            stage();
            __syncthreads();
        }
    }
    if (nsplit > 1) {
        // part: [split][batch][len][CV + 2], in natural units
        float *pp = part + (((size_t)split * gridDim.y + b) * L + qclamp) * (CV + 2);
        if (qlive) {
            if constexpr (PVS) {
#pragma unroll
                for (int j = 0; j < NVT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) pp[j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl] = __builtin_amdgcn_ldexpf(o[j][r], -e_cur - (int)POFF);
            } else if (CV == 2 && hl == 0) {
                pp[0] = o2x; pp[1] = o2y;
            }
            if (hl == 0) { pp[CV] = m_run; pp[CV + 1] = __builtin_amdgcn_ldexpf(l_run, -(int)POFF); }
        }
        return;
    }
    const float inv = 1.0f / l_run;
    if (stats && qlive && hl == 0) {     // row statistics of the softmax (max, sum): used by the column-sum pass
        stats[(tb + qi) * 2] = m_run * kLn2H;   // natural-log units for the column-sum kernel
        stats[(tb + qi) * 2 + 1] = __builtin_amdgcn_ldexpf(l_run, -(int)POFF);
    }
    if constexpr (PVS) {
        // O^T[c][query]: lane = query nl, registers = channels (r&3)+8(r>>2)+4hl of tile j
        if (qlive) {
            const float ie = __builtin_amdgcn_ldexpf(inv, -e_cur);      // the 2^15 of P cancels against l
            float *op = out + row(qi) * CV;
#pragma unroll
            for (int j = 0; j < NVT; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 4)   // registers r..r+3 are four consecutive channels
                    *reinterpret_cast<float4 *>(op + j * 32 + 8 * (r >> 2) + 4 * hl) =
                        make_float4(o[j][r] * ie, o[j][r + 1] * ie, o[j][r + 2] * ie, o[j][r + 3] * ie);
        }
    } else if constexpr (CV == 2) {
        if (qlive && hl == 0) *reinterpret_cast<float2 *>(out + row(qi) * 2) = make_float2(o2x * inv, o2y * inv);
    }
}

// Column sums of a row-softmax, given its row statistics: colsum[b][j] = sum_i exp(scale q_i.k_j - m_i) / l_i
// (pasmnet/utils.py:31,34).  One wave per 32 keys; queries on the MFMA rows (= registers), the key on the lane: the sum over
// queries is a sum over registers plus one cross-half shuffle, in a fixed order (deterministic).
template <int C>
__global__ __launch_bounds__(256, 2) void attention16_colsum_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                                    const float *__restrict__ stats, float *__restrict__ colsum, int L,
                                                                    float scale) {
    constexpr int QV4 = (32 * C / 4) / 256;         // float4 per thread of one query tile
    constexpr int SROW = kSsRowH(C);
    __shared__ __attribute__((aligned(16))) unsigned char Qs[2 * 32 * SROW];
    __shared__ float2 Ms[32];
    __shared__ __attribute__((aligned(16))) float Mx[4];
    __shared__ float Gs[1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    int bxo, b;
    xcd_order(bxo, b);
    const int j0 = (bxo * 4 + wave) * 32;
    const size_t tb = (size_t)b * L;
    const int kj = j0 + nl;
    uint4 kf[C / 16][2];                            // B fragments: this lane's key row (a workgroup's surplus waves clamp)
    float gk;
    {
        const float qs = scale * kLog2eH;
        const float *kp = k + (tb + (kj < L ? kj : L - 1)) * C + 8 * hl;
        float x[C / 16][8];
        float amax = 0.f;
#pragma unroll
        for (int st = 0; st < C / 16; ++st) {
            const float4 t0 = *reinterpret_cast<const float4 *>(kp + 16 * st), t1 = *reinterpret_cast<const float4 *>(kp + 16 * st + 4);
            x[st][0] = t0.x * qs; x[st][1] = t0.y * qs; x[st][2] = t0.z * qs; x[st][3] = t0.w * qs;
            x[st][4] = t1.x * qs; x[st][5] = t1.y * qs; x[st][6] = t1.z * qs; x[st][7] = t1.w * qs;
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(x[st][j]));
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const int ek = scale_exp_h(amax, 0);
        const float sk = pow2i_h(ek);
        gk = pow2i_h(-ek);
#pragma unroll
        for (int st = 0; st < C / 16; ++st) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[st][j] *= sk;
            split2x8h(x[st], kf[st][0], kf[st][1]);
        }
    }
    float4 qpre[QV4];
    float2 mpre = make_float2(0.f, 0.f);
    auto fetch = [&](int i0) {
#pragma unroll
        for (int i = 0; i < QV4; ++i) {
            const int f = tid + i * 256, qq = f / (C / 4), c4 = f - qq * (C / 4);
            qpre[i] = (i0 + qq < L) ? *reinterpret_cast<const float4 *>(q + (tb + i0 + qq) * C + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < 32) {
            // an out-of-range query contributes exp2(0 - inf) * 0 = 0
            mpre = (i0 + tid < L) ? *reinterpret_cast<const float2 *>(stats + (tb + i0 + tid) * 2) : make_float2(INFINITY, INFINITY);
        }
    };
    auto note_max = [&]() {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < QV4; ++i)
            m = fmaxf(m, fmaxf(fmaxf(fabsf(qpre[i].x), fabsf(qpre[i].y)), fmaxf(fabsf(qpre[i].z), fabsf(qpre[i].w))));
        m = wave_max_nonneg_h(m);
        if (lane == 0) Mx[wave] = m;
    };
    auto stage = [&]() {
        const float4 m0 = *reinterpret_cast<const float4 *>(Mx);
        const int eq = __builtin_amdgcn_readfirstlane(scale_exp_h(fmaxf(fmaxf(m0.x, m0.y), fmaxf(m0.z, m0.w)), 0));
        const float sqv = pow2i_h(eq);
        if (tid == 0) Gs[0] = pow2i_h(-eq);
#pragma unroll
        for (int i = 0; i < QV4; ++i) {
            const int f = tid + i * 256, qq = f / (C / 4), c4 = f - qq * (C / 4);
            unsigned int h0, l0, h1, l1;
            split2x2h(qpre[i].x * sqv, qpre[i].y * sqv, h0, l0);
            split2x2h(qpre[i].z * sqv, qpre[i].w * sqv, h1, l1);
            unsigned char *qd = Qs + qq * SROW + 8 * c4;
            *reinterpret_cast<uint2 *>(qd) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(qd + 32 * SROW) = make_uint2(l0, l1);
        }
        if (tid < 32) Ms[tid] = make_float2(mpre.x * kLog2eH, 1.0f / mpre.y);   // (max in log2 units, 1 / sum)
    };
    float acc = 0.f;
    fetch(0);
    note_max();
    __syncthreads();
    stage();
    __syncthreads();
    for (int i0 = 0; i0 < L; i0 += 32) {
        const bool more = i0 + 32 < L;
        if (more) fetch(i0 + 32);
        f32x16h s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        const unsigned char *qp = Qs + nl * SROW + 16 * hl;
        uint4 ac[2], an[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) ac[p] = *reinterpret_cast<const uint4 *>(qp + p * 32 * SROW);
#pragma unroll
        for (int st = 0; st < C / 16; ++st) {
            if (st + 1 < C / 16) {
#pragma unroll
                for (int p = 0; p < 2; ++p) an[p] = *reinterpret_cast<const uint4 *>(qp + p * 32 * SROW + 32 * (st + 1));
            }
            mfma_split3h(s, ac, kf[st]);
#pragma unroll
            for (int p = 0; p < 2; ++p) ac[p] = an[p];
        }
        // lane: key nl; s[r] * f = log2-domain score of query i0 + (r&3)+8(r>>2)+4hl
        const float f = Gs[0] * gk;
        float2 ml[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ml[r] = Ms[(r & 3) + 8 * (r >> 2) + 4 * hl];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc = fmaf(__builtin_amdgcn_exp2f(fmaf(s[r], f, -ml[r].x)), ml[r].y, acc);
        if (more) note_max();
        __syncthreads();                 // every wave is done with this tile; the next tile's maxima are visible
        if (more) {
            stage();
            __syncthreads();
        }
    }
    acc += __shfl_xor(acc, 32, 64);
    if (hl == 0 && kj < L) colsum[tb + kj] = acc;
}

// ---- launchers (called by the C entry points of gmflow.hip) ----------------------------------------------------------
bool attention16_enabled() {
    static const int on = [] { const char *e = getenv("CT_HIP_ATT16"); return e ? atoi(e) : 1; }();
    return on != 0;
}

void attention16_tokens128(const float *q, const float *k, const float *v, const int *region, const int *rowmap, float *out, int batch,
                           int len, int cv, float scale, int nsplit, float *ws, long long kv_shift, long long kv_total, hipStream_t s) {
    dim3 grid((len + 127) / 128, batch, nsplit);
    float *nostats = nullptr;
#define CT_ATT16(CVV, MAPPED) hipLaunchKernelGGL((attention16_tokens_kernel<128, CVV, MAPPED>), grid, dim3(256), 0, s, q, k, v, region, rowmap, out, nostats, len, scale, ws, kv_shift, kv_total)
    if (cv == 128) { if (rowmap) CT_ATT16(128, true); else CT_ATT16(128, false); }
    else { if (rowmap) CT_ATT16(2, true); else CT_ATT16(2, false); }
#undef CT_ATT16
}

void attention16_rows64(const float *q, const float *k, const float *v, float *out, float *stats, int batch, int len, float scale,
                        hipStream_t s) {
    dim3 grid((len + 127) / 128, batch);
    const int *noreg = nullptr;
    if (v) hipLaunchKernelGGL((attention16_tokens_kernel<64, 96, false>), grid, dim3(256), 0, s, q, k, v, noreg, noreg, out, stats, len, scale, (float *)nullptr, 0LL, 0LL);
    else hipLaunchKernelGGL((attention16_tokens_kernel<64, 0, false>), grid, dim3(256), 0, s, q, k, v, noreg, noreg, out, stats, len, scale, (float *)nullptr, 0LL, 0LL);
}

void attention16_colsum64(const float *q, const float *k, const float *stats, float *colsum, int batch, int len, float scale, hipStream_t s) {
    dim3 grid((len + 127) / 128, batch);
    hipLaunchKernelGGL((attention16_colsum_kernel<64>), grid, dim3(256), 0, s, q, k, stats, colsum, len, scale);
}

}  // namespace ct
