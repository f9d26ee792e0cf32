// cnn.hip -- DCMCS3DI forward building blocks on gfx950 (MI355X), exact-float32 MFMA.
//
// Replaces the ATen op sequences of the reference's CNN path:
//   B1/B2/B3/B5  Conv2d 3x3 / 1x1 (+bias, +LeakyReLU(0.01), +residual, +clamp)
//                methods/dcmcs3di.py:41-51, pasmnet/backbone.py:8-15, pasmnet/attention.py:13-16
//                -> conv_mfma_kernel<KS, MT>  (implicit GEMM, LDS-tiled, v_mfma_f32_32x32x2_f32)
//   B4           cost = Q.K/c, softmax, warp(att @ V), valid mask (column sums > 0.1)
//                pasmnet/attention.py:39-46, pasmnet/utils.py:30-35,123-125, dcmcs3di.py:58,65
//                -> pam_attend_kernel<MODE> (+ pam_valid_kernel): never writes a [H,W,W] tensor
//                unless the caller asks for the attention maps.
//
// Layout: NCHW float32, exactly what the reference's modules exchange (no layout change at the
// boundary).  The GEMM is oriented M = output channels (weights are the A operand), N = 32
// consecutive pixels of one image row (activations are the B operand), so that
//   * the B operand is read from the LDS halo tile at consecutive addresses (conflict free),
//   * the 32x32 accumulator (column = pixel on the lane, rows = channels in registers) is
//     stored as 128-byte row segments straight into the NCHW planes.
// Arithmetic is float32 in, float32 accumulate (bitwise an fmaf chain in k order): the dense
// peak this path is priced against is the FP32 matrix rate, 157.3 TFLOP/s.
#include "ct_common.h"
#include "ct_conv.h"

namespace ct {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kConvTH = 8;      // output rows per workgroup (two per wave)
constexpr int kConvTW = 32;     // output columns per workgroup (= MFMA N)
constexpr int kConvChunk = 32;  // input channels staged in LDS at a time (one pipeline stage)
constexpr int kNumCUs = 256;    // MI355X

// ---------------------------------------------------------------------------------------------
// conv_mfma_kernel: out[n][co][y][x] = epilogue( bias[co] + sum_{ci,ky,kx} w[co][ci][ky][kx] in[n][ci][y+ky-p][x+kx-p] )
//   wp   : weights packed as [tap][cin_pair][2][MT*32]  (zero padded in cin and cout)
//   grid : persistent, two workgroups per CU walking the (n, y/8, x/32) tiles; block 256;
//          dynamic LDS = halo tile + 2 weight slices (double buffer); the NEXT tile's halo is prefetched into
//          registers (43 VGPRs) while the current one is multiplied, so HBM/L2 latency hides under the MFMAs
// ---------------------------------------------------------------------------------------------

// GEN = false: the DCMCS3DI instantiations, which only know LeakyReLU(0.01) -- the full switch in the epilogue costs
// them 1.5 % (measured r01)
template <bool GEN>
__device__ __forceinline__ float conv_act(float v, int act) {
    if (!GEN) return v > 0.f ? v : 0.01f * v;
    switch (act) {
        case 1: return v > 0.f ? v : 0.01f * v;
        case 2: return v > 0.f ? v : 0.f;
        case 3: return 1.0f / (1.0f + expf(-v));
        case 4: return tanhf(v);
        case 5: return v / (1.0f + expf(-v));        // swish
        default: return v;
    }
}

template <int KH, int KW, int MT, bool VEC, bool GEN>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs a, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int PADY = KH / 2, PADX = KW / 2;
    constexpr bool HALO = (KW > 1);
    constexpr int ROWS = kConvTH + KH - 1;
    constexpr int TWP = HALO ? kConvTW + 8 : kConvTW;   // LDS row = image columns [x0-4, x0+36): the 32 centre columns are 16-byte aligned
    constexpr int COL0 = HALO ? 4 : 0;                  // LDS column of image column x0
    constexpr int CS = ROWS * TWP;                      // floats per channel in the LDS tile
    constexpr int NV4 = kConvChunk * ROWS * (kConvTW / 4);   // centre float4 of one staged (tile, 32-channel chunk)
    constexpr int PF4 = NV4 / 256;                      // 10 (3x3) / 8 (1x1) per thread, exact
    constexpr int HC = KW - 1;                          // halo columns: x0-PADX..x0-1 and x0+32..x0+31+PADX
    constexpr int NE = kConvChunk * ROWS * HC;
    constexpr int PFE = (NE + 255) / 256;               // 3 (3x3) / 4 (1x5) / 0
    static_assert(NV4 % 256 == 0, "tile geometry");
    constexpr int COUTP = MT * 32;
    constexpr int RPW = kConvTH / 4;           // output rows per wave
    constexpr int TAPS = KH * KW;
    extern __shared__ float smem[];
    float *tin = smem;                         // [kConvChunk][ROWS][TWP]
    float *stg = smem + kConvChunk * CS + (threadIdx.x >> 6) * (32 * 32);   // per-wave [32 ch][32 px] transpose buffer

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const size_t plane = (size_t)a.H * a.W;
    const int cin_pairs_total = (a.cin + 1) >> 1;
    const int n_chunks = (a.cin + kConvChunk - 1) / kConvChunk;
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n_stages = my_tiles * n_chunks;  // a stage = (tile, channel chunk)

    // ---- (tile, chunk) halo tile -> registers; the loads stay in flight while the previous stage computes ----
    // Centre columns as 16-byte loads (a 4-byte-per-lane load costs ~620 cycles per wave-instruction on this chip,
    // tools/ubench/store_rates.hip), the two halo columns as scalars.  Offsets are 32-bit relative to the chunk base.
    float4 pf4[PF4];
    float pfe[PFE > 0 ? PFE : 1];
    const unsigned int uplane = (unsigned int)plane;
    auto fetch_tile = [&](int stage) {
        const int k = stage / n_chunks, chunk = stage - k * n_chunks;
        const int t = (blockIdx.x + k * gridDim.x) / a.groups;   // the group index varies fastest: neighbours share the input tile
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int x0 = tx * kConvTW, y0 = ty * kConvTH, c0 = chunk * kConvChunk;
        const int cc = (a.cin - c0) < kConvChunk ? (a.cin - c0) : kConvChunk;
        const float *in = a.in + (size_t)n * a.in_bstride + (size_t)c0 * plane;
#pragma unroll
        for (int i = 0; i < PF4; ++i) {
            int f = tid + i * 256;
            // opaque to the optimiser: otherwise LICM hoists all index decompositions (stage-invariant) out of the
            // stage loop and keeps them alive -> hundreds of spills
            asm volatile("" : "+v"(f));
            const int c = f / (ROWS * 8), rem = f - c * (ROWS * 8);
            const int yy = rem >> 3, g = rem & 7;
            const int gy = y0 + yy - PADY, gx = x0 + 4 * g;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < cc && gy >= 0 && gy < a.H) {
                const float *p = in + (unsigned int)c * uplane + (unsigned int)(gy * a.W + gx);
                if (VEC) {
                    if (gx < a.W) v = *reinterpret_cast<const float4 *>(p);   // W % 4 == 0: all four columns or none
                } else {
                    if (gx + 0 < a.W) v.x = p[0];
                    if (gx + 1 < a.W) v.y = p[1];
                    if (gx + 2 < a.W) v.z = p[2];
                    if (gx + 3 < a.W) v.w = p[3];
                }
            }
            pf4[i] = v;
        }
        if (HALO) {
#pragma unroll
            for (int j = 0; j < PFE; ++j) {
                int e = tid + j * 256;
                asm volatile("" : "+v"(e));
                const int c = e / (ROWS * HC), rem = e - c * (ROWS * HC);
                const int yy = rem / HC, side = rem - yy * HC;
                const int gy = y0 + yy - PADY, gx = side < PADX ? x0 - PADX + side : x0 + kConvTW + side - PADX;
                float v = 0.f;
                if (e < NE && c < cc && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = in[(unsigned int)c * uplane + (unsigned int)(gy * a.W + gx)];
                pfe[j] = v;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PF4; ++i) {
            int f = tid + i * 256;
            asm volatile("" : "+v"(f));
            const int c = f / (ROWS * 8), rem = f - c * (ROWS * 8);
            const int yy = rem >> 3, g = rem & 7;
            *reinterpret_cast<float4 *>(tin + c * CS + yy * TWP + COL0 + 4 * g) = pf4[i];
        }
        if (HALO) {
#pragma unroll
            for (int j = 0; j < PFE; ++j) {
                int e = tid + j * 256;
                asm volatile("" : "+v"(e));
                const int c = e / (ROWS * HC), rem = e - c * (ROWS * HC);
                const int yy = rem / HC, side = rem - yy * HC;
                if (e < NE) tin[c * CS + yy * TWP + (side < PADX ? COL0 - PADX + side : COL0 + kConvTW + side - PADX)] = pfe[j];
            }
        }
    };
    // ---- A operand: this lane's weights of one (tap, chunk) slice, straight from L1/L2 into registers ----
    // (all waves of all workgroups read the same 147 KB: it lives in L2/L1; no LDS staging and therefore
    //  no barrier per tap -- the waves of a workgroup only meet twice per stage)
    constexpr int KSTEPS = kConvChunk / 2;
    auto load_w = [&](int grp, int chunk, int tap, float (&w)[KSTEPS][MT]) {
        const int c0 = chunk * kConvChunk;
        const int cc = (a.cin - c0) < kConvChunk ? (a.cin - c0) : kConvChunk;
        const int ccp = (cc + 1) >> 1;
        const float *src = a.wp + (((size_t)grp * TAPS + tap) * cin_pairs_total + (c0 >> 1)) * 2 * COUTP + hl * COUTP + nl;
#pragma unroll
        for (int p = 0; p < KSTEPS; ++p)
#pragma unroll
            for (int m = 0; m < MT; ++m) w[p][m] = (p < ccp) ? src[p * 2 * COUTP + m * 32] : 0.f;
    };

    if (n_stages == 0) return;
#ifdef CT_CONV_PROFILE
    // diagnostic build: per-phase cycle totals of wave 0 of each workgroup -> a.prof[block][8]
    unsigned long long pt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt0;
#define CT_STAMP(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define CT_PHASE(i) do { unsigned long long t__; CT_STAMP(t__); pt[i] += t__ - pt0; pt0 = t__; } while (0)
    CT_STAMP(pt0);
#else
#define CT_PHASE(i) do { } while (0)
#endif
    f32x16 acc[RPW][MT];
    float wa[KSTEPS][MT], wb[KSTEPS][MT];
    // ---- phase stagger --------------------------------------------------------------------------------------
    // All workgroups start together and do identical work, so left alone they reach their epilogues (a burst
    // of output stores + residual loads, no MFMA) at the same instant, chip-wide, and the matrix pipes idle.
    // The workgroup in the odd hardware wave slot of its SIMD (= the second of the two co-resident ones)
    // therefore waits for half a tile of its partner's work: from then on one of the two is always in its
    // MFMA loop while the other drains/loads.  Pure scheduling: results do not depend on it.
    if (my_tiles >= 2) {
        const unsigned int hw_wave_slot = __builtin_amdgcn_s_getreg(4 | (0 << 6) | ((4 - 1) << 11));   // HW_ID[3:0]
        if (hw_wave_slot & 1) {
            const int half_tile_cycles = n_chunks * TAPS * KSTEPS * RPW * MT * 64 / 2;
            for (int c = 0; c < half_tile_cycles; c += 64 * 64) __builtin_amdgcn_s_sleep(64);
        }
    }
    // accumulator pre-load for tile k: residual (ResB skip, act == 0) or zero.  Issued right after the previous
    // tile's stores, i.e. a whole barrier + tile-store phase before the first MFMA needs it.
    const bool res_in_acc = (a.residual != nullptr) && (a.act == 0);
    // The accumulator of a lane holds 16 channels of ONE pixel, so a direct gather of the skip tensor (and the
    // final scatter of the result) would be 4-byte-per-lane accesses -- ~620 cycles per wave-instruction on this chip
    // (tools/ubench/store_rates.hip).  With VEC both go through a per-wave 32x32 LDS transpose instead: global
    // traffic is 16 bytes per lane (8 channel rows x 128 B per instruction), LDS does the re-layout.
    auto init_acc = [&](int k) {
        const int tg = blockIdx.x + k * gridDim.x, t = tg / a.groups, grp = tg - t * a.groups;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const float *__restrict__ res = res_in_acc ? a.residual + (size_t)n * a.res_bstride + (size_t)grp * COUTP * plane : nullptr;
        const int cout_g = a.cout - grp * COUTP;   // channels of this group that exist
        if (VEC) {
            // raw float4 rows land in the accumulator registers; finish_acc() re-lays them out at tile start
            const int x4 = tx * kConvTW + 4 * (lane & 7);
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const int y = ty * kConvTH + wave * RPW + q;
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
        } else {
            const int x = tx * kConvTW + nl;
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const int y = ty * kConvTH + wave * RPW + q;
                const bool inb = res_in_acc && (y < a.H) && (x < a.W);
                const unsigned int pix = (unsigned int)(y * a.W + x);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                        float v = 0.f;
                        if (inb && co < cout_g) v = res[(unsigned int)co * uplane + pix];
                        acc[q][m][r] = v;
                    }
            }
        }
    };
    // VEC: [8j + lane/8][4*(lane%8)..+3] rows -> (pixel nl, channels (r&3)+8(r>>2)+4hl) accumulator layout
    auto finish_acc = [&]() {
        if (!VEC) return;
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
    load_w((int)blockIdx.x % a.groups, 0, 0, wa);
    fetch_tile(0);
    init_acc(0);
    store_tile();
    for (int stage = 0; stage < n_stages; ++stage) {
        const int k = stage / n_chunks, chunk = stage - k * n_chunks;
        const int grp = (int)((blockIdx.x + k * gridDim.x) % a.groups);
        if (chunk == 0) {
            finish_acc();
            // the accumulators were pre-loaded with the residual (or zero) by init_acc(); add the bias here, so
            // that the epilogue has no load to wait for
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
        CT_PHASE(0);                                       // store_tile + acc init of the previous iteration
        __syncthreads();                                   // this stage's tile is visible
        CT_PHASE(1);
        const bool next_stage = (stage + 1 < n_stages);
        const int next_chunk = (chunk + 1 == n_chunks) ? 0 : chunk + 1;
        const int next_grp = (next_chunk == 0) ? (int)((blockIdx.x + (k + 1) * gridDim.x) % a.groups) : grp;
        auto compute_tap = [&](int tap, const float (&w)[KSTEPS][MT]) {
            const int ky = tap / KW, kx = tap - ky * KW;
            const float *brow = tin + hl * CS + (wave * RPW + ky) * TWP + (COL0 - PADX) + kx + nl;
            // B operands are read one k-step ahead of the MFMAs that consume them, and the scheduler is
            // pinned to "1 LDS read, then RPW*MT MFMAs" groups: a wave that has the matrix pipe to itself
            // (partner in its epilogue) then issues back to back instead of exposing the LDS latency per step
            float bc[RPW], bn[RPW];
#pragma unroll
            for (int q = 0; q < RPW; ++q) bc[q] = brow[q * TWP];
#pragma unroll
            for (int p = 0; p < KSTEPS; ++p) {
                if (p + 1 < KSTEPS) {
#pragma unroll
                    for (int q = 0; q < RPW; ++q) bn[q] = brow[(p + 1) * 2 * CS + q * TWP];
                }
#pragma unroll
                for (int q = 0; q < RPW; ++q)
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[p][m], bc[q], acc[q][m], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < RPW; ++q) bc[q] = bn[q];
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // the (paired) LDS read of step p+1
                __builtin_amdgcn_sched_group_barrier(0x008, RPW * MT, 0);   // the MFMAs of step p
            }
        };
        // taps alternate between the two weight register sets; the set not in use is being loaded
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const bool last_tap = (tap + 1 == TAPS);
            if ((tap & 1) == 0) {
                if (!last_tap) load_w(grp, chunk, tap + 1, wb);
                else if (next_stage) load_w(next_grp, next_chunk, 0, wb);
                if (tap == 0 && next_stage) fetch_tile(stage + 1);   // next halo tile: in flight under this stage's MFMAs
                compute_tap(tap, wa);
            } else {
                if (!last_tap) load_w(grp, chunk, tap + 1, wa);
                else if (next_stage) load_w(next_grp, next_chunk, 0, wa);
                compute_tap(tap, wb);
            }
        }
        if ((TAPS & 1) == 1) {   // odd tap count: the next stage's tap-0 weights sit in wb; move them to wa
#pragma unroll
            for (int p = 0; p < KSTEPS; ++p)
#pragma unroll
                for (int m = 0; m < MT; ++m) wa[p][m] = wb[p][m];
        }
        CT_PHASE(2);                                       // tap loop (MFMAs + prefetch issue)
        if (chunk + 1 == n_chunks) {
            // ---- epilogue: lane owns pixels (y0 + wave*RPW + q, x0+nl), channels (r&3)+8(r>>2)+4hl of each 32-tile ----
            const int t = (blockIdx.x + k * gridDim.x) / a.groups;
            const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
            float *__restrict__ out = a.out + (size_t)n * a.out_bstride + (size_t)grp * COUTP * plane;
            const float *__restrict__ res = a.residual ? a.residual + (size_t)n * a.res_bstride + (size_t)grp * COUTP * plane : nullptr;
            const bool late_res = (res != nullptr) && !res_in_acc;   // an activation *and* a skip: not in these models
            const int cout_g = a.cout - grp * COUTP;
            const bool full = (cout_g >= COUTP);   // uniform: no per-channel predicate in the common case
            const bool wide = VEC && !late_res && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const int y = ty * kConvTH + wave * RPW + q;
                if (wide) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = acc[q][m][r];
                            if (a.act) v = conv_act<GEN>(v, a.act);
                            if (a.clamp) v = fminf(fmaxf(v, 0.f), 1.f);
                            stg[((r & 3) + 8 * (r >> 2) + 4 * hl) * 32 + nl] = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                        const int x4 = tx * kConvTW + 4 * (lane & 7);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int co = m * 32 + (lane >> 3) + 8 * j;
                            const float4 v = *reinterpret_cast<const float4 *>(stg + ((lane >> 3) + 8 * j) * 32 + 4 * (lane & 7));
                            if (y < a.H && x4 < a.W && (full || co < cout_g))
                                *reinterpret_cast<float4 *>(out + (unsigned int)co * uplane + (unsigned int)(y * a.W + x4)) = v;
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                } else {
                    const int x = tx * kConvTW + nl;
                    if (y < a.H && x < a.W) {
                        const unsigned int pix = (unsigned int)(y * a.W + x);
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                                float v = acc[q][m][r];
                                if (a.act) v = conv_act<GEN>(v, a.act);
                                if (late_res && (full || co < cout_g)) v += res[(unsigned int)co * uplane + pix];
                                if (a.clamp) v = fminf(fmaxf(v, 0.f), 1.f);
                                if (full || co < cout_g) out[(unsigned int)co * uplane + pix] = v;
                            }
                        }
                    }
                }
            }
            if (next_stage) init_acc(k + 1);   // the stores above are fire-and-forget; start the next tile's skip loads
        }
        CT_PHASE(3);                       // epilogue
        __syncthreads();                   // every wave is done reading this stage's tile
        CT_PHASE(4);
        if (next_stage) store_tile();      // waits for the prefetched loads, then fills the tile
    }
#ifdef CT_CONV_PROFILE
    CT_PHASE(0);
    if (tid == 0 && a.prof) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.prof[(size_t)blockIdx.x * 8 + i] = pt[i];
    }
#endif
}

template <int KH, int KW, int MT, bool GEN>
int launch_conv(const ConvArgs &a, int N, hipStream_t s) {
    constexpr int ROWS = kConvTH + KH - 1, TWP = (KW > 1) ? kConvTW + 8 : kConvTW;
    const size_t lds = (size_t)(kConvChunk * ROWS * TWP + 4 * 32 * 32) * sizeof(float);
    // 16-byte loads need every (channel, row, 4-column group) address 16-byte aligned
    const bool vec = (a.W % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.in) & 15) == 0) && (a.in_bstride % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.out) & 15) == 0) && (a.out_bstride % 4 == 0) &&
                     (!a.residual || (((reinterpret_cast<uintptr_t>(a.residual) & 15) == 0) && (a.res_bstride % 4 == 0)));
    const int tiles_x = (a.W + kConvTW - 1) / kConvTW, tiles_y = (a.H + kConvTH - 1) / kConvTH;
    const long long n_tiles = (long long)tiles_x * tiles_y * N * a.groups;
    if (n_tiles > 0x7fffffffLL) return CT_E_BADARG;
    const int grid = n_tiles < 2 * kNumCUs ? (int)n_tiles : 2 * kNumCUs;   // persistent: two workgroups per CU
    if (vec) hipLaunchKernelGGL((conv_mfma_kernel<KH, KW, MT, true, GEN>), dim3(grid), dim3(256), lds, s, a, tiles_x, tiles_y, (int)n_tiles);
    else hipLaunchKernelGGL((conv_mfma_kernel<KH, KW, MT, false, GEN>), dim3(grid), dim3(256), lds, s, a, tiles_x, tiles_y, (int)n_tiles);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

// Stride-1 "same" convolutions with 64-channel output groups (GMFlow's backbone / refinement convs, gmflow.hip).
// Returns 1 when the geometry has no fast kernel (the caller then uses its generic one).
int conv_fast(const ConvArgs &a, int N, int kh, int kw, hipStream_t s) {
    if (kh == 3 && kw == 3) return launch_conv<3, 3, 2, true>(a, N, s);
    if (kh == 1 && kw == 1) return launch_conv<1, 1, 2, true>(a, N, s);
    if (kh == 1 && kw == 5) return launch_conv<1, 5, 2, true>(a, N, s);
    if (kh == 5 && kw == 1) return launch_conv<5, 1, 2, true>(a, N, s);
    return 1;
}

// ---------------------------------------------------------------------------------------------
// Parallax attention for one direction (pasmnet/attention.py:39-46 + utils.py:30-35,123-125).
//   S[i][j] = (1/C) sum_c Q[c][h][i] K[c][h][j] ; P = softmax_j S
//   MODE 0 (attend): out[c][h][i] = sum_j P[i][j] V[c][h][j] for the CV channels of `v` and the 3 of `rgb`
//   MODE 1 (colsum): colpart[h][tile][j] = sum_{i in tile} P[i][j]      (valid mask numerator)
//   optional att[h][i][j] = P (the API's attention map), written only when the pointer is non-null
// grid = (ceil(W/32), H, N), block 256, dynamic LDS = 32*(W+1) floats + V staging
// ---------------------------------------------------------------------------------------------
struct PamArgs {
    const float *q, *k;        // [N][C][H][W]
    const float *v;            // [N][CV][H][W]      (MODE 0)
    const float *rgb;          // [N][3][H][W]       (MODE 0)
    float *out_v, *out_rgb;    // [N][CV][H][W], [N][3][H][W]
    float *colpart;            // [N][H][tiles][W]   (MODE 1)
    float *att;                // nullable [N][H][W][W]
    int C, CV, H, W;
};

constexpr int kPamKC = 128;  // keys staged per S step (one 32-key MFMA tile per wave)
constexpr int kPamJC = 64;   // keys staged per PV step
constexpr int kPamC = 64;    // channels of q/k (DCMCS3DI: channels = 64)

// Cooperative copy of rows [r][j0 .. j0+NCOLS) of a row-major [rows][plane-strided] slab into registers
// (NV4 float4 per thread), 16-byte loads when the row base is aligned, zero fill outside [0,W) x [0,rows).
template <int NROWS, int NCOLS>
struct RowChunk {
    static constexpr int V4_PER_ROW = NCOLS / 4;
    static constexpr int NV4 = (NROWS * V4_PER_ROW + 255) / 256;
    float4 v[NV4];
    // src(r) = row pointer of row r at column 0, or nullptr for a zero row
    template <typename RowPtr>
    __device__ __forceinline__ void fetch(RowPtr src, int j0, int W, bool vec, int tid) {
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            int f = tid + i * 256;
            asm volatile("" : "+v"(f));
            const int r = f / V4_PER_ROW, g = f - r * V4_PER_ROW;
            const int j = j0 + 4 * g;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            const float *p = (r < NROWS) ? src(r) : nullptr;
            if (p && j < W) {
                if (vec) x = *reinterpret_cast<const float4 *>(p + j);
                else {
                    x.x = p[j];
                    if (j + 1 < W) x.y = p[j + 1];
                    if (j + 2 < W) x.z = p[j + 2];
                    if (j + 3 < W) x.w = p[j + 3];
                }
            }
            v[i] = x;
        }
    }
    // LDS image [NROWS][LD]; LD % 4 == 0 -> one 16-byte store, else four scalar stores
    template <int LD>
    __device__ __forceinline__ void store(float *lds, int tid) const {
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int f = tid + i * 256;
            const int r = f / V4_PER_ROW, g = f - r * V4_PER_ROW;
            if (r < NROWS) {
                float *d = lds + r * LD + 4 * g;
                if (LD % 4 == 0) *reinterpret_cast<float4 *>(d) = v[i];
                else { d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w; }
            }
        }
    }
};

// TQ = queries per workgroup: 32, or 16 for wide images (one TQ x W tile of S must fit the 160 KiB LDS; the MFMA
// tile stays 32 x 32 with the upper 16 query rows idle)
template <int MODE, int TQ>
__global__ __launch_bounds__(256) void pam_attend_kernel(PamArgs a) {
    extern __shared__ float smem[];
    const int W = a.W, SW = W | 1;                 // odd row stride: column reads are conflict free
    constexpr int VST = kPamJC + 1;                // odd: the PV A-operand reads walk down a column
    float *S = smem;                               // [TQ][SW]
    float *Qs = smem + TQ * SW;                    // [64][TQ]
    float *Ks = Qs + kPamC * TQ;               // [64][128]   (S phase)  /  Vs [96][65] (PV phase), aliased
    float *Vs = Ks;
    const int i0 = blockIdx.x * TQ, h = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const size_t plane = (size_t)a.H * W;
    const float *q = a.q + (size_t)n * kPamC * plane + (size_t)h * W;
    const float *k = a.k + (size_t)n * kPamC * plane + (size_t)h * W;
    const float inv_c = 1.0f / (float)kPamC;
    const bool vec = (W % 4 == 0) && (((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k)) & 15) == 0);

    // ---- S tile = Q^T K / C : M = query i (A = Q[c][i]), N = key j (B = K[c][j]), K = channels ----
    {
        RowChunk<kPamC, TQ> qc;
        qc.fetch([&](int r) { return q + (size_t)r * plane; }, i0, W, vec, tid);
        RowChunk<kPamC, kPamKC> kc;
        kc.fetch([&](int r) { return k + (size_t)r * plane; }, 0, W, vec, tid);
        qc.template store<TQ>(Qs, tid);
        kc.template store<kPamKC>(Ks, tid);
        __syncthreads();
        float qa[kPamC / 2];                        // this lane's A operands for all 32 k-steps
#pragma unroll
        for (int p = 0; p < kPamC / 2; ++p) qa[p] = (nl < TQ) ? Qs[(2 * p + hl) * TQ + nl] : 0.f;
        for (int j0 = 0; j0 < W; j0 += kPamKC) {
            const bool more = (j0 + kPamKC < W);
            if (more) kc.fetch([&](int r) { return k + (size_t)r * plane; }, j0 + kPamKC, W, vec, tid);   // next chunk in flight
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float *brow = Ks + hl * kPamKC + wave * 32 + nl;
#pragma unroll
            for (int p = 0; p < kPamC / 2; ++p) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[p], brow[p * 2 * kPamKC], acc, 0, 0, 0);
            // D[i][j]: lane holds key column j, query rows (r&3)+8(r>>2)+4hl
            const int kj = j0 + wave * 32 + nl;
            if (kj < W) {
#pragma unroll
                for (int r = 0; r < TQ / 2; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * hl) * SW + kj] = acc[r] * inv_c;   // rows < TQ
            }
            __syncthreads();                        // everyone is done with this K chunk
            if (more) {
                kc.template store<kPamKC>(Ks, tid);
                __syncthreads();
            }
        }
    }
    // prefetch the first V chunk while the softmax runs (MODE 0)
    const int CT = a.CV + 3;                        // feature channels + the 3 RGB channels of `right`
    const float *v = MODE == 0 ? a.v + (size_t)n * a.CV * plane + (size_t)h * W : nullptr;
    const float *rgb = MODE == 0 ? a.rgb + (size_t)n * 3 * plane + (size_t)h * W : nullptr;
    const bool vecv = MODE == 0 && (W % 4 == 0) &&
                      (((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(rgb)) & 15) == 0);
    auto vrow = [&](int r) -> const float * {
        return r < a.CV ? v + (size_t)r * plane : (r < CT ? rgb + (size_t)(r - a.CV) * plane : nullptr);
    };
    RowChunk<96, kPamJC> vc;
    if (MODE == 0) vc.fetch(vrow, 0, W, vecv, tid);

    // ---- row softmax (F.softmax(dim=-1)): 8 rows per wave ----
    for (int rr = 0; rr < TQ / 4; ++rr) {
        const int row = wave * (TQ / 4) + rr;
        float *srow = S + row * SW;
        float mx = -INFINITY;
        for (int j = lane; j < W; j += 64) mx = fmaxf(mx, srow[j]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float sum = 0.f;
        for (int j = lane; j < W; j += 64) {
            const float e = expf(srow[j] - mx);
            srow[j] = e;
            sum += e;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        const float inv = 1.0f / sum;
        const bool live = (i0 + row) < W;
        for (int j = lane; j < W; j += 64) {
            const float p = live ? srow[j] * inv : 0.f;
            srow[j] = p;
            if (a.att && live) a.att[(((size_t)n * a.H + h) * W + (i0 + row)) * W + j] = p;
        }
    }
    __syncthreads();
    if (MODE == 1) {
        // column sums over this tile's (live) query rows, fixed order
        float *dst = a.colpart + (((size_t)n * a.H + h) * gridDim.x + blockIdx.x) * W;
        for (int j = tid; j < W; j += 256) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < TQ; ++r) s += S[r * SW + j];
            dst[j] = s;
        }
        return;
    }
    // ---- out[c][i] = sum_j V[c][j] P[i][j] : M = channel (A = V, staged [c][JC+1]), N = query (B = P) ----
    const int mt_total = (CT + 31) / 32;            // 3 for CV = 64
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    vc.template store<VST>(Vs, tid);
    __syncthreads();
    for (int j0 = 0; j0 < W; j0 += kPamJC) {
        const bool more = (j0 + kPamJC < W);
        if (more) vc.fetch(vrow, j0 + kPamJC, W, vecv, tid);     // next chunk in flight under the MFMAs
        if (wave < mt_total) {
            const float *arow = Vs + (wave * 32 + nl) * VST + hl;
            const float *brow = S + (nl < TQ ? nl : 0) * SW + j0 + hl;
#pragma unroll
            for (int jj = 0; jj < kPamJC; jj += 2) {
                const float bv = (nl < TQ && j0 + jj + hl < W) ? brow[jj] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[jj], bv, acc, 0, 0, 0);
            }
        }
        __syncthreads();
        if (more) {
            vc.template store<VST>(Vs, tid);
            __syncthreads();
        }
    }
    if (wave < mt_total) {
        const int x = i0 + nl;
        if (nl < TQ && x < W) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                if (c < a.CV) a.out_v[((size_t)n * a.CV + c) * plane + (size_t)h * W + x] = acc[r];
                else if (c < CT) a.out_rgb[((size_t)n * 3 + (c - a.CV)) * plane + (size_t)h * W + x] = acc[r];
            }
        }
    }
}

// valid[n][0][h][j] = (sum over query tiles of colpart) > 0.1, as 0.0 / 1.0 (bool -> float promotion of
// torch.cat, dcmcs3di.py:59); colsum (pre-threshold) is also written for the parity tests.
__global__ void pam_valid_kernel(const float *__restrict__ colpart, int tiles, int H, int W, float *__restrict__ valid,
                                 float *__restrict__ colsum) {
    const int n = blockIdx.z, h = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= W) return;
    const float *p = colpart + (((size_t)n * H + h) * tiles) * W + j;
    float s = 0.f;
    for (int t = 0; t < tiles; ++t) s += p[(size_t)t * W];
    const size_t o = ((size_t)n * H + h) * W + j;
    valid[o] = s > 0.1f ? 1.0f : 0.0f;
    if (colsum) colsum[o] = s;
}

template <int MODE, int TQ>
static int launch_pam_tq(const PamArgs &a, int N, hipStream_t s, size_t lds) {
    static DynLdsAttr attr;             // per instantiation, per device
    {
        hipError_t e = attr.ensure(reinterpret_cast<const void *>(&pam_attend_kernel<MODE, TQ>), lds);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid((a.W + TQ - 1) / TQ, a.H, N);
    hipLaunchKernelGGL((pam_attend_kernel<MODE, TQ>), grid, dim3(256), lds, s, a);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

static int pam_tq(int W) {   // queries per workgroup such that S (TQ x W) + Q tile + K chunk fit the 160 KiB LDS
    const size_t lds32 = ((size_t)32 * (W | 1) + kPamC * 32 + kPamC * kPamKC) * sizeof(float);
    const size_t lds16 = ((size_t)16 * (W | 1) + kPamC * 16 + kPamC * kPamKC) * sizeof(float);
    if (lds32 <= 160 * 1024) return 32;
    if (lds16 <= 160 * 1024) return 16;
    return 0;
}

template <int MODE>
static int launch_pam(const PamArgs &a, int N, hipStream_t s) {
    static_assert(96 * (kPamJC + 1) <= kPamC * kPamKC, "V chunk must fit the K chunk region");
    if (a.C != kPamC) return CT_E_BADARG;
    const int tq = pam_tq(a.W);
    if (tq == 0) return CT_E_BADARG;   // W > 1982: needs the streaming (online-softmax) variant
    const size_t lds = ((size_t)tq * (a.W | 1) + kPamC * tq + kPamC * kPamKC) * sizeof(float);
    return tq == 32 ? launch_pam_tq<MODE, 32>(a, N, s, lds) : launch_pam_tq<MODE, 16>(a, N, s, lds);
}

}  // namespace ct

extern "C" {

int ct_conv2d_f32(const float *in, const float *wp, const float *bias, const float *residual, float *out, int n,
                  int cin, int cout, int h, int w, int ksize, long long in_bstride, long long out_bstride,
                  long long res_bstride, int act, int clamp, void *stream) {
    if (!in || !wp || !bias || !out || n < 0 || cin < 1 || cout < 1 || cout > 64 || h < 0 || w < 0) return CT_E_BADARG;
    if ((ksize != 1 && ksize != 3) || (act != 0 && act != 1)) return CT_E_BADARG;
    if (n == 0 || h == 0 || w == 0) return CT_OK;
    ct::ConvArgs a;
    a.in = in; a.in2 = nullptr; a.cin1 = cin; a.in2_bstride = 0; a.wp = wp; a.bias = bias; a.residual = residual; a.out = out;
    a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.in_bstride = in_bstride; a.out_bstride = out_bstride; a.res_bstride = res_bstride;
    a.act = act; a.clamp = clamp; a.groups = 1; a.prof = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const int mt = cout > 32 ? 2 : 1;
    if (ksize == 3) return mt == 2 ? ct::launch_conv<3, 3, 2, false>(a, n, s) : ct::launch_conv<3, 3, 1, false>(a, n, s);
    return mt == 2 ? ct::launch_conv<1, 1, 2, false>(a, n, s) : ct::launch_conv<1, 1, 1, false>(a, n, s);
}

#ifdef CT_CONV_PROFILE
// diagnostic build only (tools/prof_conv.py): 3x3 64->64 conv with per-phase cycle stamps, prof[grid][8]
int ct_conv2d_prof_f32(const float *in, const float *wp, const float *bias, const float *residual, float *out, int n,
                       int cin, int cout, int h, int w, unsigned long long *prof, void *stream) {
    ct::ConvArgs a;
    a.in = in; a.in2 = nullptr; a.cin1 = cin; a.in2_bstride = 0; a.wp = wp; a.bias = bias; a.residual = residual; a.out = out;
    a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.in_bstride = (long long)cin * h * w; a.out_bstride = (long long)cout * h * w; a.res_bstride = a.out_bstride;
    a.act = 0; a.clamp = 0; a.groups = 1; a.prof = prof;
    return ct::launch_conv<3, 3, 2, false>(a, n, (hipStream_t)stream);
}
#endif

size_t ct_pam_workspace_bytes(int n, int h, int w) {
    if (n < 0 || h < 0 || w < 0) return 0;
    const int tq = ct::pam_tq(w);
    if (tq == 0) return 0;
    return (size_t)n * h * ((w + tq - 1) / tq) * w * sizeof(float);
}

int ct_pam_attend_f32(const float *q, const float *k, const float *v, const float *rgb, float *out_v, float *out_rgb,
                      float *att, int n, int c, int cv, int h, int w, void *stream) {
    if (!q || !k || !v || !rgb || !out_v || !out_rgb || n < 0 || c < 1 || cv < 1 || cv > 93 || h < 0 || w < 0)
        return CT_E_BADARG;
    if (n == 0 || h == 0 || w == 0) return CT_OK;
    ct::PamArgs a;
    a.q = q; a.k = k; a.v = v; a.rgb = rgb; a.out_v = out_v; a.out_rgb = out_rgb; a.colpart = nullptr; a.att = att;
    a.C = c; a.CV = cv; a.H = h; a.W = w;
    return ct::launch_pam<0>(a, n, (hipStream_t)stream);
}

int ct_pam_valid_f32(const float *q, const float *k, float *valid, float *colsum, float *att, int n, int c, int h, int w,
                     void *ws, size_t ws_bytes, void *stream) {
    if (!q || !k || !valid || n < 0 || c < 1 || h < 0 || w < 0) return CT_E_BADARG;
    if (n == 0 || h == 0 || w == 0) return CT_OK;
    if (!ws || ws_bytes < ct_pam_workspace_bytes(n, h, w)) return CT_E_WORKSPACE;
    ct::PamArgs a;
    a.q = q; a.k = k; a.v = nullptr; a.rgb = nullptr; a.out_v = nullptr; a.out_rgb = nullptr;
    a.colpart = reinterpret_cast<float *>(ws); a.att = att;
    a.C = c; a.CV = 0; a.H = h; a.W = w;
    int rc = ct::launch_pam<1>(a, n, (hipStream_t)stream);
    if (rc) return rc;
    const int tiles = (w + ct::pam_tq(w) - 1) / ct::pam_tq(w);
    hipLaunchKernelGGL(ct::pam_valid_kernel, dim3((w + 255) / 256, h, n), dim3(256), 0, (hipStream_t)stream,
                       (const float *)ws, tiles, h, w, valid, colsum);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // extern "C"
