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
struct ConvArgs {
    const float *in;
    const float *wp;
    const float *bias;      // [MT*32], zero padded
    const float *residual;  // nullable, same shape/strides as out
    float *out;
    int cin, cout, H, W;
    long long in_bstride, out_bstride, res_bstride;   // elements between images of a batch
    int act;     // 0 none, 1 LeakyReLU(0.01)
    int clamp;   // 1 = clamp to [0,1]
};

template <int KS, int MT>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs a, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int PAD = KS / 2;
    constexpr int ROWS = kConvTH + KS - 1;
    constexpr int TWP = kConvTW + KS - 1;
    constexpr int CS = ROWS * TWP;             // floats per channel in the LDS tile
    constexpr int ELEMS = kConvChunk * CS;     // floats of one staged (tile, channel chunk)
    constexpr int PF = (ELEMS + 255) / 256;    // prefetch registers per thread (85 for 3x3, 64 for 1x1)
    constexpr int COUTP = MT * 32;
    constexpr int RPW = kConvTH / 4;           // output rows per wave
    constexpr int TAPS = KS * KS;
    constexpr int WSLICE = (kConvChunk / 2) * 2 * COUTP;   // floats of one (tap, chunk) weight slice
    constexpr int WV4 = WSLICE / 4 / 256;      // float4 per thread per slice
    extern __shared__ float smem[];
    float *tin = smem;                         // [kConvChunk][ROWS][TWP]
    float *tw0 = smem + ELEMS;                 // 2 x [kConvChunk/2][2][COUTP]  (double buffer)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const size_t plane = (size_t)a.H * a.W;
    const int cin_pairs_total = (a.cin + 1) >> 1;
    const int n_chunks = (a.cin + kConvChunk - 1) / kConvChunk;
    const int my_tiles = (n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int n_stages = my_tiles * n_chunks;  // a stage = (tile, channel chunk)

    // ---- (tile, chunk) halo tile -> registers; the loads stay in flight while the previous stage computes ----
    // Offsets are 32-bit relative to the (image, chunk) base: a chunk spans < 2^31 elements.
    float pf[PF];
    const unsigned int uplane = (unsigned int)plane;
    auto fetch_tile = [&](int stage) {
        const int k = stage / n_chunks, chunk = stage - k * n_chunks;
        const int t = blockIdx.x + k * gridDim.x;
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int x0 = tx * kConvTW, y0 = ty * kConvTH, c0 = chunk * kConvChunk;
        const int cc = (a.cin - c0) < kConvChunk ? (a.cin - c0) : kConvChunk;
        const float *in = a.in + (size_t)n * a.in_bstride + (size_t)c0 * plane;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            {
                int e = tid + i * 256;
                // opaque to the optimiser: otherwise LICM hoists all PF index decompositions (c, yy, xx are
                // stage-invariant) out of the stage loop and keeps 3*PF values alive -> hundreds of spills
                asm volatile("" : "+v"(e));
                const int c = e / CS, rem = e - c * CS;
                const int yy = rem / TWP, xx = rem - yy * TWP;
                const int gy = y0 + yy - PAD, gx = x0 + xx - PAD;
                const bool ok = (e < ELEMS) && (c < cc) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                // exec-masked load into a pre-zeroed register: no predicate has to outlive the load
                float v = 0.f;
                if (ok) v = in[(unsigned int)c * uplane + (unsigned int)(gy * a.W + gx)];
                pf[i] = v;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int e = tid + i * 256;
            if (e < ELEMS) tin[e] = pf[i];
        }
    };
    // ---- weight slice (tap, chunk) -> registers (zero beyond the chunk's channel pairs) ----
    float4 wreg[WV4];
    auto fetch_w = [&](int chunk, int tap) {
        const int c0 = chunk * kConvChunk;
        const int cc = (a.cin - c0) < kConvChunk ? (a.cin - c0) : kConvChunk;
        const int nv = ((cc + 1) >> 1) * 2 * COUTP / 4;
        const float4 *src = reinterpret_cast<const float4 *>(a.wp + ((size_t)tap * cin_pairs_total + (c0 >> 1)) * 2 * COUTP);
#pragma unroll
        for (int i = 0; i < WV4; ++i) {
            const int e = tid + i * 256;
            wreg[i] = (e < nv) ? src[e] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_w = [&](int buf) {
        float4 *dst = reinterpret_cast<float4 *>(tw0 + buf * WSLICE);
#pragma unroll
        for (int i = 0; i < WV4; ++i) dst[tid + i * 256] = wreg[i];
    };

    if (n_stages == 0) return;
    f32x16 acc[RPW][MT];
    fetch_tile(0);
    fetch_w(0, 0);
    store_tile();
    int wbuf = 0;
    for (int stage = 0; stage < n_stages; ++stage) {
        const int k = stage / n_chunks, chunk = stage - k * n_chunks;
        const int cc = (a.cin - chunk * kConvChunk) < kConvChunk ? (a.cin - chunk * kConvChunk) : kConvChunk;
        const int ccp = (cc + 1) >> 1;
        if (chunk == 0) {
#pragma unroll
            for (int q = 0; q < RPW; ++q)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;
        }
        store_w(wbuf);
        __syncthreads();                                   // this stage's tile and tap-0 weights are visible
        const bool next_stage = (stage + 1 < n_stages);
        if (next_stage) fetch_tile(stage + 1);            // the next halo tile: in flight under this stage's MFMAs
        for (int tap = 0; tap < TAPS; ++tap) {
            const bool last_tap = (tap + 1 == TAPS);
            if (!last_tap) fetch_w(chunk, tap + 1);
            else if (next_stage) fetch_w((chunk + 1 == n_chunks) ? 0 : chunk + 1, 0);
            const int ky = tap / KS, kx = tap - ky * KS;
            const float *brow = tin + hl * CS + (wave * RPW + ky) * TWP + kx + nl;
            const float *arow = tw0 + wbuf * WSLICE + hl * COUTP + nl;
            auto kstep = [&](int p) {
                float w[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) w[m] = arow[p * 2 * COUTP + m * 32];
#pragma unroll
                for (int q = 0; q < RPW; ++q) {
                    const float b = brow[p * 2 * CS + q * TWP];
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[q][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[m], b, acc[q][m], 0, 0, 0);
                }
            };
            if (ccp == kConvChunk / 2) {   // the common full chunk: compile-time trip count, fully unrolled
#pragma unroll
                for (int p = 0; p < kConvChunk / 2; ++p) kstep(p);
            } else {
                for (int p = 0; p < ccp; ++p) kstep(p);
            }
            wbuf ^= 1;
            if (!last_tap) {
                store_w(wbuf);             // the other buffer: nobody reads it during this tap
                __syncthreads();
            }
        }
        if (chunk + 1 == n_chunks) {
            // ---- epilogue: lane owns pixels (y0 + wave*RPW + q, x0+nl), channels (r&3)+8(r>>2)+4hl of each 32-tile ----
            const int t = blockIdx.x + k * gridDim.x;
            const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
            const int x = tx * kConvTW + nl;
#pragma unroll
            for (int q = 0; q < RPW; ++q) {
                const int y = ty * kConvTH + wave * RPW + q;
                if (y < a.H && x < a.W) {
                    // 32-bit element offsets inside one image (cout * plane < 2^32): keeps the 64 stores cheap in registers
                    float *out = a.out + (size_t)n * a.out_bstride;
                    const float *res = a.residual ? a.residual + (size_t)n * a.res_bstride : nullptr;
                    const unsigned int pix = (unsigned int)(y * a.W + x);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hl;
                            if (co < a.cout) {
                                const unsigned int off = (unsigned int)co * uplane + pix;
                                float v = acc[q][m][r] + a.bias[co];
                                if (a.act == 1) v = v > 0.f ? v : 0.01f * v;
                                if (res) v += res[off];
                                if (a.clamp) v = fminf(fmaxf(v, 0.f), 1.f);
                                out[off] = v;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();                   // every wave is done reading this stage's tile
        if (next_stage) store_tile();      // waits for the prefetched loads, then fills the tile
    }
}

template <int KS, int MT>
static int launch_conv(const ConvArgs &a, int N, hipStream_t s) {
    constexpr int ROWS = kConvTH + KS - 1, TWP = kConvTW + KS - 1;
    const size_t lds = (size_t)(kConvChunk * ROWS * TWP + 2 * (kConvChunk / 2) * 2 * MT * 32) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<KS, MT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int tiles_x = (a.W + kConvTW - 1) / kConvTW, tiles_y = (a.H + kConvTH - 1) / kConvTH;
    const long long n_tiles = (long long)tiles_x * tiles_y * N;
    if (n_tiles > 0x7fffffffLL) return CT_E_BADARG;
    const int grid = n_tiles < 2 * kNumCUs ? (int)n_tiles : 2 * kNumCUs;   // persistent: two workgroups per CU
    hipLaunchKernelGGL((conv_mfma_kernel<KS, MT>), dim3(grid), dim3(256), lds, s, a, tiles_x, tiles_y, (int)n_tiles);
    CT_CHECK_LAUNCH();
    return CT_OK;
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

constexpr int kPamTQ = 32;   // queries per workgroup
constexpr int kPamJC = 64;   // keys staged per PV step

template <int MODE>
__global__ __launch_bounds__(256) void pam_attend_kernel(PamArgs a) {
    extern __shared__ float smem[];
    const int W = a.W, SW = W | 1;                 // odd row stride: column reads are conflict free
    float *S = smem;                               // [32][SW]
    float *Vs = smem + kPamTQ * SW;                // [96][kPamJC + 1]
    const int i0 = blockIdx.x * kPamTQ, h = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nl = lane & 31, hl = lane >> 5;
    const size_t plane = (size_t)a.H * W;
    const float *q = a.q + (size_t)n * a.C * plane + (size_t)h * W;
    const float *k = a.k + (size_t)n * a.C * plane + (size_t)h * W;
    const float inv_c = 1.0f / (float)a.C;

    // ---- S tile = Q^T K / C : M = query i (A = Q[c][i]), N = key j (B = K[c][j]), K = channels ----
    {
        const int qi = i0 + nl;
        const int ntile = (W + 31) / 32;
        for (int jt = wave; jt < ntile; jt += 4) {
            const int kj = jt * 32 + nl;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int c = 0; c < a.C; c += 2) {
                const int ch = c + hl;
                const float av = (qi < W && ch < a.C) ? q[(size_t)ch * plane + qi] : 0.f;
                const float bv = (kj < W && ch < a.C) ? k[(size_t)ch * plane + kj] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
            // D[i][j]: lane holds key column j = jt*32+nl, query rows (r&3)+8(r>>2)+4hl
            if (kj < W) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[((r & 3) + 8 * (r >> 2) + 4 * hl) * SW + kj] = acc[r] * inv_c;
            }
        }
    }
    __syncthreads();
    // ---- row softmax (F.softmax(dim=-1)): 8 rows per wave ----
    for (int rr = 0; rr < kPamTQ / 4; ++rr) {
        const int row = wave * (kPamTQ / 4) + rr;
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
            for (int r = 0; r < kPamTQ; ++r) s += S[r * SW + j];
            dst[j] = s;
        }
        return;
    }
    // ---- out[c][i] = sum_j V[c][j] P[i][j] : M = channel (A = V, staged [c][JC+1]), N = query (B = P) ----
    const int CT = a.CV + 3;                        // feature channels + the 3 RGB channels of `right`
    const int mt_total = (CT + 31) / 32;            // 3 for CV = 64
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float *v = a.v + (size_t)n * a.CV * plane + (size_t)h * W;
    const float *rgb = a.rgb + (size_t)n * 3 * plane + (size_t)h * W;
    constexpr int VST = kPamJC + 1;
    for (int j0 = 0; j0 < W; j0 += kPamJC) {
        __syncthreads();
        for (int idx = tid; idx < mt_total * 32 * kPamJC; idx += 256) {
            const int c = idx / kPamJC, jj = idx - c * kPamJC, j = j0 + jj;
            float val = 0.f;
            if (j < W) {
                if (c < a.CV) val = v[(size_t)c * plane + j];
                else if (c < CT) val = rgb[(size_t)(c - a.CV) * plane + j];
            }
            Vs[c * VST + jj] = val;
        }
        __syncthreads();
        if (wave < mt_total) {
            const float *arow = Vs + (wave * 32 + nl) * VST + hl;
            const float *brow = S + nl * SW + j0 + hl;
#pragma unroll
            for (int jj = 0; jj < kPamJC; jj += 2) {
                const float bv = (j0 + jj + hl < W) ? brow[jj] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[jj], bv, acc, 0, 0, 0);
            }
        }
    }
    if (wave < mt_total) {
        const int x = i0 + nl;
        if (x < W) {
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

template <int MODE>
static int launch_pam(const PamArgs &a, int N, hipStream_t s) {
    const int SW = a.W | 1;
    const size_t lds = ((size_t)kPamTQ * SW + 96 * (kPamJC + 1)) * sizeof(float);
    if (lds > 160 * 1024) return CT_E_BADARG;
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&pam_attend_kernel<MODE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr = lds;
    }
    dim3 grid((a.W + kPamTQ - 1) / kPamTQ, a.H, N);
    hipLaunchKernelGGL((pam_attend_kernel<MODE>), grid, dim3(256), lds, s, a);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // namespace ct

extern "C" {

int ct_conv2d_f32(const float *in, const float *wp, const float *bias, const float *residual, float *out, int n,
                  int cin, int cout, int h, int w, int ksize, long long in_bstride, long long out_bstride,
                  long long res_bstride, int act, int clamp, void *stream) {
    if (!in || !wp || !bias || !out || n < 0 || cin < 1 || cout < 1 || cout > 64 || h < 0 || w < 0) return CT_E_BADARG;
    if (ksize != 1 && ksize != 3) return CT_E_BADARG;
    if (n == 0 || h == 0 || w == 0) return CT_OK;
    ct::ConvArgs a;
    a.in = in; a.wp = wp; a.bias = bias; a.residual = residual; a.out = out;
    a.cin = cin; a.cout = cout; a.H = h; a.W = w;
    a.in_bstride = in_bstride; a.out_bstride = out_bstride; a.res_bstride = res_bstride;
    a.act = act; a.clamp = clamp;
    hipStream_t s = (hipStream_t)stream;
    const int mt = cout > 32 ? 2 : 1;
    if (ksize == 3) return mt == 2 ? ct::launch_conv<3, 2>(a, n, s) : ct::launch_conv<3, 1>(a, n, s);
    return mt == 2 ? ct::launch_conv<1, 2>(a, n, s) : ct::launch_conv<1, 1>(a, n, s);
}

size_t ct_pam_workspace_bytes(int n, int h, int w) {
    if (n < 0 || h < 0 || w < 0) return 0;
    return (size_t)n * h * ((w + ct::kPamTQ - 1) / ct::kPamTQ) * w * sizeof(float);
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
    const int tiles = (w + ct::kPamTQ - 1) / ct::kPamTQ;
    hipLaunchKernelGGL(ct::pam_valid_kernel, dim3((w + 255) / 256, h, n), dim3(256), 0, (hipStream_t)stream,
                       (const float *)ws, tiles, h, w, valid, colsum);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // extern "C"
