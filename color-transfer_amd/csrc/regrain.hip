// regrain.hip -- the "regrain" post-process of automated_color_grading (reference methods/iterative.py:62-138) on gfx950.
//
//   _regrain  (iterative.py:62-78)   multigrid recursion: both images are halved with skimage.transform.resize down to
//                                    ~20 pixels, the coarse solution is resized back up and relaxed on every level
//   _solve    (iterative.py:81-117)  nbit Jacobi sweeps of a 5-point gradient-preserving relaxation (psi / phi weights
//                                    from the gradients of the ORIGINAL image)
//   skimage.transform.resize (third party, scikit-image 0.18.3 _warps.py:resize/warp + scipy.ndimage.gaussian_filter):
//       down: Gaussian anti-aliasing, sigma = (factor - 1) / 2 per axis, radius int(4 sigma + 0.5), 'mirror' boundary,
//             axis 0 then axis 1; then bilinear sampling at factor * (i + 0.5) - 0.5 with 'reflect' boundary
//       up:   the bilinear sampling alone
//
// All arithmetic is float64 (the reference's `img_arr_col` is the float64 IDT output; a float32 target is promoted here
// where the reference resizes it in float32 -- a 1e-7-level difference, see tests).  HWC layout, one thread per pixel;
// the relaxation runs k = 8 sweeps per launch on LDS-resident tiles with a recomputed halo (rg_sweepk_kernel: 31 sweep launches
// for a 1080p frame instead of 244, bitwise the same result); the resizes are plain streaming kernels.
#include "ct_common.h"

namespace ct {

constexpr int kRgMaxLevels = 8;

__device__ __forceinline__ int mirror_idx(int i, int n) {          // scipy.ndimage 'mirror': d c b | a b c d | c b a
    if (n == 1) return 0;
    i = i < 0 ? -i : i;
    const int p = 2 * (n - 1);
    i %= p;
    return i >= n ? p - i : i;
}

// skimage _warp_fast coord_map(dim, coord, 'R')
__device__ __forceinline__ int reflect_idx(int coord, int dim) {
    const int cmax = dim - 1;
    if (dim == 1) return 0;
    if (coord < 0) {
        const int n = -coord;
        return ((n / cmax) % 2 != 0) ? cmax - (n % cmax) : n % cmax;
    }
    if (coord > cmax) return ((coord / cmax) % 2 != 0) ? cmax - (coord % cmax) : coord % cmax;
    return coord;
}

// one axis of scipy.ndimage.gaussian_filter on an [H][W][3] image: out = sum_k w[k] in[mirror(i + k - r)]
__global__ __launch_bounds__(kBlock) void rg_gauss_kernel(const double *__restrict__ in, double *__restrict__ out, int H, int W, int axis,
                                                         int radius, const double w0, const double w1, const double w2) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int r = (int)(i / W), c = (int)(i % W);
    const double wk[3] = {w0, w1, w2};                      // symmetric kernel: w[|k|], radius <= 2
    double acc[3] = {0.0, 0.0, 0.0};
    for (int k = -radius; k <= radius; ++k) {
        const int rr = axis == 0 ? mirror_idx(r + k, H) : r, cc = axis == 1 ? mirror_idx(c + k, W) : c;
        const double *p = in + ((size_t)rr * W + cc) * 3;
        const double g = wk[k < 0 ? -k : k];
        acc[0] += g * p[0]; acc[1] += g * p[1]; acc[2] += g * p[2];
    }
    double *o = out + (size_t)i * 3;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2];
}

// skimage warp(order=1, mode='reflect') of an [Hi][Wi][3] image to [Ho][Wo][3]: source = factor * (i + 0.5) - 0.5
__global__ __launch_bounds__(kBlock) void rg_bilinear_kernel(const double *__restrict__ in, double *__restrict__ out, int Hi, int Wi, int Ho,
                                                            int Wo, double fr, double fc) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)Ho * Wo) return;
    const int r = (int)(i / Wo), c = (int)(i % Wo);
    const double sr = fr * ((double)r + 0.5) - 0.5, sc = fc * ((double)c + 0.5) - 0.5;
    const int minr = (int)floor(sr), minc = (int)floor(sc), maxr = (int)ceil(sr), maxc = (int)ceil(sc);
    const double dr = sr - (double)minr, dc = sc - (double)minc;
    const int r0 = reflect_idx(minr, Hi), r1 = reflect_idx(maxr, Hi), c0 = reflect_idx(minc, Wi), c1 = reflect_idx(maxc, Wi);
    const double *p00 = in + ((size_t)r0 * Wi + c0) * 3, *p01 = in + ((size_t)r0 * Wi + c1) * 3;
    const double *p10 = in + ((size_t)r1 * Wi + c0) * 3, *p11 = in + ((size_t)r1 * Wi + c1) * 3;
    double *o = out + (size_t)i * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const double top = (1.0 - dc) * p00[ch] + dc * p01[ch];
        const double bottom = (1.0 - dc) * p10[ch] + dc * p11[ch];
        o[ch] = (1.0 - dr) * top + dr * bottom;
    }
}


// iterative.py:91-98: gradient magnitude of the original image -> psi, phi  (wt[i] = {psi, phi})
__global__ __launch_bounds__(kBlock) void rg_weights_kernel(const double *__restrict__ in, double2 *__restrict__ wt, int H, int W, double phi_scale) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int r = (int)(i / W), c = (int)(i % W);
    const double *pr = in + ((size_t)r * W + min(c + 1, W - 1)) * 3, *pl = in + ((size_t)r * W + max(c - 1, 0)) * 3;
    const double *pd = in + ((size_t)min(r + 1, H - 1) * W + c) * 3, *pu = in + ((size_t)max(r - 1, 0) * W + c) * 3;
    double s = 0.0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const double dx = pr[ch] - pl[ch], dy = pd[ch] - pu[ch];
        s += dx * dx + dy * dy;
    }
    const double delta = sqrt(s);
    double psi = 256.0 * delta / 5.0;
    psi = psi > 1.0 ? 1.0 : psi;
    wt[i] = make_double2(psi, phi_scale / (1.0 + 10.0 * delta));
}

// one sweep of iterative.py:106-115 (Jacobi: everything on the right-hand side is the previous iterate)
__global__ __launch_bounds__(kBlock) void rg_sweep_kernel(const double *__restrict__ prev, const double *__restrict__ in, const double *__restrict__ col,
                                                         const double2 *__restrict__ wt, double *__restrict__ next, int H, int W) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int r = (int)(i / W), c = (int)(i % W);
    const size_t iR = (size_t)r * W + min(c + 1, W - 1), iL = (size_t)r * W + max(c - 1, 0);
    const size_t iD = (size_t)min(r + 1, H - 1) * W + c, iU = (size_t)max(r - 1, 0) * W + c;
    const double2 w0 = wt[i];
    const double psi = w0.x, phi = w0.y;
    const double phi1 = (wt[iR].y + phi) / 2, phi2 = (wt[iD].y + phi) / 2, phi3 = (wt[iL].y + phi) / 2, phi4 = (wt[iU].y + phi) / 2;
    const double den = psi + phi1 + phi2 + phi3 + phi4;
    const double eps = 1e-6, rho = 1 / 5.0;
    // (num / (den + eps)) * (1 - rho) of the reference as num * scale: the quotient does not depend on the channel or the sweep
    // (one float64 division per pixel instead of three per sweep; the last bit may differ from the reference's order, the
    // relaxation is contractive: 1e-15 on the result, tests/test_regrain.py asserts 1e-9)
    const double scale = (1 - rho) / (den + eps);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const double x = in[i * 3 + ch];
        const double num = psi * col[i * 3 + ch] + phi1 * (prev[iR * 3 + ch] - in[iR * 3 + ch] + x) + phi2 * (prev[iD * 3 + ch] - in[iD * 3 + ch] + x) +
                           phi3 * (prev[iL * 3 + ch] - in[iL * 3 + ch] + x) + phi4 * (prev[iU * 3 + ch] - in[iU * 3 + ch] + x);
        next[i * 3 + ch] = num * scale + rho * prev[i * 3 + ch];
    }
}

// K sweeps of rg_sweep_kernel in ONE launch (temporal blocking with a recomputed halo): a workgroup owns a th x tw tile of
// the level, loads it with a halo of k pixels and runs k Jacobi sweeps on the WHOLE region, synchronising with
// barriers instead of launch boundaries.  Region pixels whose neighbours lie outside the region use the nearest region pixel
// instead: that garbage travels inward one pixel per sweep and after k sweeps has eaten exactly the halo; the tile itself is
// what rg_sweep_kernel would have produced, operation for operation (the per-pixel arithmetic below is the same code).
// Neighbours outside the IMAGE are the pixel itself in both kernels (min / max clamping, iterative.py:100-104).
// A 1080p frame: 244 sweep launches become 31 (nbits 4, 16, 32, 64, 64, 64 with k = 8).
constexpr int kRgKMax = 8;
constexpr int kRgMaxRegion = 1536;                      // (16 + 2 k) x (32 + 2 k) at k = 8
constexpr int kRgPxPerThread = kRgMaxRegion / kBlock;   // 6

// LDS holds ONE quantity per pixel and channel, d = iterate - in (what a neighbour needs: the formula reads prev[k] - in[k]),
// twice for the ping-pong; everything else a pixel needs in every sweep lives in the registers of the thread that owns it:
// in, the current iterate, psi * col, the four neighbour weights and the hoisted scale.
__global__ __launch_bounds__(kBlock) void rg_sweepk_kernel(const double *__restrict__ prev, const double *__restrict__ in, const double *__restrict__ col,
                                                          const double2 *__restrict__ wt, double *__restrict__ next, int H, int W, int th, int tw,
                                                          int k, int tiles_x) {
    extern __shared__ double lds[];
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * th, x0 = tx * tw;
    // region = tile + halo, clipped to the image
    const int ry0 = max(y0 - k, 0), ry1 = min(y0 + th + k, H), rx0 = max(x0 - k, 0), rx1 = min(x0 + tw + k, W);
    const int rh = ry1 - ry0, rw = rx1 - rx0, rn = rh * rw;
    double *dA = lds, *dB = lds + 3 * kRgMaxRegion, *phiL = lds + 6 * kRgMaxRegion;
    double inr[kRgPxPerThread][3], cur[kRgPxPerThread][3], pc[kRgPxPerThread][3], ph[kRgPxPerThread][4], sc[kRgPxPerThread];
    int nb[kRgPxPerThread][4];
#pragma unroll
    for (int j = 0; j < kRgPxPerThread; ++j) {
        const int p = threadIdx.x + j * kBlock;
        if (p < rn) {
            const int r = p / rw, c = p - r * rw;
            const size_t g = (size_t)(ry0 + r) * W + (rx0 + c);
            const double2 w0 = wt[g];
            ph[j][0] = w0.x;                                   // psi for now
            phiL[p] = w0.y;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                inr[j][ch] = in[g * 3 + ch]; cur[j][ch] = prev[g * 3 + ch]; pc[j][ch] = col[g * 3 + ch];
                dA[p * 3 + ch] = cur[j][ch] - inr[j][ch];
            }
            // image clamping == region clamping wherever the region edge is an image edge; elsewhere: halo garbage
            nb[j][0] = r * rw + min(c + 1, rw - 1); nb[j][1] = min(r + 1, rh - 1) * rw + c;
            nb[j][2] = r * rw + max(c - 1, 0);      nb[j][3] = max(r - 1, 0) * rw + c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kRgPxPerThread; ++j) {
        const int p = threadIdx.x + j * kBlock;
        if (p < rn) {
            const double psi = ph[j][0], phi = phiL[p];
            const double phi1 = (phiL[nb[j][0]] + phi) / 2, phi2 = (phiL[nb[j][1]] + phi) / 2, phi3 = (phiL[nb[j][2]] + phi) / 2,
                         phi4 = (phiL[nb[j][3]] + phi) / 2;
            const double den = psi + phi1 + phi2 + phi3 + phi4;
            const double eps = 1e-6, rho = 1 / 5.0;
            sc[j] = (1 - rho) / (den + eps);
            ph[j][0] = phi1; ph[j][1] = phi2; ph[j][2] = phi3; ph[j][3] = phi4;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) pc[j][ch] = psi * pc[j][ch];
        }
    }
    double *dc = dA, *dn = dB;
    for (int sweep = 0; sweep < k; ++sweep) {
#pragma unroll
        for (int j = 0; j < kRgPxPerThread; ++j) {
            const int p = threadIdx.x + j * kBlock;
            if (p < rn) {
                const double rho = 1 / 5.0;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const double x = inr[j][ch];
                    const double num = pc[j][ch] + ph[j][0] * (dc[nb[j][0] * 3 + ch] + x) + ph[j][1] * (dc[nb[j][1] * 3 + ch] + x) +
                                       ph[j][2] * (dc[nb[j][2] * 3 + ch] + x) + ph[j][3] * (dc[nb[j][3] * 3 + ch] + x);
                    cur[j][ch] = num * sc[j] + rho * cur[j][ch];
                    dn[p * 3 + ch] = cur[j][ch] - x;
                }
            }
        }
        __syncthreads();
        double *t = dc; dc = dn; dn = t;
    }
    // the tile (the part of the region that lies k pixels away from every non-image edge of it)
#pragma unroll
    for (int j = 0; j < kRgPxPerThread; ++j) {
        const int p = threadIdx.x + j * kBlock;
        if (p < rn) {
            const int r = p / rw, c = p - r * rw, gy = ry0 + r, gx = rx0 + c;
            if (gy >= y0 && gy < min(y0 + th, H) && gx >= x0 && gx < min(x0 + tw, W)) {
                const size_t g = (size_t)gy * W + gx;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) next[g * 3 + ch] = cur[j][ch];
            }
        }
    }
}

struct RgLevel { int h, w; };

static int rg_levels(int H, int W, int n_nbits, RgLevel *lv) {      // iterative.py:62-68: recurse while len(nbits) > 1 and h2 > 20 and w2 > 20
    int n = 0;
    lv[n++] = {H, W};
    while (n < n_nbits) {
        const int h2 = (lv[n - 1].h + 1) / 2, w2 = (lv[n - 1].w + 1) / 2;
        if (!(h2 > 20 && w2 > 20)) break;
        lv[n++] = {h2, w2};
    }
    return n;
}

static inline dim3 rg_grid(int64_t n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

// skimage.transform.resize(src [hi][wi][3] -> dst [ho][wo][3]); tmp: two [hi][wi][3] buffers (down-scaling only)
static int rg_resize(const double *src, int hi, int wi, double *dst, int ho, int wo, double *tmp0, double *tmp1, hipStream_t s) {
    const double fr = (double)hi / (double)ho, fc = (double)wi / (double)wo;
    int radius[2] = {0, 0};
    double wts[2][3] = {{1.0, 0.0, 0.0}, {1.0, 0.0, 0.0}};
    bool filtered[2] = {false, false};
    for (int axis = 0; axis < 2; ++axis) {
        const double f = axis == 0 ? fr : fc;
        const double sigma = f > 1.0 ? (f - 1.0) / 2.0 : 0.0;
        if (!(sigma > 1e-15)) continue;                               // scipy skips such axes
        radius[axis] = (int)(4.0 * sigma + 0.5);
        if (radius[axis] > 2) return CT_E_BADARG;                     // halving pyramids never get here
        double sum = 0.0;
        for (int k = -radius[axis]; k <= radius[axis]; ++k) sum += exp(-0.5 / (sigma * sigma) * (double)(k * k));
        for (int k = 0; k <= radius[axis]; ++k) wts[axis][k] = exp(-0.5 / (sigma * sigma) * (double)(k * k)) / sum;
        filtered[axis] = true;
    }
    // (a fused gaussian + gaussian + bilinear kernel was tried: bitwise the same, 0.7 ms SLOWER per 1080p frame -- its 100 strided
    // float64 gathers per output pixel coalesce badly; the three streaming launches stay)
    const double *cur = src;
    double *bufs[2] = {tmp0, tmp1};
    int nb = 0;
    for (int axis = 0; axis < 2; ++axis) {
        if (!filtered[axis]) continue;
        hipLaunchKernelGGL(rg_gauss_kernel, rg_grid((int64_t)hi * wi), dim3(kBlock), 0, s, cur, bufs[nb], hi, wi, axis, radius[axis], wts[axis][0],
                           wts[axis][1], wts[axis][2]);
        CT_CHECK_LAUNCH();
        cur = bufs[nb];
        nb ^= 1;
    }
    hipLaunchKernelGGL(rg_bilinear_kernel, rg_grid((int64_t)ho * wo), dim3(kBlock), 0, s, cur, dst, hi, wi, ho, wo, fr, fc);
    CT_CHECK_LAUNCH();
    return CT_OK;
}

}  // namespace ct

extern "C" {

size_t ct_regrain_workspace_bytes(int height, int width) {
    if (height < 1 || width < 1) return 0;
    ct::RgLevel lv[ct::kRgMaxLevels];
    const int n = ct::rg_levels(height, width, 6, lv);
    size_t doubles = (size_t)height * width * 3 * 2;                   // two resize scratch images at the finest size
    for (int l = 0; l < n; ++l) doubles += (size_t)lv[l].h * lv[l].w * (3 * 4 + 2);   // in, col, two iterates, {psi, phi}
    return doubles * sizeof(double) + 256;
}

// img_in, img_col, out: device [height][width][3] float64 (out may not alias the inputs); nbits: host array (the reference's
// default is {4, 16, 32, 64, 64, 64}); one frame per call.
int ct_regrain_f64(const double *img_in, const double *img_col, double *out, int height, int width, const int *nbits, int n_nbits,
                   void *ws, size_t ws_bytes, void *stream) {
    using namespace ct;
    if (!img_in || !img_col || !out || !nbits || height < 1 || width < 1 || n_nbits < 1 || n_nbits > kRgMaxLevels) return CT_E_BADARG;
    if (!ws || (reinterpret_cast<uintptr_t>(ws) & 15) || ws_bytes < ct_regrain_workspace_bytes(height, width)) return CT_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    static DynLdsAttr attr;             // per device
    if (attr.ensure(reinterpret_cast<const void *>(rg_sweepk_kernel), 7 * kRgMaxRegion * sizeof(double)) != hipSuccess) return CT_E_BADARG;
    RgLevel lv[kRgMaxLevels];
    const int n = rg_levels(height, width, n_nbits, lv);
    double *p = reinterpret_cast<double *>(ws);
    double *tmp0 = p; p += (size_t)height * width * 3;
    double *tmp1 = p; p += (size_t)height * width * 3;
    const double *in_l[kRgMaxLevels], *col_l[kRgMaxLevels];
    double *it_a[kRgMaxLevels], *it_b[kRgMaxLevels];
    double2 *wt_l[kRgMaxLevels];
    for (int l = 0; l < n; ++l) {
        const size_t px = (size_t)lv[l].h * lv[l].w;
        double *a = p; p += px * 3;
        double *b = p; p += px * 3;
        it_a[l] = p; p += px * 3;
        it_b[l] = p; p += px * 3;
        wt_l[l] = reinterpret_cast<double2 *>(p); p += px * 2;
        if (l == 0) { in_l[0] = img_in; col_l[0] = img_col; }
        else {
            int rc = rg_resize(in_l[l - 1], lv[l - 1].h, lv[l - 1].w, a, lv[l].h, lv[l].w, tmp0, tmp1, s);
            if (rc) return rc;
            if ((rc = rg_resize(col_l[l - 1], lv[l - 1].h, lv[l - 1].w, b, lv[l].h, lv[l].w, tmp0, tmp1, s))) return rc;
            in_l[l] = a; col_l[l] = b;
        }
    }
    const double *coarse = nullptr;                                    // the solution of level l + 1
    for (int l = n - 1; l >= 0; --l) {
        const int h = lv[l].h, w = lv[l].w;
        const int64_t px = (int64_t)h * w;
        const double *start;
        if (coarse == nullptr) start = in_l[l];                        // iterative.py:74: img_arr_out = img_arr_in
        else {
            const int rc = rg_resize(coarse, lv[l + 1].h, lv[l + 1].w, it_a[l], h, w, tmp0, tmp1, s);
            if (rc) return rc;
            start = it_a[l];
        }
        hipLaunchKernelGGL(rg_weights_kernel, rg_grid(px), dim3(kBlock), 0, s, in_l[l], wt_l[l], h, w, 30.0 * exp2(-(double)l));
        CT_CHECK_LAUNCH();
        const double *prev = start;
        static const int kblock = [] { const char *e = getenv("CT_HIP_REGRAIN_K"); const int v = e ? atoi(e) : kRgKMax; return v < 1 ? 1 : (v > kRgKMax ? kRgKMax : v); }();
        // small levels: small tiles (a launch lasts as long as its slowest workgroup: k sweeps of region / 256 pixels per thread)
        const int th = px > 200000 ? 16 : 8, tw = px > 200000 ? 32 : 16;
        const int tiles_x = (w + tw - 1) / tw, tiles_y = (h + th - 1) / th;
        for (int i = 0; i < nbits[l];) {
            const int kk = nbits[l] - i < kblock ? nbits[l] - i : kblock;
            // ping-pong; the LAST sweep of level 0 lands in the caller's buffer
            double *next = (l == 0 && i + kk == nbits[l]) ? out : ((prev == it_b[l]) ? it_a[l] : it_b[l]);
            if (kk == 1) {
                hipLaunchKernelGGL(rg_sweep_kernel, rg_grid(px), dim3(kBlock), 0, s, prev, in_l[l], col_l[l], wt_l[l], next, h, w);
            } else {
                hipLaunchKernelGGL(rg_sweepk_kernel, dim3(tiles_x * tiles_y), dim3(kBlock), (size_t)7 * kRgMaxRegion * sizeof(double), s, prev, in_l[l],
                                   col_l[l], wt_l[l], next, h, w, th, tw, kk, tiles_x);
            }
            CT_CHECK_LAUNCH();
            prev = next;
            i += kk;
        }
        if (nbits[l] == 0 && l == 0) {                                 // no sweep on the finest level: the start image is the result
            if (hipMemcpyAsync(out, prev, (size_t)px * 3 * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return (int)hipGetLastError();
        }
        coarse = prev;
    }
    return CT_OK;
}

}  // extern "C"
