/*
 * ct_hip.h -- C ABI of libct_hip.so: MI355X (gfx950) kernels for the per-frame
 * colour-transfer hot path of egorchistov/color-transfer.
 *
 * The reference is pure Python (numpy / scipy / scikit-image / torch) and has no
 * native layer; each entry point below replaces the numpy expression(s) cited next
 * to it (paths relative to the reference root).  A binding only needs ctypes (see
 * INTEGRATION.md).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller, except where the name
 *     says host; nothing is allocated, freed or synchronised inside (one exception:
 *     none -- the workspace comes from the caller, sized by ct_workspace_bytes());
 *   - every entry takes the hipStream_t to launch on (as void*; NULL = default stream)
 *     and returns immediately after enqueueing (asynchronous, graph-capturable);
 *   - images are interleaved HWC, C = 3, contiguous: pixel p of image b starts at
 *     base + (b * n_pixels + p) * 3 elements;
 *   - return value: 0 = CT_OK, negative = CT_E_* argument error, positive = hipError_t;
 *   - arithmetic is float64 internally (see ct_set_lab_mode for the table-driven float32-image path); "_f32"/"_f64" name the I/O element type.
 */
#ifndef CT_HIP_H
#define CT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CT_OK 0
#define CT_E_BADARG (-1)     /* null pointer, negative size, unknown enum */
#define CT_E_WORKSPACE (-2)  /* workspace too small / misaligned */
#define CT_E_ALIGN (-3)      /* image base not aligned to its element size */

/* bumped whenever an entry point changes its argument list (2: ct_attention_tokens_f32 gained kv_shift; 3: round 3; 4: ct_conv2d_split_rows_f32, res_pre_act, ct_linear_ws16_f32, layernorm partials, f16 form of ct_conv2d_split*;
 * 5: ct_reinhard_persist_*, ct_reinhard_psnr_u8, CT_WS_REINHARD_PERSIST; 6: ct_conv2d_split_f32 gained scratch / scratch_bytes;
 * 7: ct_device_status);
 * the ctypes binding refuses a library whose ct_abi_version() differs */
#define CT_ABI_VERSION 9

/* doubles per image in a stats record written by ct_lab_stats / ct_rgb_meancov */
#define CT_LAB_STATS_STRIDE 8  /* mean[3], std[3] (population, ddof 0), n, 0           */
#define CT_RGB_STATS_STRIDE 16 /* mean[3], cov[9] row-major (ddof 1), n, 0, 0, 0       */

enum ct_workspace_kind {
    CT_WS_LAB_STATS = 0, /* ct_lab_stats_*: n_images = number of images in the call    */
    CT_WS_RGB_MEANCOV = 1,
    CT_WS_REINHARD = 2,  /* ct_reinhard_*:  n_images = batch (pairs)                    */
    CT_WS_IDT = 3,       /* ct_idt_*:       n_images = batch (pairs)                    */
    CT_WS_REINHARD_PSNR = 4, /* ct_reinhard_psnr_f32: n_images = batch (pairs)          */
    CT_WS_REINHARD_PERSIST = 5 /* ct_reinhard_persist_f32 / ct_reinhard_psnr_u8: n_images = batch (pairs); 0 = frame size not supported */
};

int ct_abi_version(void);

/* Sticky status of the CURRENT device, for callers that never synchronise per call (video loops, graphs): a bit mask, 0 = all well.
 *   bit 0: a bounded spin of a persistent Reinhard launch gave up (a workgroup of its grid never became resident: that call's
 *          frames and PSNR records are NaN);
 *   bit 1: a stream-K consumer of ct_conv2d_split_f32 gave up waiting for its producer (that launch's tile is wrong).
 * Neither can happen while the launch's workgroups are all resident; both are bounded so that a queue never hangs.  clear != 0
 * resets the bits read.  Blocks the calling thread for two 4-byte copies and does NOT wait for running work: synchronise the
 * streams of interest first.  Returns -1 when the device cannot be read.  New in ABI 7; the reference has no counterpart
 * (its CPU path cannot fail this way). */
int ct_device_status(int clear);

/* Lab arithmetic of the float32 entries (ct_lab_stats_f32, ct_reinhard_*_f32, ct_reinhard_lab_f32):
 *   CT_LAB_TABLE (default)  the power functions of skimage's rgb2lab / lab2rgb (methods/linear.py:25,26,40) are LDS
 *                           look-ups + short polynomials; Lab agrees with the float64 path to ~5e-7, statistics to ~1e-7;
 *                           values outside [0,1], NaNs and degenerate statistics fall back to the exact code per wave;
 *   CT_LAB_EXACT            float64 arithmetic with hardware seeds and one Newton correction (~1e-11 relative).
 * float64 images always take the exact path.  ct_set_lab_mode: the process-wide default (atomic; env CT_HIP_LAB=exact presets
 * it).  ct_set_lab_mode_thread: an override for the CALLING thread only (-1 = back to the default) -- host threads that drive
 * different streams with different arithmetic each set their own; every entry reads the mode once, on the calling thread.
 * ct_get_lab_mode: what the calling thread's next call will use.                                                          */
#define CT_LAB_TABLE 0
#define CT_LAB_EXACT 1
int ct_set_lab_mode(int mode);
int ct_set_lab_mode_thread(int mode);
int ct_get_lab_mode(void);
/* Measurement hook: four hipEvent_t (or NULL = off) that the library records on the launch stream immediately before /
 * after moments_kernel<T,true> and reinhard_apply_kernel of the following ct_lab_stats / ct_reinhard* calls, so that a
 * caller can time exactly those kernels with hipEventElapsedTime (bench.py `roofline`).  When a call takes the persistent
 * launch, the first pair is recorded back to back and the second pair brackets reinhard_persist_kernel.  Per calling thread. */
void ct_profile_events(void *moments_start, void *moments_stop, void *apply_start, void *apply_stop);
/* Human readable text for a return code of this library (never NULL). */
const char *ct_error_string(int code);
/* Bytes of device workspace an entry of `kind` needs for `n_images` images of `n_pixels`. */
size_t ct_workspace_bytes(int kind, int64_t n_pixels, int n_images);

/* ---- A1: rgb2lab + np.mean/np.std  (methods/linear.py:25-26,33-36; skimage rgb2lab) ----
 * For image i in [0, n_images): stats[i*8 ..] = {mean L,a,b ; std L,a,b (ddof 0) ; n ; 0}.
 * Deterministic: fixed-shape tree reduction, no float atomics.                          */
int ct_lab_stats_f32(const float *rgb, int64_t n_pixels, int n_images, double *stats,
                     void *ws, size_t ws_bytes, void *stream);
int ct_lab_stats_f64(const double *rgb, int64_t n_pixels, int n_images, double *stats,
                     void *ws, size_t ws_bytes, void *stream);

/* ---- A2: (lab - mu_t) * sigma_r / sigma_t + mu_r ; lab2rgb   (methods/linear.py:38-40) ----
 * stats_t / stats_r: records written by ct_lab_stats (device memory, no host sync).
 * out may alias target.  Output clipped to [0,1] like skimage's xyz2rgb.                */
int ct_reinhard_apply_f32(const float *target, const double *stats_t, const double *stats_r,
                          float *out, int64_t n_pixels, int batch, void *stream);
int ct_reinhard_apply_f64(const double *target, const double *stats_t, const double *stats_r,
                          double *out, int64_t n_pixels, int batch, void *stream);
/* Same affine map but the result is left in Lab (parity probe for the 1e-4 Lab gate).   */
int ct_reinhard_lab_f32(const float *target, const double *stats_t, const double *stats_r,
                        float *out_lab, int64_t n_pixels, int batch, void *stream);

/* ---- a1 fused: methods.linear.color_transfer_between_images (methods/linear.py:8-42) ----
 * batch pairs per call: stats of all 2*batch images in one launch, finalize, apply.
 * ws: ct_workspace_bytes(CT_WS_REINHARD, n_pixels, batch).
 * stats_out: NULL, or device [2*batch][8] doubles receiving the Lab stats records
 * (targets first, then references) -- the per-frame metrics a caller may gather.       */
int ct_reinhard_f32(const float *target, const float *reference, float *out,
                    int64_t n_pixels, int batch, double *stats_out, void *ws, size_t ws_bytes,
                    void *stream);
int ct_reinhard_f64(const double *target, const double *reference, double *out,
                    int64_t n_pixels, int batch, double *stats_out, void *ws, size_t ws_bytes,
                    void *stream);
/* The same + the per-frame PSNR of Runner.test_step (methods/__init__.py:30-32,37) against ground-truth frames gt
 * ([batch][n_pixels][3] like the images): psnr_out[i] = {mse, 10 log10(1 / mse)} of the (clipped) result of pair i.
 * With the table arithmetic the squared error is accumulated by the apply sweep while it writes the result -- the result is
 * not read back from HBM.  ws: ct_workspace_bytes(CT_WS_REINHARD_PSNR, n_pixels, batch).                               */
int ct_reinhard_psnr_f32(const float *target, const float *reference, const float *gt, float *out, double *psnr_out,
                         int64_t n_pixels, int batch, double *stats_out, void *ws, size_t ws_bytes, void *stream);

/* ---- a1 as ONE persistent launch (csrc/reinhard_persist.hip): methods/linear.py:8-42 for `batch` pairs.  One workgroup per CU
 * owns 1 / CUs of every frame; the cube-root-domain image of (up to 18 of) its target tiles waits in LDS for the statistics of
 * the whole frame, which travel between the workgroups as 64-bit integer atomics one pair ahead of their use (the rest of its
 * target tiles are fetched a second time); no launch boundary, no finishing kernels, one pass over the reference.
 * Frame sizes: 256 pixels .. 16 Mpixel (ct_reinhard_persist_supported).  Measured at 1080p, 16 pairs per call: float32 frames
 * 39 - 41 k pairs/s (the two sweeps of ct_reinhard_psnr_f32: 41.5 - 43.6 k -- this form is bound by vector instruction issue and
 * executes more instructions, so the fused float32 entries keep the two sweeps unless CT_HIP_REINHARD_PERSIST=1);
 * uint8 frames 46.6 - 49.4 k pairs/s, which no other entry serves.
 * out must NOT overlap target, reference or gt (CT_E_BADARG): a tile with a pixel near a kink of the inverse is redone with the
 * exact code from the input frame AFTER its fast-path result has been stored (the two-sweep entries above do allow out == target).
 * gt = NULL: no metric (psnr_out unused); else psnr_out[i] = {mse, PSNR} like ct_reinhard_psnr_f32.
 * ws: ct_workspace_bytes(CT_WS_REINHARD_PERSIST, n_pixels, batch).  After the call has completed, the first 32-bit word of
 * ws is 0, or 1 when a workgroup of the grid never became resident within 2 s (results are then NaN).
 * Bitwise reproducible run to run and independent of `batch`.                                                           */
int ct_reinhard_persist_supported(int64_t n_pixels);
/* 1 when ct_reinhard_f32 / ct_reinhard_psnr_f32 take the persistent launch for frames of this size in the current Lab mode */
int ct_reinhard_takes_persist(int64_t n_pixels);
int ct_reinhard_persist_f32(const float *target, const float *reference, const float *gt, float *out, double *psnr_out,
                            int64_t n_pixels, int batch, double *stats_out, void *ws, size_t ws_bytes, void *stream);
/* uint8 frames as the reference's datasets deliver them (utils/data.py:84,106,125: `read_image(...).float() / 255`): the
 * kernel reads the bytes (a quarter of the float32 traffic) and takes k / 255 (IEEE float32 division) and its gamma
 * expansion from 256-entry tables filled by the float32 kernel's own functions: per pixel the arithmetic is that of
 * ct_reinhard_persist_f32 on `u8.float() / 255`; the frame statistics agree to their last float32 rounding (the moment terms
 * are added in another lane order), results within 2e-7.  out: float32.                                               */
int ct_reinhard_psnr_u8(const uint8_t *target, const uint8_t *reference, const uint8_t *gt, float *out, double *psnr_out,
                        int64_t n_pixels, int batch, double *stats_out, void *ws, size_t ws_bytes, void *stream);

/* ---- A3: np.mean(axis=0) + np.cov(x.T)   (methods/linear.py:64-67,103-106) ----
 * stats[i*16 ..] = {mean[3] ; cov[9] row-major, ddof 1 ; n ; 0 0 0}.                     */
int ct_rgb_meancov_f32(const float *rgb, int64_t n_pixels, int n_images, double *stats,
                       void *ws, size_t ws_bytes, void *stream);
int ct_rgb_meancov_f64(const double *rgb, int64_t n_pixels, int n_images, double *stats,
                       void *ws, size_t ws_bytes, void *stream);

/* ---- A4 (sync-free): the 3x3 algebra of monge_kantorovitch_color_transfer on the device (methods/linear.py:108-118) ----
 * stats_t / stats_r: records of ct_rgb_meancov; decomposition 0 "MK", 1 "sqrt", 2 "cholesky"; writes the 16-double coef
 * records ct_affine3x3_* consumes (T is applied as x @ T).  float64 Jacobi eigen-decomposition: the SPD matrix square root
 * is unique, so it equals scipy.linalg.sqrtm to rounding.  (Xiao's SVD-based matrix depends on LAPACK's sign
 * convention and is computed on the host.)                                                                            */
int ct_mk_coef_f64(const double *stats_t, const double *stats_r, int decomposition, int batch, double *coef,
                   void *stream);

/* ---- a3 fused: methods.linear.monge_kantorovitch_color_transfer (methods/linear.py:85-124), batch pairs per call,
 * everything on the device (moments of all 2*batch images in one sweep, finishing kernel, ct_mk_coef, affine apply).
 * ws: ct_workspace_bytes(CT_WS_REINHARD, n_pixels, batch).  Output unclipped like the reference.                      */
int ct_mk_f32_f32(const float *target, const float *reference, float *out, int64_t n_pixels, int batch,
                  int decomposition, void *ws, size_t ws_bytes, void *stream);
int ct_mk_f32_f64(const float *target, const float *reference, double *out, int64_t n_pixels, int batch,
                  int decomposition, void *ws, size_t ws_bytes, void *stream);
int ct_mk_f64_f64(const double *target, const double *reference, double *out, int64_t n_pixels, int batch,
                  int decomposition, void *ws, size_t ws_bytes, void *stream);

/* ---- A5: (x - mu_t) @ A + mu_r   (methods/linear.py:80,122) ----
 * coef: device, 16 doubles per image: A[9] row-major such that out_j = sum_i (x_i-mu_t_i)*A[i][j]
 * (pass T for MK, T.T for Xiao), mu_t[3], mu_r[3], pad.  No clipping (the reference does
 * not clip either; the caller clamps, methods/__init__.py:30).                           */
int ct_affine3x3_f32_f64(const float *in, const double *coef, double *out, int64_t n_pixels,
                         int batch, void *stream);
int ct_affine3x3_f64_f64(const double *in, const double *coef, double *out, int64_t n_pixels,
                         int batch, void *stream);
int ct_affine3x3_f32_f32(const float *in, const double *coef, float *out, int64_t n_pixels,
                         int batch, void *stream);

/* ---- regrain: the second half of methods.iterative.automated_color_grading (methods/iterative.py:62-138) ----------------
 * img_in (the original target), img_col (the IDT result), out: device [height][width][3] float64, one frame per call; out
 * must not alias the inputs.  Multigrid as in the reference: both images are halved with skimage.transform.resize
 * semantics (Gaussian anti-aliasing sigma (factor-1)/2 with 'mirror' boundary, then bilinear sampling with 'reflect'
 * boundary; scikit-image 0.18.3) while (h+1)/2 > 20 and (w+1)/2 > 20 and nbits has entries left; on every level nbits[level]
 * Jacobi sweeps of iterative.py:106-115.  nbits: HOST array, the reference's default is {4, 16, 32, 64, 64, 64}.
 * ws: ct_regrain_workspace_bytes(height, width) (16-byte aligned).                                                     */
size_t ct_regrain_workspace_bytes(int height, int width);
int ct_regrain_f64(const double *img_in, const double *img_col, double *out, int height, int width, const int *nbits,
                   int n_nbits, void *ws, size_t ws_bytes, void *stream);

/* ---- test-set distortions (utils/data.py:12-22,120-125): torchvision.transforms.functional.adjust_* on a uint8 frame ----
 * in: uint8 [3][height][width] (what read_image returns).  kind: 0 identity, 1 brightness, 2 contrast, 3 saturation
 * (param = factor >= 0), 4 hue (param = hue_factor in [-0.5, 0.5]), 5 gamma (param = gamma >= 0, gain 1).
 * out_u8 (optional): the distorted uint8 frame; out_f32 (optional): that frame / 255 as float32 [3][H][W] -- the
 * `target / 255` the dataset hands over.  torchvision's tensor-backend arithmetic restated (float32, truncating casts; the
 * Python-float factor and 1 - factor are each rounded to float32, hence the double parameter).
 * ws: >= 8 bytes, 8-byte aligned.                                                                                      */
int ct_distort_u8(const uint8_t *in, int height, int width, int kind, double param, uint8_t *out_u8, float *out_f32,
                  void *ws, size_t ws_bytes, void *stream);

/* ---- per-frame metric (first step): PSNR as Runner.test_step logs it (methods/__init__.py:32,37) ----
 * a, b: [batch][n_elems] float32 (any layout, same for both); out[i] = {mse, 10 log10(1/mse)} (data range 1).
 * Deterministic float64 reduction.  ws: batch * 1024 doubles (ct_workspace_bytes(CT_WS_LAB_STATS, ., batch) suffices). */
int ct_frame_psnr_f32(const float *a, const float *b, int64_t n_elems, int batch, double *out, void *ws,
                      size_t ws_bytes, void *stream);

/* ---- per-frame SSIM and iCID, the other metrics of Runner.test_step (methods/__init__.py:33,35) --------------------------
 * a (result, already clamped to [0,1] by the caller like methods/__init__.py:30), b (ground truth): [batch][3][height][width]
 * float32, NCHW planes; out[i] = the metric of frame i.
 *   ct_frame_ssim_f32  piq.ssim defaults: average-pool by f = max(1, round(min(H,W)/256)), 11x11 Gaussian (sigma 1.5) on
 *                      "valid" windows, k1 0.01, k2 0.03, mean over channels and positions; needs min(H,W)/f >= 11
 *   ct_frame_icid_f32  utils/icid.py:28-152 (intent "perceptual", all seven maps, downsampling on): bilinear resize by 1/f,
 *                      Lab, eleven 11x11 sigma-2 Gaussian moments with reflect padding, 1 - mean(product of the maps)
 * One fused tile kernel per metric + a fixed-order finishing kernel; float32 arithmetic like the reference's torch code,
 * float64 sums.  ws: ct_metric_workspace_bytes(height, width, batch).                                                  */
size_t ct_metric_workspace_bytes(int height, int width, int batch);
int ct_frame_ssim_f32(const float *a, const float *b, int height, int width, int batch, double *out, void *ws,
                      size_t ws_bytes, void *stream);
int ct_frame_icid_f32(const float *a, const float *b, int height, int width, int batch, double *out, void *ws,
                      size_t ws_bytes, void *stream);

/* ---- a4: methods.iterative.iterative_distribution_transfer (methods/iterative.py:8-59) ----
 * Per iteration: projection on the rotated axes (float64, fma chain), exact lo/hi, 2x3 histograms
 * with numpy's bin rule (LDS-binned integer atomics), cumulative LUT, np.interp apply with the
 * `left=0` quirk, back-rotation.  The rotation matrices are DATA drawn by the caller (the
 * reference draws them from numpy's global RNG, iterative.py:32):
 *   rot, rinv : device [batch][n_iter][9] float64 row-major, rinv = inverse(rot)
 *   out       : device [batch][n_t][3] float64 (the reference returns float64); also the working image
 *   round_dr_f32 : 1 reproduces `d_r = np.empty_like(target.T)` being float32 on iteration 0 for
 *                  float32 callers (iterative.py:36)
 *   bins <= 1024, n_iter >= 1;  ws: ct_idt_workspace_bytes(batch, n_iter, bins) bytes
 *   dbg  : NULL, or device buffers receiving the integer/LUT state (parity probes):
 *          hist [batch][n_iter][2][3][bins] u32 (0 = target, 1 = reference), lut [batch][n_iter][3][bins][2]
 *          (f, slope), par [batch][n_iter][3][4] (lo, hi, step, bins/(hi-lo)), binidx [batch][n_iter][3][n_t] u16 */
typedef struct ct_idt_debug {
    unsigned int *hist;
    double *lut;
    double *par;
    unsigned short *binidx;
} ct_idt_debug;

size_t ct_idt_workspace_bytes(int batch, int n_iter, int bins);
int ct_idt_f32(const float *target, int64_t n_t, const float *reference, int64_t n_r, int batch,
               const double *rot, const double *rinv, int n_iter, int bins, int round_dr_f32,
               double *out, void *ws, size_t ws_bytes, const ct_idt_debug *dbg, void *stream);
int ct_idt_f64(const double *target, int64_t n_t, const double *reference, int64_t n_r, int batch,
               const double *rot, const double *rinv, int n_iter, int bins, int round_dr_f32,
               double *out, void *ws, size_t ws_bytes, const ct_idt_debug *dbg, void *stream);

/* ---- a5-a9: DCMCS3DI forward building blocks (methods/dcmcs3di.py:41-66, pasmnet/ *.py) ----
 * NCHW float32 tensors as the reference's modules exchange them; exact-f32 MFMA arithmetic.
 *
 * ct_conv2d_f32: torch.nn.Conv2d(cin, cout, ksize, padding=ksize/2) + bias, then optionally
 *   LeakyReLU(0.01) (act=1; pasmnet/backbone.py:10), `+ residual` (ResB skip, backbone.py:15),
 *   clamp to [0,1] (dcmcs3di.py:61).  ksize in {1,3}, cout <= 64.
 *   wp   : weights packed [ksize*ksize][ceil(cin/2)][2][32*ceil(cout/32)] (zero padded):
 *          wp[ky*ksize+kx][ci/2][ci%2][co] = weight[co][ci][ky][kx]
 *   bias : [32*ceil(cout/32)] zero padded
 *   *_bstride : elements between consecutive images of the batch (lets in/out be channel slices)   */
int ct_conv2d_f32(const float *in, const float *wp, const float *bias, const float *residual,
                  float *out, int n, int cin, int cout, int h, int w, int ksize,
                  long long in_bstride, long long out_bstride, long long res_bstride, int act,
                  int clamp, void *stream);

/* The same convolution family on the bf16 matrix pipe with float32-grade accuracy (csrc/conv_split.hip): every float32
 * operand is split into three bf16 pieces and a product is the sum of the six partial products above 2^-16 of the
 * leading one, accumulated in float32 (error ~ one float32 rounding per product; NOT bitwise the fmaf chain of
 * ct_conv2d_f32).  Stride 1, padding k/2, kernel (kh,kw) in {3x3, 1x1, 1x5, 5x1} (f16 form also 2x2 with padding 1 on the top / left
 * only: see ct_space_to_depth2_f32); w % 4 == 0; in / out / residual
 * 16-byte aligned with batch strides % 4 == 0 (CT_E_BADARG otherwise: use ct_conv2d_f32 / ct_gconv2d_f32 then).
 * wp_split: bf16 bit patterns [ceil(cout/64)][ceil(cin/16)][kh*kw][piece hi,mid,lo][m][k-half][cout%32][8 channels];
 * bias: zero padded to 64*ceil(cout/64).  act: 0 none, 1 LeakyReLU(0.01), 2 ReLU, 3 sigmoid, 4 tanh, 5 swish.
 * in2 != NULL: input channels [cin1, cin) come from in2 (cin1 % 16 == 0) -- torch.cat([a, b], dim=1) without the copy
 * (reg_refine.py:43,72,75: the GRU's hx / [r*h, x] and the motion encoder's [cor, flo]).
 * in3 != NULL (needs in2): channels [cin2, cin) come from in3 (cin2 % 16 == 0, cin1 < cin2 < cin): DCMCS3DI's
 * transfer[0] reads cat([fea_left, fea_warped, valid_left]) (methods/dcmcs3di.py:59,47) from its three tensors.
 * f16 != 0: wp_split is the TWO-piece fp16 image of weight * 2^w_exp instead (ct_hip.pack_conv_weight_split16: same index order
 * with [piece hi,lo]; three v_mfma_f32_32x32x16_f16 per product, 2^-22 relative dropped; every staged 16-channel input tile
 * carries a running power-of-two scale per output tile (exponent clamped to +-100), so inputs with |x| < 2^111 are in
 * range; a tile maximum at or above 2^112 overflows fp16: inf / NaN results) -- half the matrix work.
 * post_op (f16 form only; p1 / p2 dense tensors with out's strides): 1 = the activated result times p1 (the GRU's r * h,
 * reg_refine.py:50), 2 = (1 - p1) * p2 + p1 * result (its gate h = (1 - z) h + z q, reg_refine.py:47,55), before the clamp.
 * residual: added after the activation (ResB skip), or -- res_pre_act != 0 -- BEFORE it: a pre-computed partial convolution
 * (the SepConvGRU's loop-invariant `inp` channels, reg_refine.py:25-55: conv(cat([h, inp, motion])) = conv_inp(inp) + conv(rest)).
 * scratch (nullable; f16 form; 16-byte aligned, >= ct_conv_split_scratch_bytes(), ALL ZERO before its first use, then owned by
 * the launches of ONE stream: each launch leaves its flag words zero again): lets a launch whose (8x32-pixel, 64-channel) units
 * do not fill a whole number of rounds of the 512 resident workgroups share the units' input-channel loops between neighbouring
 * workgroups (stream-K: e.g. 544 units take 1.06 rounds instead of 2).  The float32 sums of a shared unit are added in a fixed
 * order, so results are deterministic; they differ from the scratch == NULL result by float32 rounding only.
 * CT_E_WORKSPACE: scratch misaligned or too small. */
size_t ct_conv_split_scratch_bytes(void);
int ct_conv2d_split_f32(const float *in, const float *in2, int cin1, const float *in3, int cin2, const void *wp_split,
                        const float *bias, const float *residual, float *out, int n, int cin, int cout, int h, int w, int kh,
                        int kw, long long in_bstride, long long in2_bstride, long long in3_bstride, long long out_bstride,
                        long long res_bstride, int act, int clamp, int res_pre_act, int f16, int w_exp, int post_op,
                        const float *p1, const float *p2, void *scratch, long long scratch_bytes, void *stream);

/* ct_conv2d_split_f32 of one input tensor with the result stored as TOKEN ROWS: out_rows[(n*h + y)*w + x][rows_c0 + co] of a
 * [n*h, w, rows_channels] tensor (rows_channels, rows_c0, cout multiples of 4; rows_c0 + cout <= rows_channels).  The query / key /
 * value 1x1 convolutions of the parallax attention (pasmnet/attention.py:39-40,44-45, dcmcs3di.py:58) write the layout
 * ct_attention_rows64_f32 reads, so their NCHW tensors and the transposes never exist.  No residual / clamp in this form. */
int ct_conv2d_split_rows_f32(const float *in, const void *wp_split, const float *bias, float *out_rows, int n, int cin, int cout,
                             int h, int w, int kh, int kw, long long in_bstride, int rows_channels, int rows_c0, int act,
                             int f16, int w_exp, void *stream);

/* The same for the 1x1 convolutions in front of the parallax attention (pasmnet/attention.py:39-40,44-45: query / key; dcmcs3di.py:58: value),
 * 64 input channels -> cout <= 64 (a multiple of 4), as a STREAMING kernel of its own (csrc/conv1x1_rows.hip; new in ABI 9): exact float32
 * products on the float32 matrix pipe (v_mfma_f32_32x32x2_f32), the 64 x 64 weights in a wave's registers, no LDS, no barrier -- a 1x1
 * convolution is HBM-bound by a factor of two even there.  weight: the Conv2d weight itself, float32 [cout][64]; bias [cout];
 * out_rows / rows_channels / rows_c0 / act (0 .. 4) as above; out_rows and weight 16-byte aligned. */
int ct_conv1x1_rows_f32(const float *in, const float *weight, const float *bias, float *out_rows, int n, int cin, int cout, int h, int w,
                        long long in_bstride, int rows_channels, int rows_c0, int act, void *stream);

/* The ResB convolutions (3x3, stride 1, padding 1, 32 < cin <= 64; reference pasmnet/backbone.py:8-15, unimatch/backbone.py
 * residual blocks) on the weight-stationary kernel of csrc/conv_ws.hip with float32 operands as TWO fp16 pieces and three
 * v_mfma_f32_32x32x16_f16 per 16-channel product (float32 accumulation; 2^-22 relative is dropped: float32-grade, not bitwise
 * the fmaf chain).  Every staged input row is scaled by a power of two of its own (exponent clamped to +-100), the
 * weights by 2^w_exp, so inputs with |x| < 2^111 are in range (a row maximum at or above 2^112 overflows fp16: inf / NaN
 * results; the bf16 three-piece and exact-f32 modes have no such bound); 64 * h * w must be below 2^32 (32-bit indexing).  wp16: fp16 bit patterns [ceil(cout/64)][ceil(cin/16)][9][piece hi/lo][m][k-half][cout%32][8]
 * of weight * 2^w_exp; bias zero-padded to 64 * ceil(cout/64); act / clamp / residual as ct_conv2d_split_f32. */
int ct_conv3x3_ws16_f32(const float *in, const void *wp16, int w_exp, const float *bias, const float *residual, float *out, int n,
                        int cin, int cout, int h, int w, long long in_bstride, long long out_bstride, long long res_bstride, int act,
                        int clamp, void *stream);

/* The same convolution as Winograd F(2x2, 3x3) on the two fp16 pieces (csrc/conv_wino.hip): 16 instead of 36 multiplications per
 * 2x2 output tile and channel pair, i.e. 2.25x fewer matrix instructions -- the weight-stationary kernel above runs at the chip's
 * power limit on real data.  Same arguments, bounds and float32-grade rounding (3e-7 of the output range against float64; results
 * are not bitwise those of ct_conv3x3_ws16_f32); every tile row of four input rows carries one power-of-two scale.
 * wq16: fp16 bit patterns [ceil(cout/64)][16 positions][4 cout blocks][2 cin chunks][piece hi/lo][64 lanes][8] of
 * (G g G^T) * 2^w_exp in the A-fragment order of v_mfma_f32_16x16x32_f16 (ct_hip.pack_conv_weight_wino16). */
int ct_conv3x3_wino16_f32(const float *in, const void *wq16, int w_exp, const float *bias, const float *residual, float *out, int n,
                          int cin, int cout, int h, int w, long long in_bstride, long long out_bstride, long long res_bstride, int act,
                          int clamp, void *stream);
/* Which kernel ct_conv3x3_wino16_f32 launches (new in ABI 9; process-wide, atomic; env CT_HIP_WINO_FORM=1 presets 1):
 *   0 (default)  csrc/conv_wino4.hip: four waves of 512 registers per CU, the transformed weights in the accumulation registers, the
 *                input transform of the next two output rows inside the matrix phase of the current ones (images of 64 * h * w * 4
 *                bytes below 2^30; larger ones take form 1 by themselves);
 *   1            csrc/conv_wino.hip: the eight-wave, three-phase kernel of round 5.
 * Both are float32-grade and deterministic; they sum the sixteen positions in different orders, so they agree to rounding, not bitwise. */
int ct_set_conv_wino_form(int form);

/* Parallax attention, one direction (pasmnet/attention.py:39-41, utils.py:30, utils.py:123-125):
 *   P = softmax_j( sum_c q[c][h][i] k[c][h][j] / c ) ;  out_v[c][h][i] = sum_j P[i][j] v[c][h][j],
 *   out_rgb likewise for the 3 channels of `rgb` (dcmcs3di.py:58,65).  att: NULL, or [n][h][w][w]
 *   receiving P (the API's att_right2left).  w <= ~1000 (one 32-query tile of S lives in LDS).    */
int ct_pam_attend_f32(const float *q, const float *k, const float *v, const float *rgb,
                      float *out_v, float *out_rgb, float *att, int n, int c, int cv, int h, int w,
                      void *stream);
/* valid mask of the opposite direction (utils.py:31,34-35): valid[n][0][h][j] = 1.0 if
 * sum_i softmax_j(q.k/c)[i][j] > 0.1 else 0.0; colsum (nullable) receives the sums themselves.
 * ws: ct_pam_workspace_bytes(n,h,w).  Column sums are added in a fixed order (deterministic).   */
size_t ct_pam_workspace_bytes(int n, int h, int w);
int ct_pam_valid_f32(const float *q, const float *k, float *valid, float *colsum, float *att,
                     int n, int c, int h, int w, void *ws, size_t ws_bytes, void *stream);

/* ---- a10-a16: GMFlow / UniMatch matcher building blocks (unimatch/ *.py, as called by methods/dmsct.py:85-94) ----
 * float32, exact-f32 MFMA for the contractions.  "tokens" = channels-last [batch][H*W][C] (the layout the
 * reference's transformer uses, transformer.py:238-239); everything else NCHW.
 *
 * ct_gconv2d_f32: Conv2d with any kernel / stride / padding / channel count (backbone.py:14-17,53,67;
 *   trident_conv.py:64-72; reg_refine.py:16-17,33-39,65-69,104-107; unimatch.py:59). act: 0 none, 1 LeakyReLU(0.01),
 *   2 ReLU, 3 sigmoid, 4 tanh.  wp: output channels in groups of 64, [ceil(cout/64)][kh*kw][ceil(cin/2)][2][64] (zero
 *   padded); bias padded to 64*ceil(cout/64), or NULL.  Stride-1 "same" convolutions with a 3x3 / 1x1 / 1x5 / 5x1
 *   kernel and a bias run on the LDS-tiled persistent kernel of ct_conv2d_f32, everything else on a generic one.  */
int ct_gconv2d_f32(const float *in, const float *wp, const float *bias, float *out, int n, int cin,
                   int cout, int h, int w, int kh, int kw, int stride, int pad_h, int pad_w,
                   long long in_bstride, long long out_bstride, int act, void *stream);
/* InstanceNorm2d(affine=False) over `planes` planes of `plane` elements (backbone.py:10,34-39):
 *   mode 0: IN(x)   1: relu(IN(x))   2: relu(skip + relu(IN(x))).  ws (nullable, 8-byte aligned,
 *   ct_instance_norm_workspace_bytes(planes)): lets few large planes be split over several workgroups each.       */
size_t ct_instance_norm_workspace_bytes(int planes);
int ct_instance_norm_f32(const float *x, const float *skip, float *y, int planes, int plane, float eps,
                         int mode, void *ws, size_t ws_bytes, void *stream);
/* Space-to-depth by 2 of an NCHW tensor (h even, w % 8 == 0, 16-byte aligned): out[n][(2 sy + sx) c_total + c][y][x] =
 * in[n][c][2y + sy][2x + sx], out dense [n][4c][h/2][w/2].  With it a stride-2 3x3 "same" convolution (unimatch/backbone.py:14-17,
 * 53,67; trident_conv.py:64-72) is ct_conv2d_split_f32 with kh = kw = 2 (taps at block offsets -1 / 0, f16 form only) over 4c
 * channels, and a stride-2 1x1 convolution is the 1x1 convolution of its first c channels. */
int ct_space_to_depth2_f32(const float *in, float *out, int n, int c, int h, int w, long long in_bstride, void *stream);
/* elementwise: op 0 a+b, 1 a*b, 2 (1-a)*b + a*c (GRU update, reg_refine.py:47,55), 3 normalize_img (utils.py:26-34),
 *   4 a*s0, 5 tanh on channels < split / relu on the rest (unimatch.py:320-323)                                   */
int ct_eltwise_f32(const float *a, const float *b, const float *c, float *y, long long n, int op,
                   int plane, int chans, int split, float s0, void *stream);
/* nn.Linear on tokens: out[t][n] = act(x[t][:] . w[n][:] + bias[n]); w in PyTorch layout [n][k]; k % 16 == 0;
 *   act 0 none, 6 exact GELU (transformer.py:26-41; attention.py:181-182).  x2 == NULL: x is [tokens][k], k1 == k.
 *   x2 != NULL: the input row is [x[t][0:k1] | x2[t][0:k-k1]], k1 % 32 == 0 -- torch.cat([source, message], -1) in
 *   front of the FFN (transformer.py:131) without materialising it.                                               */
int ct_linear_tokens_f32(const float *x, const float *x2, int k1, const float *w, const float *bias, float *out,
                         long long tokens, int k, int n, int act, void *stream);
/* the same layer on the bf16 matrix pipe with float32-grade accuracy (operands as three bf16 pieces, six MFMAs per product;
 *   csrc/conv_split.hip's arithmetic): wp = the weight pre-split on the host as bf16 bit patterns
 *   [ceil(n/128)][k/32][piece hi/mid/lo][8-channel group 0..3][feature row 0..127][8 channels], zero rows beyond n
 *   (ct_hip.pack_linear_weight_split).  k % 32 == 0.                                                              */
int ct_linear_tokens_split_f32(const float *x, const float *x2, int k1, const void *wp, const float *bias, float *out,
                               long long tokens, int k, int n, int act, void *stream);
/* The FFN-shaped linears (transformer.py:33-35,131-133: mlp = Linear(256 -> 1024) . GELU . Linear(1024 -> 128), no bias) with the
 * weight slice RESIDENT in LDS and float32 operands as two fp16 pieces (csrc/linear_ws16.hip; three MFMAs per product, float32
 * accumulation, 2^-22 relative dropped; power-of-two scales: one per layer on the weights, a running one per 32-token tile on the
 * activations).  Two shapes:
 *   k == 256, n % 128 == 0, n / 128 in {1,2,4,8}: out[tokens][n] = act(x' w^T + bias); x' = [x[t][0:128] | x2[t][0:128]] (k1 = 128)
 *        or x[t][0:256] (x2 = NULL, k1 = 256);
 *   n == 128, k % 256 == 0, k / 256 in {1,2,4,8}, x2 = NULL, act = 0: out = PARTIAL slabs [k/256][tokens][128] whose sum is the
 *        result (bias in slab 0) -- ct_layernorm128_f32(partials = k/256) adds them in slab order on its way in;
 *   k == 128, n % 128 == 0, n / 128 in {1,2,3,4}, x2 = NULL: several 128 -> 128 layers reading the same tokens in one launch (the
 *        q / k / v projections, transformer.py:26-31; w = their weights stacked): out = slabs [n/128][tokens][128], one per layer.
 * wp16: ct_hip.pack_linear_weight_ws16: fp16 bit patterns [slice][piece hi/lo][k step 0..15][lane half][feature 0..127][8 channels]
 * of w * 2^w_exp, channel of (step s, half h, j) = 128 h + 8 s + j within the slice (k == 128: 8 steps, 64 h + 8 s + j).
 * act: 0 none, 6 exact GELU.  ln_gamma / ln_beta != NULL (k = n = 128 only): out = [ln_residual +] LayerNorm_128(x w^T + bias) -- the
 * merge projection with norm1 and the skip of transformer.py:120-127,139-147 in one launch.                                   */
int ct_linear_ws16_f32(const float *x, const float *x2, int k1, const void *wp16, int w_exp, const float *bias, float *out,
                       long long tokens, int k, int n, int act, const float *ln_gamma, const float *ln_beta,
                       const float *ln_residual, void *stream);
/* LayerNorm(128, eps 1e-5, affine) on tokens, out = residual + LN(x) when residual != NULL (transformer.py:139-147);
 * partials > 1: x is [partials][tokens][128] and the normalised input is the sum of the slabs (added in slab order) */
int ct_layernorm128_f32(const float *x, const float *gamma, const float *beta, const float *residual,
                        float *out, long long tokens, int partials, void *stream);
/* single-head attention on tokens (C = 128), streaming softmax: out = softmax(q k^T * scale + mask) v.
 *   cv = 128 (swin window attention, attention.py:48-107) or 2 (global correlation -> expected coordinate,
 *   matching.py:10-39; flow propagation, attention.py:199-216).  region: NULL, or int32 [batch][len] ids;
 *   pairs with different ids get the additive -100 of the shifted-window mask (utils.py:87-111).
 *   rowmap: NULL (token (b, i) is row b*len + i of q/k/v/out), or int32 [batch][len] row indices into q/k/v/out: the
 *   (shifted) window partition of attention.py:60-92,100-107 as a gather/scatter table instead of roll + permute.
 *   nsplit > 1: the keys are split over nsplit workgroups per query tile and merged by a second kernel (same result up
 *   to float rounding); ws needs ct_attention_workspace_bytes(batch, len, cv, nsplit).  Use it when batch*len/128
 *   workgroups cannot fill the 256 CUs.
 *   kv_shift (rowmap launches; else 0): the keys / values of a token are read kv_shift rows further (mod batch*len) than
 *   its query -- the cross attention of transformer.py:281-287 attends every image to the other half of the batch
 *   (kv_shift = batch*len/2) without materialising torch.cat(chunk(2)[::-1]).
 *   Arithmetic (this entry and the two ct_attention_*64_f32 below): both products on the 16-bit matrix pipe with float32
 *   operands as two fp16 pieces and power-of-two scales per query row / staged key tile / running per value tile
 *   (csrc/attention16.hip; 2^-22 relative dropped; a probability below 2^-29 of its row's maximum keeps an absolute error of
 *   2^-40 of it); env CT_HIP_ATT16=0 selects the three-piece bf16 kernels of csrc/gmflow.hip.                      */
size_t ct_attention_workspace_bytes(int batch, int len, int cv, int nsplit);
int ct_attention_tokens_f32(const float *q, const float *k, const float *v, const int *region, const int *rowmap,
                            float *out, int batch, int len, int cv, float scale, int nsplit, float *ws,
                            size_t ws_bytes, long long kv_shift, void *stream);
/* Streaming parallax attention on 64-channel row tokens (pasmnet/attention.py:39-46, utils.py:30-35,123-125), for
 * image widths whose score tile does not fit LDS (ct_pam_* need w <= 1982) and as the faster path in general:
 *   ct_attention_rows64_f32 : out[b][i][0:96] = softmax_j(q_i.k_j*scale) v[b][j][0:96]   (v != NULL), and/or the row
 *                             statistics stats[b][i] = (max, sum) of that softmax (v == NULL: statistics only)
 *   ct_attention_colsum64_f32: colsum[b][j] = sum_i exp(q_i.k_j*scale - max_i) / sum_i  (the valid-mask numerator),
 *                             fixed summation order.  batch = N*H rows, len = W, tokens channels-last.                */
/* layout moves for the row attention: rows[(b*h + y)*w + x][c0 + ch] = nchw[b][ch][y][x] for ch < c (row_channels
 * floats per token in `rows`), and back.  nchw_bstride: elements between images (channel slices are fine).           */
int ct_nchw_to_rows_f32(const float *nchw, float *rows, int batch, int c, int h, int w, long long nchw_bstride,
                        int row_channels, int c0, void *stream);
int ct_rows_to_nchw_f32(const float *rows, float *nchw, int batch, int c, int h, int w, long long nchw_bstride,
                        int row_channels, int c0, void *stream);
int ct_attention_rows64_f32(const float *q, const float *k, const float *v, float *out, float *stats,
                            int batch, int len, float scale, void *stream);
int ct_attention_colsum64_f32(const float *q, const float *k, const float *stats, float *colsum,
                              int batch, int len, float scale, void *stream);
/* matching.py:42-86: flow[b][2][h][w] from the softmax over the (2r+1)^2 integer neighbourhood; f0,f1 tokens       */
int ct_local_corr_softmax_f32(const float *f0, const float *f1, float *flow, int batch, int h, int w,
                              int radius, void *stream);
/* matching.py:89-126: corr[b][(2r+1)^2][h][w] = f0 . bilinear(f1, pos + window + flow) / sqrt(128); radius <= 4   */
int ct_local_corr_flow_f32(const float *f0, const float *f1, const float *flow, float *corr, int batch,
                           int h, int w, int radius, void *stream);
/* attention.py:220-256: (2r+1)^2 local window attention of flow; q = q_proj(f), k = k_proj(f) as tokens          */
int ct_local_attn_prop_f32(const float *q, const float *k, const float *flow, float *out, int batch, int h,
                           int w, int radius, void *stream);
/* F.interpolate(bilinear, align_corners=True) to (ho, wo); channel 0 scaled by mul0, the others by mul1           */
int ct_bilinear_resize_f32(const float *in, float *out, int n, int c, int h, int w, int ho, int wo,
                           float mul0, float mul1, void *stream);
/* geometry.py:68-75 flow_warp: bilinear sample at (x,y)+flow, zeros padding, align_corners=True                   */
int ct_flow_warp_f32(const float *img, const float *flow, float *out, int n, int c, int h, int w,
                     void *stream);
/* utils.py:137-155 convex upsampling by `factor` (mask [b][9*factor^2][h][w])                                     */
int ct_convex_upsample_f32(const float *flow, const float *mask, float *out, int b, int h, int w,
                           int factor, void *stream);
/* geometry.py:78-99 given flow_warp(bwd, fwd) and flow_warp(fwd, bwd): occlusion masks as 0/1 [b][h][w]          */
int ct_fb_check_f32(const float *fwd, const float *bwd, const float *warped_bwd, const float *warped_fwd,
                    float *fwd_occ, float *bwd_occ, int b, int h, int w, float alpha, float beta,
                    void *stream);

/* piq.fsim(result, gt) of Runner.test_step (methods/__init__.py:34; FSIMc, piq defaults; third-party, restated: parity
 *   unpinned).  ct_fsim_setup_f32 builds, once per frame size, the log-Gabor filter bank ([16][hp*wp] float32,
 *   orientation-major; hp x wp = ct_fsim_pooled_size) and its three noise constants per orientation ([4][3] float64) on
 *   the device; ct_frame_fsim_f32 scores `batch` frames [batch][3][h][w] in [0,1]: out[b] float64.  The 1 + 16 FFTs per
 *   image run in this library's own batched 2-D transform (ct_fft2d_c2c_f32 below; no vendor FFT, no plan, no state);
 *   ws: 256-byte aligned, ct_fsim_workspace_bytes(batch, h, w) (0: pooled frame beyond 4096 points on an axis).       */
/* Batched in-place 2-D complex DFT (csrc/fft2d.hip): `planes` planes [hp][wp] of interleaved (re, im) float32 -- what
 *   torch.fft.fft2 (inverse = 0) and torch.fft.ifft2 x hp x wp (inverse = 1: unnormalised) compute inside piq.fsim.  Mixed-radix
 *   Stockham transforms of whole lines in LDS (radices 4, 2, 3, 5 in registers, any other prime factor as a plain butterfly):
 *   every size with both axes <= 4096; asynchronous, no workspace.                                                    */
int ct_fft2d_c2c_f32(void *data, int hp, int wp, int planes, int inverse, void *stream);
int ct_fsim_pooled_size(int h, int w, int *hp, int *wp);
size_t ct_fsim_workspace_bytes(int batch, int h, int w);
int ct_fsim_setup_f32(int h, int w, float *filters, double *consts, void *ws, size_t ws_bytes, void *stream);
int ct_frame_fsim_f32(const float *a, const float *b, double *out, int batch, int h, int w, const float *filters,
                      const double *consts, void *ws, size_t ws_bytes, void *stream);

/* ---- f4: DMSCT's colour-correction network (methods/dmsct.py:34-56,96-116): segmentation_models_pytorch's EfficientNet-B2
 * encoder (efficientnet_pytorch MBConv blocks), UnetDecoder and SegmentationHead.  Third-party, absent offline: the
 * structure is restated in oracle/smp_unet.py ("parity unpinned").  float32 NCHW.  The 1x1 / 3x3 convolutions of these
 * layers are ct_gconv2d_f32 / ct_conv2d_split_f32 with the BatchNorm folded into weight and bias (act 5 = swish).
 *
 * ct_gconv2d_pad_f32: ct_gconv2d_f32's generic kernel with explicit top / left zero padding and output size (the bottom /
 *   right padding is what the output size implies): efficientnet_pytorch's static TF-"SAME" padding, e.g. (0, 1) for the
 *   stride-2 stem (Conv2dStaticSamePadding).
 * ct_dwconv_f32: depthwise k x k convolution (k 3 / 5, stride 1 / 2), same padding convention, BatchNorm folded into
 *   w [c][k*k] and bias [c], act 0 / 5 (`swish(bn1(depthwise_conv(x)))`, MBConvBlock.forward).  tile_sums (nullable):
 *   [n][c][ct_dwconv_tiles(out_h, out_w)] receives the sum of every output tile -- the squeeze of the SE gate.
 * ct_se_gate_f32: gate[n][c] = sigmoid(se_expand(swish(se_reduce(mean)))) with mean = sum(tile_sums) / plane (float64,
 *   fixed order); w_reduce [nsq][c], w_expand [c][nsq].
 * ct_scale_planes_f32: x[p][:] *= gate[p], in place (`torch.sigmoid(x_squeezed) * x`).
 * ct_upsample2_concat_f32: DecoderBlock.forward's `cat([interpolate(x, scale_factor=2, mode="nearest"), skip], 1)`.    */
int ct_gconv2d_pad_f32(const float *in, const float *wp, const float *bias, float *out, int n, int cin, int cout, int h,
                       int w, int kh, int kw, int stride, int pad_top, int pad_left, int out_h, int out_w,
                       long long in_bstride, long long out_bstride, int act, void *stream);
int ct_dwconv_tiles(int out_h, int out_w);
int ct_dwconv_f32(const float *in, const float *w, const float *bias, float *out, int n, int c, int h, int wd, int k,
                  int stride, int pad_top, int pad_left, int out_h, int out_w, int act, float *tile_sums, void *stream);
int ct_se_gate_f32(const float *tile_sums, int tiles, int plane, const float *w_reduce, const float *b_reduce,
                   const float *w_expand, const float *b_expand, float *gate, int n, int c, int nsq, void *stream);
int ct_scale_planes_f32(float *x, const float *gate, int planes, int plane, void *stream);
int ct_upsample2_concat_f32(const float *x, const float *skip, float *out, int n, int cx, int cs, int h, int w,
                            void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CT_HIP_H */
