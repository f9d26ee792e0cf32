// ct_reinhard_persist.h -- host-side interface of reinhard_persist.hip (the one-launch Reinhard transfer) for linear.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace ct {
namespace rp {

// true when a batch of n_pixels-sized float32 / uint8 frames can take the persistent launch on the current device:
// every workgroup's share of the target (n_pixels / 256 / CUs tiles of 3 KB) must fit beside the tables in one CU's LDS.
// `any_size`: accept images too small to occupy every wave (the explicit ct_reinhard_persist_* entries: tests);
// the automatic dispatch of ct_reinhard_f32 / ct_reinhard_psnr_f32 asks for any_size = false.
bool eligible(int64_t n_pixels, bool any_size);

// device workspace the launch needs for `batch` pairs (0 when not eligible)
size_t ws_bytes(int64_t n_pixels, int batch);

// Enqueue color_transfer_between_images (methods/linear.py:8-42) for `batch` pairs, optionally with the per-frame squared
// error against gt (psnr_out: {mse, PSNR} per pair).  T = float (frames in [0,1]) or uint8_t (k / 255 as float32, the
// reference's `.float() / 255`, utils/data.py:84,106,125).  Returns CT_OK or an error code; the caller checked eligible().
// ev_start / ev_stop: optional events recorded immediately around the persistent kernel.
template <typename T>
int launch(const T *target, const T *reference, const T *gt, float *out, double *psnr_out, int64_t n_pixels, int batch,
           double *stats_out, void *ws, size_t ws_bytes, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop);

// sticky status of the current device (bit 0: a bounded spin of some persistent launch gave up since the last clear); -1 = the
// device could not be read.  Blocks the calling thread for one 4-byte copy; does not wait for running work.
int read_status(bool clear);

// first word of the workspace after a launch: 0 = ok, otherwise a bounded spin gave up (a workgroup of the grid was not
// resident); the results of that call are NaN
constexpr int kErrWord = 0;

}  // namespace rp
}  // namespace ct
