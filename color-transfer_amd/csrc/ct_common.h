// ct_common.h -- shared launch / reduction helpers for libct_hip.so (gfx950 only).
#pragma once
#include <atomic>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ct_hip.h"

namespace ct {

constexpr int kBlock = 256;        // 4 waves of 64
constexpr int kWave = 64;
constexpr int kMaxBlocksPerImage = 1024;
// total workgroups a streaming launch aims for: 256 CUs x 8 (guide: cap ~2048, grid-stride the rest)
constexpr int kTargetBlocks = 2048;
inline int target_blocks() {
    static int v = [] { const char *e = getenv("CT_HIP_TARGET_BLOCKS"); int x = e ? atoi(e) : 0; return x > 0 ? x : kTargetBlocks; }();
    return v;
}

inline int blocks_per_image(int64_t n_chunks, int n_images) {
    int64_t want = (n_chunks + kBlock - 1) / kBlock;
    int64_t cap = target_blocks() / (n_images > 0 ? n_images : 1);
    if (cap < 8) cap = 8;
    if (cap > kMaxBlocksPerImage) cap = kMaxBlocksPerImage;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)want;
}

// ---- per-device launch state ----------------------------------------------------------------------------------------------------
// Function attributes and the CU count belong to a DEVICE: a process that drives a second GPU must set them there as well.
constexpr int kMaxDevices = 64;
inline int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev >= 0 && dev < kMaxDevices ? dev : 0;
}
// hipFuncAttributeMaxDynamicSharedMemorySize of one kernel, raised once per device (one static instance per kernel symbol;
// safe from concurrent host threads: the worst case is the same attribute set twice)
struct DynLdsAttr {
    std::atomic<int> have[kMaxDevices];
    hipError_t ensure(const void *fn, size_t bytes) {
        const int dev = current_device();
        if (have[dev].load(std::memory_order_acquire) >= (int)bytes) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) have[dev].store((int)bytes, std::memory_order_release);
        else (void)hipGetLastError();
        return e;
    }
};

#define CT_CHECK_LAUNCH()                           \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

// ---- zeroing workspace words on the launch stream ---------------------------------------------------------------------------
// A kernel of this library instead of hipMemsetAsync: captured in a hipGraph (torch.cuda.graph), a memset node in front of a
// kernel was seen not to have run when that kernel read the words on replay, once any other call of the entry had preceded
// the capture (ROCm 7.2, round 4; csrc/reinhard_persist.hip found it: stale arrival counts, NaN results).  A kernel node
// keeps the stream order.  p: 4-byte aligned, bytes a multiple of 4.
static __global__ void ct_zero_words_kernel(unsigned int *__restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline int zero_async(void *p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    const size_t n = (bytes + 3) / 4;
    size_t g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(ct_zero_words_kernel, dim3((unsigned)g), dim3(256), 0, s, reinterpret_cast<unsigned int *>(p), n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// ---- 4-pixel (12 element) vector I/O of interleaved HWC data -------------------------------
// A lane owns 4 whole pixels = 48 B (f32) / 96 B (f64) of contiguous memory, fetched as
// 16-byte vectors.  `vec` = the image base is 16-byte aligned (wave-uniform).
template <typename T>
__device__ __forceinline__ void load12(const T *p, bool vec, double (&v)[12]);

template <>
__device__ __forceinline__ void load12<float>(const float *p, bool vec, double (&v)[12]) {
    if (vec) {
        const float4 *q = reinterpret_cast<const float4 *>(p);
        const float4 a = q[0], b = q[1], c = q[2];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = p[i];
    }
}

template <>
__device__ __forceinline__ void load12<double>(const double *p, bool vec, double (&v)[12]) {
    if (vec) {
        const double2 *q = reinterpret_cast<const double2 *>(p);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double2 a = q[i];
            v[2 * i] = a.x;
            v[2 * i + 1] = a.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = p[i];
    }
}

// raw (unconverted) form, so that a chunk can be prefetched without paying for 12 doubles
template <typename T>
struct Raw12 {
    T e[12];
};

template <typename T>
__device__ __forceinline__ void load12_raw(const T *p, bool vec, Raw12<T> &r) {
    constexpr int VE = 16 / sizeof(T);  // elements per 16-byte vector
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    if (vec) {
        const vec_t *q = reinterpret_cast<const vec_t *>(p);
#pragma unroll
        for (int i = 0; i < 12 / VE; ++i) {
            const vec_t a = q[i];
#pragma unroll
            for (int j = 0; j < VE; ++j) r.e[i * VE + j] = a[j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) r.e[i] = p[i];
    }
}

template <typename T>
__device__ __forceinline__ void unpack12(const Raw12<T> &r, double (&v)[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = (double)r.e[i];
}

template <typename T>
__device__ __forceinline__ void store12(T *p, bool vec, const T (&v)[12]);

template <>
__device__ __forceinline__ void store12<float>(float *p, bool vec, const float (&v)[12]) {
    if (vec) {
        float4 *q = reinterpret_cast<float4 *>(p);
        q[0] = make_float4(v[0], v[1], v[2], v[3]);
        q[1] = make_float4(v[4], v[5], v[6], v[7]);
        q[2] = make_float4(v[8], v[9], v[10], v[11]);
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) p[i] = v[i];
    }
}

template <>
__device__ __forceinline__ void store12<double>(double *p, bool vec, const double (&v)[12]) {
    if (vec) {
        double2 *q = reinterpret_cast<double2 *>(p);
#pragma unroll
        for (int i = 0; i < 6; ++i) q[i] = make_double2(v[2 * i], v[2 * i + 1]);
    } else {
#pragma unroll
        for (int i = 0; i < 12; ++i) p[i] = v[i];
    }
}

// ---- deterministic block sum of NV doubles per thread ------------------------------------
// wave: __shfl_down tree (fixed shape); block: 4 wave leaders through LDS, added in wave
// order by thread 0.  Result valid in thread 0 only.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *lds /* [4][NV] */) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += __shfl_down(v[i], off, kWave);
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) lds[wid * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = ((lds[i] + lds[NV + i]) + lds[2 * NV + i]) + lds[3 * NV + i];
    }
}

}  // namespace ct
