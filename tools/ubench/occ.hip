// occupancy the runtime reports for workgroups of 256 threads with a given static LDS footprint (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES> __global__ __launch_bounds__(256) void k(float *o) {
    __shared__ float s[BYTES / 4];
    s[threadIdx.x] = o[threadIdx.x];
    __syncthreads();
    o[threadIdx.x] = s[(threadIdx.x * 7) % (BYTES / 4)];
}
template <int BYTES> void q() {
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<BYTES>, 256, 0);
    printf("LDS %6d B/WG -> %d WG/CU (%s)\n", BYTES, n, hipGetErrorString(e));
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: CUs %d, sharedMemPerBlock %zu, sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerMultiprocessor %d\n",
           p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock, p.sharedMemPerMultiprocessor, p.maxSharedMemoryPerMultiProcessor, p.regsPerMultiprocessor);
    q<16384>(); q<32768>(); q<36864>(); q<49920>(); q<65536>();
    return 0;
}
