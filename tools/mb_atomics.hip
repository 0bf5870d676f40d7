// Microbenchmark: rate of global int32 atomic adds as a function of the working set they hit (the question behind the
// y path for K >= 9: is a histogram of 4^K bins cheaper after the k-mers have been partitioned so that all workgroups
// work on one small slice of bins at a time?).  Every thread does `per` adds at hashed indices inside a window of
// `ws` bins; the window moves every `ws * dwell` adds of the whole grid (dwell = adds per bin before moving on).
// hipcc --offload-arch=gfx950 -O3 tools/mb_atomics.hip -o tools/mb_atomics.bin && tools/mb_atomics.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// MODE 0: atomicAdd without return (agent scope); 1: workgroup-scope atomic (same instruction, narrower scope bits);
// 2: plain load+store (not atomic: the ceiling of an uncontended read-modify-write)
template <int MODE>
__global__ __launch_bounds__(512) void k(int *bins, unsigned ws_mask, unsigned n_windows_mask, unsigned ws_log2, int per,
                                         unsigned window_adds_log2) {
    const unsigned gid = blockIdx.x * 512u + threadIdx.x, nthreads = gridDim.x * 512u;
    for (int j = 0; j < per; ++j) {
        const unsigned long long seq = (unsigned long long)j * nthreads + gid;      // global order of adds
        const unsigned win = (unsigned)(seq >> window_adds_log2) & n_windows_mask;  // which window of bins
        const unsigned idx = (win << ws_log2) | (hash32((unsigned)seq) & ws_mask);
        if (MODE == 0)
            __hip_atomic_fetch_add(&bins[idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1)
            __hip_atomic_fetch_add(&bins[idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else
            bins[idx] += 1;
    }
}

template <int MODE>
void run(const char *name, int *bins, unsigned total_log2, unsigned ws_log2) {
    const int nblk = 2048, per = 256;  // 2^28 adds
    const unsigned ws_mask = (1u << ws_log2) - 1u, nwin_mask = (1u << (total_log2 - ws_log2)) - 1u;
    // each window gets 2^28 / n_windows adds, in one visit
    const unsigned window_adds_log2 = 28 - (total_log2 - ws_log2);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(512), 0, 0, bins, ws_mask, nwin_mask, ws_log2, per, window_adds_log2);
    CK(hipEventRecord(e0));
    const int reps = 3;
    for (int it = 0; it < reps; ++it)
        hipLaunchKernelGGL((k<MODE>), dim3(nblk), dim3(512), 0, 0, bins, ws_mask, nwin_mask, ws_log2, per, window_adds_log2);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double adds = (double)nblk * 512 * per;
    printf("%-18s bins 2^%2u  window 2^%2u bins (%8.0f KB)  %8.3f ms  %7.2f G adds/s\n", name, total_log2, ws_log2,
           (double)(4ull << ws_log2) / 1024, ms, adds / ms / 1e6);
    fflush(stdout);
}

int main() {
    int *bins;
    CK(hipMalloc(&bins, (size_t)4 << 28));
    CK(hipMemset(bins, 0, (size_t)4 << 28));
    for (unsigned ws : {28u, 24u, 22u, 20u, 18u, 16u, 14u, 12u}) run<0>("agent atomic", bins, 28, ws);
    for (unsigned ws : {24u, 20u, 16u}) run<0>("agent atomic", bins, 24, ws);
    for (unsigned ws : {28u, 24u, 20u, 16u}) run<1>("workgroup atomic", bins, 28, ws);
    for (unsigned ws : {28u, 24u, 20u, 16u}) run<2>("plain rmw", bins, 28, ws);
    return 0;
}
