// Microbenchmark: cost of one workgroup barrier interval on gfx950, by workgroup size and what happens between barriers.
// hipcc --offload-arch=gfx950 -O3 tools/mb_barrier.hip -o tools/mb_barrier.bin && tools/mb_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MODE 0: bare barrier; 1: one LDS broadcast read + readfirstlane + barrier; 2: mode 1 + 6 returning LDS atomics in ONE wave only;
// 3: mode 2 in all waves; 4: mode 1 with __syncthreads()
template <int NT, int MODE>
__global__ __launch_bounds__(NT) void k(unsigned iters, unsigned *out) {
    __shared__ unsigned h[4096];
    __shared__ unsigned s[64];
    const unsigned tid = threadIdx.x;
    for (unsigned i = tid; i < 4096; i += NT) h[i] = 0;
    if (tid < 64) s[tid] = tid;
    __syncthreads();
    unsigned x = tid * 2654435761u + blockIdx.x, acc = 0;
    for (unsigned it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            const unsigned v = (unsigned)__builtin_amdgcn_readfirstlane((int)s[it & 63]);
            acc += v;
        }
        if (MODE == 2 || MODE == 3) {
            if (MODE == 3 || (tid >> 6) == (it & (NT / 64 - 1))) {
                unsigned o[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) {
                    x = x * 1664525u + 1013904223u;
                    o[u] = atomicOr(&h[(x >> 10) & 4095], 1u << (x & 31));
                }
#pragma unroll
                for (int u = 0; u < 6; ++u) acc += o[u];
            }
        }
        if (MODE == 4)
            __syncthreads();
        else
            lds_sync();
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int NT, int MODE>
void run(const char *name, unsigned *out) {
    const unsigned iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NT, MODE>), dim3(256), dim3(NT), 0, 0, iters, out);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NT, MODE>), dim3(256), dim3(NT), 0, 0, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s NT %4d  %7.3f ms  %7.1f ns per barrier interval\n", name, NT, ms, ms * 1e6 / iters);
    fflush(stdout);
}

int main() {
    unsigned *out;
    CK(hipMalloc(&out, 256));
    run<1024, 0>("bare barrier", out);
    run<512, 0>("bare barrier", out);
    run<256, 0>("bare barrier", out);
    run<1024, 1>("lds read + readfirstlane + barrier", out);
    run<512, 1>("lds read + readfirstlane + barrier", out);
    run<1024, 4>("lds read + readfirstlane + __syncthreads", out);
    run<1024, 2>("+ 6 or_rtn in one wave", out);
    run<512, 2>("+ 6 or_rtn in one wave", out);
    run<1024, 3>("+ 6 or_rtn in all waves", out);
    run<512, 3>("+ 6 or_rtn in all waves", out);
    return 0;
}
