// Microbenchmark: rate of LDS atomics on gfx950 (per CU), by form and address pattern.
// hipcc --offload-arch=gfx950 -O3 tools/mb_ldsatomic.hip -o tools/mb_ldsatomic.bin && tools/mb_ldsatomic.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: ds_add (no return); 1: ds_add_rtn; 2: ds_or_rtn; 3: plain ds_write; 4: plain ds_read; 5: ds_and_rtn + dependent ds_add_rtn
// PAT 0: random word in a table of WORDS; 1: conflict-free (lane-strided); 2: all lanes one address
template <int NT, int MODE, int PAT, int WORDS>
__global__ __launch_bounds__(NT) void k(unsigned iters, unsigned *out) {
    __shared__ unsigned h[WORDS];
    const unsigned tid = threadIdx.x;
    for (unsigned i = tid; i < WORDS; i += NT) h[i] = 0;
    __syncthreads();
    unsigned x = tid * 2654435761u + blockIdx.x, acc = 0;
    for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            unsigned a;
            if (PAT == 0) a = (x >> 10) % WORDS;
            if (PAT == 1) a = (tid + (x >> 28) * NT) % WORDS;
            if (PAT == 2) a = (x >> 31);
            if (MODE == 0) atomicAdd(&h[a], 1u);
            if (MODE == 1) acc += atomicAdd(&h[a], 1u);
            if (MODE == 2) acc += atomicOr(&h[a], 1u << (x & 31));
            if (MODE == 3) h[a] = x;
            if (MODE == 4) acc += h[a];
            if (MODE == 5) { unsigned o = atomicAnd(&h[a], ~(0xFFFFu << ((x & 1) * 16))); if (o & 0x10000) acc += atomicAdd(&h[(a * 7) % WORDS], 1u); }
        }
    }
    __syncthreads();
    if (acc == 0x12345678u) out[0] = acc + h[tid % WORDS];
}

template <int NT, int MODE, int PAT, int WORDS>
void run(const char *name, unsigned *out) {
    const unsigned iters = 2000;
    const int blocks = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NT, MODE, PAT, WORDS>), dim3(blocks), dim3(NT), 0, 0, iters, out);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NT, MODE, PAT, WORDS>), dim3(blocks), dim3(NT), 0, 0, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)iters * 8 * NT;   // per workgroup = per CU
    printf("%-34s NT %4d words %6d  %7.3f ms  %6.2f lane-ops/ns/CU  (%5.1f ns per wave-instruction)\n", name, NT, WORDS, ms,
           ops / (ms * 1e6), ms * 1e6 / (ops / 64));
    fflush(stdout);
}

int main() {
    unsigned *out;
    CK(hipMalloc(&out, 256));
    run<1024, 0, 0, 32768>("add      random 128K", out);
    run<1024, 1, 0, 32768>("add_rtn  random 128K", out);
    run<1024, 2, 0, 2048>("or_rtn   random 8K", out);
    run<1024, 0, 0, 256>("add      random 1K (256 ctrs)", out);
    run<1024, 1, 0, 256>("add_rtn  random 1K (256 ctrs)", out);
    run<1024, 0, 1, 32768>("add      conflict-free", out);
    run<1024, 1, 1, 32768>("add_rtn  conflict-free", out);
    run<1024, 0, 2, 32768>("add      one address", out);
    run<1024, 3, 0, 32768>("write    random 128K", out);
    run<1024, 4, 0, 32768>("read     random 128K", out);
    run<1024, 3, 1, 32768>("write    conflict-free", out);
    run<1024, 4, 1, 32768>("read     conflict-free", out);
    run<1024, 5, 0, 32768>("and_rtn + dep add_rtn 128K", out);
    run<256, 0, 0, 32768>("add      random 128K", out);
    run<256, 1, 0, 32768>("add_rtn  random 128K", out);
    run<512, 1, 0, 16384>("add_rtn  random 64K", out);
    return 0;
}
