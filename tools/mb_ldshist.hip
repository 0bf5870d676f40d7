// Microbenchmark: one workgroup per contiguous slice of a u32 array, every entry -> one LDS atomic add at (entry & mask).
// Separates the cost of streaming the entries from the cost of the LDS atomics in k_count_y_sorted.
// hipcc --offload-arch=gfx950 -O3 tools/mb_ldshist.hip -o tools/mb_ldshist.bin && tools/mb_ldshist.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_fill(unsigned *p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u;
        x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
        p[i] = x;
    }
}

// MODE 0: read only; 1: LDS atomic add; 2: LDS plain read-modify-write (racy: rate only); 3: global atomic into a per-block slice
template <int NT, int LOG2_BINS, int MODE, int U>
__global__ __launch_bounds__(NT) void k(const unsigned *ent, size_t per_block, unsigned *out, unsigned *gbins) {
    constexpr unsigned NB = 1u << LOG2_BINS;
    __shared__ unsigned h[NB];
    const unsigned tid = threadIdx.x;
    for (unsigned i = tid; i < NB; i += NT) h[i] = 0;
    __syncthreads();
    const uint4 *e4 = reinterpret_cast<const uint4 *>(ent + blockIdx.x * per_block);
    const unsigned q1 = (unsigned)(per_block / 4);
    unsigned acc = 0;
    unsigned *gb = gbins + (size_t)blockIdx.x * NB;
    auto add = [&](unsigned v) {
        const unsigned b = v & (NB - 1);
        if (MODE == 0) acc += b;
        if (MODE == 1) atomicAdd(&h[b], 1u);
        if (MODE == 2) h[b] += 1u;
        if (MODE == 3) atomicAdd(&gb[b], 1u);
    };
    for (unsigned q0 = tid; q0 < q1; q0 += NT * U) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (q0 + u * NT < q1) ? e4[q0 + u * NT] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) { add(v[u].x); add(v[u].y); add(v[u].z); add(v[u].w); }
    }
    __syncthreads();
    for (unsigned i = tid; i < NB; i += NT) acc += h[i];
    if (acc == 0x12345678u) out[0] = acc;
}

template <int NT, int LOG2_BINS, int MODE, int U>
void run(const char *name, const unsigned *ent, size_t n, int blocks, unsigned *out, unsigned *gbins) {
    const size_t per = (n / blocks) & ~(size_t)3;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<NT, LOG2_BINS, MODE, U>), dim3(blocks), dim3(NT), 0, 0, ent, per, out, gbins);
    CK(hipEventRecord(e0));
    const int reps = 3;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<NT, LOG2_BINS, MODE, U>), dim3(blocks), dim3(NT), 0, 0, ent, per, out, gbins);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-28s NT %4d bins 2^%2d blocks %5d unroll %d  %7.3f ms  %6.2f G entries/s  %5.2f TB/s\n", name, NT, LOG2_BINS, blocks, U, ms,
           (double)per * blocks / ms / 1e6, (double)per * blocks * 4 / ms / 1e9);
    fflush(stdout);
}

int main() {
    const size_t n = (size_t)1 << 30;
    unsigned *ent, *out, *gbins;
    CK(hipMalloc(&ent, n * 4)); CK(hipMalloc(&out, 256)); CK(hipMalloc(&gbins, (size_t)8192 * 32768 * 4));
    CK(hipMemset(gbins, 0, (size_t)8192 * 32768 * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, ent, n);
    CK(hipDeviceSynchronize());
    run<1024, 15, 0, 4>("read only", ent, n, 256, out, gbins);
    run<1024, 15, 0, 4>("read only", ent, n, 8192, out, gbins);
    run<1024, 15, 1, 4>("lds atomic", ent, n, 256, out, gbins);
    run<1024, 15, 1, 4>("lds atomic", ent, n, 8192, out, gbins);
    run<1024, 10, 1, 4>("lds atomic", ent, n, 256, out, gbins);
    run<1024, 10, 1, 4>("lds atomic", ent, n, 2048, out, gbins);
    run<512, 10, 1, 4>("lds atomic", ent, n, 2048, out, gbins);
    run<256, 10, 1, 4>("lds atomic", ent, n, 4096, out, gbins);
    run<1024, 15, 2, 4>("lds plain rmw", ent, n, 256, out, gbins);
    run<1024, 10, 2, 4>("lds plain rmw", ent, n, 2048, out, gbins);
    run<1024, 13, 1, 4>("lds atomic", ent, n, 1024, out, gbins);
    run<512, 13, 1, 4>("lds atomic", ent, n, 2048, out, gbins);
    run<1024, 15, 3, 4>("global atomic, block slice", ent, n, 256, out, gbins);
    run<1024, 12, 3, 4>("global atomic, block slice", ent, n, 2048, out, gbins);
    return 0;
}
