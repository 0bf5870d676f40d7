// Microbenchmark: the store pattern of one radix pass without any of its logic.  512 persistent workgroups, each walks
// `per` tiles; a tile writes RUN consecutive 8-byte entries into each of 256 streams (stream s of workgroup b is the
// contiguous region [ (s*nblk + b) * per*RUN , ... ) of the output), optionally reading a coalesced 64 KB tile first.
// hipcc --offload-arch=gfx950 -O3 tools/mb_scatter.hip -o /tmp/mb_scatter && /tmp/mb_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int RUN, bool READ, int SKEW>
__global__ __launch_bounds__(512) void k(const u64 *in, u64 *out, int per, int nblk) {
    const int tid = threadIdx.x, b = blockIdx.x;
    constexpr int TILE = RUN * 256;
    for (int t = 0; t < per; ++t) {
        u64 acc = 0;
        if (READ) {
            const u64 *src = in + ((size_t)b * per + t) * TILE;
            for (int i = tid; i < TILE; i += 512) acc += src[i];
        }
        for (int i = tid; i < TILE; i += 512) {
            const int s = i / RUN, r = i % RUN;
            const size_t pos = ((size_t)s * nblk + b) * ((size_t)per * RUN + SKEW) + (size_t)t * RUN + r;
            out[pos] = acc + i;
        }
    }
}

// Same bytes, but every wave store instruction covers one 512-byte-aligned window of the output: the lanes that fall
// outside the run are masked off (partial lines only at the two ends of a run).
template <int RUN, int SKEW>
__global__ __launch_bounds__(512) void k_al(u64 *out, int per, int nblk) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, b = blockIdx.x;
    for (int t = 0; t < per; ++t) {
        for (int s = w; s < 256; s += 8) {       // a wave writes whole runs
            const size_t g0 = ((size_t)s * nblk + b) * ((size_t)per * RUN + SKEW) + (size_t)t * RUN, g1 = g0 + RUN;
            for (size_t c = g0 & ~(size_t)63; c < g1; c += 64) {
                const size_t pos = c + lane;
                if (pos >= g0 && pos < g1) out[pos] = pos;
            }
        }
    }
}
template <int RUN, int SKEW>
void run_al(const char *name, u64 *out, size_t n) {
    const int nblk = 512;
    const int per = (int)(n / ((size_t)nblk * RUN * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_al<RUN, SKEW>), dim3(nblk), dim3(512), 0, 0, out, per, nblk);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_al<RUN, SKEW>), dim3(nblk), dim3(512), 0, 0, out, per, nblk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)per * nblk * RUN * 256 * 8;
    printf("%-40s run=%4d B  %7.3f ms  write %6.2f TB/s\n", name, RUN * 8, ms, bytes / ms / 1e9);
}

template <int RUN, bool READ, int SKEW>
void run(const char *name, u64 *in, u64 *out, size_t n) {
    const int nblk = 512;
    const int per = (int)(n / ((size_t)nblk * RUN * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<RUN, READ, SKEW>), dim3(nblk), dim3(512), 0, 0, in, out, per, nblk);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<RUN, READ, SKEW>), dim3(nblk), dim3(512), 0, 0, in, out, per, nblk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double bytes = (double)per * nblk * RUN * 256 * 8;
    printf("%-40s run=%4d B  %7.3f ms  write %6.2f TB/s%s\n", name, RUN * 8, ms, bytes / ms / 1e9, READ ? " (+ equal read)" : "");
}

int main() {
    const size_t n = 400u * 1000 * 1000;
    u64 *in, *out;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, (n + (1u << 24)) * 8));
    CK(hipMemset(in, 1, n * 8));
    run<32, false, 0>("256 streams/wg, aligned", in, out, n);
    run<32, false, 5>("256 streams/wg, misaligned (skew 40 B)", in, out, n);
    run<64, false, 5>("256 streams/wg, misaligned", in, out, n);
    run<128, false, 5>("256 streams/wg, misaligned", in, out, n);
    run<32, false, 8>("256 streams/wg, skew 64 B", in, out, n);
    run<32, false, 16>("256 streams/wg, skew 128 B", in, out, n);
    run<32, false, 4>("256 streams/wg, skew 32 B", in, out, n);
    run_al<32, 5>("window-aligned instr, skew 40 B", out, n);
    run_al<64, 5>("window-aligned instr, skew 40 B", out, n);
    run_al<128, 5>("window-aligned instr, skew 40 B", out, n);
    run_al<32, 0>("window-aligned instr, aligned", out, n);
    run<16, false, 0>("256 streams/wg, aligned", in, out, n);
    run<16, true, 0>("read + 256 streams/wg, aligned", in, out, n);
    run<32, true, 0>("read + 256 streams/wg, aligned", in, out, n);
    run<32, true, 5>("read + 256 streams/wg, misaligned", in, out, n);
    run<64, true, 5>("read + 256 streams/wg, misaligned", in, out, n);
    return 0;
}
