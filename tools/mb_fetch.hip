// tools/mb_fetch.hip -- what does TCC_EA0_RDREQ (and FETCH_SIZE = RDREQ x 64 B) count per byte of a streaming read?
// Four kernels read the same 1 GiB once: 4-byte and 16-byte loads per lane, from a 128-byte-aligned base and from a base
// that is off by 16 bytes (a wave's kilobyte then straddles nine lines), plus the access shape of ys_count_range
// (1024-thread workgroups, four 16 KB pieces in flight, every workgroup its own contiguous range of arbitrary start).
//   hipcc --offload-arch=gfx950 -O3 tools/mb_fetch.hip -o tools/mb_fetch.bin
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d out -- tools/mb_fetch.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void rd4(const uint32_t *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void rd16(const uint4 *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
// every workgroup reads its own contiguous range [b * per + skew_b, ...) in pieces of 4 x 16 KB (ys_count_range's shape)
__global__ __launch_bounds__(1024) void rd_ranges(const uint4 *p, size_t per, uint32_t *sink) {
    const size_t begin = (size_t)blockIdx.x * per + (blockIdx.x * 37u) % 61u, end = (size_t)(blockIdx.x + 1) * per;
    uint32_t acc = 0;
    for (size_t qb = begin; qb < end; qb += 4096) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t q = qb + u * 1024 + threadIdx.x;
            v[u] = q < end ? p[q] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char *buf;
    uint32_t *sink;
    if (hipMalloc(&buf, bytes + 4096) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes + 4096);
    hipDeviceSynchronize();
    rd4<<<4096, 256>>>((const uint32_t *)buf, bytes / 4, sink);
    rd4<<<4096, 256>>>((const uint32_t *)(buf + 16), bytes / 4, sink);
    rd16<<<4096, 256>>>((const uint4 *)buf, bytes / 16, sink);
    rd16<<<4096, 256>>>((const uint4 *)(buf + 16), bytes / 16, sink);
    rd_ranges<<<512, 1024>>>((const uint4 *)buf, bytes / 16 / 512, sink);
    hipDeviceSynchronize();
    printf("read 1 GiB five times: rd4 aligned, rd4 +16, rd16 aligned, rd16 +16, rd_ranges\n");
    return 0;
}
