// model.hip -- SURVEY.md section 8(f) row 4: the step after the path.  X stays in HBM as kmers_hip_build_x left it and
// the two sparse products of the reference's logp_grad (kmerexpr/utils.py:76-108) run on it:
//     xtheta = xnnz . theta                       (utils.py:92,  dot_product_mkl(xnnz, theta, out=scratch))
//     val    = sum ynnz * log(xtheta)             (utils.py:93,  _a_dot_logb)
//     w      = ynnz / xtheta                      (utils.py:102, _divide)
//     grad   = w . xnnz                           (utils.py:103, dot_product_mkl(yxTtheta, xnnz))
// with xnnz = x[y != 0] and ynnz = y[y != 0] / sum(y) built once per model, as multinomial_simplex_model.__init__
// does (kmerexpr/multinomial_simplex_model.py:56-62).  The (beta - 1) terms of utils.py:94-96,104-106 are O(T) host
// arithmetic on the caller's theta and stay with the caller (kmers_amd/multinomial_simplex_model.py).
//
// Layout.  xnnz is a CSR with 4-byte row pointers, 4-byte columns and float32 values (8 bytes per non-zero instead of
// the ABI's 12); a second, transposed copy (cptr / trow / tval) serves the transposed product, so both products are
// the same gather kernel and neither needs float atomics: results are deterministic.  The transposed copy is sorted
// by (row slab, column), 2 MB of w per slab, so that its gathers stay in L2 when there are millions of masked rows;
// it is made once per model with rocPRIM's stable radix sort (a set-up step, not the hot loop).
//
// k_spmv splits the NON-ZEROS evenly (2048 or 8192 per workgroup, 8 consecutive ones per thread), whatever the row lengths
// are (K=8: ~4000 per row; K=14: 1-2 per row, many empty rows): a thread walks its 8 products along the row
// pointers, writes rows that end inside its range, and hands the open row's partial sum to its successors through a
// segmented scan over the workgroup; a row that started in an earlier workgroup is finished by k_spmv_fixup from
// the per-tile carries.  Sums are accumulated in double and rounded to float32 once.
// Algorithmic bytes per non-zero: 8 (column + value) + the gathered vector (L2-resident up to a few MB).
#include "kmers_hip.h"
#include "kernels.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>  // before rocprim: one of its iterator headers calls memset without including this
#include <new>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

namespace {

using u32 = uint32_t;
using u64 = uint64_t;

// Threads per workgroup of k_spmv, chosen per product when the model is built: 1024 when the rows are long (>= 512
// non-zeros on average: K <= 9), else 256.  Same-run A/B: K=8 2.18 -> 2.04 ms per logp_grad with 1024, K=12 / K=14
// 2.95 -> 3.01 / 2.61 -> 2.94 ms (short rows: the walks diverge and a bigger workgroup waits longer at its barriers).
constexpr int SP_NT_SHORT = 256, SP_NT_LONG = 1024;
constexpr u32 SP_LONG_ROWS = 512;
#ifndef SP_IPT_
#define SP_IPT_ 8
#endif
constexpr int SP_IPT = SP_IPT_;         // consecutive non-zeros per thread (a multiple of 4: 16-byte loads)
constexpr u32 SP_TILE_MAX = SP_NT_LONG * SP_IPT;  // non-zeros per workgroup: threads x SP_IPT
#ifndef SP_PTR_LDS_
#define SP_PTR_LDS_ 4096
#endif
constexpr u32 SP_PTR_LDS = SP_PTR_LDS_;  // row pointers of one tile staged in LDS when they fit
constexpr u32 NO_ROW = 0xFFFFFFFFu;
#ifndef SP_XCDS_
#define SP_XCDS_ 8
#endif
constexpr u32 SP_XCDS = SP_XCDS_;       // XCDs of an MI355X (1 = plain blockIdx order, for A/B runs)
constexpr int TERM_BLOCKS = 2048;       // k_logp_terms: partial sums, added on the host in index order

#define MD_CHECK(expr)                                                                                              \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) {                                                                                     \
            fprintf(stderr, "libkmers: HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
            return KMERS_ERR_HIP;                                                                                   \
        }                                                                                                           \
    } while (0)

struct TileCarry {
    double tail;          // partial sum of the row still open at the end of the tile
    double head_partial;  // this tile's share of the row that started in an earlier tile and ends here
    u32 head_row;         // that row, or NO_ROW
    u32 flushed;          // 1 when some row ended inside the tile (the tail starts after it)
};

// last r in [lo, hi] with ptr[r] <= k   (ptr[lo] <= k is the caller's invariant)
template <class P>
__device__ __forceinline__ u32 last_row_le(const P &ptr, u32 lo, u32 hi, u32 k) {
    while (lo < hi) {
        const u32 mid = lo + (hi - lo + 1) / 2;
        if (ptr(mid) <= k) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// first / last row with a non-zero in every tile, found once per model (the row structure never changes)
__global__ __launch_bounds__(256) void k_tile_rows(const u32 *__restrict__ ptr, u32 n, u32 nnz, u32 n_tiles, u32 tile_nnz,
                                                   uint2 *__restrict__ tile_rows) {
    const u32 tile = blockIdx.x * 256 + threadIdx.x;
    if (tile >= n_tiles) return;
    auto gptr = [&](u32 r) { return ptr[r]; };
    const u32 base = tile * tile_nnz, tile_end = min(base + tile_nnz, nnz);
    const u32 lo = tile == 0 ? 0u : last_row_le(gptr, 0u, n - 1, base);  // tile 0 also owns the empty rows in front
    tile_rows[tile] = make_uint2(lo, last_row_le(gptr, lo, n - 1, tile_end - 1));
}

// out[r] = sum_{k in [ptr[r], ptr[r+1])} val[k] * vec[idx[k]]     for r < n;  ptr[n] == nnz > 0
// WHICH only names the instantiation in profiles: 0 = xnnz . theta, 1 = w . xnnz (on the column-sorted copy)
template <int WHICH, class OutT, int SP_NT>
__global__ __launch_bounds__(SP_NT) void k_spmv(const u32 *__restrict__ ptr, const u32 *__restrict__ idx,
                                                const float *__restrict__ val, const float *__restrict__ vec,
                                                const uint2 *__restrict__ tile_rows, u32 n, u32 nnz, u32 n_tiles,
                                                OutT *__restrict__ out, TileCarry *__restrict__ carry) {
    __shared__ u32 s_ptr[SP_PTR_LDS];
    constexpr u32 SP_TILE = SP_NT * SP_IPT;
    __shared__ double s_wave_a[SP_NT / 64];
    __shared__ u32 s_wave_f[SP_NT / 64];
    // Workgroups go to the 8 XCDs round-robin; give every XCD one contiguous eighth of the tiles, so that the part of
    // the gathered vector a run of tiles shares (a row slab of w: 2 MB) is fetched into one L2 instead of eight.
    const u32 t = threadIdx.x;
    // (The forward product gathers theta, which every L2 holds whole: plain order is 2 % faster there.)
    const u32 per_xcd = (n_tiles + SP_XCDS - 1) / SP_XCDS;
    const u32 tile = WHICH == 1 ? (blockIdx.x % SP_XCDS) * per_xcd + blockIdx.x / SP_XCDS : blockIdx.x;
    if (tile >= n_tiles) return;
    const u32 base = tile * SP_TILE;
    const u32 tile_end = min(base + SP_TILE, nnz);
    const u32 k0 = base + t * SP_IPT;
    const u32 k1 = min(k0 + SP_IPT, tile_end);
    // the streaming loads and the gathers depend on nothing but k0: in flight before the row structure is looked at
    u32 c[SP_IPT];
    float v[SP_IPT], x[SP_IPT];
    if (k0 + SP_IPT <= nnz) {
#pragma unroll
        for (int q = 0; q < SP_IPT / 4; ++q) {
            const uint4 cq = *reinterpret_cast<const uint4 *>(idx + k0 + 4 * q);
            const float4 vq = *reinterpret_cast<const float4 *>(val + k0 + 4 * q);
            c[4 * q] = cq.x, c[4 * q + 1] = cq.y, c[4 * q + 2] = cq.z, c[4 * q + 3] = cq.w;
            v[4 * q] = vq.x, v[4 * q + 1] = vq.y, v[4 * q + 2] = vq.z, v[4 * q + 3] = vq.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < SP_IPT; ++j) {
            const bool in = k0 + j < nnz;
            c[j] = in ? idx[k0 + j] : 0u;
            v[j] = in ? val[k0 + j] : 0.f;
        }
    }
#pragma unroll
#ifdef SP_NOGATHER
    for (int j = 0; j < SP_IPT; ++j) x[j] = 1.0f + (float)(c[j] & 1);
#else
    for (int j = 0; j < SP_IPT; ++j) x[j] = vec[c[j]];  // c is 0 (a valid index) past the end
#endif

    const uint2 tr = tile_rows[tile];
    const u32 r_lo = tr.x, r_hi = tr.y;
    const u32 n_stage = (r_hi - r_lo + 2 <= SP_PTR_LDS) ? r_hi - r_lo + 2 : 0u;  // ptr[r_lo .. r_hi + 1]
    for (u32 i = t; i < n_stage; i += SP_NT) s_ptr[i] = ptr[r_lo + i];
    __syncthreads();
    auto P = [&](u32 r) { return (r - r_lo) < n_stage ? s_ptr[r - r_lo] : ptr[r]; };

    double acc = 0.0, first_partial = 0.0;
    u32 first_row = NO_ROW, flushed = 0;
    if (k0 < tile_end) {
        u32 r = (tile == 0 && t == 0) ? 0u : last_row_le(P, r_lo, r_hi, k0);
        u32 end = P(r + 1);
        auto flush = [&]() {
            if (!flushed) {
                first_row = r;
                first_partial = acc;
                flushed = 1;
            } else {
                out[r] = (OutT)acc;
            }
            acc = 0.0;
        };
#pragma unroll
        for (int j = 0; j < SP_IPT; ++j) {
            const u32 k = k0 + j;
            if (k < k1) {
                while (k >= end) {  // k < nnz == ptr[n], so r + 1 <= n here
                    flush();
                    ++r;
                    end = P(r + 1);
                }
                acc += (double)v[j] * (double)x[j];
            }
        }
        // rows that end exactly at k1 (the open one, and the empty ones behind it) are this thread's to write
        while (end <= k1) {
            flush();
            ++r;
            if (r == n) break;
            end = P(r + 1);
        }
    }

    // segmented inclusive scan of (flushed, tail) over the workgroup: carry-in of a thread = tails of the
    // predecessors back to (and including) the nearest one that flushed
    const u32 lane = t & 63, wave = t >> 6;
    double a = acc;
    u32 f = flushed;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double oa = __shfl_up(a, d);
        const u32 of = __shfl_up(f, d);
        if (lane >= (u32)d) {
            if (!f) a += oa;
            f |= of;
        }
    }
    if (lane == 63) {
        s_wave_a[wave] = a;
        s_wave_f[wave] = f;
    }
    double ex_a = __shfl_up(a, 1);
    u32 ex_f = __shfl_up(f, 1);
    if (lane == 0) ex_a = 0.0, ex_f = 0;
    __syncthreads();
    double pre_a = 0.0;  // fold of the waves in front of this one
    u32 pre_f = 0;
    for (u32 w = 0; w < wave; ++w) {
        pre_a = s_wave_f[w] ? s_wave_a[w] : pre_a + s_wave_a[w];
        pre_f |= s_wave_f[w];
    }
    const double carry_in = ex_f ? ex_a : pre_a + ex_a;

    const bool head_open = ptr[r_lo] < base;  // row r_lo started in an earlier tile
    if (flushed) {
        const double vsum = first_partial + carry_in;
        if (head_open && first_row == r_lo) carry[tile].head_partial = vsum;
        else out[first_row] = (OutT)vsum;
    }
    if (t == SP_NT - 1) {
        // inclusive value of the last thread == fold over the whole tile
        const double tot_a = f ? a : pre_a + a;
        carry[tile].tail = tot_a;
        carry[tile].flushed = f | pre_f;
    }
    if (t == 0) carry[tile].head_row = (head_open && P(r_lo + 1) <= tile_end) ? r_lo : NO_ROW;
}

// rows that span tiles: add the tails of the tiles in front, nearest first, back to the tile the row started in
template <class OutT>
__global__ void k_spmv_fixup(const TileCarry *__restrict__ carry, u32 n_tiles, OutT *__restrict__ out) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles || i == 0) return;
    const u32 row = carry[i].head_row;
    if (row == NO_ROW) return;
    double s = carry[i].head_partial;
    for (u32 j = i; j-- > 0;) {
        s += carry[j].tail;
        if (carry[j].flushed) break;
    }
    out[row] = (OutT)s;
}

// grad[c] = sum over the row slabs of part[slab][c], in slab order
__global__ __launch_bounds__(256) void k_reduce_slabs(const double *__restrict__ part, u32 n_slabs, u32 n_cols,
                                                      float *__restrict__ grad) {
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols) return;
    double s = 0.0;
    for (u32 b = 0; b < n_slabs; ++b) s += part[(size_t)b * n_cols + c];
    grad[c] = (float)s;
}

// val partials and w = ynnz / xtheta (float64 / float32 -> float32, as utils.py:102 stores it).  ynnz[i] is formed
// here as ycnt[i] / sum(y) in float64 -- the value multinomial_simplex_model.py:62 stores -- from the 4-byte count.
__global__ __launch_bounds__(256) void k_logp_terms(const float *__restrict__ xtheta, const int32_t *__restrict__ ycnt,
                                                    double sum_y, u32 n, float *__restrict__ w,
                                                    double *__restrict__ partials) {
    __shared__ double s_part[4];
    double s = 0.0;
    const u32 n4 = n / 4;  // 16-byte loads and stores: the three arrays are hipMalloc'ed
    for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q < n4; q += (u64)gridDim.x * 256) {
        const float4 x4 = reinterpret_cast<const float4 *>(xtheta)[q];
        const int4 c4 = reinterpret_cast<const int4 *>(ycnt)[q];
        const double x0 = x4.x, x1 = x4.y, x2 = x4.z, x3 = x4.w;
        const double y0 = c4.x / sum_y, y1 = c4.y / sum_y, y2 = c4.z / sum_y, y3 = c4.w / sum_y;
        s += y0 * log(x0);
        s += y1 * log(x1);
        s += y2 * log(x2);
        s += y3 * log(x3);
        reinterpret_cast<float4 *>(w)[q] = make_float4((float)(y0 / x0), (float)(y1 / x1), (float)(y2 / x2), (float)(y3 / x3));
    }
    if (blockIdx.x == 0 && threadIdx.x < n - 4 * n4) {
        const u32 i = 4 * n4 + threadIdx.x;
        const double x = (double)xtheta[i], y = (double)ycnt[i] / sum_y;
        s += y * log(x);
        w[i] = (float)(y / x);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// ---- set-up kernels (once per model) ------------------------------------------------------------------------

struct NonZero {
    __device__ bool operator()(int32_t v) const { return v != 0; }
};

// stats[0] += sum of y   (the count of non-zeros comes from the selection)
__global__ __launch_bounds__(256) void k_y_sum(const int32_t *__restrict__ y, u64 n, unsigned long long *stats) {
    __shared__ long long s_part[4];
    long long s = 0;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) s += y[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(stats, (unsigned long long)(s_part[0] + s_part[1] + s_part[2] + s_part[3]));
}

// lens[i] = length of X's row rows[i]; ycnt[i] = y[rows[i]]; stats[1] += sum of lens
__global__ __launch_bounds__(256) void k_row_lens(const u32 *__restrict__ rows, u32 n_m, const u64 *__restrict__ row_ind,
                                                  const int32_t *__restrict__ y, u32 *__restrict__ lens,
                                                  int32_t *__restrict__ ycnt, unsigned long long *stats) {
    __shared__ unsigned long long s_part[4];
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    unsigned long long len = 0;
    if (i < n_m) {
        const u64 r = rows[i];
        len = row_ind[r + 1] - row_ind[r];
        lens[i] = (u32)len;  // the host rejects the model when the total does not fit 32 bits
        ycnt[i] = y[r];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) len += __shfl_down(len, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = len;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(stats + 1, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

// non-zero k of xnnz <- X: column narrowed to 32 bits; for the sort of the transposed copy the key
// (row slab, column) and the payload (masked row, value bits)
__global__ __launch_bounds__(256) void k_gather_rows(const u32 *__restrict__ xptr, const u32 *__restrict__ rows, u32 n_m,
                                                     u32 nnz, const u64 *__restrict__ row_ind, const u64 *__restrict__ col_ind,
                                                     const float *__restrict__ data, u64 first_col, u32 slab_shift, u32 n_cols,
                                                     u32 *__restrict__ xcol, float *__restrict__ xval,
                                                     u32 *__restrict__ keys, u64 *__restrict__ payload,
                                                     unsigned long long *stats) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nnz) return;
    auto P = [&](u32 r) { return xptr[r]; };
    const u32 i = last_row_le(P, 0u, n_m - 1, k);
    const u64 src = row_ind[rows[i]] + (k - xptr[i]);
    const float v = data[src];
    const u64 gcol = col_ind[src] - first_col;
    if (gcol >= n_cols) stats[2] = 1;  // a column outside [first_col, first_col + n_cols): the host rejects the model
    const u32 col = gcol < n_cols ? (u32)gcol : 0u;
    xcol[k] = col;
    xval[k] = v;
    keys[k] = (i >> slab_shift) * n_cols + col;
    payload[k] = ((u64)i << 32) | __float_as_uint(v);
}

__global__ __launch_bounds__(256) void k_unpack(const u64 *__restrict__ payload, u32 nnz, u32 *__restrict__ trow,
                                                float *__restrict__ tval) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nnz) return;
    const u64 p = payload[k];
    trow[k] = (u32)(p >> 32);
    tval[k] = __uint_as_float((u32)p);
}

// cptr[j] = first k with keys[k] >= j   (keys sorted ascending), j = 0 .. n_cols
__global__ __launch_bounds__(256) void k_col_ptr(const u32 *__restrict__ keys, u32 nnz, u32 n_cols, u32 *__restrict__ cptr) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j > n_cols) return;
    u32 lo = 0, hi = nnz;
    while (lo < hi) {
        const u32 mid = lo + (hi - lo) / 2;
        if (keys[mid] < j) lo = mid + 1;
        else hi = mid;
    }
    cptr[j] = lo;
}

struct Dev {  // a device allocation that frees itself
    void *p = nullptr;
    ~Dev() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
    void reset() {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
};

static inline u32 div_up(u64 a, u32 b) { return (u32)((a + b - 1) / b); }

}  // namespace

struct kmers_model {
    int device = 0;
    u64 n_rows = 0;
    u32 n_cols = 0, n_m = 0, nnz = 0;
    u32 x_nt = SP_NT_SHORT, t_nt = SP_NT_SHORT, x_tiles = 0, t_tiles = 0;  // workgroup size and tile count per product
    u32 slab_shift = 31, n_slabs = 1, n_trows = 0;  // transposed copy: n_slabs row slabs x n_cols columns = n_trows rows
    long long sum_y = 0;
    u32 *rows = nullptr, *xptr = nullptr, *xcol = nullptr, *cptr = nullptr, *trow = nullptr;
    float *xval = nullptr, *tval = nullptr, *xtheta = nullptr, *w = nullptr;
    int32_t *ycnt = nullptr;  // y[ymask]; ynnz = ycnt / sum_y
    double *partials = nullptr, *h_partials = nullptr;
    TileCarry *carry = nullptr;
    double *gpart = nullptr;  // n_slabs > 1: per-slab partial gradients
    uint2 *xtiles = nullptr, *ttiles = nullptr;  // k_tile_rows of xnnz and of its column-sorted copy
};

extern "C" void kmers_hip_model_free(kmers_model *m) {
    if (!m) return;
    for (void *p : {(void *)m->rows, (void *)m->xptr, (void *)m->xcol, (void *)m->cptr, (void *)m->trow, (void *)m->xval,
                    (void *)m->tval, (void *)m->xtheta, (void *)m->w, (void *)m->ycnt, (void *)m->partials, (void *)m->carry,
                    (void *)m->xtiles, (void *)m->ttiles, (void *)m->gpart})
        if (p) (void)hipFree(p);
    if (m->h_partials) (void)hipHostFree(m->h_partials);
    delete m;
}

extern "C" uint64_t kmers_hip_model_n_masked(const kmers_model *m) { return m ? m->n_m : 0; }
extern "C" uint64_t kmers_hip_model_nnz(const kmers_model *m) { return m ? m->nnz : 0; }
extern "C" int64_t kmers_hip_model_sum_y(const kmers_model *m) { return m ? m->sum_y : 0; }
extern "C" const uint32_t *kmers_hip_model_rows(const kmers_model *m) { return m ? m->rows : nullptr; }
extern "C" const float *kmers_hip_model_xtheta(const kmers_model *m) { return m ? m->xtheta : nullptr; }

extern "C" int kmers_hip_model_read(const kmers_model *m, int what, void *host_dst) {
    if (!m || !host_dst || what < 0 || what > 1) return KMERS_ERR_BAD_ARG;
    MD_CHECK(hipMemcpy(host_dst, what == 0 ? (const void *)m->rows : (const void *)m->xtheta, (size_t)m->n_m * 4,
                       hipMemcpyDeviceToHost));
    return KMERS_OK;
}

extern "C" int kmers_hip_model_create_shard(const float *d_data, const uint64_t *d_col_ind, const uint64_t *d_row_ind,
                                            uint64_t n_rows, uint64_t first_col, uint64_t n_cols, const int32_t *d_y,
                                            void *stream, kmers_model **out) {
    if (!out) return KMERS_ERR_BAD_ARG;
    *out = nullptr;
    if (!d_row_ind || !d_y || n_rows == 0 || n_rows > (1ull << 32) || n_cols == 0 || n_cols >= 0xFFFFFFFFull)
        return KMERS_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    kmers_model *m = new (std::nothrow) kmers_model;
    if (!m) return KMERS_ERR_NOMEM;
    struct Guard {
        kmers_model *m;
        ~Guard() { kmers_hip_model_free(m); }
    } guard{m};
    MD_CHECK(hipGetDevice(&m->device));
    m->n_rows = n_rows;
    m->n_cols = (u32)n_cols;

    // 1. sum(y), the masked rows
    Dev stats, count, tmp;
    MD_CHECK(stats.alloc(24));
    MD_CHECK(count.alloc(8));
    MD_CHECK(hipMemsetAsync(stats.p, 0, 24, s));
    hipLaunchKernelGGL(k_y_sum, dim3(2048), dim3(256), 0, s, d_y, (u64)n_rows, stats.as<unsigned long long>());
    MD_CHECK(hipGetLastError());
    Dev all_rows;  // selection output is sized for the worst case, then trimmed
    MD_CHECK(all_rows.alloc((size_t)n_rows * 4));
    auto flags = rocprim::make_transform_iterator(d_y, NonZero());
    rocprim::counting_iterator<u32> ids(0);
    size_t tmp_bytes = 0;
    MD_CHECK(rocprim::select(nullptr, tmp_bytes, ids, flags, all_rows.as<u32>(), count.as<size_t>(), (size_t)n_rows, s));
    MD_CHECK(tmp.alloc(tmp_bytes));
    MD_CHECK(rocprim::select(tmp.p, tmp_bytes, ids, flags, all_rows.as<u32>(), count.as<size_t>(), (size_t)n_rows, s));
    size_t n_sel = 0;
    unsigned long long h_stats[3] = {0, 0, 0};
    MD_CHECK(hipMemcpyAsync(&n_sel, count.p, sizeof n_sel, hipMemcpyDeviceToHost, s));
    MD_CHECK(hipMemcpyAsync(h_stats, stats.p, 8, hipMemcpyDeviceToHost, s));
    MD_CHECK(hipStreamSynchronize(s));
    m->sum_y = (long long)h_stats[0];
    if (n_sel == 0 || n_sel >= 0xFFFFFFFFull) return n_sel ? KMERS_ERR_TOO_LARGE : KMERS_ERR_BAD_ARG;  // y all zero
    m->n_m = (u32)n_sel;
    MD_CHECK(hipMalloc((void **)&m->rows, (size_t)m->n_m * 4));
    MD_CHECK(hipMemcpyAsync(m->rows, all_rows.p, (size_t)m->n_m * 4, hipMemcpyDeviceToDevice, s));

    // 2. row pointers of xnnz, ynnz
    MD_CHECK(hipMalloc((void **)&m->xptr, ((size_t)m->n_m + 1) * 4));
    MD_CHECK(hipMalloc((void **)&m->ycnt, (size_t)m->n_m * 4));
    MD_CHECK(hipMemsetAsync(m->xptr + m->n_m, 0, 4, s));
    hipLaunchKernelGGL(k_row_lens, dim3(div_up(m->n_m, 256)), dim3(256), 0, s, (const u32 *)m->rows, m->n_m, d_row_ind, d_y,
                       m->xptr, m->ycnt, stats.as<unsigned long long>());
    MD_CHECK(hipGetLastError());
    MD_CHECK(hipMemcpyAsync(h_stats, stats.p, 16, hipMemcpyDeviceToHost, s));
    MD_CHECK(hipStreamSynchronize(s));
    all_rows.reset();
    if (h_stats[1] >= 0xFFFFFFFFull - SP_TILE_MAX) return KMERS_ERR_TOO_LARGE;
    m->nnz = (u32)h_stats[1];
    m->x_nt = (m->nnz / m->n_m >= SP_LONG_ROWS) ? SP_NT_LONG : SP_NT_SHORT;
    m->x_tiles = div_up(m->nnz, m->x_nt * SP_IPT);
    {
        size_t b = 0;
        Dev t2;
        MD_CHECK(rocprim::exclusive_scan(nullptr, b, m->xptr, m->xptr, 0u, (size_t)m->n_m + 1, rocprim::plus<u32>(), s));
        MD_CHECK(t2.alloc(b));
        MD_CHECK(rocprim::exclusive_scan(t2.p, b, m->xptr, m->xptr, 0u, (size_t)m->n_m + 1, rocprim::plus<u32>(), s));
        MD_CHECK(hipStreamSynchronize(s));
    }

    // 3. buffers of the hot loop
    MD_CHECK(hipMalloc((void **)&m->xtheta, (size_t)m->n_m * 4));
    MD_CHECK(hipMalloc((void **)&m->w, (size_t)m->n_m * 4));
    MD_CHECK(hipMalloc((void **)&m->partials, TERM_BLOCKS * 8));
    MD_CHECK(hipHostMalloc((void **)&m->h_partials, TERM_BLOCKS * 8, hipHostMallocDefault));
    MD_CHECK(hipMalloc((void **)&m->carry, ((size_t)div_up(m->nnz, SP_NT_SHORT * SP_IPT) + 1) * sizeof(TileCarry)));
    // Row slabs of the transposed copy: the transposed product gathers w[row]; with more masked rows than an L2 holds
    // (K >= 11 or so) every gather would be an HBM access of its own (measured at K=14: 128 bytes fetched per
    // non-zero, 5.1 ms).  Sorting by (row slab, column) with 2 MB of w per slab keeps the gathers of one tile in L2;
    // the product then yields one partial gradient per slab, added up in slab order by k_reduce_slabs.
    {
        u32 shift = 19;  // 2^19 rows = 2 MB of w: half of one XCD's L2
        if (const char *e = getenv("KMERS_MODEL_SLAB_LOG2")) shift = (u32)atoi(e);
        if (shift < 8) shift = 8;
        if (shift > 31) shift = 31;
        while (shift < 31 && (u64)div_up(m->n_m, 1u << shift) * m->n_cols >= 0xFFFFFFF0ull) ++shift;
        m->slab_shift = shift;
        m->n_slabs = div_up(m->n_m, 1u << shift);
        if ((u64)m->n_slabs * m->n_cols >= 0xFFFFFFF0ull) return KMERS_ERR_TOO_LARGE;
        m->n_trows = m->n_slabs * m->n_cols;
    }
    if (m->n_slabs > 1) MD_CHECK(hipMalloc((void **)&m->gpart, (size_t)m->n_trows * 8));
    MD_CHECK(hipMalloc((void **)&m->cptr, ((size_t)m->n_trows + 1) * 4));
    const size_t nz = m->nnz ? m->nnz : 1;
    MD_CHECK(hipMalloc((void **)&m->xcol, nz * 4 + 64));
    MD_CHECK(hipMalloc((void **)&m->xval, nz * 4 + 64));
    MD_CHECK(hipMalloc((void **)&m->trow, nz * 4 + 64));
    MD_CHECK(hipMalloc((void **)&m->tval, nz * 4 + 64));
    if (m->nnz == 0) {
        MD_CHECK(hipMemsetAsync(m->cptr, 0, ((size_t)m->n_trows + 1) * 4, s));
        MD_CHECK(hipStreamSynchronize(s));
        guard.m = nullptr;
        *out = m;
        return KMERS_OK;
    }
    if (!d_data || !d_col_ind) return KMERS_ERR_BAD_ARG;

    // 4. xnnz = x[ymask], and its copy sorted by column (stable: rows ascend inside a column)
    Dev pay_in, pay_out, keys_in, keys_out, t3;
    MD_CHECK(pay_in.alloc(nz * 8));
    MD_CHECK(pay_out.alloc(nz * 8));
    MD_CHECK(keys_in.alloc(nz * 4));
    MD_CHECK(keys_out.alloc(nz * 4));
    hipLaunchKernelGGL(k_gather_rows, dim3(div_up(m->nnz, 256)), dim3(256), 0, s, (const u32 *)m->xptr, (const u32 *)m->rows,
                       m->n_m, m->nnz, d_row_ind, d_col_ind, d_data, (u64)first_col, m->slab_shift, m->n_cols, m->xcol, m->xval, keys_in.as<u32>(),
                       pay_in.as<u64>(), stats.as<unsigned long long>());
    MD_CHECK(hipGetLastError());
    unsigned bits = 1;
    while (bits < 32 && (1ull << bits) < m->n_trows) ++bits;
    size_t sb = 0;
    MD_CHECK(rocprim::radix_sort_pairs(nullptr, sb, (const u32 *)keys_in.as<u32>(), keys_out.as<u32>(), (const u64 *)pay_in.as<u64>(),
                                       pay_out.as<u64>(), (size_t)m->nnz, 0u, bits, s));
    MD_CHECK(t3.alloc(sb));
    MD_CHECK(rocprim::radix_sort_pairs(t3.p, sb, (const u32 *)keys_in.as<u32>(), keys_out.as<u32>(), (const u64 *)pay_in.as<u64>(),
                                       pay_out.as<u64>(), (size_t)m->nnz, 0u, bits, s));
    hipLaunchKernelGGL(k_unpack, dim3(div_up(m->nnz, 256)), dim3(256), 0, s, (const u64 *)pay_out.as<u64>(), m->nnz, m->trow,
                       m->tval);
    hipLaunchKernelGGL(k_col_ptr, dim3(div_up((u64)m->n_trows + 1, 256)), dim3(256), 0, s, (const u32 *)keys_out.as<u32>(),
                       m->nnz, m->n_trows, m->cptr);
    m->t_nt = (m->nnz / m->n_trows >= SP_LONG_ROWS) ? SP_NT_LONG : SP_NT_SHORT;
    m->t_tiles = div_up(m->nnz, m->t_nt * SP_IPT);
    MD_CHECK(hipMalloc((void **)&m->xtiles, (size_t)m->x_tiles * sizeof(uint2)));
    MD_CHECK(hipMalloc((void **)&m->ttiles, (size_t)m->t_tiles * sizeof(uint2)));
    hipLaunchKernelGGL(k_tile_rows, dim3(div_up(m->x_tiles, 256)), dim3(256), 0, s, (const u32 *)m->xptr, m->n_m, m->nnz,
                       m->x_tiles, m->x_nt * SP_IPT, m->xtiles);
    hipLaunchKernelGGL(k_tile_rows, dim3(div_up(m->t_tiles, 256)), dim3(256), 0, s, (const u32 *)m->cptr, m->n_trows, m->nnz,
                       m->t_tiles, m->t_nt * SP_IPT, m->ttiles);
    MD_CHECK(hipGetLastError());
    MD_CHECK(hipMemcpyAsync(h_stats, stats.p, 24, hipMemcpyDeviceToHost, s));
    MD_CHECK(hipStreamSynchronize(s));
    if (h_stats[2]) return KMERS_ERR_BAD_ARG;  // a column outside this model's range: the products would gather past theta
    guard.m = nullptr;
    *out = m;
    return KMERS_OK;
}

// Event pairs around the kernels of one logp_grad while kmers_hip_profile_enable is on (read after the sync).
struct Marks {
    hipStream_t s;
    bool on;
    int n = 0;
    hipEvent_t ev[12];
    const char *label[12];
    explicit Marks(hipStream_t s_, bool enable = true) : s(s_), on(enable && kk::profile_on()) { mark(nullptr); }
    void mark(const char *name) {
        if (!on || n == 12) return;
        if (hipEventCreate(&ev[n]) != hipSuccess) {
            on = false;
            return;
        }
        (void)hipEventRecord(ev[n], s);
        label[n++] = name;
    }
    ~Marks() {  // the stream has been synchronised by then
        for (int i = 1; i < n; ++i) {
            float ms = 0.f;
            if (on && label[i] && hipEventElapsedTime(&ms, ev[i - 1], ev[i]) == hipSuccess) kk::profile_add(label[i], ms);
        }
        for (int i = 0; i < n; ++i) (void)hipEventDestroy(ev[i]);
    }
};

template <int WHICH, class OutT>
static int spmv(const kmers_model *m, const u32 *ptr, const u32 *idx, const float *val, const float *vec,
                const uint2 *tiles, u32 n, OutT *out, hipStream_t s, Marks &marks, const char *label) {
    if (m->nnz == 0) {
        MD_CHECK(hipMemsetAsync(out, 0, (size_t)n * sizeof(OutT), s));
        return KMERS_OK;
    }
    const u32 nt = WHICH == 0 ? m->x_nt : m->t_nt, n_tiles = WHICH == 0 ? m->x_tiles : m->t_tiles;
    const dim3 grid(div_up(n_tiles, SP_XCDS) * SP_XCDS);
    if (nt == SP_NT_LONG)
        hipLaunchKernelGGL((k_spmv<WHICH, OutT, SP_NT_LONG>), grid, dim3(SP_NT_LONG), 0, s, ptr, idx, val, vec, tiles, n, m->nnz,
                           n_tiles, out, m->carry);
    else
        hipLaunchKernelGGL((k_spmv<WHICH, OutT, SP_NT_SHORT>), grid, dim3(SP_NT_SHORT), 0, s, ptr, idx, val, vec, tiles, n, m->nnz,
                           n_tiles, out, m->carry);
    marks.mark(label);
    hipLaunchKernelGGL(k_spmv_fixup<OutT>, dim3(div_up(n_tiles, 256)), dim3(256), 0, s, (const TileCarry *)m->carry, n_tiles,
                       out);
    marks.mark("k_spmv_fixup");
    MD_CHECK(hipGetLastError());
    return KMERS_OK;
}

extern "C" int kmers_hip_model_create(const float *d_data, const uint64_t *d_col_ind, const uint64_t *d_row_ind,
                                      uint64_t n_rows, uint64_t n_cols, const int32_t *d_y, void *stream,
                                      kmers_model **out) {
    return kmers_hip_model_create_shard(d_data, d_col_ind, d_row_ind, n_rows, 0, n_cols, d_y, stream, out);
}

// the model's memory lives on the device that was current when it was made
static bool on_models_device(const kmers_model *m) {
    int dev = -1;
    return hipGetDevice(&dev) == hipSuccess && dev == m->device;
}

// phase 1: xtheta = xnnz . theta (asynchronous)
static int phase_xtheta(kmers_model *m, const float *d_theta, float *d_xtheta, hipStream_t s, Marks &marks) {
    return spmv<0>(m, m->xptr, m->xcol, m->xval, d_theta, m->xtiles, m->n_m, d_xtheta, s, marks, "k_spmv<x.theta>");
}

// phase 2: val, w and the gradient from a given xtheta; synchronises
static int phase_grad(kmers_model *m, const float *d_xtheta, int nograd, double *val_out, float *d_grad, hipStream_t s,
                      Marks &marks) {
    hipLaunchKernelGGL(k_logp_terms, dim3(TERM_BLOCKS), dim3(256), 0, s, d_xtheta, (const int32_t *)m->ycnt, (double)m->sum_y, m->n_m,
                       m->w, m->partials);
    marks.mark("k_logp_terms");
    MD_CHECK(hipGetLastError());
    MD_CHECK(hipMemcpyAsync(m->h_partials, m->partials, TERM_BLOCKS * 8, hipMemcpyDeviceToHost, s));
    if (!nograd) {
        marks.mark(nullptr);  // the copy is not a kernel
        int rc;
        if (m->n_slabs == 1) {
            rc = spmv<1>(m, m->cptr, m->trow, m->tval, m->w, m->ttiles, m->n_cols, d_grad, s, marks, "k_spmv<w.x>");
            if (rc) return rc;
        } else {
            rc = spmv<1>(m, m->cptr, m->trow, m->tval, m->w, m->ttiles, m->n_trows, m->gpart, s, marks, "k_spmv<w.x>");
            if (rc) return rc;
            hipLaunchKernelGGL(k_reduce_slabs, dim3(div_up(m->n_cols, 256)), dim3(256), 0, s, (const double *)m->gpart,
                               m->n_slabs, m->n_cols, d_grad);
            marks.mark("k_reduce_slabs");
            MD_CHECK(hipGetLastError());
        }
    }
    MD_CHECK(hipStreamSynchronize(s));
    double v = 0.0;
    for (int i = 0; i < TERM_BLOCKS; ++i) v += m->h_partials[i];
    *val_out = v;
    return KMERS_OK;
}

extern "C" int kmers_hip_model_logp_grad(kmers_model *m, const float *d_theta, int nograd, double *val_out, float *d_grad,
                                         void *stream) {
    if (!m || !d_theta || !val_out || (!nograd && !d_grad) || !on_models_device(m)) return KMERS_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    Marks marks(s);
    const int rc = phase_xtheta(m, d_theta, m->xtheta, s, marks);
    if (rc) return rc;
    return phase_grad(m, m->xtheta, nograd, val_out, d_grad, s, marks);
}

extern "C" int kmers_hip_model_xtheta_into(kmers_model *m, const float *d_theta, float *d_xtheta, void *stream) {
    if (!m || !d_theta || !d_xtheta || !on_models_device(m)) return KMERS_ERR_BAD_ARG;
    Marks marks((hipStream_t)stream, false);  // asynchronous: nothing to read the events after
    return phase_xtheta(m, d_theta, d_xtheta, (hipStream_t)stream, marks);
}

extern "C" int kmers_hip_model_grad_from(kmers_model *m, const float *d_xtheta, int nograd, double *val_out, float *d_grad,
                                         void *stream) {
    if (!m || !d_xtheta || ((uintptr_t)d_xtheta & 15) || !val_out || (!nograd && !d_grad) || !on_models_device(m))
        return KMERS_ERR_BAD_ARG;
    Marks marks((hipStream_t)stream);
    return phase_grad(m, d_xtheta, nograd, val_out, d_grad, (hipStream_t)stream, marks);
}
