// kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels for the k-mer shredding / counting hot path.
//
// What they replace in the reference (src/libkmers.cpp):
//   valid_kmer + kmer_to_id per window (:222-227, :321-326, :331-345)  -> window_at(): one 64-bit bit-field
//                                                                         extract from the 2-bit packed store
//   count_kmers, atomic ++ into a shared int array (:213-216, :320-328) -> k_count_y*
//   collect_kmers (unordered_map per isoform) + fill_indices_coo (:218-238)
//   + ThreadPoolLocalData::to_csr (sort by row, gather, sort by col, :77-169)
//        -> "shred, stable radix partition by k-mer id, run-length reduce":
//           k_hist_bases / k_scatter (pass 0, straight from the packed bases), k_hist_keys / k_scatter
//           (further 8-bit digits), k_rle_count / k_rle_emit (one CSR entry per run of equal (k-mer, column),
//           value float(run length) / float(valid windows of the column), row pointer by gap filling).
//
// Why a sort and not a hash: every window instance becomes an 8-byte (k-mer id, column) entry in position
// order, i.e. column-ascending.  A STABLE least-significant-digit partition by k-mer id therefore leaves,
// inside every row, the columns ascending and the instances of one (row, column) adjacent -- exactly the CSR
// order the reference produces, with no per-isoform hash table, no long-isoform special case and a memory
// access pattern that is fully coalesced (HBM-bound, no MFMA: nothing here is a contraction).
//
// Geometry: 512-thread workgroups (8 wave64), a tile is 8192 entries.  Wave w of a tile owns the contiguous
// chunk [w*1024, (w+1)*1024) and walks it in 16 rounds of 64 consecutive entries, so the order
// (wave, round, lane) is the entry order; ranks inside a digit come from a wave-wide ballot match, which is
// deterministic (no atomics decide an order anywhere).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace kk {

constexpr int NT = 512;          // threads per workgroup
constexpr int NW = NT / 64;      // waves per workgroup
constexpr int IPT = 16;          // rounds per wave
constexpr int TILE = NT * IPT;   // entries per tile (8192)
constexpr int WCHUNK = 64 * IPT; // entries per wave (1024)
constexpr int RADIX = 256;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int FILL_INLINE = 32;          // row-pointer gaps up to this many rows are filled by the finding lane
constexpr uint64_t FILL_CHUNK = 1 << 16; // longer gaps become tasks of at most this many rows

static_assert(TILE % 64 == 0, "tile must be a whole number of wave rounds");

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ uint64_t lanes_below(uint32_t lane) { return (1ull << lane) - 1ull; }

// ------------------------------------------------------------------------------------------------------
// Packed-bases tile: staging + record lookup + window extraction
// ------------------------------------------------------------------------------------------------------

struct BasesArgs {
    const uint32_t *bases;
    const uint32_t *valid;
    const uint64_t *rec_start;  // n_rec + 1
    uint64_t bases_words, valid_words;
    uint64_t n_pos, n_rec;
    int K;
};

struct BasesLds {
    uint32_t sb[TILE / 16 + 2];     // 2-bit codes of the tile (+ overlap for windows that start near the end)
    uint32_t sv[TILE / 32 + 2];     // validity bits
    uint32_t sflag[TILE / 32];      // bit o set: a record starts at tile offset o (o > 0)
    uint32_t spre[TILE / 64];       // record starts before 64-block b (exclusive prefix)
    uint32_t c0;                    // record that contains the tile's first position
};

// Stage one tile.  All NT threads call it; ends with a barrier.  with_records = false skips the record table
// (the y path needs no column).
__device__ __forceinline__ void bases_tile_load(const BasesArgs &a, uint64_t tile_start, BasesLds &L, bool with_records) {
    const uint32_t tid = threadIdx.x;
    const uint64_t bw0 = tile_start / 16, vw0 = tile_start / 32;
    for (uint32_t i = tid; i < TILE / 16 + 2; i += NT) {
        const uint64_t w = bw0 + i;
        L.sb[i] = (w < a.bases_words) ? a.bases[w] : 0u;
    }
    for (uint32_t i = tid; i < TILE / 32 + 2; i += NT) {
        const uint64_t w = vw0 + i;
        L.sv[i] = (w < a.valid_words) ? a.valid[w] : 0u;
    }
    if (!with_records) {
        __syncthreads();
        return;
    }
    for (uint32_t i = tid; i < TILE / 32; i += NT) L.sflag[i] = 0u;
    // record containing tile_start: last r with rec_start[r] <= tile_start (uniform search, scalar loads)
    uint64_t lo = 0, hi = a.n_rec;  // invariant: rec_start[lo] <= tile_start, answer in [lo, hi)
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a.rec_start[mid] <= tile_start)
            lo = mid;
        else
            hi = mid;
    }
    if (tid == 0) L.c0 = (uint32_t)lo;
    __syncthreads();
    // mark the record starts inside (tile_start, tile_start + TILE)
    const uint64_t tile_end = tile_start + TILE;
    for (uint64_t base = lo + 1;; base += NT) {
        const uint64_t c = base + tid;
        bool in = false;
        if (c < a.n_rec) {
            const uint64_t rs = a.rec_start[c];
            if (rs < tile_end) {
                in = true;
                const uint32_t o = (uint32_t)(rs - tile_start);
                atomicOr(&L.sflag[o >> 5], 1u << (o & 31));
            }
        }
        if (!__syncthreads_or(in && tid == NT - 1)) break;  // starts ascend: go on only while the last one was inside
    }
    __syncthreads();
    if (tid < TILE / 64) {
        uint32_t s = 0;
        for (uint32_t b = 0; b < tid; ++b) s += __popc(L.sflag[2 * b]) + __popc(L.sflag[2 * b + 1]);
        L.spre[tid] = s;
    }
    __syncthreads();
}

// Window starting at tile offset o: true iff its K positions are all valid; key = the reference's k-mer id.
__device__ __forceinline__ bool window_at(const BasesLds &L, uint32_t o, int K, uint32_t &key) {
    const uint32_t vi = o >> 5;
    const uint64_t v = ((uint64_t)L.sv[vi + 1] << 32) | L.sv[vi];
    const uint32_t need = (K == 32) ? 0xFFFFFFFFu : ((1u << K) - 1u);
    const bool ok = ((uint32_t)(v >> (o & 31)) & need) == need;
    const uint32_t bi = o >> 4;
    const uint64_t x = ((uint64_t)L.sb[bi] << 32) | L.sb[bi + 1];
    const uint32_t sh = 64u - 2u * (o & 15) - 2u * (uint32_t)K;  // >= 2 for K <= 16
    const uint32_t mask = (K == 16) ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
    key = (uint32_t)(x >> sh) & mask;
    return ok;
}

// Local record index (0-based inside this store) of tile offset o.
__device__ __forceinline__ uint32_t record_at(const BasesLds &L, uint32_t o) {
    const uint32_t b = o >> 6, l = o & 63;
    const uint64_t f = ((uint64_t)L.sflag[2 * b + 1] << 32) | L.sflag[2 * b];
    const uint64_t upto = (l == 63) ? ~0ull : ((2ull << l) - 1ull);  // bits 0..l
    return L.c0 + L.spre[b] + (uint32_t)__popcll(f & upto);
}

// ------------------------------------------------------------------------------------------------------
// y: dense histogram of valid windows
// ------------------------------------------------------------------------------------------------------

// One tile per loop iteration, grid-stride.  Global atomics: the table (4^K int32) lives in L2 / Infinity Cache
// for K <= 12; int32 sums are order-independent, so the result is bit-exact whatever the interleaving.
__global__ __launch_bounds__(NT) void k_count_y_global(BasesArgs a, int32_t *counts, uint64_t n_tiles) {
    __shared__ BasesLds L;
    const uint32_t w = threadIdx.x >> 6, lane = lane_id();
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        bases_tile_load(a, tile * TILE, L, false);
#pragma unroll 4
        for (int j = 0; j < IPT; ++j) {
            const uint32_t o = w * WCHUNK + j * 64 + lane;
            uint32_t key;
            if (window_at(L, o, a.K, key)) atomicAdd(&counts[key], 1);
        }
        __syncthreads();
    }
}

// Small tables (4^K <= 16384 bins, K <= 7): privatise the histogram in LDS, flush once per workgroup.
__global__ __launch_bounds__(NT) void k_count_y_lds(BasesArgs a, int32_t *counts, uint64_t n_tiles, uint32_t n_bins) {
    __shared__ BasesLds L;
    __shared__ uint32_t h[16384];
    const uint32_t w = threadIdx.x >> 6, lane = lane_id();
    for (uint32_t i = threadIdx.x; i < n_bins; i += NT) h[i] = 0;
    __syncthreads();
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        bases_tile_load(a, tile * TILE, L, false);
#pragma unroll 4
        for (int j = 0; j < IPT; ++j) {
            const uint32_t o = w * WCHUNK + j * 64 + lane;
            uint32_t key;
            if (window_at(L, o, a.K, key)) atomicAdd(&h[key], 1u);
        }
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n_bins; i += NT) {
        const uint32_t v = h[i];
        if (v) atomicAdd(&counts[i], (int32_t)v);
    }
}

// ------------------------------------------------------------------------------------------------------
// X pass structure: per-tile digit histogram -> exclusive scan -> stable scatter
// ------------------------------------------------------------------------------------------------------

// Lanes of the wave whose (ok, digit) equals this lane's; 0 for lanes that are not ok.
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool ok) {
    uint64_t m = __ballot(ok);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return ok ? m : 0ull;
}

// Exclusive scan over the NT threads' values (thread order); *total gets the sum.  All threads call it.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *stmp /* NW + 1 */, uint32_t &total) {
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += t;
    }
    __syncthreads();  // stmp may still be read from a previous call
    if (lane == 63) stmp[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t t = stmp[i];
        if ((uint32_t)i < w) base += t;
        tot += t;
    }
    total = tot;
    return base + inc - v;
}

// Pass 0 histogram, straight from the packed bases.  Also accumulates the number of valid windows per record
// (the reference's total_count, src/libkmers.cpp:220-227) with one atomic per (wave round, record).
__global__ __launch_bounds__(NT) void k_hist_bases(BasesArgs a, uint32_t *tilehist, uint32_t n_tiles, uint32_t *rec_total) {
    __shared__ BasesLds L;
    __shared__ uint32_t h[NW][RADIX];
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t tile = blockIdx.x;
    for (uint32_t i = tid; i < NW * RADIX; i += NT) (&h[0][0])[i] = 0;
    bases_tile_load(a, (uint64_t)tile * TILE, L, true);
    for (int j = 0; j < IPT; ++j) {
        const uint32_t o = w * WCHUNK + j * 64 + lane;
        uint32_t key;
        const bool ok = window_at(L, o, a.K, key);
        const uint32_t rec = record_at(L, o);
        if (ok) atomicAdd(&h[w][key & 255u], 1u);
        // valid windows per record: lanes are position-ordered, so records form contiguous lane segments
        const uint64_t okm = __ballot(ok);
        if (okm) {
            const uint32_t prev = __shfl_up(rec, 1);
            const bool first = (lane == 0) || (rec != prev);
            const uint64_t fm = __ballot(first);
            if (first) {
                const uint64_t above = (lane == 63) ? 0ull : (fm & ~((2ull << lane) - 1ull));
                const uint32_t endl = above ? (uint32_t)__ffsll((long long)above) - 1u : 64u;
                const uint64_t seg = ((endl == 64) ? ~0ull : ((1ull << endl) - 1ull)) & ~lanes_below(lane);
                const uint32_t cnt = (uint32_t)__popcll(okm & seg);
                if (cnt) atomicAdd(&rec_total[rec], cnt);
            }
        }
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += h[i][tid];
        tilehist[(uint64_t)tid * n_tiles + tile] = s;
    }
}

// Later-pass histogram over stored keys.
__global__ __launch_bounds__(NT) void k_hist_keys(const uint32_t *keys, const uint32_t *p_count, int shift, uint32_t *tilehist,
                                                  uint32_t n_tiles) {
    __shared__ uint32_t h[NW][RADIX];
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t tile = blockIdx.x;
    const uint32_t E = *p_count;
    const uint64_t base = (uint64_t)tile * TILE;
    for (uint32_t i = tid; i < NW * RADIX; i += NT) (&h[0][0])[i] = 0;
    __syncthreads();
    if (base < E) {
        for (int j = 0; j < IPT; ++j) {
            const uint64_t i = base + w * WCHUNK + j * 64 + lane;
            if (i < E) atomicAdd(&h[w][(keys[i] >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t s = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += h[i][tid];
        tilehist[(uint64_t)tid * n_tiles + tile] = s;
    }
}

// Stable scatter of one tile by the 8-bit digit at `shift`.
//   FROM_BASES: entries are generated from the packed store (pass 0); otherwise read from keys_in / cols_in.
//   tilebase = exclusive scan of tilehist (digit-major), i.e. the global position of the first entry of
//   (digit, tile).
template <bool FROM_BASES>
__global__ __launch_bounds__(NT) void k_scatter(BasesArgs a, const uint32_t *keys_in, const uint32_t *cols_in,
                                                const uint32_t *p_count, int shift, const uint32_t *tilebase, uint32_t n_tiles,
                                                uint32_t *keys_out, uint32_t *cols_out) {
    __shared__ uint32_t skey[TILE];
    __shared__ uint32_t scol[TILE];
    __shared__ uint32_t wcnt[NW][RADIX];
    __shared__ uint32_t tstart[RADIX];   // first tile-local slot of each digit
    __shared__ uint32_t gadj[RADIX];     // global base of (digit, tile) minus tstart[digit]
    __shared__ uint32_t stmp[NW + 1];
    __shared__ BasesLds L;               // only touched when FROM_BASES

    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t tile = blockIdx.x;
    const uint64_t base = (uint64_t)tile * TILE;

    uint32_t E = 0;
    if (!FROM_BASES) {
        E = *p_count;
        if (base >= E) return;
    }
    for (uint32_t i = tid; i < NW * RADIX; i += NT) (&wcnt[0][0])[i] = 0;
    if (FROM_BASES)
        bases_tile_load(a, base, L, true);
    else
        __syncthreads();

    uint32_t key[IPT], col[IPT], rank[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        bool ok;
        if (FROM_BASES) {
            const uint32_t o = w * WCHUNK + j * 64 + lane;
            ok = window_at(L, o, a.K, key[j]);
            col[j] = record_at(L, o);
        } else {
            const uint64_t i = base + w * WCHUNK + j * 64 + lane;
            ok = i < E;
            key[j] = ok ? keys_in[i] : 0u;
            col[j] = ok ? cols_in[i] : 0u;
        }
        const uint32_t d = (key[j] >> shift) & 255u;
        const uint64_t m = match_digit(d, ok);
        uint32_t before = 0;
        if (ok) before = wcnt[w][d];
        const uint32_t below = (uint32_t)__popcll(m & lanes_below(lane));
        rank[j] = ok ? (before + below) : NONE;
        if (ok && below == 0) wcnt[w][d] = before + (uint32_t)__popcll(m);  // the lowest lane of each group
        asm volatile("" ::: "memory");  // next round must re-read wcnt
    }
    __syncthreads();
    // digit d: exclusive scan of the per-wave counts over waves, then of the tile totals over digits
    uint32_t dtot = 0;
    if (tid < RADIX) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const uint32_t t = wcnt[i][tid];
            wcnt[i][tid] = dtot;
            dtot += t;
        }
    }
    uint32_t tile_count;
    const uint32_t dstart = block_excl_scan(dtot, stmp, tile_count);
    if (tid < RADIX) {
        tstart[tid] = dstart;
        gadj[tid] = tilebase[(uint64_t)tid * n_tiles + tile] - dstart;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        if (rank[j] != NONE) {
            const uint32_t d = (key[j] >> shift) & 255u;
            const uint32_t slot = tstart[d] + wcnt[w][d] + rank[j];
            skey[slot] = key[j];
            scol[slot] = col[j];
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < tile_count; i += NT) {
        const uint32_t k = skey[i];
        const uint32_t dst = gadj[(k >> shift) & 255u] + i;
        keys_out[dst] = k;
        cols_out[dst] = scol[i];
    }
}

// ------------------------------------------------------------------------------------------------------
// Exclusive scan of a uint32 array (three launches; the array is the digit-major tile histogram)
// ------------------------------------------------------------------------------------------------------

constexpr int SCAN_IPT = 8;
constexpr int SCAN_CHUNK = NT * SCAN_IPT;  // 4096 values per workgroup

__global__ __launch_bounds__(NT) void k_scan_reduce(const uint32_t *in, uint64_t n, uint32_t *block_sums) {
    __shared__ uint32_t stmp[NW + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_IPT;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j)
        if (base + j < n) s += in[base + j];
    uint32_t tot;
    (void)block_excl_scan(s, stmp, tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// One workgroup: exclusive scan of block_sums in place; grand total to *p_total.
__global__ __launch_bounds__(NT) void k_scan_block_sums(uint32_t *block_sums, uint32_t n_blocks, uint32_t *p_total) {
    __shared__ uint32_t stmp[NW + 1];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_blocks; base += NT) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = (i < n_blocks) ? block_sums[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, stmp, tot);
        if (i < n_blocks) block_sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) *p_total = carry;
}

__global__ __launch_bounds__(NT) void k_scan_apply(const uint32_t *in, uint64_t n, const uint32_t *block_sums, uint32_t *out) {
    __shared__ uint32_t stmp[NW + 1];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_IPT;
    uint32_t v[SCAN_IPT];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        v[j] = (base + j < n) ? in[base + j] : 0u;
        s += v[j];
    }
    uint32_t tot;
    uint32_t run = block_sums[blockIdx.x] + block_excl_scan(s, stmp, tot);
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        if (base + j < n) out[base + j] = run;
        run += v[j];
    }
}

// ------------------------------------------------------------------------------------------------------
// Run-length reduce of the sorted (k-mer, column) instances -> CSR
// ------------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool is_head(const uint32_t *keys, const uint32_t *cols, uint64_t i, uint32_t E) {
    if (i >= E) return false;
    if (i == 0) return true;
    return keys[i] != keys[i - 1] || cols[i] != cols[i - 1];
}

// Heads per tile and the first head of each tile.
__global__ __launch_bounds__(NT) void k_rle_count(const uint32_t *keys, const uint32_t *cols, const uint32_t *p_count,
                                                  uint32_t *tile_heads, uint32_t *first_head) {
    __shared__ uint32_t s_cnt[NW], s_first[NW];
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t tile = blockIdx.x;
    const uint32_t E = *p_count;
    const uint64_t base = (uint64_t)tile * TILE;
    uint32_t cnt = 0, first = NONE;
    if (base < E) {
        for (int j = 0; j < IPT; ++j) {
            const uint64_t i = base + w * WCHUNK + j * 64 + lane;
            const uint64_t hm = __ballot(is_head(keys, cols, i, E));
            if (hm) {
                if (first == NONE) first = (uint32_t)(base + w * WCHUNK + j * 64) + (uint32_t)__ffsll((long long)hm) - 1u;
                cnt += (uint32_t)__popcll(hm);
            }
        }
    }
    if (lane == 0) {
        s_cnt[w] = cnt;
        s_first[w] = first;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t c = 0, f = NONE;
        for (int i = 0; i < NW; ++i) {
            c += s_cnt[i];
            if (f == NONE) f = s_first[i];
        }
        tile_heads[tile] = c;
        first_head[tile] = f;
    }
}

struct FillTask {
    uint64_t begin, end, value;
};

// row_ind[lo .. hi) = value, by this lane when short, queued for k_fill_tasks when long.
__device__ __forceinline__ void fill_rows(uint64_t *row_ind, uint64_t lo, uint64_t hi, uint64_t value, FillTask *tasks,
                                          uint32_t *p_ntasks, uint32_t task_cap, uint32_t *p_err) {
    if (hi - lo <= (uint64_t)FILL_INLINE) {
        for (uint64_t r = lo; r < hi; ++r) row_ind[r] = value;
        return;
    }
    for (uint64_t b = lo; b < hi; b += FILL_CHUNK) {
        const uint32_t t = atomicAdd(p_ntasks, 1u);
        if (t >= task_cap) {
            *p_err = 1u;
            return;
        }
        tasks[t].begin = b;
        tasks[t].end = (hi - b > FILL_CHUNK) ? b + FILL_CHUNK : hi;
        tasks[t].value = value;
    }
}

// One CSR entry per head: column, value, and the row pointer entries of every row that starts (or is empty)
// between the previous head's k-mer and this one.
//   p_scalars[0] = E (instances), p_scalars[1] = nnz (heads)
__global__ __launch_bounds__(NT) void k_rle_emit(const uint32_t *keys, const uint32_t *cols, const uint32_t *p_count,
                                                 const uint32_t *p_nnz, const uint32_t *tile_headbase, const uint32_t *first_head,
                                                 uint32_t n_tiles, const uint32_t *rec_total, uint64_t first_col, uint64_t M,
                                                 uint64_t max_nz, float *data, uint64_t *col_ind, uint64_t *row_ind,
                                                 FillTask *tasks, uint32_t *p_ntasks, uint32_t task_cap, uint32_t *p_err) {
    __shared__ uint32_t sidx[TILE];
    __shared__ uint32_t s_wave[NW];
    __shared__ uint32_t s_next;
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t tile = blockIdx.x;
    const uint32_t E = *p_count;
    const uint64_t nnz = *p_nnz;
    if (nnz > max_nz) return;  // caller's buffers are too small: write nothing (host reports -2)
    if (E == 0) {              // no instance at all: every row pointer is 0
        if (tile == 0 && tid == 0) fill_rows(row_ind, 0, M + 1, 0, tasks, p_ntasks, task_cap, p_err);
        return;
    }
    const uint64_t base = (uint64_t)tile * TILE;
    if (base >= E) return;

    uint64_t hm[IPT];
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const uint64_t i = base + w * WCHUNK + j * 64 + lane;
        hm[j] = __ballot(is_head(keys, cols, i, E));
        cnt += (uint32_t)__popcll(hm[j]);
    }
    if (lane == 0) s_wave[w] = cnt;
    if (tid == 0) {  // first head after this tile (runs may span tiles)
        uint32_t t = tile + 1;
        while (t < n_tiles && first_head[t] == NONE) ++t;
        s_next = (t < n_tiles) ? first_head[t] : E;
    }
    __syncthreads();
    uint32_t wbase = 0, n_heads = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t t = s_wave[i];
        if ((uint32_t)i < w) wbase += t;
        n_heads += t;
    }
    uint32_t run = wbase;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        if ((hm[j] >> lane) & 1ull)
            sidx[run + (uint32_t)__popcll(hm[j] & lanes_below(lane))] = (uint32_t)(base + w * WCHUNK + j * 64 + lane);
        run += (uint32_t)__popcll(hm[j]);
    }
    __syncthreads();
    const uint64_t out0 = tile_headbase[tile];
    for (uint32_t r = tid; r < n_heads; r += NT) {
        const uint32_t i = sidx[r];
        const uint32_t nxt = (r + 1 < n_heads) ? sidx[r + 1] : s_next;
        const uint32_t k = keys[i], c = cols[i];
        const uint64_t o = out0 + r;
        // reference: float(count) / total_count with total_count uint32_t -> both rounded to float, float divide.
        // Dividing the two floats in double and rounding once more is the correctly rounded float quotient
        // (53 >= 2*24 + 2), independent of the device's float-division mode.
        const float fc = (float)(nxt - i), ft = (float)rec_total[c];
        data[o] = (float)((double)fc / (double)ft);
        col_ind[o] = first_col + c;
        if (i == 0) {
            fill_rows(row_ind, 0, (uint64_t)k + 1, o, tasks, p_ntasks, task_cap, p_err);
        } else {
            const uint32_t pk = keys[i - 1];
            if (pk != k) fill_rows(row_ind, (uint64_t)pk + 1, (uint64_t)k + 1, o, tasks, p_ntasks, task_cap, p_err);
        }
        if (nxt == E) fill_rows(row_ind, (uint64_t)k + 1, M + 1, nnz, tasks, p_ntasks, task_cap, p_err);
    }
}

__global__ __launch_bounds__(NT) void k_fill_tasks(uint64_t *row_ind, const FillTask *tasks, const uint32_t *p_ntasks,
                                                   uint32_t task_cap) {
    uint32_t n = *p_ntasks;
    if (n > task_cap) n = task_cap;
    for (uint32_t t = blockIdx.x; t < n; t += gridDim.x) {
        const FillTask ft = tasks[t];
        for (uint64_t r = ft.begin + threadIdx.x; r < ft.end; r += NT) row_ind[r] = ft.value;
    }
}

// ------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------

// Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline object).
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfEntry> g_prof;

void profile_enable(bool on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on;
}
void profile_reset() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.clear();
}
std::vector<ProfEntry> profile_snapshot() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return g_prof;
}

struct PhaseTimer {
    hipStream_t s;
    bool on;
    std::vector<hipEvent_t> ev;
    std::vector<const char *> label;
    explicit PhaseTimer(hipStream_t s_) : s(s_), on(g_prof_on) { mark(nullptr); }
    void mark(const char *name) {  // closes the phase `name` (first call opens the timeline)
        if (!on) return;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
            on = false;
            return;
        }
        (void)hipEventRecord(e, s);
        ev.push_back(e);
        label.push_back(name);
    }
    void finish() {  // call after the stream has been synchronised
        if (on) {
            std::lock_guard<std::mutex> lk(g_prof_mu);
            for (size_t i = 1; i < ev.size(); ++i) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev[i - 1], ev[i]) != hipSuccess) continue;
                bool found = false;
                for (auto &p : g_prof)
                    if (p.name == label[i]) {
                        p.ms += ms;
                        p.calls += 1;
                        found = true;
                        break;
                    }
                if (!found) g_prof.push_back(ProfEntry{label[i], (double)ms, 1});
            }
        }
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear();
    }
    ~PhaseTimer() { finish(); }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct XLayout {
    uint32_t n_tiles;
    uint64_t hist_len;       // RADIX * n_tiles
    uint32_t scan_blocks;
    uint32_t task_cap;
    size_t off_keyA, off_colA, off_keyB, off_colB, off_hist, off_blocksums, off_total, off_heads, off_first, off_scalars,
        off_tasks, bytes;
};

static XLayout x_layout(uint64_t n_pos, uint64_t n_rec, int K) {
    XLayout l;
    const uint64_t emax = n_pos ? n_pos : 1;
    l.n_tiles = (uint32_t)((emax + TILE - 1) / TILE);
    l.hist_len = (uint64_t)RADIX * l.n_tiles;
    l.scan_blocks = (uint32_t)((l.hist_len + SCAN_CHUNK - 1) / SCAN_CHUNK);
    const uint64_t M = 1ull << (2 * K);
    // long gaps: each holds > FILL_INLINE rows and follows a row head, plus the splits of very long ones
    uint64_t cap = (M + 1) / (FILL_INLINE + 1) + 2;
    if (cap > emax + 2) cap = emax + 2;
    cap += (M + 1) / FILL_CHUNK + 2;
    l.task_cap = (uint32_t)(cap > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : cap);
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = align_up(o + bytes, 256);
        return at;
    };
    l.off_keyA = take(emax * 4);
    l.off_colA = take(emax * 4);
    l.off_keyB = take(emax * 4);
    l.off_colB = take(emax * 4);
    l.off_hist = take(l.hist_len * 4);
    l.off_blocksums = take((size_t)l.scan_blocks * 4 + 4);
    l.off_total = take((n_rec ? n_rec : 1) * 4);
    l.off_heads = take((size_t)l.n_tiles * 4 + 4);
    l.off_first = take((size_t)l.n_tiles * 4 + 4);
    l.off_scalars = take(64);
    l.off_tasks = take((size_t)l.task_cap * sizeof(FillTask));
    l.bytes = o;
    return l;
}

size_t x_workspace_bytes(uint64_t n_pos, uint64_t n_rec, int K) { return x_layout(n_pos, n_rec, K).bytes; }

#define KK_CHECK(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            fprintf(stderr, "libkmers: HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
            return -11;                                                                                  \
        }                                                                                                \
    } while (0)

static int scan_u32(const uint32_t *in, uint64_t n, uint32_t *out, uint32_t *block_sums, uint32_t n_blocks, uint32_t *p_total,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_scan_reduce, dim3(n_blocks), dim3(NT), 0, s, in, n, block_sums);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(NT), 0, s, block_sums, n_blocks, p_total);
    hipLaunchKernelGGL(k_scan_apply, dim3(n_blocks), dim3(NT), 0, s, in, n, block_sums, out);
    KK_CHECK(hipGetLastError());
    return 0;
}

int count_y(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words, uint64_t n_pos, int K,
            int32_t *d_counts, hipStream_t s) {
    if (n_pos == 0) return 0;
    BasesArgs a{d_bases, d_valid, nullptr, bases_words, valid_words, n_pos, 0, K};
    const uint64_t n_tiles = (n_pos + TILE - 1) / TILE;
    const uint32_t n_bins = 1u << (2 * (K > 7 ? 7 : K));
    if (K <= 7) {
        const uint32_t grid = (uint32_t)(n_tiles < 512 ? n_tiles : 512);
        hipLaunchKernelGGL(k_count_y_lds, dim3(grid), dim3(NT), 0, s, a, d_counts, n_tiles, n_bins);
    } else {
        const uint32_t grid = (uint32_t)(n_tiles < 4096 ? n_tiles : 4096);
        hipLaunchKernelGGL(k_count_y_global, dim3(grid), dim3(NT), 0, s, a, d_counts, n_tiles);
    }
    KK_CHECK(hipGetLastError());
    return 0;
}

int build_x(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words, const uint64_t *d_rec_start,
            uint64_t n_pos, uint64_t n_rec, uint64_t first_col, int K, float *d_data, uint64_t *d_col_ind, uint64_t *d_row_ind,
            uint64_t max_nz, void *d_ws, size_t ws_bytes, uint64_t *nnz_out, hipStream_t s) {
    const XLayout l = x_layout(n_pos, n_rec, K);
    if (ws_bytes < l.bytes) return -14;
    const uint64_t M = 1ull << (2 * K);
    char *ws = (char *)d_ws;
    uint32_t *keyA = (uint32_t *)(ws + l.off_keyA), *colA = (uint32_t *)(ws + l.off_colA);
    uint32_t *keyB = (uint32_t *)(ws + l.off_keyB), *colB = (uint32_t *)(ws + l.off_colB);
    uint32_t *hist = (uint32_t *)(ws + l.off_hist), *bsum = (uint32_t *)(ws + l.off_blocksums);
    uint32_t *total = (uint32_t *)(ws + l.off_total);
    uint32_t *heads = (uint32_t *)(ws + l.off_heads), *first = (uint32_t *)(ws + l.off_first);
    uint32_t *scal = (uint32_t *)(ws + l.off_scalars);  // [0]=E [1]=nnz [2]=ntasks [3]=err
    FillTask *tasks = (FillTask *)(ws + l.off_tasks);

    PhaseTimer pt(s);
    KK_CHECK(hipMemsetAsync(scal, 0, 64, s));
    if (n_pos == 0 || n_rec == 0) {  // nothing to shred: all-zero row pointer
        KK_CHECK(hipMemsetAsync(d_row_ind, 0, (M + 1) * sizeof(uint64_t), s));
        KK_CHECK(hipStreamSynchronize(s));
        *nnz_out = 0;
        return 0;
    }
    KK_CHECK(hipMemsetAsync(total, 0, n_rec * 4, s));

    BasesArgs a{d_bases, d_valid, d_rec_start, bases_words, valid_words, n_pos, n_rec, K};
    const dim3 grid(l.n_tiles), block(NT);
    const int n_pass = (2 * K + 7) / 8;

    // pass 0: packed bases -> (key, col) instances partitioned by the lowest digit
    pt.mark("memset");
    hipLaunchKernelGGL(k_hist_bases, grid, block, 0, s, a, hist, l.n_tiles, total);
    pt.mark("k_hist_bases");
    if (int rc = scan_u32(hist, l.hist_len, hist, bsum, l.scan_blocks, &scal[0], s)) return rc;
    pt.mark("k_scan");
    hipLaunchKernelGGL(k_scatter<true>, grid, block, 0, s, a, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                       (const uint32_t *)&scal[0], 0, (const uint32_t *)hist, l.n_tiles, keyA, colA);
    pt.mark("k_scatter<bases>");
    uint32_t *ki = keyA, *ci = colA, *ko = keyB, *co = colB;
    for (int p = 1; p < n_pass; ++p) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(k_hist_keys, grid, block, 0, s, (const uint32_t *)ki, (const uint32_t *)&scal[0], shift, hist, l.n_tiles);
        pt.mark("k_hist_keys");
        if (int rc = scan_u32(hist, l.hist_len, hist, bsum, l.scan_blocks, &scal[4], s)) return rc;
        pt.mark("k_scan");
        hipLaunchKernelGGL(k_scatter<false>, grid, block, 0, s, a, (const uint32_t *)ki, (const uint32_t *)ci,
                           (const uint32_t *)&scal[0], shift, (const uint32_t *)hist, l.n_tiles, ko, co);
        pt.mark("k_scatter<keys>");
        uint32_t *t;
        t = ki, ki = ko, ko = t;
        t = ci, ci = co, co = t;
    }
    // run-length reduce
    hipLaunchKernelGGL(k_rle_count, grid, block, 0, s, (const uint32_t *)ki, (const uint32_t *)ci, (const uint32_t *)&scal[0], heads,
                       first);
    pt.mark("k_rle_count");
    {
        const uint32_t nb = (l.n_tiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
        if (int rc = scan_u32(heads, l.n_tiles, heads, bsum, nb, &scal[1], s)) return rc;
    }
    pt.mark("k_scan");
    hipLaunchKernelGGL(k_rle_emit, grid, block, 0, s, (const uint32_t *)ki, (const uint32_t *)ci, (const uint32_t *)&scal[0],
                       (const uint32_t *)&scal[1], (const uint32_t *)heads, (const uint32_t *)first, l.n_tiles,
                       (const uint32_t *)total, first_col, M, max_nz, d_data, d_col_ind, d_row_ind, tasks, &scal[2], l.task_cap,
                       &scal[3]);
    pt.mark("k_rle_emit");
    hipLaunchKernelGGL(k_fill_tasks, dim3(2048), block, 0, s, d_row_ind, (const FillTask *)tasks, (const uint32_t *)&scal[2],
                       l.task_cap);
    pt.mark("k_fill_tasks");
    KK_CHECK(hipGetLastError());
    uint32_t h[4] = {0, 0, 0, 0};
    KK_CHECK(hipMemcpyAsync(h, scal, sizeof h, hipMemcpyDeviceToHost, s));
    KK_CHECK(hipStreamSynchronize(s));
    pt.finish();
    *nnz_out = h[1];
    if ((uint64_t)h[1] > max_nz) return -2;
    if (h[3]) {
        fprintf(stderr, "libkmers: internal error: row-pointer fill queue overflow\n");
        return -11;
    }
    return 0;
}

}  // namespace kk
