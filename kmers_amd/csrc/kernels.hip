// kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels for the k-mer shredding / counting hot path.
//
// What they replace in the reference (src/libkmers.cpp):
//   valid_kmer + kmer_to_id per window (:222-227, :321-326, :331-345)  -> window_at() / word_window(): a bit-field
//                                                                         extract from the 2-bit packed store
//   count_kmers, atomic ++ into a shared int array (:213-216, :320-328)
//        -> y, K <= 8: k_count_y_lds (LDS tables, windows shifted out of registers);
//           y, K >= 9: partition by the top 8 / 16 id bits (k_hist_words_y, k_yscatter, for K >= 13 also
//           k_hist_entries, k_scatter mode 2, k_bucket_bounds), then per-slice LDS tables (k_count_y_sorted,
//           k_count_y_overflow); k_count_y_global is the plain-atomics kernel kept for A-B runs.
//   collect_kmers (unordered_map per isoform) + fill_indices_coo (:218-238)
//   + ThreadPoolLocalData::to_csr (sort by row, gather, sort by col, :77-169)
//        -> "shred, stable radix partition by k-mer id, run-length reduce":
//           k_hist_bases / k_scatter<true> (pass 0, straight from the packed bases), k_hist_entries /
//           k_scatter<false> (further 8-bit digits), k_cscan_* (tile histogram -> positions), k_rle (one CSR entry per
//           run of equal (k-mer, column), value float(run length) / float(valid windows of the column), row pointer
//           by gap filling), k_fill_tasks.
//
// Why a sort and not a hash: every window instance becomes a (k-mer id, column) entry in position order, i.e.
// column-ascending.  A STABLE least-significant-digit partition by k-mer id therefore leaves, inside every row, the
// columns ascending and the instances of one (row, column) adjacent -- exactly the CSR order the reference produces,
// with no per-isoform hash table, no long-isoform special case and coalesced traffic (HBM-bound, no MFMA: nothing
// here is a contraction).  Entries are 8 bytes (id << 32 | column), or 4 bytes for K <= 8 (see "narrow entries").
//
// Geometry: 512-thread workgroups (8 wave64), a tile is 8192 entries.  Wave w of a tile owns the contiguous
// chunk [w*1024, (w+1)*1024) and walks it in 16 rounds of 64 consecutive entries, so the order
// (wave, round, lane) is the entry order; ranks inside a digit come from a wave-wide match (an LDS OR-mask per
// (wave, digit)), which is deterministic (no atomic decides an order on the X path).  The partition kernels are
// persistent: a workgroup walks a contiguous range of tiles and fetches tile t+1 (into registers) while tile t is
// ranked and written, so HBM latency is paid once per workgroup, not once per tile.  The y path needs no order at
// all (a histogram follows), so there ranks are what a returning LDS atomic hands out.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>
#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace kk {

constexpr int NT = 512;          // threads per workgroup
constexpr int NW = NT / 64;      // waves per workgroup
constexpr int IPT = 16;          // rounds per wave
constexpr int TILE = NT * IPT;   // entries per tile (8192)
constexpr int WCHUNK = 64 * IPT; // entries per wave (1024)
#ifndef RLE_IPT_NARROW
#define RLE_IPT_NARROW 24   // k_rle: rounds per wave with 4-byte / 8-byte entries, threads per workgroup
#endif
#ifndef RLE_IPT_WIDE
#define RLE_IPT_WIDE 16
#endif
#ifndef RLE_NT
#define RLE_NT 1024
#endif
static_assert(RLE_NT * RLE_IPT_NARROW >= NT * IPT && RLE_NT * RLE_IPT_WIDE >= NT * IPT,
              "the chained-scan status array holds one word per 8192-entry tile");
constexpr int RADIX = 256;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int FILL_INLINE = 32;          // row-pointer gaps up to this many rows are filled by the finding lane
constexpr uint64_t FILL_CHUNK = 1 << 16; // longer gaps become tasks of at most this many rows
#ifndef KK_HIST_BLOCKS
#define KK_HIST_BLOCKS 8192
#endif
#ifndef KK_PERSIST_BLOCKS
#define KK_PERSIST_BLOCKS 8192
#endif
constexpr int PERSIST_BLOCKS = KK_PERSIST_BLOCKS;  // workgroups of the partition kernels: each walks a contiguous tile range (512 = one resident set: 1.41 ms; 8192: 1.17 ms for k_scatter<entries> on config 2 -- the tail balances)

static_assert(TILE / 16 == NT, "one packed-bases word per thread per tile");
static_assert(TILE % 64 == 0, "tile must be a whole number of wave rounds");

typedef unsigned long long u64;

// Diagnostic build only (-DKK_STAMPS): per-phase cycle sums of the scatter kernels, wave 0 lane 0 of every workgroup.
#ifdef KK_STAMPS
__device__ int g_ablate;   // diagnostic: 1 = no write-out stores, 2 = no entry prefetch loads, 4 = no ranking
#define KK_ABLATE(bit) (g_ablate & (bit))
__device__ u64 g_stamps[16];
#define KK_STAMP(k)                                                         \
    do {                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                  \
        if (threadIdx.x == 0) {                                             \
            const u64 now_ = __builtin_amdgcn_s_memtime();                  \
            stamp_acc_[k] += now_ - stamp_last_;                            \
            stamp_last_ = now_;                                             \
        }                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                  \
    } while (0)
#define KK_STAMP_BEGIN()  u64 stamp_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u64 stamp_last_ = __builtin_amdgcn_s_memtime()
#define KK_STAMP_END(off)                                                                        \
    do {                                                                                         \
        if (threadIdx.x == 0)                                                                    \
            for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&g_stamps[(off) + q_], stamp_acc_[q_]);    \
    } while (0)
#else
#define KK_ABLATE(bit) 0
#define KK_STAMP(k) do {} while (0)
#define KK_STAMP_BEGIN() do {} while (0)
#define KK_STAMP_END(off) do {} while (0)
#endif

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }
__device__ __forceinline__ u64 lanes_below(uint32_t lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ u64 make_entry(uint32_t key, uint32_t col) { return ((u64)key << 32) | col; }
__device__ __forceinline__ uint32_t entry_key(u64 e) { return (uint32_t)(e >> 32); }
__device__ __forceinline__ uint32_t entry_col(u64 e) { return (uint32_t)e; }

// ---- narrow entries (K <= 8 and at most 2^24 records): 4 bytes = one digit byte << 24 | column.
// The partition is least-significant-digit first, so after a pass the digit that was just sorted on is implied by
// the REGION an entry sits in (regions[d] = position of the first entry of digit d, regions[256] = entry count):
//   pass 0 writes   (k-mer id >> 8) << 24 | column         into the regions of the low digit,
//   pass 1 reads it, partitions on the top byte and writes   low digit << 24 | column   into the regions of the high digit,
//   the run-length reduce rebuilds  k-mer id = region << 8 | top byte  (or = region when 2K <= 8 and there is one pass).
// Half the bytes of the 8-byte form on every intermediate array; the 8-byte form serves K >= 9 or more records.
__device__ __forceinline__ uint32_t region_search(const uint32_t *sreg, uint32_t i) {  // largest r with sreg[r] <= i
    uint32_t lo = 0, hi = RADIX;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t mid = (lo + hi) >> 1;
        if (sreg[mid] <= i)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}
// region of i given a lower bound `hint` (i < sreg[RADIX])
__device__ __forceinline__ uint32_t region_from(const uint32_t *sreg, uint32_t i, uint32_t hint) {
    uint32_t r = hint;
    while (sreg[r + 1] <= i) ++r;
    return r;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release/acquire fence around
// s_barrier, and on gfx950 global loads, stores and LDS ops are released by s_waitcnt vmcnt(0) / lgkmcnt(0): it would
// drain every global prefetch and every write-out store at each barrier.  No kernel here passes GLOBAL data between
// the waves of a workgroup, so waiting for this wave's LDS operations before the barrier is all that is needed.
__device__ __forceinline__ void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Inclusive scan over the 64 lanes of a wave in registers: four row_shr steps inside each row of 16 lanes, then the
// row totals cross with row_bcast:15 / row_bcast:31 (DPP: a few cycles per step; the six ds_bpermute steps of a
// __shfl_up scan are six dependent LDS round trips, ~1 us per tile in the partition kernels' digit scan).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
#define KK_DPP_ADD(ctrl, rows, bound) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xf, bound)
    KK_DPP_ADD(0x111, 0xf, true);   // row_shr:1
    KK_DPP_ADD(0x112, 0xf, true);   // row_shr:2
    KK_DPP_ADD(0x114, 0xf, true);   // row_shr:4
    KK_DPP_ADD(0x118, 0xf, true);   // row_shr:8
    KK_DPP_ADD(0x142, 0xa, false);  // row_bcast:15 -> rows 1 and 3
    KK_DPP_ADD(0x143, 0xc, false);  // row_bcast:31 -> rows 2 and 3
#undef KK_DPP_ADD
    return v;
}

// Exclusive scan over the NT threads' values (thread order); total gets the sum.  All threads call it.
// PRE_SYNCED: the caller has a workgroup barrier between the last read of stmp and this call (saves one here).
template <bool PRE_SYNCED = false>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *stmp /* NW */, uint32_t &total) {
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v);
    if (!PRE_SYNCED) lds_sync();  // stmp may still be read from a previous call
    if (lane == 63) stmp[w] = inc;
    lds_sync();
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

// ------------------------------------------------------------------------------------------------------
// Packed-bases tile: prefetch to registers, commit to LDS, window extraction, record lookup
// ------------------------------------------------------------------------------------------------------

struct BasesArgs {
    const uint32_t *bases;
    const uint32_t *valid;
    const uint32_t *rec_flags;  // bit p set: a record starts at position p (built by k_record_tables); may be null (y)
    const uint32_t *tile_cb;    // [n_tiles + 1]: record starts strictly before tile t's first position
    uint64_t bases_words, valid_words;
    uint64_t n_pos;
    int K;
    uint64_t tile0;  // y in chunks: the kernel's tile t is tile tile0 + t of the store (X: 0)
    int low_bits;    // y for K >= 9: the k-mer id is split into a sort key (id >> low_bits) and a payload (the low bits)
};

struct TileLds {
    uint32_t sb[TILE / 16 + 2];   // 2-bit codes of the tile (+ overlap for windows that start near the end)
    uint32_t sv[TILE / 32 + 2];   // validity bits
    uint32_t sflag[TILE / 32];    // record-start bits
    uint32_t spre[TILE / 64];     // record starts before 64-block b of the tile (exclusive prefix)
    uint32_t cb;                  // record starts before the tile
};

struct TilePf {  // one tile's words, held in registers between fetch and commit
    uint32_t b0, b1, v, f, cb;
};

template <bool WITH_REC>
__device__ __forceinline__ TilePf tile_fetch(const BasesArgs &a, uint64_t local_tile) {
    const uint32_t tid = threadIdx.x;
    TilePf p;
    const uint64_t tile = local_tile + a.tile0;
    const uint64_t bw = tile * (TILE / 16) + tid;
    p.b0 = (bw < a.bases_words) ? a.bases[bw] : 0u;
    p.b1 = 0u;
    if (tid < 2) {
        const uint64_t w2 = tile * (TILE / 16) + NT + tid;
        p.b1 = (w2 < a.bases_words) ? a.bases[w2] : 0u;
    }
    const uint64_t vw = tile * (TILE / 32) + tid;
    p.v = (tid < TILE / 32 + 2 && vw < a.valid_words) ? a.valid[vw] : 0u;
    p.f = 0u;
    p.cb = 0u;
    if (WITH_REC) {
        p.f = (tid < TILE / 32 && vw < a.valid_words) ? a.rec_flags[vw] : 0u;
        p.cb = a.tile_cb[local_tile];
    }
    return p;
}

// All NT threads call it; ends with a barrier.
template <bool WITH_REC>
__device__ __forceinline__ void tile_commit(TileLds &L, const TilePf &p, uint32_t *stmp) {
    const uint32_t tid = threadIdx.x;
    L.sb[tid] = p.b0;
    if (tid < 2) L.sb[NT + tid] = p.b1;
    if (tid < TILE / 32 + 2) L.sv[tid] = p.v;
    if (WITH_REC) {
        if (tid < TILE / 32) L.sflag[tid] = p.f;
        if (tid == 0) L.cb = p.cb;
        // record starts per 64-position block: lanes 2b, 2b+1 hold its two flag words
        const uint32_t pc = (tid < TILE / 32) ? (uint32_t)__popc(p.f) : 0u;
        const uint32_t pair = pc + __shfl_xor(pc, 1);
        uint32_t tot;
        // every caller has at least one barrier between a tile's last use of stmp and the next tile's commit
        const uint32_t ex = block_excl_scan<true>((tid & 1) ? 0u : pair, stmp, tot);
        if (tid < TILE / 32 && !(tid & 1)) L.spre[tid >> 1] = ex;
    }
    lds_sync();
}

// Window starting at tile offset o: true iff its K positions are all valid; key = the reference's k-mer id.
__device__ __forceinline__ bool window_at(const TileLds &L, uint32_t o, int K, uint32_t &key) {
    const uint32_t vi = o >> 5;
    const u64 v = ((u64)L.sv[vi + 1] << 32) | L.sv[vi];
    const uint32_t need = (1u << K) - 1u;  // K <= 16
    const bool ok = ((uint32_t)(v >> (o & 31)) & need) == need;
    const uint32_t bi = o >> 4;
    const u64 x = ((u64)L.sb[bi] << 32) | L.sb[bi + 1];
    const uint32_t sh = 64u - 2u * (o & 15) - 2u * (uint32_t)K;  // >= 2 for K <= 16
    const uint32_t mask = (K == 16) ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
    key = (uint32_t)(x >> sh) & mask;
    return ok;
}

// Local record index (0-based inside this store) of tile offset o.
__device__ __forceinline__ uint32_t record_at(const TileLds &L, uint32_t o) {
    const uint32_t b = o >> 6, l = o & 63;
    const u64 f = ((u64)L.sflag[2 * b + 1] << 32) | L.sflag[2 * b];
    const u64 upto = (l == 63) ? ~0ull : ((2ull << l) - 1ull);  // bits 0..l
    return L.cb + L.spre[b] + (uint32_t)__popcll(f & upto) - 1u;
}

// Record lookup for a wave's rounds without per-lane LDS traffic: round j of wave w covers the 64 positions of flag
// block w*IPT + j, so its two flag words and its record base are wave-uniform.  Lane l < IPT holds round l's values;
// a round fetches them with v_readlane.  Nearly every block has no record start (a boundary every ~2400 positions):
// then the whole round is one record.
struct RecLanes {
    uint32_t f_lo, f_hi, rb;  // flag words of block w*IPT + lane; records that start before it (tile base included)
};
__device__ __forceinline__ RecLanes rec_lanes_load(const TileLds &L, uint32_t w, uint32_t lane) {
    RecLanes r{0u, 0u, 0u};
    if (lane < (uint32_t)IPT) {
        const uint32_t b = w * IPT + lane;
        r.f_lo = L.sflag[2 * b];
        r.f_hi = L.sflag[2 * b + 1];
        r.rb = L.cb + L.spre[b];
    }
    return r;
}
// Local record index of this lane's position in round j (wave-uniform when the block has no record start).
__device__ __forceinline__ uint32_t rec_lanes_at(const RecLanes &r, int j, uint32_t lane, u64 &f_out) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)r.f_lo, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)r.f_hi, j);
    const uint32_t rb = (uint32_t)__builtin_amdgcn_readlane((int)r.rb, j);
    const u64 f = ((u64)hi << 32) | lo;
    f_out = f;
    uint32_t rec = rb - 1u;
    if (f) {  // wave-uniform branch
        const u64 upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        rec += (uint32_t)__popcll(f & upto);
    }
    return rec;
}

// One launch for the two record tables.  Threads [0, n_rec): bit p of rec_flags := a record starts at p (rec_flags is
// zeroed beforehand).  Threads [n_rec, n_rec + n_tiles]: tile_cb[t] = number of records that start strictly before
// position t*TILE, t in [0, n_tiles].
__global__ void k_record_tables(const uint64_t *rec_start, uint64_t n_rec, uint32_t *rec_flags, uint32_t n_tiles, uint32_t *tile_cb) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n_rec) {
        const uint64_t p = rec_start[g];
        atomicOr(&rec_flags[p >> 5], 1u << (p & 31));
        return;
    }
    const uint64_t t = g - n_rec;
    if (t > n_tiles) return;
    const uint64_t target = t * TILE;
    uint64_t lo = 0, hi = n_rec;  // first r with rec_start[r] >= target
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (rec_start[mid] < target)
            lo = mid + 1;
        else
            hi = mid;
    }
    tile_cb[t] = (uint32_t)lo;
}

// ------------------------------------------------------------------------------------------------------
// y: dense histogram of valid windows
// ------------------------------------------------------------------------------------------------------

// Global atomics: int32 sums are order-independent, so the result is bit-exact whatever the interleaving.
__global__ __launch_bounds__(NT) void k_count_y_global(BasesArgs a, int32_t *counts, uint64_t n_tiles) {
    __shared__ TileLds L;
    __shared__ uint32_t stmp[NW];
    const uint32_t w = threadIdx.x >> 6, lane = lane_id();
    uint64_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    TilePf pf = tile_fetch<false>(a, tile);
    for (; tile < n_tiles; tile += gridDim.x) {
        tile_commit<false>(L, pf, stmp);
        if (tile + gridDim.x < n_tiles) pf = tile_fetch<false>(a, tile + gridDim.x);
#pragma unroll 4
        for (int j = 0; j < IPT; ++j) {
            const uint32_t o = w * WCHUNK + j * 64 + lane;
            uint32_t key;
            if (window_at(L, o, a.K, key)) atomicAdd(&counts[key], 1);
        }
        lds_sync();
    }
}

// ---- order-free window decode: a histogram does not care in which order windows arrive, so a thread takes the 16
// consecutive windows that start in ONE packed word (positions 16 g .. 16 g + 15) and shifts them out of registers:
// no LDS tile, no barrier, two VALU instructions per k-mer id.
struct WordWindows {
    uint32_t b0, b1;  // bases words g, g + 1 (a window of K <= 16 positions never reaches word g + 2)
    uint32_t v;       // validity of positions 16 g .. 16 g + 31
};
__device__ __forceinline__ WordWindows word_fetch(const BasesArgs &a, uint64_t g) {
    WordWindows x;
    x.b0 = (g < a.bases_words) ? a.bases[g] : 0u;
    x.b1 = (g + 1 < a.bases_words) ? a.bases[g + 1] : 0u;
    const uint64_t vi = g >> 1;  // valid word of position 16 g
    const uint32_t v0 = (vi < a.valid_words) ? a.valid[vi] : 0u, v1 = (vi + 1 < a.valid_words) ? a.valid[vi + 1] : 0u;
    x.v = (g & 1) ? ((v0 >> 16) | (v1 << 16)) : v0;
    return x;
}
// window j (0..15) of the word: valid iff its K positions are; key = the reference's k-mer id
__device__ __forceinline__ bool word_window(const WordWindows &x, int j, int K, uint32_t &key) {
    const uint32_t need = (1u << K) - 1u;  // K <= 16
    const bool ok = ((x.v >> j) & need) == need;
    const uint32_t top = j ? __builtin_amdgcn_alignbit(x.b0, x.b1, 32u - 2u * (uint32_t)j) : x.b0;  // 32 bits from base j on
    key = top >> (32u - 2u * (uint32_t)K);
    return ok;
}

// y, K <= 8: tables of up to 32768 bins per sweep are privatised in LDS (u32 bins, no overflow) and flushed once per
// workgroup: K <= 7 in one sweep; K = 8 in two sweeps over the store (bins [0,32768) then [32768,65536)): the packed
// bases are 3 bits per position, so reading them twice costs far less than 4e8 contended global atomics.
// A thread adds a run of equal consecutive k-mers (poly-A) once, with its length.
constexpr int YL_NT = 1024;
__global__ __launch_bounds__(YL_NT) void k_count_y_lds(BasesArgs a, int32_t *counts, uint64_t n_words, uint32_t n_bins) {
    __shared__ uint32_t h[32768];
    const uint32_t sweep_bins = n_bins < 32768u ? n_bins : 32768u;
    const int K = a.K;
    for (uint32_t lo = 0; lo < n_bins; lo += sweep_bins) {
        for (uint32_t i = threadIdx.x; i < sweep_bins; i += YL_NT) h[i] = 0;
        __syncthreads();
        const uint64_t stride = (uint64_t)gridDim.x * YL_NT;
        uint64_t g = (uint64_t)blockIdx.x * YL_NT + threadIdx.x;
        WordWindows nx = word_fetch(a, g);
        for (; g < n_words; g += stride) {
            const WordWindows x = nx;
            if (g + stride < n_words) nx = word_fetch(a, g + stride);
            uint32_t run_bin = NONE, run_len = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                uint32_t key;
                const bool ok = word_window(x, j, K, key);
                const uint32_t b = ok ? key - lo : NONE;  // >= sweep_bins: not in this sweep (or no k-mer)
                if (b == run_bin) {
                    ++run_len;
                } else {
                    if (run_bin < sweep_bins) atomicAdd(&h[run_bin], run_len);
                    run_bin = b;
                    run_len = 1;
                }
            }
            if (run_bin < sweep_bins) atomicAdd(&h[run_bin], run_len);
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < sweep_bins; i += YL_NT) {
            const uint32_t v = h[i];
            if (v) atomicAdd(&counts[lo + i], (int32_t)v);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// X pass structure: per-tile digit histogram -> exclusive scan -> stable scatter
// ------------------------------------------------------------------------------------------------------

// Pass 0 histogram, straight from the packed bases.  Also accumulates the number of valid windows per record
// (the reference's total_count, src/libkmers.cpp:220-227).
//
// One WAVE per tile and no workgroup barrier: a lane takes one packed word (the 16 windows that start in it, shifted
// out of registers as in the y kernels -- a histogram does not care about the order), the wave has a 256-bin table of
// its own in LDS (the LDS executes one wave's operations in order), and after the tile's 8 groups of 64 words each lane
// moves 4 bins to the tile's histogram row.  rec_total: the record starts are bits of rec_flags; words without a start
// (all but ~1 in 150 on a transcriptome) add their windows to a wave-wide sum that is carried in a scalar from group to
// group and flushed with ONE atomic when a record ends or the tile does; words with starts split their windows between
// the records around them with one segmented sum over the wave's lanes.
#ifndef HB_BLOCKS
#define HB_BLOCKS 8192
#endif
#ifndef YS16_BLOCKS
#define YS16_BLOCKS 65536
#endif
#ifndef YS16_NT
#define YS16_NT 1024
#endif
#ifndef HW_SUB
#define HW_SUB 4
#endif
#ifndef HH_SUB
#define HH_SUB 4
#endif

#ifndef YS32_BLOCKS
#define YS32_BLOCKS 256
#endif
#ifndef HW_BLOCKS
#define HW_BLOCKS 4096
#endif
constexpr int HB_NT = 256, HB_NW = HB_NT / 64;
constexpr int HB_GROUPS = TILE / (64 * 16);  // groups of 64 words per tile

// bit j of the result = bits j .. j + K - 1 of v are all set (1 <= K <= 16); only bits 0..15 are meaningful for a 32-bit v
__device__ __forceinline__ uint32_t runs_of(uint32_t v, int K) {
    uint32_t r = v;
    int s = 1;
    while (2 * s <= K) {  // K is wave-uniform: scalar loop, at most 4 rounds
        r &= r >> s;
        s *= 2;
    }
    return r & (r >> (K - s));
}

template <int BITS>
__global__ __launch_bounds__(HB_NT) void k_hist_bases(BasesArgs a, uint32_t *tilehist, uint32_t n_tiles, uint32_t *rec_total) {
    constexpr int BINS = 1 << BITS;  // digit 0 = the low BITS bits of the k-mer id
    __shared__ uint32_t hs[HB_NW][BINS];
    const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t *const h = hs[w];
    const uint32_t n_waves = gridDim.x * HB_NW, wave = blockIdx.x * HB_NW + w;
    const int K = a.K;
    const uint32_t ksh = 32u - 2u * (uint32_t)K;
#pragma unroll
    for (int i = 0; i < BINS / 64; ++i) h[lane + 64 * i] = 0u;
    for (uint32_t tile = wave; tile < n_tiles; tile += n_waves) {
        uint32_t rec_base = __builtin_amdgcn_readfirstlane(a.tile_cb[tile]);  // record starts strictly before the group
        uint32_t carry = 0;  // valid windows of record rec_base - 1 seen in this tile and not yet added (wave-uniform)
        const uint64_t g0 = (uint64_t)tile * (TILE / 16) + lane;
        WordWindows nx = word_fetch(a, g0);
        uint32_t nfw = ((g0 >> 1) < a.valid_words) ? a.rec_flags[g0 >> 1] : 0u;
        for (int i = 0; i < HB_GROUPS; ++i) {
            const uint64_t g = g0 + (uint64_t)i * 64;
            const WordWindows x = nx;
            const uint32_t fl = (g & 1) ? (nfw >> 16) : (nfw & 0xFFFFu);  // record starts among positions 16 g .. 16 g + 15
            if (i + 1 < HB_GROUPS) {
                nx = word_fetch(a, g + 64);
                nfw = (((g + 64) >> 1) < a.valid_words) ? a.rec_flags[(g + 64) >> 1] : 0u;
            }
            const uint32_t okm = runs_of(x.v, K) & 0xFFFFu;  // bit j: window j of the word is a k-mer
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t top = j ? __builtin_amdgcn_alignbit(x.b0, x.b1, 32u - 2u * (uint32_t)j) : x.b0;
                if ((okm >> j) & 1u) atomicAdd(&h[(top >> ksh) & (uint32_t)(BINS - 1)], 1u);
            }
            // ---- valid windows per record
            const uint32_t call = (uint32_t)__popc(okm);
            const u64 hb = __ballot(fl != 0u);
            if (hb == 0ull) {  // no record starts in these 1024 positions
                uint32_t t = call;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) t += __shfl_xor(t, d);
                carry += __builtin_amdgcn_readfirstlane(t);
                continue;
            }
            // flagged words: windows before the first start (A) belong to the record that is open, windows from the
            // last start on (B) to the record it opens, windows between two starts of one word are added at once
            const uint32_t nfl = (uint32_t)__popc(fl);
            uint32_t cntA = 0, cntB = 0;
            if (fl) {
                cntA = (uint32_t)__popc(okm & ((fl & (0u - fl)) - 1u));
                cntB = (uint32_t)__popc(okm & ~((1u << (31 - __clz(fl))) - 1u));
            }
            const uint32_t sc = wave_incl_scan((fl ? 0u : call) | (nfl << 16));
            const uint32_t p_incl = sc & 0xFFFFu, f_incl = sc >> 16;
            if (nfl >= 2u) {  // records that start and end inside one word
                uint32_t f = fl, rec = rec_base + f_incl - nfl;
                while (f & (f - 1u)) {
                    const uint32_t lo = (uint32_t)__ffs((int)f) - 1u;
                    f &= f - 1u;
                    const uint32_t hi = (uint32_t)__ffs((int)f) - 1u;
                    const uint32_t c = (uint32_t)__popc(okm & ((1u << hi) - (1u << lo)));
                    if (c) atomicAdd(&rec_total[rec], c);
                    ++rec;
                }
            }
            const u64 above = (lane == 63u) ? 0ull : (hb & ~((2ull << lane) - 1ull));
            const uint32_t e = above ? (uint32_t)__ffsll((long long)above) - 1u : 64u;  // next flagged lane
            const uint32_t p_em1 = __shfl(p_incl, (int)((e - 1u) & 63u)), a_e = __shfl(cntA, (int)(e & 63u));
            const uint32_t seg = cntB + (p_em1 - p_incl) + (e < 64u ? a_e : 0u);  // (flagged lanes) the record opened here
            if (fl && e < 64u && seg) atomicAdd(&rec_total[rec_base + f_incl - 1u], seg);
            // head: the open record up to the first flagged lane
            const uint32_t e0 = __builtin_amdgcn_readfirstlane((uint32_t)__ffsll((long long)hb) - 1u);
            uint32_t head = carry + (uint32_t)__builtin_amdgcn_readlane((int)cntA, (int)e0);
            if (e0) head += (uint32_t)__builtin_amdgcn_readlane((int)p_incl, (int)(e0 - 1u));
            if (head && lane == 0u) atomicAdd(&rec_total[rec_base - 1u], head);  // (head > 0 implies a record is open)
            const uint32_t last = __builtin_amdgcn_readfirstlane(63u - (uint32_t)__clzll((long long)hb));
            carry = (uint32_t)__builtin_amdgcn_readlane((int)seg, (int)last);
            rec_base += (uint32_t)__builtin_amdgcn_readlane((int)f_incl, 63);
        }
        if (carry && lane == 0u) atomicAdd(&rec_total[rec_base - 1u], carry);
        // the wave's table -> the tile's row; zeroed for the next tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < BINS / 64; ++i) {
            const uint32_t v = h[lane + 64 * i];
            tilehist[(uint64_t)tile * BINS + lane + 64 * i] = v;
            h[lane + 64 * i] = 0u;
        }
    }
}

// y for K >= 9, pass 0 histogram: digit = low byte of the sort key id >> low_bits, per tile of BNT packed words
// (= 16 BNT positions, one word per thread, decoded from registers: the order inside a tile does not matter here).
// The chunk holds n_words8 = 512 * (its number of 8192-position tiles) words; a last tile may be partly outside.
template <int BNT>
__device__ __forceinline__ WordWindows yword_fetch(const BasesArgs &a, uint32_t tile, uint64_t n_words8) {
    const uint64_t lw = (uint64_t)tile * BNT + threadIdx.x;  // word inside the chunk
    return (lw < n_words8) ? word_fetch(a, a.tile0 * NT + lw) : WordWindows{0u, 0u, 0u};
}

template <int BNT>
__global__ __launch_bounds__(BNT) void k_hist_words_y(BasesArgs a, uint32_t *tilehist, uint64_t n_words8, uint32_t n_tiles) {
    // ONE table per tile for all 16 waves (two, used in turn: one is zeroed while the other is filled): LDS atomics of
    // different waves never meet inside one LDS instruction, so per-wave copies bought nothing and cost a 16-way sum by
    // a quarter of the threads between two barriers on every tile.
    // HW_SUB copies of the table, lane l adds to copy l % HW_SUB (stride RADIX + 1: other banks): lanes of one LDS
    // instruction that hold the same digit meet at one address less often (y12: 3.19 -> 2.90 ms per step with 4 copies,
    // 2.87 with 8)
    constexpr uint32_t SUBW = HW_SUB * (RADIX + 1);
    __shared__ uint32_t h[2][SUBW];
    const uint32_t tid = threadIdx.x, sub = (tid % HW_SUB) * (RADIX + 1);
    const uint32_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    if (t0 >= t1) return;
    WordWindows nx = yword_fetch<BNT>(a, t0, n_words8);
    const uint32_t dshift = 32u - 2u * (uint32_t)a.K + (uint32_t)a.low_bits;  // digit = (id >> low_bits) & 255, id = top 2K bits
    for (uint32_t i = tid; i < 2 * SUBW; i += BNT) (&h[0][0])[i] = 0;
    lds_sync();
    for (uint32_t tile = t0; tile < t1; ++tile) {
        uint32_t *const hr = h[(tile - t0) & 1u], *const hc = hr + sub;
        const WordWindows x = nx;
        if (tile + 1 < t1) nx = yword_fetch<BNT>(a, tile + 1, n_words8);
        // One predicated LDS add per window and nothing else: the run-merging loop this replaces (equal consecutive
        // digits added once) cost ~8 VALU + 9 SALU + 3 branches per window for the sake of poly-A.  A word whose 16
        // digits are all equal (24 equal bases from its first position on, every window valid) still adds once.
        const uint32_t okm = (runs_of(x.v, a.K) & 0xFFFFu);
        const bool mono = okm == 0xFFFFu && __builtin_amdgcn_alignbit(x.b0, x.b1, 30u) == x.b0 && ((x.b0 ^ x.b1) >> 16) == 0u;
        if (mono) {
            atomicAdd(&hc[(x.b0 >> dshift) & 255u], 16u);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t top = j ? __builtin_amdgcn_alignbit(x.b0, x.b1, 32u - 2u * (uint32_t)j) : x.b0;
                if ((okm >> j) & 1u) atomicAdd(&hc[(top >> dshift) & 255u], 1u);
            }
        }
        lds_sync();
        // thread t < 256 sums bin t of the copies and clears them for the tile after next: no second barrier, the next
        // tile's atomics go to the other table and these words are written again only after its barrier
        if (tid < RADIX) {
            uint32_t sum = 0;
#pragma unroll
            for (uint32_t c = 0; c < HW_SUB; ++c) {
                sum += hr[c * (RADIX + 1) + tid];
                hr[c * (RADIX + 1) + tid] = 0u;
            }
            tilehist[(uint64_t)tile * RADIX + tid] = sum;
        }
    }
}

// y for K >= 9, pass 0 partition: sort key = id >> low_bits, payload = the low bits, no records.  K <= 12: 2-byte
// entries (the payload; the sort key is the region); K >= 13: 4-byte entries ((sort key >> 8) << 24 | low bits).
// Order-free: a histogram follows, not a run-length reduce, so the windows come from registers (one packed word per
// thread) and the place of an entry inside its (digit, tile) group is what ONE returning LDS atomic hands out.
// Round 3: the tile's digit counts are not counted again here -- they are the difference of two rows of the scanned
// tile histogram (row t + 1 minus row t; the region table closes the last tile), so ONE wave turns them into the
// digit starts of tile t + 1 while tile t is written out, the atomic returns the entry's final slot in the staging
// array and the entry is stored there at once: no rank registers, no second decode, two barriers per tile instead
// of six (2.28 -> see DESIGN.md section 4 per 2^31 positions with 2-byte entries).
// 1024-thread workgroups on tiles of 16,384 positions: the write-out is store-bound and its runs are twice as long as
// with 8192-position tiles -- 2.17 -> 1.79 ms with 4-byte entries (round 1) and 3.07 -> 2.60 ms per 2^31 positions
// with 2-byte entries (round 2: 64-byte runs wrote every 128-byte line in two halves, 2.0x WRITE_SIZE in the PMC pass).
template <class ET, int BNT, int WPT>
__global__ __launch_bounds__(BNT) void k_yscatter(BasesArgs a, const uint32_t *tilebase, const uint32_t *regions, uint64_t n_words8,
                                                  uint32_t n_tiles, ET *out) {
    constexpr int BNW = BNT / 64, BTILE = BNT * WPT * 16;  // WPT packed words per thread
    // staged as 32-bit words, the entry with its digit in bits 16..23 (free in both forms: a 2-byte payload, or
    // (key byte << 24 | at most 16 low bits)): ONE LDS store per window where rounds 1-2 kept a digit byte beside every
    // entry, and the write-out finds the entry's global offset from the word it reads anyway (y12: 10.0 -> 9.6 ms per
    // step; deriving the digit from the digit ENDS in a per-wave walk instead was built too: 10.6 - 13.8 ms)
    __shared__ __attribute__((aligned(16))) uint32_t sent[BTILE];
    __shared__ __attribute__((aligned(16))) uint32_t cnt[2][RADIX], gadj[2][RADIX];
    __shared__ uint32_t s_count[2];
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    if (t0 >= t1) return;
    const uint32_t low_mask = (1u << a.low_bits) - 1u;
    const uint32_t dshift = 32u - 2u * (uint32_t)a.K + (uint32_t)a.low_bits, ksh = 32u - 2u * (uint32_t)a.K;
    auto fetch = [&](WordWindows *dst, uint32_t tile) {
#pragma unroll
        for (int wi = 0; wi < WPT; ++wi) {
            const uint64_t lw = (uint64_t)tile * (BNT * WPT) + (uint32_t)wi * BNT + tid;  // word inside the chunk
            dst[wi] = (lw < n_words8) ? word_fetch(a, a.tile0 * NT + lw) : WordWindows{0u, 0u, 0u};
        }
    };
    WordWindows nx[WPT];
    fetch(nx, t0);
    // the scan wave (the last one) keeps two rows of the scanned histogram, digits 4 lane .. 4 lane + 3: row_lo = the
    // bases of the tile it sets up next, row_hi = the row after it
    const bool scan_wave = w == BNW - 1;
    auto load_row = [&](uint32_t t) -> uint4 {  // t <= n_tiles
        if (t < n_tiles) return reinterpret_cast<const uint4 *>(tilebase + (uint64_t)t * RADIX)[lane];
        return make_uint4(regions[4 * lane + 1], regions[4 * lane + 2], regions[4 * lane + 3], regions[4 * lane + 4]);
    };
    uint4 row_lo = make_uint4(0, 0, 0, 0), row_hi = row_lo;
    auto setup = [&](uint32_t buf) {  // digit starts of the tile whose rows are row_lo / row_hi
        const uint32_t c0 = row_hi.x - row_lo.x, c1 = row_hi.y - row_lo.y, c2 = row_hi.z - row_lo.z, c3 = row_hi.w - row_lo.w;
        const uint32_t sum = c0 + c1 + c2 + c3, inc = wave_incl_scan(sum), s0 = inc - sum, s1 = s0 + c0, s2 = s1 + c1, s3 = s2 + c2;
        reinterpret_cast<uint4 *>(cnt[buf])[lane] = make_uint4(s0, s1, s2, s3);
        reinterpret_cast<uint4 *>(gadj[buf])[lane] = make_uint4(row_lo.x - s0, row_lo.y - s1, row_lo.z - s2, row_lo.w - s3);
        if (lane == 63u) s_count[buf] = inc;
    };
    if (scan_wave) {
        row_lo = load_row(t0);
        row_hi = load_row(t0 + 1);
        setup(0);
        row_lo = row_hi;
    }
    lds_sync();
    for (uint32_t tile = t0; tile < t1; ++tile) {
        const uint32_t buf = (tile - t0) & 1u;
        WordWindows xs[WPT];
#pragma unroll
        for (int wi = 0; wi < WPT; ++wi) xs[wi] = nx[wi];
        if (tile + 1 < t1) {
            fetch(nx, tile + 1);
            if (scan_wave) row_hi = load_row(tile + 2);
        }
#pragma unroll
        for (int wi = 0; wi < WPT; ++wi) {
            const WordWindows x = xs[wi];
            const uint32_t okm = runs_of(x.v, a.K) & 0xFFFFu;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if ((okm >> j) & 1u) {
                    const uint32_t top = j ? __builtin_amdgcn_alignbit(x.b0, x.b1, 32u - 2u * (uint32_t)j) : x.b0;
                    const uint32_t d = (top >> dshift) & 255u, key = top >> ksh;
                    const uint32_t slot = atomicAdd(&cnt[buf][d], 1u);
                    if constexpr (sizeof(ET) == 2)
                        sent[slot] = (key & low_mask) | (d << 16);
                    else
                        sent[slot] = (((key >> a.low_bits) >> 8) << 24) | (d << 16) | (key & low_mask);
                }
            }
        }
        lds_sync();
        if (scan_wave && tile + 1 < t1) {  // the next tile's digit starts, while this one leaves
            setup(buf ^ 1u);
            row_lo = row_hi;
        }
        const uint32_t tile_count = s_count[buf];
#if !defined(YS_ABL)  // (YS_ABL: timing-only builds without the write-out)
        for (uint32_t i = tid; i < tile_count; i += BNT) {
            const uint32_t v = sent[i];
            out[gadj[buf][(v >> 16) & 255u] + i] = (ET)(v & 0xFF00FFFFu);
        }
#else
        if (tile_count == 0xFFFFFFFFu) out[tid] = (ET)sent[tid];
#endif
        lds_sync();
    }
}

// Later-pass histogram over stored entries.  Persistent: workgroup b walks a contiguous range of tiles and loads
// tile t+1 while tile t is counted (two LDS tables: one is zeroed while the other is filled); tiles past the entry count get zeros.
template <class ET, int BITS = 8>
__global__ __launch_bounds__(NT) void k_hist_entries(const ET *ent, const uint32_t *p_count, int shift, uint32_t *tilehist,
                                                     uint32_t n_tiles) {
    constexpr bool NARROW = sizeof(ET) == 4;
    constexpr int RADIX = 1 << BITS;  // (shadows the 256 of the file scope: this kernel's number of bins)
    static_assert(RADIX <= NT && (!NARROW || BITS == 8), "one thread per bin when a tile's row is written");
    __shared__ uint32_t h[2][RADIX];  // one table per tile for all waves (see k_hist_words_y)
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t E = *p_count;
    const uint32_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    if (t0 >= t1) return;
    ET e[IPT];
    auto load_tile = [&](uint32_t t) {
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint64_t i = (uint64_t)t * TILE + w * WCHUNK + j * 64 + lane;
            e[j] = (i < E) ? __builtin_nontemporal_load(&ent[i]) : (ET)0;  // (same-run A/B: 0.365 -> 0.315, 1.87 -> 1.72 ms)
        }
    };
    load_tile(t0);
    for (uint32_t i = tid; i < 2 * RADIX; i += NT) (&h[0][0])[i] = 0;
    lds_sync();
    for (uint32_t tile = t0; tile < t1; ++tile) {
        uint32_t *const hc = h[(tile - t0) & 1u];
        uint32_t *const hn = h[((tile - t0) & 1u) ^ 1u];
        const uint64_t base = (uint64_t)tile * TILE;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint64_t i = base + w * WCHUNK + j * 64 + lane;
            uint32_t d;
            if constexpr (NARROW)
                d = (uint32_t)e[j] >> 24;
            else
                d = (entry_key((u64)e[j]) >> shift) & (uint32_t)(RADIX - 1);
            if (i < E) atomicAdd(&hc[d], 1u);
        }
        if (tile + 1 < t1) load_tile(tile + 1);  // e[] is free again
        if (tid < RADIX) hn[tid] = 0;  // the other table, for the next tile
        lds_sync();
        // (no second barrier: thread tid zeroes hc[tid] itself, in the next iteration, after this read)
        if (tid < RADIX) tilehist[(uint64_t)tile * RADIX + tid] = hc[tid];
    }
}

// Stable scatter by the 8-bit digit of the k-mer id at `shift`.  Persistent: workgroup b walks the contiguous tile
// range [b*per, (b+1)*per) and fetches the next tile while the current one is ranked and written.
//   FROM_BASES: entries are generated from the packed store (pass 0); otherwise read from `in`.
//   tilebase = exclusive scan of the digit-major tile histogram: global position of the first entry of (digit, tile).
//   ORDER_FREE = 2 (second pass of the y path for K >= 13): the entries of a (digit, tile) may land in any order -- a
//   histogram follows, not a run-length reduce -- so the rank is simply what one returning LDS atomic on the tile's
//   digit counter hands out; only in tiles that lie inside one region of the input, i.e. all but ~256 of them (the
//   output must stay ordered by the byte the first pass sorted on).
template <bool FROM_BASES, class ET, int ORDER_FREE = 0>
__global__ __launch_bounds__(NT, 4) void k_scatter(BasesArgs a, const ET *in, const uint32_t *p_count, int shift,
                                                   const uint32_t *tilebase, uint32_t n_tiles, ET *out, const uint32_t *regions) {
    constexpr bool NARROW = sizeof(ET) == 4;
    static_assert(ORDER_FREE == 0 || (ORDER_FREE == 2 && !FROM_BASES && NARROW), "mode 2 is for the y path's second pass");
    // entry staging; its first NW*RADIX 8-byte words double as the match masks (below)
    __shared__ __attribute__((aligned(16))) ET sent[TILE];
    __shared__ uint8_t saux[NARROW ? TILE : 4];  // narrow: pass 0 the partition digit, pass 1 the byte that replaces it
    __shared__ uint32_t sreg[NARROW ? RADIX + 1 : 1];
    __shared__ uint32_t wcnt[NW][RADIX];
    static_assert(NW * RADIX * 8 <= TILE * 4, "match masks alias the entry staging");
    __shared__ uint32_t gadj[RADIX];    // global base of (digit, tile) minus the digit's first tile-local slot
    __shared__ uint32_t stmp[NW];
    __shared__ TileLds L;               // only touched when FROM_BASES

    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    uint32_t E = 0, nt = n_tiles;
    if (!FROM_BASES) {
        E = *p_count;
        nt = (uint32_t)(((uint64_t)E + TILE - 1) / TILE);
    }
    const uint32_t per = (nt + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = (t0 + per < nt) ? t0 + per : nt;
    if (t0 >= t1) return;
    u64 *const smask = reinterpret_cast<u64 *>(sent);

    ET e[IPT];
    uint32_t aux[IPT];   // narrow pass 1: the region (= previous digit) of each entry, found when the load is issued
    TilePf pf;
    // entries of tile t (the wave's 16 rounds) into e[]; narrow: their regions into aux[].  A wave's 1024 entries lie
    // in one region nearly always (256 regions): one wave-uniform test instead of an LDS walk per entry.
    auto load_tile = [&](uint32_t t) {
        const uint32_t c0 = t * TILE + w * WCHUNK;  // the wave's first entry
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint64_t i = (uint64_t)c0 + j * 64 + lane;
            e[j] = (i < E) ? in[i] : (ET)0;
        }
        if constexpr (NARROW) {
            const uint32_t r0 = region_search(sreg, c0 < E ? c0 : (E ? E - 1u : 0u));  // wave-uniform
            const bool one = sreg[r0 + 1] >= c0 + WCHUNK || sreg[r0 + 1] >= E;
            if (one) {
#pragma unroll
                for (int j = 0; j < IPT; ++j) aux[j] = r0;
            } else {
                uint32_t r = r0;
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const uint32_t i = c0 + j * 64 + lane;
                    if (i < E) r = region_from(sreg, i, r);
                    aux[j] = r;
                }
            }
        }
    };
    if (FROM_BASES) {
        pf = tile_fetch<true>(a, t0);
    } else {
        if constexpr (NARROW) {
            for (uint32_t i = tid; i <= RADIX; i += NT) sreg[i] = regions[i];
            lds_sync();
        }
        load_tile(t0);
    }
    uint32_t gb = (tid < RADIX) ? tilebase[(uint64_t)t0 * RADIX + tid] : 0u;

    KK_STAMP_BEGIN();
    for (uint32_t tile = t0; tile < t1; ++tile) {
        const uint64_t base = (uint64_t)tile * TILE;
        bool of = false;  // workgroup-uniform
        if constexpr (ORDER_FREE == 2) {
            const uint32_t r = region_search(sreg, (uint32_t)base);
            const uint64_t tile_last = (base + TILE < E) ? base + TILE : E;
            of = sreg[r + 1] >= tile_last;
        }
        if (of) {
            if (tid < RADIX) wcnt[0][tid] = 0;
        } else {
            for (uint32_t i = tid; i < NW * RADIX; i += NT) {
                (&wcnt[0][0])[i] = 0;
                smask[i] = 0ull;  // the match masks must start (and, by construction, end) a tile all zero
            }
        }
        if (FROM_BASES) {
            tile_commit<true>(L, pf, stmp);
            if (tile + 1 < t1) pf = tile_fetch<true>(a, tile + 1);
        } else {
            lds_sync();
        }
        // lane offset inside the tile, made opaque once per tile: otherwise the per-round LDS addresses and shift
        // counts (loop-invariant across tiles) are hoisted out of the tile loop and spilled around it
        uint32_t wl = w * WCHUNK + lane;
        asm volatile("" : "+v"(wl));
        KK_STAMP(0);  // zero + commit/barrier
        const uint32_t n_valid = FROM_BASES ? (uint32_t)TILE : (uint32_t)((E - base < TILE) ? E - base : TILE);
        uint32_t *const wrow = &wcnt[w][0];
        u64 *const wmask = &smask[w * RADIX];
        uint32_t rank[IPT];  // bits 0..15 rank inside (wave, digit); narrow: bits 16..23 the aux byte; NONE = no entry
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            bool ok;
            uint32_t d, ab = 0;
            if (FROM_BASES) {
                const uint32_t o = wl + j * 64;
                uint32_t key;
                uint32_t col;
                ok = window_at(L, o, a.K, key);
                col = record_at(L, o);  // (a wave-uniform readlane lookup costs 3 VGPRs here: spills, slower)
                asm volatile("" : "+v"(col));  // finish the column now: sunk to its use it keeps 4 LDS words per round alive
                if constexpr (NARROW) {
                    d = key & 255u;
                    ab = d;
                    e[j] = ((key >> 8) << 24) | col;
                } else {
                    d = __builtin_amdgcn_ubfe(key, (uint32_t)shift, 8u);
                    e[j] = make_entry(key, col);
                }
            } else {
                ok = wl + j * 64 < n_valid;
                if constexpr (NARROW) {
                    d = (uint32_t)e[j] >> 24;
                    ab = aux[j];
                } else {
                    d = __builtin_amdgcn_ubfe(entry_key((u64)e[j]), (uint32_t)shift, 8u);
                }
            }
            // Match by LDS: every ok lane ORs its lane bit into the 64-bit word of (wave, digit); reading the word
            // back gives the lanes of this round that share the digit.  The LDS executes a wave's instructions in
            // order, so the read sees all 64 ORs; OR is order-independent, so the ranks are deterministic.  The lowest
            // lane of each group advances the (wave, digit) counter and clears the word for the next round.
            // (An 8-ballot match in registers gives the same masks for ~45 more VALU instructions per round.)
            if constexpr (ORDER_FREE != 0) {
                if (of) {
                    uint32_t r = 0;
                    if (ok) r = atomicAdd(&wcnt[0][d], 1u);
                    rank[j] = ok ? (r | (ab << 16)) : NONE;
                    continue;
                }
            }
            u64 m = 0;
            uint32_t before = 0;
            if (ok) {
                atomicOr(&wmask[d], 1ull << lane);
                m = wmask[d];
                before = wrow[d];
            }
            const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
            rank[j] = ok ? ((before + below) | (ab << 16)) : NONE;
            if (ok && below == 0) {
                wrow[d] = before + (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
                wmask[d] = 0ull;
            }
            asm volatile("" ::: "memory");  // the next round must re-read wcnt
            __builtin_amdgcn_sched_barrier(0);  // keep rounds apart: interleaving them only lengthens live ranges
        }
        KK_STAMP(1);  // ranking rounds (wave 0)
        lds_sync();
        KK_STAMP(2);  // barrier after ranking
        // digit d: exclusive scan of the per-wave counts over waves, then of the tile totals over digits;
        // afterwards wcnt[w][d] = tile-local slot of the first entry of (wave w, digit d)
        uint32_t dtot = 0;
        uint32_t wc[NW];
        if (tid < RADIX) {
            if (of) {
                dtot = wcnt[0][tid];
            } else {
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    wc[i] = dtot;
                    dtot += wcnt[i][tid];
                }
            }
        }
        uint32_t tile_count;
        const uint32_t dstart = block_excl_scan<true>(dtot, stmp, tile_count);  // (the barrier after the ranking)
        if (tid < RADIX) {
            if (of) {
                wcnt[0][tid] = dstart;
            } else {
#pragma unroll
                for (int i = 0; i < NW; ++i) wcnt[i][tid] = dstart + wc[i];
            }
            gadj[tid] = gb - dstart;
            if (tile + 1 < t1) gb = tilebase[(uint64_t)(tile + 1) * RADIX + tid];
        }
        lds_sync();
        KK_STAMP(3);  // digit scan
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            if (rank[j] != NONE) {
                uint32_t d;
                if constexpr (NARROW)
                    d = FROM_BASES ? ((rank[j] >> 16) & 255u) : ((uint32_t)e[j] >> 24);
                else
                    d = __builtin_amdgcn_ubfe(entry_key((u64)e[j]), (uint32_t)shift, 8u);
                const uint32_t slot = (of ? wcnt[0][d] : wrow[d]) + (rank[j] & 0xFFFFu);
                sent[slot] = e[j];
                if constexpr (NARROW) saux[slot] = (uint8_t)(rank[j] >> 16);
            }
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (!FROM_BASES && tile + 1 < t1 && !KK_ABLATE(2)) {  // e[] is free again: fetch the next tile behind the write-out
            load_tile(tile + 1);
        }
        KK_STAMP(4);  // LDS scatter + prefetch issue
        lds_sync();
        KK_STAMP(5);  // barrier
        for (uint32_t i = tid; i < tile_count; i += NT) {
            ET v = sent[i];
            uint32_t d;
            if constexpr (NARROW) {
                const uint32_t ab = saux[i];
                if (FROM_BASES) {
                    d = ab;
                } else {
                    d = (uint32_t)v >> 24;
                    v = ((uint32_t)v & 0xFFFFFFu) | (ab << 24);  // the digit sorted on is implied by the region from now on
                }
            } else {
                d = __builtin_amdgcn_ubfe(entry_key((u64)v), (uint32_t)shift, 8u);
            }
            if (KK_ABLATE(1)) continue;
            out[gadj[d] + i] = v;
        }
        KK_STAMP(6);  // write-out
        lds_sync();
        KK_STAMP(7);  // end barrier
    }
    KK_STAMP_END(FROM_BASES ? 0 : 8);
}

// Later passes of the 8-byte-entry path (K >= 9) with digits of up to 9 bits: the same stable scatter as k_scatter<false>
// (OR-mask match per (wave, digit), LDS staging, coalesced runs out), 2^BITS bins.  Round 3: K = 13, 14 take three
// passes (9 + 9 + 8 and 10 + 9 + 9 bits) instead of four of 8 bits, K = 9 two instead of three -- a pass costs a
// histogram plus a scatter whatever its digit width, as long as two workgroups still fit a CU: with 512 bins the
// per-(wave, digit) counters are 16-bit (a count is at most 1024, a tile-local slot below 8192) and the 64-bit match
// words (32 KB) still alias the 64 KB entry staging; 1024 bins would need all of it and one workgroup per CU.
// `bits` <= BITS is the width of this pass's digit (a narrower last digit leaves bins empty).
template <int BITS>
__global__ __launch_bounds__(NT, 4) void k_scatter_wide(const u64 *in, const uint32_t *p_count, int shift, int bits,
                                                        const uint32_t *tilebase, u64 *out) {
    constexpr int BINS = 1 << BITS;
    using CT = typename std::conditional<(BITS > 8), uint16_t, uint32_t>::type;
    static_assert(BINS <= NT, "one thread per digit in the scan");
    __shared__ __attribute__((aligned(16))) u64 sent[TILE];
    __shared__ CT wcnt[NW][BINS];
    static_assert(NW * BINS * 8 <= TILE * 8, "match masks alias the entry staging");
    __shared__ uint32_t gadj[BINS];
    __shared__ uint32_t stmp[NW];
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t E = *p_count;
    const uint32_t nt = (uint32_t)(((uint64_t)E + TILE - 1) / TILE);
    const uint32_t per = (nt + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = (t0 + per < nt) ? t0 + per : nt;
    if (t0 >= t1) return;
    u64 *const smask = sent;
    u64 e[IPT];
    auto load_tile = [&](uint32_t t) {
        const uint32_t c0 = t * TILE + w * WCHUNK;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint64_t i = (uint64_t)c0 + j * 64 + lane;
            e[j] = (i < E) ? in[i] : 0ull;
        }
    };
    load_tile(t0);
    uint32_t gb = (tid < BINS) ? tilebase[(uint64_t)t0 * BINS + tid] : 0u;
    for (uint32_t tile = t0; tile < t1; ++tile) {
        const uint64_t base = (uint64_t)tile * TILE;
        for (uint32_t i = tid; i < NW * BINS; i += NT) {
            (&wcnt[0][0])[i] = 0;
            smask[i] = 0ull;
        }
        lds_sync();
        uint32_t wl = w * WCHUNK + lane;
        asm volatile("" : "+v"(wl));
        const uint32_t n_valid = (uint32_t)((E - base < TILE) ? E - base : TILE);
        CT *const wrow = &wcnt[w][0];
        u64 *const wmask = &smask[w * BINS];
        uint32_t rank[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const bool ok = wl + j * 64 < n_valid;
            const uint32_t d = __builtin_amdgcn_ubfe(entry_key(e[j]), (uint32_t)shift, (uint32_t)bits);
            u64 m = 0;
            uint32_t before = 0;
            if (ok) {
                atomicOr(&wmask[d], 1ull << lane);
                m = wmask[d];
                before = wrow[d];
            }
            const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
            rank[j] = ok ? (before + below) : NONE;
            if (ok && below == 0) {
                wrow[d] = (CT)(before + (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi));
                wmask[d] = 0ull;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_sync();
        uint32_t dtot = 0;
        uint32_t wc[NW];
        if (tid < BINS) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                wc[i] = dtot;
                dtot += wcnt[i][tid];
            }
        }
        uint32_t tile_count;
        const uint32_t dstart = block_excl_scan<true>(dtot, stmp, tile_count);
        if (tid < BINS) {
#pragma unroll
            for (int i = 0; i < NW; ++i) wcnt[i][tid] = (CT)(dstart + wc[i]);
            gadj[tid] = gb - dstart;
            if (tile + 1 < t1) gb = tilebase[(uint64_t)(tile + 1) * BINS + tid];
        }
        lds_sync();
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            if (rank[j] != NONE) {
                const uint32_t d = __builtin_amdgcn_ubfe(entry_key(e[j]), (uint32_t)shift, (uint32_t)bits);
                sent[(uint32_t)wrow[d] + rank[j]] = e[j];
            }
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (tile + 1 < t1) load_tile(tile + 1);
        lds_sync();
        for (uint32_t i = tid; i < tile_count; i += NT) {
            const u64 v = sent[i];
            const uint32_t d = __builtin_amdgcn_ubfe(entry_key(v), (uint32_t)shift, (uint32_t)bits);
            out[gadj[d] + i] = v;
        }
        lds_sync();
    }
}

// X pass 0, round 3: the order INSIDE one column does not matter.
//
// The later passes are stable and the run-length reduce only asks that the instances of one (k-mer, column) end up
// adjacent and that, inside a row, the columns ascend.  Two instances of the SAME column that share the pass-0 digit are
// told apart only by their higher digits, which the later (stable) passes sort on anyway, so pass 0 may leave them in any
// order: what it has to keep is the order BETWEEN columns.  That removes the two things that made k_scatter<true> cost
// 19 LDS instructions per 64 windows: the windows come out of registers (a thread shifts the 16 windows that start in
// its packed word, as the y kernels do -- no LDS tile, no record lookup per window) and the rank inside (wave, digit) is
// what ONE returning LDS atomic on the wave's own counter hands out.  The LDS runs a wave's instructions in order, so
// all it takes to keep the columns apart is to issue the windows of one column before those of the next: a wave whose
// 1024 positions hold no record start (two in three on a transcriptome) ranks in one sweep of 16 atomics, one with s
// starts in s + 1 sweeps.  Waves stay ordered through their private counters (scan over the waves, as before).
// Tiles are BNT * 16 positions = BNT / 512 rows of the (8192-position) tile histogram: in the scanned table the rows of
// one digit follow each other, so the first row's base serves the whole tile.
// BITS = width of digit 0 (the low bits of the k-mer id): 8 for the 4-byte entries of K <= 8; 8, 9 or 10 for the 8-byte
// entries of K >= 9 (see k_scatter_wide).  Above 8 bits the per-(wave, digit) counters are packed two to a word (a wave
// has 1024 windows: a count fits 16 bits, and so does a tile-local slot), and at 10 bits the staged entry is 4 + 2 bytes
// instead of 8 -- (id >> 10) << 14 | column relative to the tile's first, and the digit beside it; a tile of 8192
// positions holds at most 8192 record starts, and 2 K - 10 <= 18 bits are left of the id for K <= 14 -- so that the
// 1024 counters still leave room for two workgroups per CU.
template <class ET, int BNT, int BITS>
__global__ __launch_bounds__(BNT) void k_xscatter0(BasesArgs a, const uint32_t *tilebase, uint32_t n_tiles8, ET *out) {
    constexpr bool NARROW = sizeof(ET) == 4;
    constexpr int BINS = 1 << BITS;
    constexpr bool PACKED = BITS > 8, COMPACT = BITS == 10;
    constexpr int BNW = BNT / 64, BTILE = BNT * 16, TPB = BTILE / TILE;
    constexpr int CW = PACKED ? BINS / 2 : BINS;  // counter words per wave
    static_assert(BNT % NT == 0, "a tile is a whole number of histogram rows");
    static_assert(!NARROW || BITS == 8, "4-byte entries keep ONE other byte of the id");
    static_assert(!COMPACT || (BTILE == 8192 && !NARROW), "the compact staging holds a 14-bit relative column");
    static_assert(CW <= BNT, "one counter word per thread in the scan");
    using ST = typename std::conditional<COMPACT, uint32_t, ET>::type;
    __shared__ __attribute__((aligned(16))) ST sent[BTILE];
    __shared__ uint8_t saux[NARROW ? BTILE : 4];      // narrow: the digit of every staged entry (the entry holds the OTHER byte)
    __shared__ uint16_t saux16[COMPACT ? BTILE : 2];  // compact: the same, 10 bits
    __shared__ uint32_t wcnt[BNW][CW];
    __shared__ uint32_t gadj[BINS], swave[BNW], sdig[BNW];
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t nbt = (n_tiles8 + TPB - 1) / TPB;
    const uint32_t per = (nbt + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = (t0 + per < nbt) ? t0 + per : nbt;
    if (t0 >= t1) return;
    const int K = a.K;
    auto fetch_flags = [&](uint64_t g) -> uint32_t {  // record starts among positions 16 g .. 16 g + 15
        const uint32_t fw = ((g >> 1) < a.valid_words) ? a.rec_flags[g >> 1] : 0u;
        return (g & 1) ? (fw >> 16) : (fw & 0xFFFFu);
    };
    // the scanned histogram row of a tile: thread t takes bin t, or (packed) bins 2 t and 2 t + 1
    auto fetch_base = [&](uint32_t tile, uint32_t &b0, uint32_t &b1) {
        const uint32_t *row = tilebase + (uint64_t)tile * TPB * BINS;
        if constexpr (PACKED) {
            if (tid < (uint32_t)CW) {
                const uint2 v = *reinterpret_cast<const uint2 *>(row + 2 * tid);
                b0 = v.x, b1 = v.y;
            }
        } else {
            if (tid < (uint32_t)BINS) b0 = row[tid];
        }
    };
    WordWindows nx = word_fetch(a, (uint64_t)t0 * BNT + tid);
    uint32_t nfl = fetch_flags((uint64_t)t0 * BNT + tid);
    uint32_t ncb = a.tile_cb[(uint64_t)t0 * TPB];
    uint32_t gb0 = 0, gb1 = 0;
    fetch_base(t0, gb0, gb1);
    uint32_t *const wrow = &wcnt[w][0];
    // rank of a window inside (wave, digit) = what the returning atomic on the wave's counter hands out
    auto take_rank = [&](uint32_t d) -> uint32_t {
        if constexpr (PACKED) {
            const uint32_t sh = (d & 1u) * 16u;
            return (atomicAdd(&wrow[d >> 1], 1u << sh) >> sh) & 0xFFFFu;
        } else {
            return atomicAdd(&wrow[d], 1u);
        }
    };
    for (uint32_t tile = t0; tile < t1; ++tile) {
        for (uint32_t i = tid; i < BNW * CW; i += BNT) (&wcnt[0][0])[i] = 0;
        const WordWindows x = nx;
        const uint32_t fl = nfl, cb_tile = ncb;
        if (tile + 1 < t1) {
            nx = word_fetch(a, (uint64_t)(tile + 1) * BNT + tid);
            nfl = fetch_flags((uint64_t)(tile + 1) * BNT + tid);
            ncb = a.tile_cb[(uint64_t)(tile + 1) * TPB];
        }
        // record starts before this thread's word: in the wave (-> sweeps), in the tile (-> column numbers)
        const uint32_t nf = (uint32_t)__popc(fl);
        const uint32_t finc = wave_incl_scan(nf);
        const uint32_t wave_starts = (uint32_t)__builtin_amdgcn_readlane((int)finc, 63);
        if (lane == 63) swave[w] = finc;
        lds_sync();
        uint32_t before_w = 0;
#pragma unroll
        for (int i = 0; i < BNW; ++i)
            if ((uint32_t)i < w) before_w += swave[i];
        const uint32_t lc0 = finc - nf;                    // starts in this wave's chunk before the word
        const uint32_t rel0 = before_w + lc0;              // starts in this TILE before the word
        const uint32_t colm1 = cb_tile - 1u;               // column of a position in front of the tile's first start
        const uint32_t okm = runs_of(x.v, K) & 0xFFFFu;    // bit j: window j of the word is a k-mer
        uint32_t rank[16];  // bits 0..15 the rank inside (wave, digit), bits 16..25 the digit; NONE = no k-mer
        auto digit_of = [&](int j) -> uint32_t {
            const uint32_t top = j ? __builtin_amdgcn_alignbit(x.b0, x.b1, 32u - 2u * (uint32_t)j) : x.b0;
            return (top >> (32u - 2u * (uint32_t)K)) & (uint32_t)(BINS - 1);
        };
#ifdef XS0_NOPHASE  // diagnostic builds (wrong results)
        if (true) {
#else
        if (wave_starts == 0) {  // (wave-uniform) one column: any order will do
#endif
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                rank[j] = NONE;
                if ((okm >> j) & 1u) {
                    const uint32_t d = digit_of(j);
                    rank[j] = take_rank(d) | (d << 16);
                }
            }
        } else {  // column after column (sweep ph takes the windows that follow exactly ph starts of the wave's chunk)
#pragma unroll
            for (int j = 0; j < 16; ++j) rank[j] = NONE;
            for (uint32_t ph = 0; ph <= wave_starts; ++ph) {
                if (__ballot(ph >= lc0 && ph <= lc0 + nf) == 0ull) continue;  // no word of the wave reaches this column
                // (opaque copies: otherwise the 16 digits and 16 start counts are hoisted out of the sweep loop and spill)
                uint32_t b0 = x.b0, b1 = x.b1, flo = fl;
                asm volatile("" : "+v"(b0), "+v"(b1), "+v"(flo));
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t lc = lc0 + (uint32_t)__popc(flo & ((2u << j) - 1u));
                    if (((okm >> j) & 1u) && lc == ph) {
                        const uint32_t top = j ? __builtin_amdgcn_alignbit(b0, b1, 32u - 2u * (uint32_t)j) : b0;
                        const uint32_t d = (top >> (32u - 2u * (uint32_t)K)) & (uint32_t)(BINS - 1);
                        rank[j] = take_rank(d) | (d << 16);
                    }
                }
            }
        }
        lds_sync();
        // exclusive scan of the per-wave counts over the waves (per digit), then of the tile totals over the digits:
        // afterwards the wave's counter of digit d holds the tile-local slot of the wave's first entry of d
        uint32_t t_lo = 0, t_hi = 0;  // tile totals of this thread's digit(s)
        if (tid < (uint32_t)CW) {
#pragma unroll
            for (int i = 0; i < BNW; ++i) {
                const uint32_t c = wcnt[i][tid];
                if constexpr (PACKED) {
                    wcnt[i][tid] = t_lo | (t_hi << 16);
                    t_lo += c & 0xFFFFu;
                    t_hi += c >> 16;
                } else {
                    wcnt[i][tid] = t_lo;
                    t_lo += c;
                }
            }
        }
        const uint32_t dtot = t_lo + t_hi;
        const uint32_t dinc = wave_incl_scan(dtot);
        if (lane == 63) sdig[w] = dinc;
        lds_sync();
        uint32_t dbase = 0, tile_count = 0;
#pragma unroll
        for (int i = 0; i < BNW; ++i) {
            const uint32_t t = sdig[i];
            if ((uint32_t)i < w) dbase += t;
            tile_count += t;
        }
        if (tid < (uint32_t)CW) {
            const uint32_t s_lo = dbase + dinc - dtot, s_hi = s_lo + t_lo;
#pragma unroll
            for (int i = 0; i < BNW; ++i) wcnt[i][tid] += PACKED ? (s_lo | (s_hi << 16)) : s_lo;
            if constexpr (PACKED) {
                gadj[2 * tid] = gb0 - s_lo;
                gadj[2 * tid + 1] = gb1 - s_hi;
            } else {
                gadj[tid] = gb0 - s_lo;
            }
        }
        if (tile + 1 < t1) fetch_base(tile + 1, gb0, gb1);
        lds_sync();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (rank[j] != NONE) {
                const uint32_t top = j ? __builtin_amdgcn_alignbit(x.b0, x.b1, 32u - 2u * (uint32_t)j) : x.b0;
                const uint32_t key = top >> (32u - 2u * (uint32_t)K);
                const uint32_t rel = rel0 + (uint32_t)__popc(fl & ((2u << j) - 1u));  // starts of the tile at or before the window
                const uint32_t d = (rank[j] >> 16) & (uint32_t)(BINS - 1);
                uint32_t start;
                if constexpr (PACKED)
                    start = (wrow[d >> 1] >> ((d & 1u) * 16u)) & 0xFFFFu;
                else
                    start = wrow[d];
                const uint32_t slot = start + (rank[j] & 0xFFFFu);
                if constexpr (NARROW) {
                    sent[slot] = ((key >> 8) << 24) | (colm1 + rel);
                    saux[slot] = (uint8_t)d;
                } else if constexpr (COMPACT) {
                    sent[slot] = ((key >> BITS) << 14) | rel;   // rel <= 8192
                    saux16[slot] = (uint16_t)d;
                } else {
                    sent[slot] = make_entry(key, colm1 + rel);
                }
            }
        }
        lds_sync();
        for (uint32_t i = tid; i < tile_count; i += BNT) {
            const ST v = sent[i];
            uint32_t d;
            ET o;
            if constexpr (NARROW) {
                d = saux[i];
                o = v;
            } else if constexpr (COMPACT) {
                d = saux16[i];
                o = make_entry(((v >> 14) << BITS) | d, colm1 + (v & 0x3FFFu));
            } else {
                d = entry_key((u64)v) & (uint32_t)(BINS - 1);
                o = v;
            }
#ifdef XS0_NOSTORE  // diagnostic builds (wrong results)
            if (d == 0x12345678u)
#endif
            out[gadj[d] + i] = o;
        }
        lds_sync();
    }
}

// ------------------------------------------------------------------------------------------------------
// y for K >= 9: histogram of partitioned k-mers
// ------------------------------------------------------------------------------------------------------
//
// 4^K bins do not fit in LDS and scattered global atomics run at ~2e10/s whatever the working set
// (tools/mb_atomics.hip), so the windows are first partitioned by the top 8 (K <= 12) or 16 (K >= 13) bits of the
// k-mer id -- the narrow-entry passes of the X path with "column" = the low bits of the id -- and then every
// workgroup owns a slice of 2^BPB_LOG2 consecutive bins: it finds the entries of its slice (they are contiguous),
// counts them in LDS and adds the slice to `counts` with plain coalesced read-modify-writes (no other workgroup of
// the launch touches those bins).
//   entries: one pass   v = low bits                       sort key = region
//            two passes v = (sort key & 255) << 24 | low   sort key = region << 8 | v >> 24

// Two passes: bounds[k] = index of the first entry whose 16-bit sort key (region << 8 | top byte) is >= k, for
// k = 0 .. 65536 -- one thread and one binary search per key (inside a region the entries are ordered by that byte).
// The slices' workgroups read their entry ranges from this table instead of searching (two dependent chains of global
// loads per workgroup, ~8 us each, were most of a small slice's life).
__global__ void k_bucket_bounds(const uint32_t *ent, const uint32_t *regions, uint32_t *bounds) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > 65536u) return;
    if (k == 65536u) {
        bounds[k] = regions[RADIX];
        return;
    }
    const uint32_t r = k >> 8, c = k & 255u;
    uint32_t lo = regions[r], hi = regions[r + 1];
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((ent[mid] >> 24) >= c)
            hi = mid;
        else
            lo = mid + 1;
    }
    bounds[k] = lo;
}

#ifndef YS_UNROLL_N
#define YS_UNROLL_N 4
#endif
constexpr int YS_NT = 1024, YS_UNROLL = YS_UNROLL_N;
constexpr uint32_t YS_OVF_CHUNK = 1u << 18;  // entries per workgroup pass in the overflow kernel

// What a slice's bins are: everything needed to turn an entry into a bin of the slice's LDS table.
struct YSlice {
    uint64_t bin0;      // first bin of the slice
    uint32_t k_lo, nk;  // first sort key of the slice / sort keys it covers (1: the slice is part of one key)
    uint32_t c_lo;      // two passes: top byte of the first entry value of the slice
    uint32_t low_mask, sub0;
    int low_bits;
};
template <int BPB_LOG2>
__device__ __forceinline__ YSlice ys_slice(uint32_t slice, int two_pass, int low_bits) {
    YSlice y;
    y.bin0 = (uint64_t)slice << BPB_LOG2;
    y.k_lo = (uint32_t)(y.bin0 >> low_bits);
    y.nk = BPB_LOG2 >= low_bits ? 1u << (BPB_LOG2 - low_bits) : 1u;
    y.low_mask = (1u << low_bits) - 1u;
    y.sub0 = (uint32_t)y.bin0 & y.low_mask;  // non-zero when the slice is part of a key
    y.c_lo = two_pass ? (y.k_lo & 255u) : 0u;
    y.low_bits = low_bits;
    return y;
}

// Count entries [begin, end) into h.  All YS_NT threads call it with the same arguments.
// ET = uint32_t: (sort-key byte << 24 | low bits); ET = uint16_t (one pass, K <= 12): the low bits alone.
// A lane whose 16-byte load holds no two equal neighbours (nearly all of them) makes one predicated LDS add per entry.
// A lane that meets an equal pair takes that value as its HOT entry and from then on counts its occurrences in a
// register (added to the table once, at the end or when a later load with a pair no longer holds it).  Why: 0.7 % of the
// bench's windows are one k-mer (poly-A), they all sit in one slice, and 64 same-address LDS atomics serialise (29 ns
// per wave instruction): with per-thread merging of adjacent equal entries only (rounds 1-2; in the partitioned array a
// (digit, tile) group mixes poly-A with other AAAA... k-mers, so the runs are 2-3 long) the poly-A slice's workgroup
// alone set the duration of the launch -- 1.75 ms per 2^31 positions whatever else was changed in this loop.
template <uint32_t NB, class ET>
__device__ __forceinline__ void ys_count_range(uint32_t *h, const ET *ent, uint32_t begin, uint32_t end, const YSlice &y) {
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t PER = 16 / sizeof(ET);  // entries per 16-byte load
    auto bin_of = [&](uint32_t v) {
        if constexpr (sizeof(ET) == 2)
            return v - y.sub0;
        else
            return ((((v >> 24) - y.c_lo) << y.low_bits) | (v & y.low_mask)) - y.sub0;
    };
    auto add1 = [&](uint32_t bin) {
        if (bin < NB) atomicAdd(&h[bin], 1u);
    };
    // 16-byte loads over the aligned body, YS_UNROLL of them in flight per lane; the ragged ends one entry at a time
    const uint32_t a0 = (begin + PER - 1u) & ~(PER - 1u), a1 = end & ~(PER - 1u);
    if (a0 < a1) {
        if (begin + tid < a0) add1(bin_of(ent[begin + tid]));
        if (a1 + tid < end) add1(bin_of(ent[a1 + tid]));
        const uint4 *ent4 = reinterpret_cast<const uint4 *>(ent);
        const uint32_t q1 = a1 / PER;
        // the loads of batch i + 1 are issued before batch i is counted: one workgroup per CU (the table fills the LDS) has
        // only 16 waves to hide the memory latency with, and without this every wave alternated between waiting and
        // counting (a slice's workgroup streamed its 15-18 MB at ~10 GB/s)
        uint4 v[YS_UNROLL], nx[YS_UNROLL];
        uint32_t hot_v = NONE, hot_n = 0;  // this lane's hot entry value and its occurrences not yet added (see below)
        auto fetch = [&](uint4 *dst, uint32_t qb) {
#pragma unroll
            for (int u = 0; u < YS_UNROLL; ++u) {
                const uint32_t q = qb + u * YS_NT + tid;
                dst[u] = (q < q1) ? ent4[q] : make_uint4(0, 0, 0, 0);
            }
        };
        fetch(nx, a0 / PER);
        for (uint32_t qb = a0 / PER; qb < q1; qb += YS_NT * YS_UNROLL) {
#pragma unroll
            for (int u = 0; u < YS_UNROLL; ++u) v[u] = nx[u];
            if (qb + YS_NT * YS_UNROLL < q1) fetch(nx, qb + YS_NT * YS_UNROLL);
#pragma unroll
            for (int u = 0; u < YS_UNROLL; ++u) {
                if (qb + u * YS_NT + tid >= q1) continue;
                const uint32_t word[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                uint32_t e[PER];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (sizeof(ET) == 2)
                        e[2 * c] = word[c] & 0xFFFFu, e[2 * c + 1] = word[c] >> 16;
                    else
                        e[c] = word[c];
                }
                bool any_pair = false;
#pragma unroll
                for (uint32_t c = 0; c + 1 < PER; ++c) any_pair |= e[c] == e[c + 1];
                if (!any_pair) {  // (all but a few lanes in a thousand on random sequence)
#pragma unroll
                    for (uint32_t c = 0; c < PER; ++c) add1(bin_of(e[c]));
                } else {
                    // the lane's hot bin: the value of an adjacent equal pair of this load, unless the load still holds
                    // the one chosen before
                    uint32_t rep = NONE;
                    bool has_hot = false;
#pragma unroll
                    for (uint32_t c = 0; c < PER; ++c) {
                        has_hot |= e[c] == hot_v;
                        if (c + 1 < PER && e[c] == e[c + 1]) rep = e[c];
                    }
                    if (!has_hot) {
                        const uint32_t hb = bin_of(hot_v);
                        if (hot_n && hb < NB) atomicAdd(&h[hb], hot_n);
                        hot_v = rep;
                        hot_n = 0;
                    }
#pragma unroll
                    for (uint32_t c = 0; c < PER; ++c) {
                        if (e[c] == hot_v)
                            ++hot_n;
                        else
                            add1(bin_of(e[c]));
                    }
                }
            }
        }
        if (hot_n) {
            const uint32_t hb = bin_of(hot_v);
            if (hb < NB) atomicAdd(&h[hb], hot_n);
        }
    } else {
        for (uint32_t i = begin + tid; i < end; i += YS_NT) add1(bin_of(ent[i]));
    }
}

struct YOverflow {
    uint32_t slice, begin, end, pad;
};

// One workgroup per slice.  A slice with more than `cap` entries (a hot bucket: poly-A) keeps its first `cap` and
// queues the rest for k_count_y_overflow, which spreads it over many workgroups.
// `regions`: one pass the 257 region starts; two passes the 65537 bucket bounds of k_bucket_bounds.
template <int BPB_LOG2, class ET>
__global__ __launch_bounds__(YS_NT) void k_count_y_sorted(const ET *ent, const uint32_t *regions, int two_pass, int low_bits,
                                                          int32_t *counts, uint32_t cap, uint32_t *ovf_count, YOverflow *ovf,
                                                          uint32_t ovf_cap) {
    constexpr uint32_t NB = 1u << BPB_LOG2;
    __shared__ __attribute__((aligned(16))) uint32_t h[NB];
    __shared__ uint32_t s_range[2];
    const uint32_t tid = threadIdx.x;
    const YSlice y = ys_slice<BPB_LOG2>(blockIdx.x, two_pass, low_bits);
    for (uint32_t i = tid; i < NB; i += YS_NT) h[i] = 0;
    // one pass: the host guarantees nk == 1 (the slice is a region or half of one); two passes: nk buckets
    if (tid < 2) s_range[tid] = regions[y.k_lo + (tid ? y.nk : 0u)];
    lds_sync();
    const uint32_t begin = s_range[0];
    uint32_t end = s_range[1];
    lds_sync();
    if (end - begin > cap) {  // workgroup-uniform
        if (tid == 0) {
            const uint32_t idx = atomicAdd(ovf_count, 1u);
            if (idx < ovf_cap) {
                ovf[idx] = YOverflow{blockIdx.x, begin + cap, end, 0u};
                s_range[1] = begin + cap;
            }  // (a full queue cannot happen -- it holds every slice that can exceed cap -- but then: do it all here)
        }
        lds_sync();
        end = s_range[1];
    }
    ys_count_range<NB, ET>(h, ent, begin, end, y);
    lds_sync();
    // counts[slice] += h: this launch's only writer of these bins, so plain read-modify-writes -- four bins per lane
    // and all of a thread's loads issued before its first store (one dependent load-add-store per bin kept a small
    // slice's workgroup alive for ~50 us: 32 round trips)
    int32_t *const dst = counts + y.bin0;
    if ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        constexpr uint32_t G = NB / 4, PER_T = (G + YS_NT - 1) / YS_NT;
        const uint4 *h4 = reinterpret_cast<const uint4 *>(h);
        int4 *dst4 = reinterpret_cast<int4 *>(dst);
        uint4 c[PER_T];
        int4 v[PER_T];
#pragma unroll
        for (uint32_t u = 0; u < PER_T; ++u) {
            const uint32_t g = tid + u * YS_NT;
            c[u] = (g < G) ? h4[g] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t u = 0; u < PER_T; ++u) {
            const uint32_t g = tid + u * YS_NT;
            v[u] = (c[u].x | c[u].y | c[u].z | c[u].w) ? dst4[g] : make_int4(0, 0, 0, 0);
        }
#pragma unroll
        for (uint32_t u = 0; u < PER_T; ++u) {
            const uint32_t g = tid + u * YS_NT;
            if (c[u].x | c[u].y | c[u].z | c[u].w)
                dst4[g] = make_int4(v[u].x + (int)c[u].x, v[u].y + (int)c[u].y, v[u].z + (int)c[u].z, v[u].w + (int)c[u].w);
        }
    } else {  // a caller's array that is not 16-byte aligned
        for (uint32_t i = tid; i < NB; i += YS_NT) {
            const uint32_t c = h[i];
            if (c) dst[i] += (int32_t)c;
        }
    }
}

// The queued remainders, YS_OVF_CHUNK entries per workgroup pass, chunks dealt round-robin over the grid; these
// share their bins with other workgroups, so they add to `counts` with atomics (non-zero bins only).
template <int BPB_LOG2, class ET>
__global__ __launch_bounds__(YS_NT) void k_count_y_overflow(const ET *ent, int two_pass, int low_bits, int32_t *counts,
                                                            const uint32_t *ovf_count, const YOverflow *ovf, uint32_t ovf_cap) {
    constexpr uint32_t NB = 1u << BPB_LOG2;
    __shared__ uint32_t h[NB];
    const uint32_t tid = threadIdx.x;
    uint32_t n = *ovf_count;
    if (n > ovf_cap) n = ovf_cap;
    uint32_t next = blockIdx.x;  // global index of this workgroup's next chunk
    uint32_t base = 0;           // global index of the current item's first chunk
    for (uint32_t i = 0; i < n; ++i) {
        const YOverflow it = ovf[i];
        const uint32_t nch = (it.end - it.begin + YS_OVF_CHUNK - 1) / YS_OVF_CHUNK;
        const YSlice y = ys_slice<BPB_LOG2>(it.slice, two_pass, low_bits);
        for (; next < base + nch; next += gridDim.x) {
            const uint32_t b = it.begin + (next - base) * YS_OVF_CHUNK;
            const uint32_t e = (it.end - b > YS_OVF_CHUNK) ? b + YS_OVF_CHUNK : it.end;
            for (uint32_t j = tid; j < NB; j += YS_NT) h[j] = 0;
            lds_sync();
            ys_count_range<NB, ET>(h, ent, b, e, y);
            lds_sync();
            for (uint32_t j = tid; j < NB; j += YS_NT) {
                const uint32_t c = h[j];
                if (c) atomicAdd(&counts[y.bin0 + j], (int32_t)c);
            }
            lds_sync();
        }
        base += nch;
    }
}

// ------------------------------------------------------------------------------------------------------
// Scan of the tile histogram
// ------------------------------------------------------------------------------------------------------
//
// hist[tile][digit] (tile-major: a tile's 256 counts are one 1 KB row, written and read coalesced) becomes, in place,
// base[tile][digit] = position of the first entry of (digit, tile) in the partitioned array
//                   = (entries of smaller digits, all tiles) + (entries of this digit in earlier tiles),
// i.e. the exclusive scan in digit-major order, computed as 256 column scans plus a 256-value scan of the totals.
// Three launches: sums per chunk of CSCAN_TILES tiles; one workgroup turns them into chunk bases; apply.

constexpr int CSCAN_TILES = 64;

template <int RADIX>  // (the bins of the pass: 256, or 512 / 1024 for the wide passes of K >= 9)
__global__ __launch_bounds__(RADIX) void k_cscan_reduce(const uint32_t *hist, uint32_t n_tiles, uint32_t *chunk_sums) {
    const uint32_t d = threadIdx.x, c = blockIdx.x;
    const uint32_t t0 = c * CSCAN_TILES, t1 = (t0 + CSCAN_TILES < n_tiles) ? t0 + CSCAN_TILES : n_tiles;
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t t = t0; t < t1; ++t) s += hist[(uint64_t)t * RADIX + d];
    chunk_sums[(uint64_t)c * RADIX + d] = s;
}

// One workgroup of 1024 threads = RADIX * CSCAN_SEG: thread (seg, d) owns a CSCAN_SEG-th of digit d's chunks.
template <int RADIX>
__global__ __launch_bounds__(1024) void k_cscan_mid(uint32_t *chunk_sums, uint32_t n_chunks, uint32_t *p_total) {
    constexpr int CSCAN_SEG = 1024 / RADIX;
    __shared__ uint32_t seg_sum[CSCAN_SEG][RADIX];
    __shared__ uint32_t dbase[RADIX];
    __shared__ uint32_t stmp[RADIX * CSCAN_SEG / 64];
    const uint32_t d = threadIdx.x % RADIX, seg = threadIdx.x / RADIX;
    const uint32_t per = (n_chunks + CSCAN_SEG - 1) / CSCAN_SEG;
    const uint32_t c0 = seg * per < n_chunks ? seg * per : n_chunks, c1 = (c0 + per < n_chunks) ? c0 + per : n_chunks;
    uint32_t s = 0;
#pragma unroll 8
    for (uint32_t c = c0; c < c1; ++c) s += chunk_sums[(uint64_t)c * RADIX + d];
    seg_sum[seg][d] = s;
    __syncthreads();
    // digit totals -> exclusive scan over the digits (threads of segment 0 carry the values, the others zero)
    uint32_t total = 0;
    if (seg == 0)
        for (int i = 0; i < CSCAN_SEG; ++i) total += seg_sum[i][d];
    // block_excl_scan is written for NT threads; this workgroup has RADIX * CSCAN_SEG: scan by hand
    {
        const uint32_t lane = lane_id(), w = threadIdx.x >> 6;
        uint32_t incl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl += t;
        }
        if (lane == 63) stmp[w] = incl;
        __syncthreads();
        uint32_t wbase = 0, all = 0;
        for (uint32_t i = 0; i < RADIX * CSCAN_SEG / 64; ++i) {
            const uint32_t t = stmp[i];
            if (i < w) wbase += t;
            all += t;
        }
        if (seg == 0) dbase[d] = wbase + incl - total;
        if (threadIdx.x == 0) *p_total = all;
    }
    __syncthreads();
    uint32_t run = dbase[d];
    for (uint32_t i = 0; i < seg; ++i) run += seg_sum[i][d];
#pragma unroll 8
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t v = chunk_sums[(uint64_t)c * RADIX + d];
        chunk_sums[(uint64_t)c * RADIX + d] = run;
        run += v;
    }
}

// regions (may be null): regions[d] = position of the first entry of digit d = tile 0's scanned row, regions[256] = the
// entry count *p_count (no launch of its own).
template <int RADIX>
__global__ __launch_bounds__(RADIX) void k_cscan_apply(uint32_t *hist, uint32_t n_tiles, const uint32_t *chunk_base, uint32_t *regions,
                                                       const uint32_t *p_count) {
    const uint32_t d = threadIdx.x, c = blockIdx.x;
    const uint32_t t0 = c * CSCAN_TILES, t1 = (t0 + CSCAN_TILES < n_tiles) ? t0 + CSCAN_TILES : n_tiles;
    uint32_t run = chunk_base[(uint64_t)c * RADIX + d];
    if (regions && c == 0) {
        regions[d] = run;
        if (d == 0) regions[RADIX] = *p_count;
    }
#pragma unroll 8
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t v = hist[(uint64_t)t * RADIX + d];
        hist[(uint64_t)t * RADIX + d] = run;
        run += v;
    }
}

// ------------------------------------------------------------------------------------------------------
// Run-length reduce of the sorted (k-mer, column) instances -> CSR, one kernel
// ------------------------------------------------------------------------------------------------------

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

// Chained scan across tiles ("decoupled look-back") of one number per tile, the tile's count of run heads.
// status[t] = flag << 62 | value; flag 1: value = heads of tile t alone, flag 2: value = heads of tiles 0..t.
// The whole state of a tile is ONE naturally aligned 8-byte word written by one relaxed agent-scope atomic store
// and polled with relaxed agent-scope atomic loads (both bypass the non-coherent per-CU L1), so no fence is needed.
// Tiles are numbered by an atomic ticket, so every predecessor of a tile has at least started and publishes its
// flag-1 word without waiting for anything: the spin below always ends; it is bounded anyway (p_err on timeout).
constexpr u64 ST_AGG = 1ull << 62, ST_PREFIX = 2ull << 62, ST_MASK = (1ull << 62) - 1ull;
constexpr uint32_t SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ u64 st_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_store(u64 *p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Called by wave 0 of tile `tile` (all 64 lanes).  Returns the number of heads in tiles 0..tile-1.
__device__ __forceinline__ u64 lookback_exclusive(u64 *status, uint32_t tile, u64 mine, uint32_t *p_err) {
    const uint32_t lane = lane_id();
    if (tile == 0) {
        if (lane == 0) st_store(&status[0], ST_PREFIX | mine);
        return 0;
    }
    if (lane == 0) st_store(&status[tile], ST_AGG | mine);
    u64 excl = 0;
    int64_t hi = (int64_t)tile - 1;  // nearest predecessor not yet summed
    uint32_t spins = 0;
    while (hi >= 0) {
        const int64_t t = hi - (int64_t)lane;
        const u64 v = (t >= 0) ? st_load(&status[t]) : ST_PREFIX;  // before tile 0: an empty prefix
        const u64 ready = __ballot((v >> 62) != 0);
        const u64 pref = __ballot((v >> 62) == 2);
        // usable lanes: from lane 0 up to the first PREFIX lane, all of which must be ready
        const uint32_t stop = pref ? (uint32_t)__ffsll((long long)pref) - 1u : 63u;
        const u64 need = (stop == 63) ? ~0ull : ((2ull << stop) - 1ull);
        if ((ready & need) != need) {
            if (++spins > SPIN_LIMIT) {
                if (lane == 0) *p_err = 2u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        u64 part = (lane <= stop) ? (v & ST_MASK) : 0ull;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
        excl += part;
        if (pref) break;
        hi -= 64;
    }
    if (lane == 0) st_store(&status[tile], ST_PREFIX | ((excl + mine) & ST_MASK));
    return excl;
}

// One CSR entry per head (first instance of a (k-mer, column) run): column, value, and the row pointer entries of
// every row that starts (or is empty) between the previous head's k-mer and this one.
//   scal[0] = E (instances, input), scal[1] = nnz (output), scal[2] = ntasks, scal[3] = err, scal[5] = ticket
// Narrow entries are widened to the 8-byte form (k-mer id << 32 | column) as they are loaded (the id's upper part is
// the region the entry sits in), so everything after the loads is common to both forms.
// RNT threads, RIPT rounds per wave: a tile of this kernel is RNT * RIPT entries, larger than the partition's: every
// tile pays one chained-scan hand-off, so fewer and larger tiles win -- as long as they come from more threads rather
// than more rounds (registers).  Same-run A/B on config 2 / K=14: 128 threads 2.83 / 6.85 ms, 256: 2.45 / 4.47,
// 512: 2.05 / 3.52, 1024: 1.94 / 3.31.
template <class ET, int RIPT, int RNT>
__global__ __launch_bounds__(RNT, 4) void k_rle(const ET *ent, uint32_t *scal, u64 *status, const uint32_t *rec_total,
                                               uint64_t first_col, uint64_t M, uint64_t max_nz, float *data, uint64_t *col_ind,
                                               uint64_t *row_ind, FillTask *tasks, uint32_t task_cap, const uint32_t *regions,
                                               int key_shift) {
    constexpr bool NARROW = sizeof(ET) == 4;
    constexpr int R_TILE = RNT * RIPT, R_WCHUNK = 64 * RIPT, RNW = RNT / 64;
    __shared__ uint32_t sreg[NARROW ? RADIX + 1 : 1];
    __shared__ uint32_t s_tile;
    __shared__ uint32_t s_cnt[RNW], s_first[RNW];
    __shared__ uint32_t s_next;   // first head after this tile (index), E if none
    __shared__ u64 s_out0;        // heads before this tile
    // cooperative row-pointer fill (below): per wave, a bitmap of the rows that start in a round, its word prefix, and
    // the output index of every row head of the round
    constexpr bool COOP_FILL = !NARROW;  // (4-byte entries mean K <= 8: at most 65,536 row heads in all)
    __shared__ uint32_t s_rbits[COOP_FILL ? RNW : 1][64], s_rpre[COOP_FILL ? RNW : 1][64];
    __shared__ u64 s_rout[COOP_FILL ? RNW : 1][64];
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    uint32_t *p_nnz = &scal[1], *p_ntasks = &scal[2], *p_err = &scal[3];
    const uint32_t E = scal[0];
    if (tid == 0) s_tile = atomicAdd(&scal[5], 1u);
    if constexpr (NARROW)
        for (uint32_t i = tid; i <= RADIX; i += RNT) sreg[i] = regions[i];
    lds_sync();
    const uint32_t tile = s_tile;
    if (E == 0) {  // no instance at all: every row pointer is 0
        if (tile == 0 && tid == 0) fill_rows(row_ind, 0, M + 1, 0, tasks, p_ntasks, task_cap, p_err);
        return;
    }
    const uint64_t base = (uint64_t)tile * R_TILE;
    if (base >= E) return;
    const uint32_t wbase_i = (uint32_t)base + w * R_WCHUNK;
    const uint64_t tile_end = base + R_TILE;

    // Narrow entries are compared as they are stored (byte << 24 | column): inside a region that is the order and the
    // equality of the 8-byte form; where a region starts, the entry is a head (and a row head) whatever its value.
    // The k-mer id (region << 8 | byte, or the region alone) is rebuilt only where a row starts.
    auto key_of = [&](uint32_t i, ET v) -> uint32_t {
        if constexpr (NARROW) {
            const uint32_t r = region_search(sreg, i);
            return key_shift ? ((r << 8) | ((uint32_t)v >> 24)) : r;
        } else {
            return entry_key((u64)v);
        }
    };
    auto col_of = [&](ET v) -> uint32_t {
        if constexpr (NARROW)
            return (uint32_t)v & 0xFFFFFFu;
        else
            return entry_col((u64)v);
    };
    auto same_row = [&](ET a, ET b) -> bool {  // for entries of one region
        if constexpr (NARROW)
            return key_shift ? (((uint32_t)a >> 24) == ((uint32_t)b >> 24)) : true;
        else
            return entry_key((u64)a) == entry_key((u64)b);
    };

    // ---- loads: the wave's 16 rounds, the entry before its chunk, and (last wave) the probe behind the tile
    ET e[RIPT];
#pragma unroll
    for (int j = 0; j < RIPT; ++j) {
        const uint64_t i = (uint64_t)wbase_i + j * 64 + lane;
        e[j] = (i < E) ? ent[i] : (ET)0;
    }
    const ET carry0 = (wbase_i > 0 && lane == 0) ? ent[wbase_i - 1] : (ET)0;
    ET probe = 0;
    if (w == RNW - 1 && tile_end + lane < E) probe = ent[tile_end + lane];

    // does a region start inside this wave's chunk (or exactly at its first entry)?  Nearly never: 256 regions.
    bool has_b = false;
    if constexpr (NARROW) {
        const uint32_t r0 = region_search(sreg, wbase_i < E ? wbase_i : E - 1);
        const uint32_t nb = sreg[r0 + 1];
        has_b = (sreg[r0] == wbase_i) || (nb < wbase_i + R_WCHUNK && nb < E);
    }

    ET carry = carry0;
    u64 hm[RIPT], rm[RIPT];  // head lanes / heads that also start a new row (wave-uniform masks)
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < RIPT; ++j) {
        const uint32_t i = wbase_i + j * 64 + lane;
        ET prev = __shfl_up(e[j], 1);
        if (lane == 0) prev = carry;
        const bool in = i < E;
        bool rs = false;  // a region starts at i
        if constexpr (NARROW) {
            if (has_b) rs = in && sreg[region_search(sreg, in ? i : 0u)] == i;
        }
        const bool head = in && (i == 0 || rs || e[j] != prev);
        const bool rowh = head && (i == 0 || rs || !same_row(e[j], prev));
        hm[j] = __ballot(head);
        rm[j] = __ballot(rowh);
        cnt += (uint32_t)__popcll(hm[j]);
        carry = __shfl(e[j], 63);  // lane 0 uses it next round
    }
    uint32_t first = NONE;
#pragma unroll
    for (int j = RIPT - 1; j >= 0; --j)
        if (hm[j]) first = wbase_i + j * 64 + (uint32_t)__ffsll((long long)hm[j]) - 1u;
    if (lane == 0) {
        s_cnt[w] = cnt;
        s_first[w] = first;
    }
    // ---- the last run of the tile may go on in the following tiles: find the next head after the tile.
    // The array is sorted (k-mer id major, column minor; narrow: by value inside a region), so that head is the first
    // entry greater than the tile's last one: the 64-entry probe loaded above (the common case), then a binary search
    // -- O(log E) even when one run spans the rest of the array (a single poly-A record of any length).
    if (w == RNW - 1) {
        uint32_t nxt = E;
        if (tile_end < E) {
            const ET last = __shfl(e[RIPT - 1], 63);  // entry tile_end - 1 (the tile is full)
            uint64_t limit = E;                       // first index at or after tile_end where a region starts
            if constexpr (NARROW) limit = sreg[region_search(sreg, (uint32_t)tile_end - 1u) + 1];
            const uint64_t i = tile_end + lane;
            const u64 dm = __ballot((i < limit) ? (probe != last) : true);
            if (dm) {
                const uint64_t at = tile_end + (uint32_t)__ffsll((long long)dm) - 1u;
                nxt = (at < E) ? (uint32_t)at : E;
            } else {
                uint64_t lo = tile_end + 64, hi = limit;  // first index in [lo, hi] whose entry is > last (hi if none)
                while (lo < hi) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if (ent[mid] > last)
                        hi = mid;
                    else
                        lo = mid + 1;
                }
                nxt = (uint32_t)lo;
            }
        }
        if (lane == 0) s_next = nxt;
    }
    lds_sync();
    uint32_t wave_before = 0, n_heads = 0;
#pragma unroll
    for (int i = 0; i < RNW; ++i) {
        const uint32_t t = s_cnt[i];
        if ((uint32_t)i < w) wave_before += t;
        n_heads += t;
    }
    // ---- heads before this tile (chained scan), by wave 0
    if (w == 0) {
#ifdef KK_RLE_NOLOOKBACK
        const u64 ex = base;
#else
        const u64 ex = lookback_exclusive(status, tile, n_heads, p_err);
#endif
        if (lane == 0) s_out0 = ex;
    }
    lds_sync();
    const u64 out0 = s_out0;
    // first head after this wave's chunk
    uint32_t after = s_next;
#pragma unroll
    for (int i = RNW - 1; i >= 1; --i)
        if ((uint32_t)i > w && s_first[i] != NONE) after = s_first[i];

    // ---- emit.  nxt_after = first head in rounds > j (wave-uniform), walked backwards
    uint32_t nxt_after[RIPT];
    {
        uint32_t nh = after;
#pragma unroll
        for (int j = RIPT - 1; j >= 0; --j) {
            nxt_after[j] = nh;
            if (hm[j]) nh = wbase_i + j * 64 + (uint32_t)__ffsll((long long)hm[j]) - 1u;
        }
    }
    uint32_t run = wave_before;
#pragma unroll
    for (int j = 0; j < RIPT; ++j) {
        if (hm[j] == 0) continue;  // wave-uniform
        const uint32_t i = wbase_i + j * 64 + lane;
        ET prev = __shfl_up(e[j], 1);
        const ET prev0 = (j > 0) ? __shfl(e[j > 0 ? j - 1 : 0], 63) : carry0;
        if (lane == 0) prev = prev0;
        // ---- rows that start in this round, filled by the whole wave.  The row heads of a round cover a contiguous range
        // of rows (head h owns the rows after the previous head's k-mer up to its own), so when there are many of them
        // -- large K: most heads are row heads, and on a column shard the gaps between them are several rows long --
        // the wave writes row_ind[first .. last] in coalesced 512-byte stores: every row looks up how many row heads lie
        // below it (a bitmap of the heads' rows + word prefix) and takes that head's output index.  A lane filling its
        // own gap (fill_rows) costs one partly filled store instruction per row of the LONGEST gap in the wave.
        bool coop = false;
        constexpr uint32_t COOP_MIN_HEADS = 8, COOP_MAX_ROWS = 2048;
        const u64 rmask = rm[j];
        if (COOP_FILL && (uint32_t)__popcll(rmask) >= COOP_MIN_HEADS) {  // wave-uniform
            const bool rh = (rmask >> lane) & 1ull;
            const uint32_t kk = rh ? key_of(i, e[j]) : 0u;
            const uint64_t lo_mine = (rh && i != 0) ? (uint64_t)key_of(i - 1, prev) + 1 : 0ull;
            const uint32_t first_l = __builtin_amdgcn_readfirstlane((uint32_t)__ffsll((long long)rmask) - 1u);
            const uint32_t last_l = __builtin_amdgcn_readfirstlane(63u - (uint32_t)__clzll((long long)rmask));
            const uint32_t lo_first = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)lo_mine, (int)first_l);
            const uint32_t k_last = (uint32_t)__builtin_amdgcn_readlane((int)kk, (int)last_l);
            const uint64_t R64 = (uint64_t)k_last - lo_first + 1ull;  // rows to write (all 2^32 of them at K = 16, in principle)
            if (R64 <= COOP_MAX_ROWS) {
                const uint32_t R = (uint32_t)R64;
                coop = true;
                uint32_t *const bits = s_rbits[w], *const pre = s_rpre[w];
                u64 *const outs = s_rout[w];
                bits[lane] = 0u;
                if (rh) {
                    const uint32_t off = kk - lo_first;
                    atomicOr(&bits[off >> 5], 1u << (off & 31u));
                    outs[(uint32_t)__popcll(rmask & lanes_below(lane))] =
                        out0 + run + (uint32_t)__popcll(hm[j] & lanes_below(lane));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: the LDS keeps its operations in order)
                const uint32_t wbits = bits[lane];
                const uint32_t pc = (uint32_t)__popc(wbits);
                pre[lane] = wave_incl_scan(pc) - pc;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (uint32_t r = lane; r < R; r += 64u) {
                    const uint32_t below = pre[r >> 5] + (uint32_t)__popc(bits[r >> 5] & ((1u << (r & 31u)) - 1u));
                    row_ind[(uint64_t)lo_first + r] = outs[below];  // the first head at or after row r
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next round overwrites the three arrays
            }
        }
        if ((hm[j] >> lane) & 1ull) {
            const u64 above = (lane == 63) ? 0ull : (hm[j] >> (lane + 1));
            const uint32_t nxt = above ? i + (uint32_t)__ffsll((long long)above) : nxt_after[j];
            const u64 o = out0 + run + (uint32_t)__popcll(hm[j] & lanes_below(lane));
            const uint32_t c = col_of(e[j]);
            if (o < max_nz) {
                // reference: float(count) / total_count with total_count uint32_t: both operands are rounded to
                // float, then one IEEE float division (hipcc's default float division is correctly rounded)
#ifdef KK_RLE_NOGATHER  // diagnostic builds (wrong results): what the rec_total gather costs
                const uint32_t total = c | 1u;
#else
                const uint32_t total = rec_total[c];
#endif
#ifdef KK_RLE_NOSTORE
                if (total == 0x7FFFFFFFu)
#endif
                // (non-temporal stores: same-run A/B 3.30 -> 3.18 ms with 8-byte entries, but 1.97 -> 2.17 ms with 4-byte ones)
                const float value = (float)(nxt - i) / (float)total;
                if constexpr (NARROW)
                    data[o] = value;
                else
                    __builtin_nontemporal_store(value, &data[o]);
#ifdef KK_RLE_NOSTORE
                if (total == 0x7FFFFFFFu)
#endif
                if constexpr (NARROW)
                    col_ind[o] = first_col + c;
                else
                    __builtin_nontemporal_store((uint64_t)(first_col + c), &col_ind[o]);
            }
            const bool row_head = (rm[j] >> lane) & 1ull, last_head = nxt == E;
            if (row_head || last_head) {  // rare: at most M + 1 lanes of the whole launch
                const uint32_t k = key_of(i, e[j]);
                if (row_head && !coop) {
                    const uint64_t lo = (i == 0) ? 0ull : (uint64_t)key_of(i - 1, prev) + 1;
                    fill_rows(row_ind, lo, (uint64_t)k + 1, o, tasks, p_ntasks, task_cap, p_err);
                }
                if (last_head) {  // the last head of all
                    *p_nnz = (uint32_t)(o + 1);
                    fill_rows(row_ind, (uint64_t)k + 1, M + 1, o + 1, tasks, p_ntasks, task_cap, p_err);
                }
            }
        }
        run += (uint32_t)__popcll(hm[j]);
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
// X for K <= 8, round 3: the LAST partition pass writes the CSR arrays itself ("fused emit")
// ------------------------------------------------------------------------------------------------------
//
// After pass 0 the 4-byte entries (high digit << 24 | column) lie in 256 regions (one per low digit d), each ordered by
// column (in any order inside a column: k_xscatter0).  All instances of one (k-mer, column) are EQUAL 32-bit values
// inside one contiguous (d, column) group.  The last pass partitions a tile of a region by the high digit h, stably:
// in LDS the run of digit h then holds the tile's entries of row (h << 8 | d) with ascending columns and equal values
// next to each other -- the CSR order -- so the run-length reduce happens where the entries already are, and the pass
// stores `data` and `col_ind` instead of a sorted copy of the entries: k_scatter<entries>'s 1.6 GB of stores, k_rle's
// 1.6 GB of loads, its scan across tiles and its launch are gone (config 2: 1.17 + 1.96 ms).
//
// What it takes:
//  * the output position of the first non-zero of every (tile, h) must be known beforehand: k_heads_hist counts, per
//    tile and h, the HEADS (first instance of an equal-value group, in region order), not the instances; the usual
//    scan of that table gives the positions, nnz, and -- tiles never span two regions -- the row pointer itself:
//    row_ind[h << 8 | d] = position of (first tile of region d, h).  No k_row_counts, no gap filling.
//  * a group that straddles a tile boundary: the head is where the FIRST instance is.  A tile looks back over the
//    entries in front of it that share its first column (sseen[h]: that (h, column) already has its head) and ahead over
//    those behind it that share its last column (sextra[h]: instances to add to a head of this tile).  Both scans end
//    with the column group, a dozen entries on a transcriptome; a group longer than SCAN_LIMIT probes (one record with
//    millions of bases) raises scal[6] and the host redoes the build on the general path (k_scatter + k_rle).

constexpr uint32_t FT = TILE;               // entries per tile of the fused passes
constexpr uint32_t SCAN_LIMIT = 1u << 10;   // 64-entry probes one look-back / look-ahead may take

// tstart[d] = number of tiles of the regions before d (region d has ceil(len / FT) tiles); tstart[256] = all tiles.
__global__ __launch_bounds__(RADIX) void k_ftiles(const uint32_t *regions, uint32_t *tstart) {
    __shared__ uint32_t sw[RADIX / 64];
    const uint32_t d = threadIdx.x, lane = lane_id(), w = d >> 6;
    const uint32_t n = (regions[d + 1] - regions[d] + FT - 1) / FT;
    const uint32_t inc = wave_incl_scan(n);
    if (lane == 63) sw[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t i = 0; i < w; ++i) base += sw[i];
    tstart[d] = base + inc - n;
    if (d == RADIX - 1) tstart[RADIX] = base + inc;
}

struct FTile {
    uint32_t begin, end;     // entries of the tile
    uint32_t rbegin, rend;   // its region
    uint32_t d;
};
__device__ __forceinline__ FTile ftile_of(const uint32_t *s_ts, const uint32_t *s_reg, uint32_t tile) {
    FTile f;
    f.d = region_search(s_ts, tile);  // (empty regions share their tstart with the next one: the LARGEST d with tstart <= tile)
    f.rbegin = s_reg[f.d];
    f.rend = s_reg[f.d + 1];
    f.begin = f.rbegin + (tile - s_ts[f.d]) * FT;
    f.end = (f.rend - f.begin > FT) ? f.begin + FT : f.rend;
    return f;
}

__device__ __forceinline__ uint32_t fe_col(uint32_t v) { return v & 0xFFFFFFu; }

// previous lane's value (lane 0: undefined) without an LDS round trip: DPP wave_shr:1
__device__ __forceinline__ uint32_t lane_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }

// Heads per (tile, h).  Every WAVE works alone on 1024 consecutive entries, 64 per round in entry order; T[h] = the column
// in which an entry of digit h was last seen by this wave -- the region is ordered by column, so "T[h] == my column" says
// that my value occurred before.  Inside a round the lanes of one value are found with the OR-mask match of the
// partition kernels, restricted to the lanes of the same column (a contiguous stretch of lanes).  In front of its chunk a
// wave first looks back over the entries that share its first column.
__global__ __launch_bounds__(NT) void k_heads_hist(const uint32_t *ent, const uint32_t *regions, const uint32_t *tstart,
                                                   uint32_t *tilehist, uint32_t rows, uint32_t *scal) {
    __shared__ uint32_t s_reg[RADIX + 1], s_ts[RADIX + 1];
    __shared__ uint32_t T[NW][RADIX];
    __shared__ u64 wmask[NW][RADIX];
    // (HH_SUB copies of the tile's table by lane, as in k_hist_words_y)
    constexpr uint32_t HSUBW = HH_SUB * (RADIX + 1);
    __shared__ uint32_t hist[2][HSUBW];
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    for (uint32_t i = tid; i <= RADIX; i += NT) {
        s_reg[i] = regions[i];
        s_ts[i] = tstart[i];
    }
    for (uint32_t i = tid; i < NW * RADIX; i += NT) (&wmask[0][0])[i] = 0ull;
    for (uint32_t i = tid; i < 2 * HSUBW; i += NT) (&hist[0][0])[i] = 0u;
    const uint32_t hsub = (lane % HH_SUB) * (RADIX + 1);
    lds_sync();
    const uint32_t n_ft = s_ts[RADIX];
    const uint32_t per = (rows + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = (t0 + per < rows) ? t0 + per : rows;
    uint32_t *const Tw = T[w];
    u64 *const mw = wmask[w];
    uint32_t e[IPT], probe = 0;
    FTile ft{0, 0, 0, 0, 0};
    auto load_tile = [&](uint32_t t) {
        if (t >= n_ft) return;  // (workgroup-uniform)
        ft = ftile_of(s_ts, s_reg, t);
        const uint32_t s0 = ft.begin + w * WCHUNK;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = s0 + j * 64 + lane;
            e[j] = (i < ft.end) ? __builtin_nontemporal_load(&ent[i]) : 0u;
        }
        probe = (s0 < ft.end && s0 >= ft.rbegin + 1u + lane) ? ent[s0 - 1u - lane] : NONE;  // (NONE: no column is 2^24 - 1 ... see below)
    };
    if (t0 < t1) load_tile(t0);
    for (uint32_t tile = t0; tile < t1; ++tile) {
        uint32_t *const hr = hist[(tile - t0) & 1u], *const hc = hr + hsub;
        if (tile >= n_ft) {  // rows past the last tile: zeros (the scan runs over all `rows`)
            if (tid < RADIX) tilehist[(uint64_t)tile * RADIX + tid] = 0u;
            continue;
        }
        const FTile f = ft;
        const uint32_t s0 = f.begin + w * WCHUNK;
#pragma unroll
        for (int i = 0; i < RADIX / 64; ++i) Tw[lane + 64 * i] = NONE;
        if (s0 < f.end && s0 > f.rbegin) {  // (wave-uniform) look back over the entries that share the chunk's first column
            const uint32_t c0 = fe_col((uint32_t)__builtin_amdgcn_readfirstlane((int)e[0]));
            uint32_t at = s0, it = 0;  // the probe holds entries at - 1 - lane
            for (;;) {
                const bool in = at >= f.rbegin + 1u + lane;
                const bool same = in && fe_col(probe) == c0;
                const u64 stop = __ballot(!same);
                const uint32_t n_same = stop ? (uint32_t)__ffsll((long long)stop) - 1u : 64u;
                if (lane < n_same) Tw[probe >> 24] = c0;
                if (stop) break;
                if (++it > SCAN_LIMIT) {
                    if (lane == 0) scal[6] = 1u;
                    break;
                }
                at -= 64u;
                probe = (at >= f.rbegin + 1u + lane) ? ent[at - 1u - lane] : NONE;
            }
        }
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t i = s0 + j * 64 + lane;
            const bool ok = i < f.end;
            const uint32_t v = e[j], h = v >> 24, c = fe_col(v);
            // lanes of my column in this round: from the last column change at or below me
            const uint32_t pc = lane_prev(c);
            const u64 chg = __ballot(lane == 0 || pc != c);
            const u64 upto = chg & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
            const uint32_t seg0 = 63u - (uint32_t)__clzll((long long)upto);  // (bit 0 is always set)
            u64 m = 0;
            uint32_t t = NONE;
            if (ok) {
                atomicOr(&mw[h], 1ull << lane);
                m = mw[h];
                t = Tw[h];
            }
            const u64 mine = m & lanes_below(lane) & ~lanes_below(seg0);  // lower lanes with my value
            const bool head = ok && mine == 0ull && t != c;
            if (ok && (lane == 63 || (m >> (lane + 1)) == 0ull)) {  // the last lane of digit h in this round
                Tw[h] = c;
                mw[h] = 0ull;
            }
            if (head) atomicAdd(&hc[h], 1u);
            asm volatile("" ::: "memory");
        }
        if (tile + 1 < t1) load_tile(tile + 1);
        lds_sync();
        // (no second barrier: the next tile's atomics go to the other table; these words are cleared here, by the thread
        // that reads them, for the tile after next)
        if (tid < RADIX) {
            uint32_t sum = 0;
#pragma unroll
            for (uint32_t c = 0; c < HH_SUB; ++c) {
                sum += hr[c * (RADIX + 1) + tid];
                hr[c * (RADIX + 1) + tid] = 0u;
            }
            tilehist[(uint64_t)tile * RADIX + tid] = sum;
        }
    }
}

// row_ind from the scanned head table (see above); scal[1] = nnz.
__global__ void k_row_ptr_fused(const uint32_t *headbase, const uint32_t *tstart, const uint32_t *p_nnz, uint64_t M,
                                uint64_t *row_ind, uint32_t *scal) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > M) return;
    if (r == M) {
        row_ind[M] = *p_nnz;
        scal[1] = *p_nnz;
        return;
    }
    const uint32_t h = (uint32_t)(r >> 8), d = (uint32_t)r & 255u;
    row_ind[r] = headbase[(uint64_t)tstart[d] * RADIX + h];
}

// The last pass: stable partition of a tile by h in LDS, then every wave turns a contiguous eighth of the partitioned
// tile into CSR entries.  The valid-window totals of the tile's columns (a tile of a region covers ~900 consecutive
// columns on a transcriptome) are staged in LDS first: one gather instruction over 64 different lines per round kept the
// CU's address path busy for longer than everything else in the emit (ablation: 0.85 ms of 2.57).
#ifndef EM_WAVES
#define EM_WAVES 4   // waves per SIMD the register allocation aims at (6: three workgroups per CU)
#endif
constexpr uint32_t EM_TOT_CAP = 2048;  // columns whose totals are staged (a sparser tile gathers the rest from memory)

__global__ __launch_bounds__(NT, EM_WAVES) void k_emit(const uint32_t *ent, const uint32_t *regions, const uint32_t *tstart,
                                                const uint32_t *headbase, const uint32_t *rec_total, uint64_t first_col,
                                                uint64_t max_nz, float *data, uint64_t *col_ind, uint32_t *scal) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[FT];  // the partitioned tile; first the match masks
    __shared__ uint32_t wcnt[NW][RADIX];
    __shared__ uint32_t s_reg[RADIX + 1], s_ts[RADIX + 1];
    __shared__ uint32_t sseen[RADIX], sextra[RADIX], shb[RADIX];
    __shared__ uint32_t stot[EM_TOT_CAP];
    __shared__ uint32_t stmp[NW], s_c[2];
    static_assert(NW * RADIX * 8 <= FT * 4, "match masks alias the entry staging");
    const uint32_t tid = threadIdx.x, lane = lane_id();
    const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));  // (scalar: the emit sweep branches on it)
    for (uint32_t i = tid; i <= RADIX; i += NT) {
        s_reg[i] = regions[i];
        s_ts[i] = tstart[i];
    }
    lds_sync();
    const uint32_t n_ft = s_ts[RADIX];
    const uint32_t per = (n_ft + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = (t0 + per < n_ft) ? t0 + per : n_ft;
    if (t0 >= t1) return;
    u64 *const smask = reinterpret_cast<u64 *>(stage);
    uint32_t e[IPT], probe = NONE, e_first = 0, e_last = 0;
    FTile ft{0, 0, 0, 0, 0};
    auto load_tile = [&](uint32_t t) {
        ft = ftile_of(s_ts, s_reg, t);
        // (the tile's bounds are the same for every thread: scalar, so that the loads take a scalar base + 32-bit offset)
        ft.begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)ft.begin);
        ft.end = (uint32_t)__builtin_amdgcn_readfirstlane((int)ft.end);
        ft.rbegin = (uint32_t)__builtin_amdgcn_readfirstlane((int)ft.rbegin);
        ft.rend = (uint32_t)__builtin_amdgcn_readfirstlane((int)ft.rend);
        const uint32_t *const tp = ent + ft.begin;
        const uint32_t nn = ft.end - ft.begin;
        uint32_t o0 = w * WCHUNK + lane;
        asm volatile("" : "+v"(o0));  // (opaque: otherwise the 16 offsets are kept in registers across the tile loop)
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t o = o0 + j * 64;
            e[j] = (o < nn) ? tp[o] : 0u;
        }
        e_first = tp[0];
        e_last = tp[nn - 1];
        probe = NONE;
        if (w == 0 && ft.begin >= ft.rbegin + 1u + lane) probe = ent[ft.begin - 1u - lane];       // look back
        if (w == NW - 1 && ft.end + lane < ft.rend) probe = ent[ft.end + lane];                    // look ahead
    };
    load_tile(t0);
    uint32_t gb = (tid < RADIX) ? headbase[(uint64_t)t0 * RADIX + tid] : 0u;
    KK_STAMP_BEGIN();
    for (uint32_t tile = t0; tile < t1; ++tile) {
        const FTile f = ft;
        const uint32_t n = f.end - f.begin;  // entries of the tile (>= 1)
        for (uint32_t i = tid; i < NW * RADIX; i += NT) {
            (&wcnt[0][0])[i] = 0;
            smask[i] = 0ull;
        }
        if (tid < RADIX) {
            sseen[tid] = 0u;
            sextra[tid] = 0u;
            shb[tid] = gb;
            if (tile + 1 < t1) gb = headbase[(uint64_t)(tile + 1) * RADIX + tid];
        }
        // the totals of the tile's columns [cf, cl], on their way while the tile is ranked
        const uint32_t cf = fe_col(e_first), cl = fe_col(e_last);
        constexpr int TQ = EM_TOT_CAP / NT;
        uint32_t totq[TQ];
#pragma unroll
        for (int q = 0; q < TQ; ++q) {
            const uint32_t c = cf + tid + (uint32_t)q * NT;
            totq[q] = (c <= cl) ? rec_total[c] : 0u;
        }
        lds_sync();
        KK_STAMP(0);
        // ---- the neighbours of the tile inside its region (waves 0 and NW - 1, while the others rank)
        if (w == 0) {
            uint32_t c0 = NONE;
            if (f.begin > f.rbegin) {
                c0 = cf;
                uint32_t at = f.begin, it = 0;
                bool any = false;
                for (;;) {
                    const bool same = at >= f.rbegin + 1u + lane && fe_col(probe) == c0;
                    const u64 stop = __ballot(!same);
                    const uint32_t n_same = stop ? (uint32_t)__ffsll((long long)stop) - 1u : 64u;
                    if (lane < n_same) sseen[probe >> 24] = 1u;
                    any = any || n_same;
                    if (stop) break;
                    if (++it > SCAN_LIMIT) {
                        if (lane == 0) scal[6] = 1u;
                        break;
                    }
                    at -= 64u;
                    probe = (at >= f.rbegin + 1u + lane) ? ent[at - 1u - lane] : NONE;
                }
                if (!any) c0 = NONE;  // nothing in front shares the first column: no lane needs to ask
            }
            if (lane == 0) s_c[0] = c0;
        }
        if (w == NW - 1) {
            uint32_t c1 = NONE;
            if (f.end < f.rend) {
                c1 = cl;
                uint32_t at = f.end, it = 0;
                bool any = false;
                for (;;) {
                    const bool same = at + lane < f.rend && fe_col(probe) == c1;
                    const u64 stop = __ballot(!same);
                    const uint32_t n_same = stop ? (uint32_t)__ffsll((long long)stop) - 1u : 64u;
                    if (lane < n_same) atomicAdd(&sextra[probe >> 24], 1u);
                    any = any || n_same;
                    if (stop) break;
                    if (++it > SCAN_LIMIT) {
                        if (lane == 0) scal[6] = 1u;
                        break;
                    }
                    at += 64u;
                    probe = (at + lane < f.rend) ? ent[at + lane] : NONE;
                }
                if (!any) c1 = NONE;
            }
            if (lane == 0) s_c[1] = c1;
        }
        // ---- stable ranks inside (wave, h): the OR-mask match of k_scatter
        uint32_t wl = w * WCHUNK + lane;
        asm volatile("" : "+v"(wl));
        uint32_t *const wrow = &wcnt[w][0];
        u64 *const wmask = &smask[w * RADIX];
        uint32_t rank2[IPT / 2];  // two 16-bit ranks per register (0xFFFF: no entry); a rank is < FT = 8192
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const bool ok = wl + j * 64 < n;
            const uint32_t d = e[j] >> 24;
            u64 m = 0;
            uint32_t before = 0;
            if (ok) {
                atomicOr(&wmask[d], 1ull << lane);
                m = wmask[d];
                before = wrow[d];
            }
            const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
            const uint32_t rk = ok ? (before + below) : 0xFFFFu;
            rank2[j >> 1] = (j & 1) ? (rank2[j >> 1] | (rk << 16)) : rk;
            if (ok && below == 0) {
                wrow[d] = before + (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
                wmask[d] = 0ull;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        KK_STAMP(1);
        lds_sync();
        KK_STAMP(2);
        uint32_t dtot = 0;
        uint32_t wc[NW];
        if (tid < RADIX) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                wc[i] = dtot;
                dtot += wcnt[i][tid];
            }
        }
        uint32_t tile_count;
        const uint32_t dstart = block_excl_scan<true>(dtot, stmp, tile_count);
        if (tid < RADIX) {
#pragma unroll
            for (int i = 0; i < NW; ++i) wcnt[i][tid] = dstart + wc[i];
        }
#pragma unroll
        for (int q = 0; q < TQ; ++q) stot[tid + (uint32_t)q * NT] = totq[q];
        lds_sync();
        KK_STAMP(3);
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint32_t rk = (j & 1) ? (rank2[j >> 1] >> 16) : (rank2[j >> 1] & 0xFFFFu);
            if (rk != 0xFFFFu) stage[wrow[e[j] >> 24] + rk] = e[j];
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (tile + 1 < t1) load_tile(tile + 1);
        KK_STAMP(4);
        lds_sync();
        KK_STAMP(5);
        // Everything that was loaded for the NEXT tile is waited for HERE, before this tile's stores are issued: vmcnt
        // counts loads and stores in one queue and the compiler cannot count the stores of the emit loop, so a wait that
        // came after them would be vmcnt(0) and drain every store before the next tile could be ranked -- with all
        // workgroups in step, the chip then alternates between an LDS-bound and a store-bound phase.  (The empty asm uses
        // the registers, which makes the compiler wait for their loads now; afterwards it has nothing pending on them.)
#pragma unroll
        for (int j = 0; j < IPT; ++j) asm volatile("" : "+v"(e[j]));
        asm volatile("" : "+v"(e_first), "+v"(e_last), "+v"(probe), "+v"(gb));
        const uint32_t c0 = s_c[0], c1 = s_c[1];
        // ---- emit: wave w owns stage[a, b); v[k] = stage[a + 64 k + lane]
        const uint32_t a = w * WCHUNK;
#ifdef EM_NOEMIT
        if (a > n) {
#else
        if (a < n) {
#endif
            const uint32_t b = (a + WCHUNK < n) ? a + WCHUNK : n;
            // lanes of a round that are heads although they start a group: the group began in front of the tile
            auto not_heads = [&](uint32_t x, u64 firsts) -> u64 {
                if (c0 == NONE) return 0ull;                                    // (workgroup-uniform)
                const u64 cand = firsts & __ballot(fe_col(x) == c0);
                if (cand == 0ull) return 0ull;                                  // (wave-uniform) nearly always
                bool seen = false;
                if ((cand >> lane) & 1ull) seen = sseen[x >> 24] != 0u;
                return __ballot(seen);
            };
            // One sweep over the chunk, a round (64 entries) at a time, with the NEXT round already read: the last group of a
            // round ends where the next round's first group starts (or further on: the rare loop below).
            auto firsts_of = [&](uint32_t base, uint32_t x, uint32_t pv) -> u64 {  // lanes of [base, base + 64) that start a group
                if (base >= b) return 0ull;
                const uint32_t left = b - base;
                const u64 inm = left >= 64u ? ~0ull : ((1ull << left) - 1ull);
                return __ballot(x != pv || base + lane == 0u) & inm;
            };
            uint32_t x = stage[(a + lane) & (FT - 1)];  // (past b: whatever is there, masked)
            uint32_t pv = lane_prev(x);
            {
                const uint32_t c = stage[a ? a - 1u : 0u];
                if (lane == 0) pv = c;
            }
            u64 fmc = firsts_of(a, x, pv);
            // heads of the chunk's first digit that lie in front of the chunk (its run may start in an earlier wave's part)
            const uint32_t h_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)x) >> 24;
            uint32_t run = 0;  // heads of digit h_first in stage[dstart(h_first), a)
            {
                const uint32_t ds = wcnt[0][h_first];
                for (uint32_t q = ds; q < a; q += 64) {
                    const uint32_t i = q + lane;
                    const uint32_t y = stage[i & (FT - 1)], py = stage[(i ? i - 1u : 0u) & (FT - 1)];
                    const u64 firsts = __ballot(i < a && (y != py || i == 0u));
                    run += (uint32_t)__popcll(firsts & ~not_heads(y, firsts));
                }
            }
            // Two copies of the sweep: with every column total of the tile staged in LDS (the rule) the loop contains NO
            // global load, so nothing in it waits on the memory queue and the stores drain behind the next tile's ranking
            // (a load's wait is vmcnt(0) here: it would hold every round until the previous round's stores have landed).
            // Output position of a head = shb[h] + heads of digit h in front of it in the tile: inside the round from the
            // lane where the digit's run starts (a mask of the digit changes), else the count carried from round to round.
            auto sweep = [&](auto staged) {
                uint32_t dig_run = run;  // heads of the digit that is open at the start of the round, in front of the round
#pragma unroll 1
                for (uint32_t base = a; base < b; base += 64) {
                    const uint32_t nb = base + 64;
                    const uint32_t xn = stage[(nb + lane) & (FT - 1)];
                    uint32_t pn = lane_prev(xn);
                    const uint32_t xl = (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
                    if (lane == 0) pn = xl;
                    const u64 fmn = firsts_of(nb, xn, pn);
                    const uint32_t h = x >> 24, c = fe_col(x);
                    const u64 dchg = fmc & __ballot(h != (pv >> 24));  // lanes where another digit's run starts
                    const u64 hmk = fmc & ~not_heads(x, fmc);
                    if (fmc != 0ull) {  // (wave-uniform; a round inside one long group has nothing to emit)
                        const uint32_t hb = shb[h];
                        uint32_t total = 1u;
#ifndef EM_NOGATHER
                        if constexpr (decltype(staged)::value) total = stot[(c - cf) & (EM_TOT_CAP - 1)];  // (any lane: LDS)
#endif
                        // where the round's last group ends
                        uint32_t nxt_last;
                        if (fmn != 0ull) {
                            nxt_last = nb + (uint32_t)__ffsll((long long)fmn) - 1u;
                        } else {  // the chunk ends here, or a group longer than a round: look for the next different entry
                            const uint32_t lastv = (uint32_t)__builtin_amdgcn_readlane((int)x, (int)(63u - (uint32_t)__clzll((long long)fmc)));
                            nxt_last = n;
                            for (uint32_t q = nb; q < n; q += 64) {
                                const uint32_t i = q + lane;
                                const u64 dm = __ballot(i < n && stage[i & (FT - 1)] != lastv);
                                if (dm) {
                                    nxt_last = q + (uint32_t)__ffsll((long long)dm) - 1u;
                                    break;
                                }
                            }
                        }
                        const uint32_t i = base + lane;
                        const u64 le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                        const u64 chg_le = dchg & le;
                        const uint32_t s0 = chg_le ? 63u - (uint32_t)__clzll((long long)chg_le) : 0u;  // my digit's first lane here
                        const u64 hseg = hmk & ~lanes_below(s0);
                        const uint32_t hbefore = __builtin_amdgcn_mbcnt_hi((uint32_t)(hseg >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hseg, 0u)) +
                                                 (chg_le ? 0u : dig_run);
                        if ((hmk >> lane) & 1ull) {
                            const u64 above = (lane == 63) ? 0ull : (fmc >> (lane + 1));
                            const uint32_t nxt = above ? i + (uint32_t)__ffsll((long long)above) : nxt_last;
                            uint32_t cnt = nxt - i;
                            if (c == c1) cnt += sextra[h];  // (rare) the group goes on behind the tile
#ifndef EM_NOGATHER
                            if constexpr (!decltype(staged)::value) total = rec_total[c];  // (a head's column: in range)
#endif
#ifdef EM_LINEAR  // diagnostic builds (wrong results): every wave stores to a stream of its own, in whole lines
                            const uint64_t o = ((uint64_t)tile * NW + w) * WCHUNK + (base - a) + lane + (hb & 0u);
#else
                            const uint64_t o = (uint32_t)(hb + hbefore);
#endif
                            if (o < max_nz) {
                                // reference: float(count) / total_count with total_count uint32_t: both operands are rounded
                                // to float, then one IEEE float division (hipcc's default float division is correctly rounded)
#ifdef EM_NOSTORE
                                if (total == 0x7FFFFFFFu)
#endif
                                data[o] = (float)cnt / (float)total;
#ifdef EM_NOSTORE
                                if (total == 0x7FFFFFFFu)
#endif
                                col_ind[o] = first_col + c;
                            }
                        }
                        // the digit that is open behind this round, and its heads so far
                        if (dchg) {
                            const uint32_t sl = 63u - (uint32_t)__clzll((long long)dchg);
                            dig_run = (uint32_t)__popcll(hmk & ~lanes_below(sl));
                        } else {
                            dig_run += (uint32_t)__popcll(hmk);
                        }
                    }
                    x = xn;
                    pv = pn;
                    fmc = fmn;
                }
            };
            if (cl - cf < EM_TOT_CAP)  // (workgroup-uniform)
                sweep(std::true_type{});
            else
                sweep(std::false_type{});
        }
        KK_STAMP(6);
        lds_sync();
        KK_STAMP(7);
    }
    KK_STAMP_END(8);
}

// ------------------------------------------------------------------------------------------------------
// X for 5 <= K <= 8: per-isoform LDS tables ("count first, then transpose")
// ------------------------------------------------------------------------------------------------------
//
// What it replaces: collect_kmers + fill_indices_coo (reference src/libkmers.cpp:218-238, one unordered_map per
// isoform) and ThreadPoolLocalData::to_csr (:77-169).  For K <= 8 all 4^K k-mers of an isoform fit in one LDS table
// (16-bit fields: 128 KB at K = 8), so a workgroup counts an isoform's k-mers exactly as the reference's map does and
// emits ONE entry per distinct (k-mer, isoform) with its count.  Every entry is then a CSR non-zero: its final
// position follows from histograms alone, the last partition pass writes data / col_ind in place, and the run-length
// reduce (k_rle) and its scan across tiles are not needed at all.
//
//   k_dd<false>  per isoform: first occurrences of every k-mer (a 4^K-bit LDS set) -> number of entries per
//                (tile, low digit); valid windows per isoform (the reference's total_count)
//   scan         -> position of the first entry of every (low digit, tile)
//   k_dd<true>   per isoform: count in the LDS table, then every window claims its k-mer's field (a returning
//                atomic AND that clears it): the claimer emits  high digit << 24 | (count - 1) << colbits | column
//                into the tile's LDS staging, grouped by low digit; coalesced runs out
//   k_hist_entries, scan          (shared with the general path: histogram of the high digit)
//   k_row_counts, k_row_ptr       row_ind from the two scans
//   k_final_scatter               stable partition by the high digit that writes data / col_ind directly
//   k_fix_escapes                 counts that do not fit the entry's count field (poly-A): values patched afterwards
//
// Work split: workgroup w owns the isoforms that START in its range of tiles (a tile = DD_C positions of the store);
// a tile's entries go to histogram row tile + w, so a tile shared by two workgroups (one's last isoform ends where the
// next one's first begins) has a row per workgroup and rows stay in position order.  Isoforms are processed one after
// the other (they share the table); the order of the entries of one isoform inside a (tile, digit) group is whatever
// the LDS atomics hand out, which is harmless: an isoform has one entry per k-mer, and entries of different isoforms
// never interleave.  Which TILE a k-mer's entry belongs to is the tile of its first occurrence, in both kernels
// (tiles of an isoform are separated by barriers).
// Fields are 16 bits: a k-mer that occurs 65536 times in ONE isoform would wrap; isoforms long enough for that count
// with returning atomics and raise scal[6], upon which the host reruns the build on the general path.

constexpr int DD_NT = 1024;          // one workgroup per CU
constexpr int DD_W = 6;              // windows per thread and tile
constexpr int DD_C = DD_NT * DD_W;   // positions per tile = capacity of the entry staging
#ifndef KK_DD_GRID
#define KK_DD_GRID 1024
#endif
constexpr int DD_GRID = KK_DD_GRID;
constexpr uint32_t DD_MAX_REC = 1u << 22;   // columns fit 22 bits, so the count field has at least 2

struct DDArgs {
    const uint32_t *bases, *valid;
    uint64_t bases_words, valid_words;
    const uint64_t *rec_start;
    uint32_t n_rec;
    int K;
    const uint32_t *wg_rec;  // [grid + 1]: first record of every workgroup
};

struct DDEscape {
    uint32_t key, col, count;
};

// wg_rec[w] = first record that starts at or after the first position of workgroup w's tiles; wg_rec[grid] = n_rec.
__global__ void k_dd_ranges(const uint64_t *rec_start, uint32_t n_rec, uint64_t n_pos, uint32_t grid, uint32_t *wg_rec) {
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w > grid) return;
    const uint64_t tiles = (n_pos + DD_C - 1) / DD_C, per = (tiles + grid - 1) / grid;
    const uint64_t target = (uint64_t)w * per * DD_C;
    uint32_t lo = 0, hi = n_rec;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (rec_start[mid] < target)
            lo = mid + 1;
        else
            hi = mid;
    }
    wg_rec[w] = (w == grid) ? n_rec : lo;
}

struct DDWords {
    uint32_t b0, b1, v0, v1;
};
// the words that hold the DD_W windows starting at p0 .. p0 + DD_W - 1 (K <= 8: 13 positions, never more than two words)
__device__ __forceinline__ DDWords dd_fetch(const DDArgs &a, uint32_t p0) {
    const uint64_t bw = p0 >> 4, vw = p0 >> 5;
    DDWords x;
    x.b0 = (bw < a.bases_words) ? a.bases[bw] : 0u;
    x.b1 = (bw + 1 < a.bases_words) ? a.bases[bw + 1] : 0u;
    x.v0 = (vw < a.valid_words) ? a.valid[vw] : 0u;
    x.v1 = (vw + 1 < a.valid_words) ? a.valid[vw + 1] : 0u;
    return x;
}
__device__ __forceinline__ void dd_decode(const DDWords &x, uint32_t p0, int K, uint32_t (&key)[DD_W], uint32_t &okm) {
    const u64 B = ((u64)x.b0 << 32) | x.b1, V = ((u64)x.v1 << 32) | x.v0;
    const uint32_t ob = p0 & 15u, ov = p0 & 31u, need = (1u << K) - 1u, mask = (1u << (2 * K)) - 1u;
    okm = 0;
#pragma unroll
    for (int j = 0; j < DD_W; ++j) {
        key[j] = (uint32_t)(B >> (64u - 2u * (ob + (uint32_t)j + (uint32_t)K))) & mask;  // shift >= 64 - 2 (15 + 5 + 8) = 8
        const uint32_t ok = (((uint32_t)(V >> (ov + (uint32_t)j))) & need) == need;      // ov + j + K <= 44 bits
        okm |= ok << j;
    }
}

// The tile's packed words, staged in LDS: a thread's DD_W windows are DD_NT positions apart (see k_dd), so it reads them
// from here rather than from two registers.
struct DDTile {
    uint32_t sb[DD_C / 16 + 2];
    uint32_t sv[DD_C / 32 + 2];
};
constexpr int DD_TB = DD_C / 16 + 2, DD_TV = DD_C / 32 + 2;
static_assert(DD_TB + DD_TV <= DD_NT, "one staged word per thread");
// thread tid's word of the tile that starts at position tile_lo (a multiple of DD_C, hence of 32)
__device__ __forceinline__ uint32_t dd_tile_fetch(const DDArgs &a, uint32_t tile_lo, uint32_t tid) {
    if (tid < (uint32_t)DD_TB) {
        const uint64_t bw = (uint64_t)(tile_lo >> 4) + tid;
        return (bw < a.bases_words) ? a.bases[bw] : 0u;
    }
    if (tid < (uint32_t)(DD_TB + DD_TV)) {
        const uint64_t vw = (uint64_t)(tile_lo >> 5) + (tid - DD_TB);
        return (vw < a.valid_words) ? a.valid[vw] : 0u;
    }
    return 0u;
}
// window at tile-relative position o: valid iff its K positions are; key = the reference's k-mer id (K <= 8)
__device__ __forceinline__ bool dd_window(const DDTile &T, uint32_t o, int K, uint32_t &key) {
    const uint32_t vi = o >> 5, bi = o >> 4;
    const u64 v = ((u64)T.sv[vi + 1] << 32) | T.sv[vi];
    const u64 x = ((u64)T.sb[bi] << 32) | T.sb[bi + 1];
    const uint32_t need = (1u << K) - 1u;
    key = (uint32_t)(x >> (64u - 2u * (o & 15u) - 2u * (uint32_t)K)) & ((1u << (2 * K)) - 1u);
    return ((uint32_t)(v >> (o & 31u)) & need) == need;
}

// Calls f(integral_constant<int, n>) for the wave-uniform n in 0 .. DD_W: the per-isoform code is unrolled for exactly
// the windows an isoform's stretch reaches in this wave, not for all six.
template <class F>
__device__ __forceinline__ void dd_dispatch(uint32_t n, const F &f) {
    switch (n) {
        case 0: break;
        case 1: f(std::integral_constant<int, 1>{}); break;
        case 2: f(std::integral_constant<int, 2>{}); break;
        case 3: f(std::integral_constant<int, 3>{}); break;
        case 4: f(std::integral_constant<int, 4>{}); break;
        case 5: f(std::integral_constant<int, 5>{}); break;
        default: f(std::integral_constant<int, 6>{}); break;
    }
}

// Thread tid takes the windows at tile positions tid, tid + DD_NT, ... (DD_W of them): an isoform's stretch of the tile
// is then spread over ALL 16 waves (a stretch of 2400 positions is two or three windows per thread), instead of keeping
// 6 of them busy with six windows each while the others wait at the barrier.
template <bool SCATTER>
__global__ __launch_bounds__(DD_NT) void k_dd(DDArgs a, uint32_t *hist, uint32_t *rec_total, uint32_t *out, uint32_t colbits,
                                              DDEscape *esc, uint32_t esc_cap, uint32_t *scal) {
    constexpr uint32_t NSET = 8;                          // !SCATTER: 4^K-bit sets, one per isoform in flight
    constexpr uint32_t TABW = SCATTER ? 32768u : NSET * 2048u;  // SCATTER: 4^K 16-bit count fields
    constexpr int NRS = 255;                             // record boundaries staged at a time
    __shared__ uint32_t tab[TABW];
    __shared__ uint32_t sent[SCATTER ? DD_C : 1];
    __shared__ uint32_t cnt[RADIX], lbase[RADIX + 1], gbase[RADIX];
    __shared__ uint32_t s_rs[NRS + 1];
    __shared__ uint32_t swave[RADIX / 64];
    __shared__ DDTile T;
    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6, w = blockIdx.x;
    const uint32_t r_lo = a.wg_rec[w], r_hi = a.wg_rec[w + 1];
    if (r_lo >= r_hi) return;
    const int K = a.K;
    const uint32_t bm_words = ((1u << (2 * K)) + 31u) / 32u;  // words of one 4^K-bit set
    const uint32_t tab_words = SCATTER ? (1u << (2 * K)) / 2u : NSET * bm_words;
    const uint32_t fmax = (1u << (24u - colbits)) - 1u;
    // !SCATTER: isoform after isoform takes the next set of the ring, and NO barrier separates them (a wave goes on to the
    // next isoform as soon as its own windows are done); sets [dirty_lo, slot) belong to finished isoforms and are
    // cleared at the end of the tile, or when the ring is full
    uint32_t slot = 0, dirty_lo = 0;
    auto clear_sets = [&](uint32_t from, uint32_t n_sets) {  // all threads, between two barriers
        for (uint32_t q = 0; q < n_sets; ++q) {
            const uint32_t b0 = ((from + q) % NSET) * bm_words;
            for (uint32_t i = tid; i < bm_words; i += DD_NT) tab[b0 + i] = 0;
        }
    };
    for (uint32_t i = tid; i < tab_words; i += DD_NT) tab[i] = 0;
    if (tid < RADIX) cnt[tid] = 0;
    uint32_t r_base = r_lo;
    auto load_rs = [&]() {  // s_rs[i] = rec_start[r_base + i]; all threads
        lds_sync();
        for (uint32_t i = tid; i <= (uint32_t)NRS; i += DD_NT) {
            const uint32_t rr = r_base + i;
            s_rs[i] = (uint32_t)a.rec_start[rr < a.n_rec ? rr : a.n_rec];
        }
        lds_sync();
    };
    load_rs();
    uint32_t b_cur = s_rs[0], b_nxt = s_rs[1], b_nn = s_rs[2];  // rec_start of records r, r + 1, r + 2
    const uint32_t P_lo = s_rs[0], P_hi = (uint32_t)a.rec_start[r_hi];
    const uint32_t t_first = P_lo / DD_C, t_last = (P_hi - 1u) / DD_C;  // a record holds at least its separator
    uint32_t r = r_lo;
    const uint32_t wave_t0 = wv * 64u;  // the wave's first thread
    uint32_t nxw = dd_tile_fetch(a, t_first * DD_C, tid);
    uint32_t gb = 0, gc = 0;  // SCATTER: position of the first entry of (digit tid, this row); entries of the row
    if (SCATTER && tid < RADIX) {
        const uint64_t row = (uint64_t)t_first + w;
        gb = hist[row * RADIX + tid];
        gc = hist[(row + 1) * RADIX + tid] - gb;
    }
    KK_STAMP_BEGIN();
    for (uint32_t t = t_first; t <= t_last; ++t) {
        KK_STAMP(7);
        const uint32_t tile_lo = t * DD_C;
        const uint32_t tile_hi = (tile_lo + DD_C < P_hi) ? tile_lo + DD_C : P_hi;
        // (the previous tile ended with a barrier: T is free)
        if (tid < (uint32_t)DD_TB)
            T.sb[tid] = nxw;
        else if (tid < (uint32_t)(DD_TB + DD_TV))
            T.sv[tid - DD_TB] = nxw;
        if (t < t_last) nxw = dd_tile_fetch(a, (t + 1) * DD_C, tid);
        uint32_t inc = 0;
        if constexpr (SCATTER) {  // LDS slot of every digit's first entry = exclusive scan of the row's counts
            inc = wave_incl_scan((tid < RADIX) ? gc : 0u);
            if (tid < RADIX && lane == 63) swave[wv] = inc;
        }
        lds_sync();
        uint32_t key[DD_W], okm = 0;
#pragma unroll
        for (int j = 0; j < DD_W; ++j) okm |= (uint32_t)dd_window(T, tid + (uint32_t)j * DD_NT, K, key[j]) << j;
        // the six 16-bit ids in three registers: the per-isoform code below walks only the windows j0 .. j0 + n - 1 that
        // the isoform's stretch can reach in this wave (j0, n wave-uniform), and picks key j with a shift
        // (shifts, not a select among registers: the compiler turns such a select chain into a table in scratch memory)
        const u64 kq = (u64)(key[0] | (key[1] << 16)) | ((u64)(key[2] | (key[3] << 16)) << 32);
        const uint32_t kr = key[4] | (key[5] << 16);
        auto key_at = [kq, kr](uint32_t j) -> uint32_t {
            const uint32_t lo4 = (uint32_t)(kq >> ((j & 3u) * 16u)), hi2 = kr >> ((j & 1u) * 16u);
            return ((j < 4u) ? lo4 : hi2) & 0xFFFFu;
        };
        if constexpr (SCATTER) {
            if (tid < RADIX) {
                uint32_t before = 0;
#pragma unroll
                for (int i = 0; i < RADIX / 64; ++i)
                    if ((uint32_t)i < wv) before += swave[i];
                lbase[tid] = before + inc - gc;
                if (tid == RADIX - 1) lbase[RADIX] = before + inc;
                gbase[tid] = gb;
                cnt[tid] = 0;
                if (t < t_last) {
                    const uint64_t row = (uint64_t)t + 1 + w;
                    gb = hist[row * RADIX + tid];
                    gc = hist[(row + 1) * RADIX + tid] - gb;
                }
            }
            lds_sync();
        }
        KK_STAMP(0);  // tile start: stage + decode, digit starts
        // ---- the isoforms that have bases in this tile, one after the other
        while (r < r_hi) {
            if (r + 2 - r_base > (uint32_t)NRS) {  // (uniform) the staged boundaries are used up
                r_base = r;
                load_rs();
                b_cur = s_rs[0];
                b_nxt = s_rs[1];
                b_nn = s_rs[2];
            }
            // bases [ra, re), separator at re: the same for every thread, so in scalar registers.  (b_nn, the bound after
            // the next, was read from LDS an isoform ago: no round trip here.)
            // (kept in vector registers on purpose: the CU's one scalar unit serves all 16 waves, and a few dozen scalar
            // instructions per wave and isoform made it the kernel's bottleneck)
            const uint32_t ra = b_cur, re = b_nxt - 1u;
            if (ra >= tile_hi) break;
            const bool ends_here = re < tile_hi;
            if (re - ra >= (uint32_t)K) {  // the isoform has windows
                // the thread's windows inside the isoform's stretch [sl, sh) of the tile: positions tid + j * DD_NT, so
                // j in [jlo, jhi).  (A window that runs over the isoform's end is invalid: the separator's bit is 0.)
                const uint32_t sl = (ra > tile_lo ? ra : tile_lo) - tile_lo, sh_ = (re < tile_hi ? re : tile_hi) - tile_lo;
                const int dlo = (int)sl - (int)tid + (DD_NT - 1), dhi = (int)sh_ - (int)tid + (DD_NT - 1);
                const uint32_t jlo = (uint32_t)(dlo < 0 ? 0 : dlo / DD_NT), jhi = (uint32_t)(dhi < 0 ? 0 : dhi / DD_NT);  // <= DD_W
                const uint32_t m = okm & ((1u << jhi) - 1u) & ~((1u << jlo) - 1u);
                // the same bounds for the wave's last / first lane (jlo, jhi fall with tid): windows j0 .. j0 + nj - 1
                const int wlo = (int)sl - (int)wave_t0 - 63 + (DD_NT - 1), whi = (int)sh_ - (int)wave_t0 + (DD_NT - 1);
                const uint32_t j0 = (uint32_t)__builtin_amdgcn_readfirstlane(wlo < 0 ? 0 : wlo / DD_NT);
                const uint32_t nj = (uint32_t)__builtin_amdgcn_readfirstlane(whi < 0 ? 0 : whi / DD_NT) - j0;
                KK_STAMP(1);  // isoform bounds
                if constexpr (!SCATTER) {
                    const uint32_t bm0 = slot * bm_words;
                    // all of the thread's returning atomics first, then what depends on them (one round trip)
                    auto body = [&](auto nc) {
                        constexpr int N = decltype(nc)::value;
                        uint32_t old[N], kk_[N];
#pragma unroll
                        for (int u = 0; u < N; ++u) {
                            const uint32_t j = j0 + (uint32_t)u;
                            kk_[u] = key_at(j);
                            old[u] = ~0u;
                            if ((m >> j) & 1u)
                                old[u] = __hip_atomic_fetch_or(&tab[bm0 + (kk_[u] >> 5)], 1u << (kk_[u] & 31u), __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        // valid windows (the reference's total_count): scalar popcounts of ballots, one global add per wave
                        uint32_t nv = 0;
#pragma unroll
                        for (int u = 0; u < N; ++u) nv += (uint32_t)__popcll(__ballot((m >> (j0 + (uint32_t)u)) & 1u));
                        if (lane == 0 && nv) atomicAdd(&rec_total[r], nv);
#pragma unroll
                        for (int u = 0; u < N; ++u)
                            if (!((old[u] >> (kk_[u] & 31u)) & 1u)) atomicAdd(&cnt[kk_[u] & 255u], 1u);  // the k-mer's first occurrence
                    };
                    dd_dispatch(nj, body);
                    KK_STAMP(2);  // set + histogram atomics
                    if (ends_here) {
                        slot = (slot + 1u) % NSET;
                        if (slot == dirty_lo) {  // the ring is full (a tile of many short isoforms): clear all of it
                            lds_sync();
                            clear_sets(0, NSET);
                            lds_sync();
                        }
                        KK_STAMP(5);
                    }
                } else {
                    if (ra >= tile_lo) {  // the isoform starts here: count ALL its windows before any is claimed
                        const bool check = (re - ra) >= 65535u + (uint32_t)K;  // 65536 windows or more: a field may wrap
                        auto add = [&](const uint32_t (&kk_)[DD_W], uint32_t mm) {
#pragma unroll
                            for (int j = 0; j < DD_W; ++j) {
                                if ((mm >> j) & 1u) {
                                    const uint32_t sh = (kk_[j] & 1u) * 16u;
                                    if (!check) {
                                        atomicAdd(&tab[kk_[j] >> 1], 1u << sh);
                                    } else {
                                        const uint32_t old = atomicAdd(&tab[kk_[j] >> 1], 1u << sh);
                                        if (((old >> sh) & 0xFFFFu) == 0xFFFFu) scal[6] = 1u;
                                    }
                                }
                            }
                        };
                        auto add_body = [&](auto nc) {
                            constexpr int N = decltype(nc)::value;
#pragma unroll
                            for (int u = 0; u < N; ++u) {
                                const uint32_t j = j0 + (uint32_t)u;
                                if ((m >> j) & 1u) {
                                    const uint32_t kj = key_at(j), sh = (kj & 1u) * 16u;
                                    if (!check) {
                                        atomicAdd(&tab[kj >> 1], 1u << sh);
                                    } else {
                                        const uint32_t old = atomicAdd(&tab[kj >> 1], 1u << sh);
                                        if (((old >> sh) & 0xFFFFu) == 0xFFFFu) scal[6] = 1u;
                                    }
                                }
                            }
                        };
                        dd_dispatch(nj, add_body);
                        // the rest of a long isoform: DD_W consecutive windows per thread, straight from the store
                        for (uint32_t tt = t + 1; (uint64_t)tt * DD_C < re; ++tt) {
                            const uint32_t g0 = tt * DD_C + DD_W * tid;
                            if (g0 - DD_W * lane >= re) break;  // (wave-uniform) nothing of it in this wave's positions
                            const DDWords xw = dd_fetch(a, g0);
                            uint32_t k2[DD_W], ok2;
                            dd_decode(xw, g0, K, k2, ok2);
                            const int64_t eh = (int64_t)re - (int64_t)g0;
                            const uint32_t jh = eh < 0 ? 0u : (eh > DD_W ? (uint32_t)DD_W : (uint32_t)eh);
                            add(k2, ok2 & ((1u << jh) - 1u));
                        }
                        KK_STAMP(2);  // count
                        lds_sync();
                        KK_STAMP(3);
                    }
                    // claim: all returning atomics of the thread first.  Once every wave has its answers the table is
                    // clean again, so the barrier that lets the next isoform count comes HERE; the ranks and the staging of
                    // this isoform's entries (other LDS arrays) then run beside the next isoform's counting.
                    uint32_t cc[DD_W], kc[DD_W];
                    auto claim = [&](auto nc) {
                        constexpr int N = decltype(nc)::value;
#pragma unroll
                        for (int u = 0; u < N; ++u) {
                            const uint32_t j = j0 + (uint32_t)u;
                            kc[u] = key_at(j);
                            cc[u] = 0;
                            if ((m >> j) & 1u) {
                                const uint32_t sh = (kc[u] & 1u) * 16u;
                                const uint32_t old = __hip_atomic_fetch_and(&tab[kc[u] >> 1], ~(0xFFFFu << sh), __ATOMIC_RELAXED,
                                                                            __HIP_MEMORY_SCOPE_WORKGROUP);
                                cc[u] = (old >> sh) & 0xFFFFu;
                            }
                        }
                    };
                    dd_dispatch(nj, claim);
                    KK_STAMP(4);  // claim
                    if (ends_here) lds_sync();
                    KK_STAMP(5);
                    const uint32_t rr = r;
                    auto stage = [&](auto nc) {
                        constexpr int N = decltype(nc)::value;
                        uint32_t slot[N];
#pragma unroll
                        for (int u = 0; u < N; ++u) {
                            slot[u] = 0;
                            if (cc[u]) {  // first to claim the k-mer: this window emits its entry
                                const uint32_t d = kc[u] & 255u;
                                slot[u] = atomicAdd(&cnt[d], 1u) + lbase[d];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < N; ++u) {
                            if (cc[u]) {
                                const uint32_t f = (cc[u] - 1u < fmax) ? cc[u] - 1u : fmax;
                                if (slot[u] < (uint32_t)DD_C) sent[slot[u]] = ((kc[u] >> 8) << 24) | (f << colbits) | rr;
                                if (f == fmax) {  // a count the entry's field cannot hold (poly-A): queued, patched after the last pass
                                    const uint32_t at = atomicAdd(&scal[7], 1u);
                                    if (at < esc_cap)
                                        esc[at] = DDEscape{kc[u], rr, cc[u]};
                                    else
                                        scal[6] = 1u;
                                }
                            }
                        }
                    };
                    dd_dispatch(nj, stage);
                    KK_STAMP(5);
                }
            }
            if (!ends_here) break;
            ++r;
            b_cur = b_nxt;
            b_nxt = b_nn;
            b_nn = s_rs[(r + 2 - r_base <= (uint32_t)NRS) ? r + 2 - r_base : 0u];
        }
        // ---- end of the tile
        KK_STAMP(1);
        lds_sync();
        if constexpr (!SCATTER) {
            if (tid < RADIX) {
                hist[((uint64_t)t + w) * RADIX + tid] = cnt[tid];
                cnt[tid] = 0;
            }
            clear_sets(dirty_lo, (slot + NSET - dirty_lo) % NSET);  // the finished isoforms' sets (an unfinished one keeps `slot`)
            dirty_lo = slot;
        } else {
            if (tid < RADIX && cnt[tid] != lbase[tid + 1] - lbase[tid]) scal[3] = 3u;  // the two kernels disagree: a bug
            // a half-wave per digit (runs of ~24 entries, whole runs per store instruction); the bounds of the half-wave's
            // 8 digits are read first, then all first chunks, then stored: two LDS round trips for the common case
            constexpr int DPH = RADIX / (2 * (DD_NT / 64));  // digits per half-wave
            const uint32_t d0 = wv * (2 * DPH) + (lane >> 5) * DPH, l5 = lane & 31u;
            uint32_t s0[DPH], n[DPH], g0[DPH], v[DPH];
#pragma unroll
            for (int it = 0; it < DPH; ++it) {
                s0[it] = lbase[d0 + it];
                n[it] = lbase[d0 + it + 1] - s0[it];
                g0[it] = gbase[d0 + it];
            }
#pragma unroll
            for (int it = 0; it < DPH; ++it) v[it] = (l5 < n[it]) ? sent[s0[it] + l5] : 0u;
#pragma unroll
            for (int it = 0; it < DPH; ++it) {
                if (l5 < n[it]) out[g0[it] + l5] = v[it];
                for (uint32_t i = l5 + 32u; i < n[it]; i += 32u) out[g0[it] + i] = sent[s0[it] + i];
            }
        }
        lds_sync();
        KK_STAMP(6);  // flush
    }
    KK_STAMP_END(SCATTER ? 8 : 0);
}

// rowcount[high << 8 | low] = entries of that k-mer, from the partitioned array `ent` (regions0: the 257 region bounds
// of the low digit) and the scanned tile histogram of the high digit (base1, one extra all-zero row at the end): whole
// tiles inside a region are a difference of two scanned rows, the ragged ends are counted here.  One workgroup per region.
__global__ __launch_bounds__(RADIX) void k_row_counts(const uint32_t *ent, const uint32_t *regions0, const uint32_t *base1,
                                                      uint32_t n_hi, uint32_t *rowcount) {
    __shared__ uint32_t h[RADIX];
    const uint32_t lo = blockIdx.x, hi = threadIdx.x;
    h[hi] = 0;
    __syncthreads();
    const uint32_t begin = regions0[lo], end = regions0[lo + 1];
    const uint32_t ta = (begin + TILE - 1) / TILE, tb = end / TILE;  // whole tiles [ta, tb)
    uint32_t full = 0;
    uint32_t e0 = end, b1 = end;  // ragged ranges [begin, e0) and [b1, end)
    if (ta < tb) {
        full = base1[(uint64_t)tb * RADIX + hi] - base1[(uint64_t)ta * RADIX + hi];
        e0 = ta * TILE;
        b1 = tb * TILE;
    }
    for (uint32_t i = begin + hi; i < e0; i += RADIX) atomicAdd(&h[ent[i] >> 24], 1u);
    for (uint32_t i = b1 + hi; i < end; i += RADIX) atomicAdd(&h[ent[i] >> 24], 1u);
    __syncthreads();
    if (hi < n_hi) rowcount[(hi << 8) | lo] = full + h[hi];
}

// row_ind[k] = exclusive prefix sum of rowcount (M <= 65536 rows, one workgroup)
__global__ __launch_bounds__(1024) void k_row_ptr(const uint32_t *rowcount, uint32_t M, uint64_t *row_ind) {
    __shared__ uint32_t sw[16];
    const uint32_t tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    const uint32_t per = (M + 1023u) / 1024u, k0 = tid * per;
    uint32_t s = 0;
    for (uint32_t i = 0; i < per; ++i)
        if (k0 + i < M) s += rowcount[k0 + i];
    uint32_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t x = __shfl_up(inc, o);
        if (lane >= (uint32_t)o) inc += x;
    }
    if (lane == 63) sw[wv] = inc;
    __syncthreads();
    uint32_t run = inc - s;
    for (uint32_t i = 0; i < wv; ++i) run += sw[i];
    for (uint32_t i = 0; i < per; ++i) {
        if (k0 + i < M) {
            row_ind[k0 + i] = run;
            run += rowcount[k0 + i];
        }
    }
    if (tid == 1023) row_ind[M] = run;
}

// The last partition pass of the small-K path: stable by the high digit (entry bits 24..31), and instead of entries it
// writes the CSR arrays themselves -- column and value = float(count) / float(valid windows of the column), the
// reference's `float(count) / total_count` (src/libkmers.cpp:236).  Same ranking as k_scatter.
__global__ __launch_bounds__(NT, 4) void k_final_scatter(const uint32_t *in, const uint32_t *p_count, const uint32_t *tilebase,
                                                         const uint32_t *rec_total, uint32_t colbits, uint64_t first_col,
                                                         uint64_t max_nz, float *data, uint64_t *col_ind) {
    __shared__ __attribute__((aligned(16))) uint32_t sent[TILE];
    __shared__ uint32_t wcnt[NW][RADIX];
    __shared__ uint32_t s_n[RADIX];     // entries of every digit in the current tile
    __shared__ uint32_t stmp[NW];
    // Write-out in whole 128-byte lines: the entries of digit d go to consecutive output positions across ALL the tiles
    // of this workgroup, so the tail of a tile's run that does not fill its 32-entry group (32 floats = one line of data,
    // two of col_ind) waits here for the next tile's entries of that digit.  Only the first and the last group of a
    // digit in the workgroup's range are written in part.
    __shared__ uint32_t carry[RADIX][32];
    __shared__ uint32_t cpend[RADIX], cpos[RADIX];  // waiting entries of the digit / output position of the first of them
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = lane_id();
    const uint32_t E = *p_count;
    const uint32_t nt = (uint32_t)(((uint64_t)E + TILE - 1) / TILE);
    const uint32_t per = (nt + gridDim.x - 1) / gridDim.x;
    const uint32_t t0 = blockIdx.x * per, t1 = (t0 + per < nt) ? t0 + per : nt;
    if (t0 >= t1) return;
    u64 *const smask = reinterpret_cast<u64 *>(sent);
    const uint32_t colmask = (1u << colbits) - 1u, fmask = (1u << (24u - colbits)) - 1u;
    uint32_t e[IPT];
    auto load_tile = [&](uint32_t t) {
        const uint32_t c0 = t * TILE + w * WCHUNK;
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const uint64_t i = (uint64_t)c0 + j * 64 + lane;
            e[j] = (i < E) ? in[i] : 0u;
        }
    };
    load_tile(t0);
    if (tid < RADIX) {
        cpend[tid] = 0;
        cpos[tid] = tilebase[(uint64_t)t0 * RADIX + tid];
    }
    auto emit = [&](uint32_t pos, uint32_t v, uint32_t tot) {
#ifdef KK_FS_NOSTORE  // diagnostic builds (wrong results): what the stores cost
        if (tot == 0x7FFFFFFFu)
#endif
        if ((uint64_t)pos < max_nz) {
            // both operands rounded to float, one IEEE division (hipcc's default float division is correctly rounded);
            // an escaped count (f == fmask) is patched by k_fix_escapes
            const uint32_t f = (v >> colbits) & fmask;
            data[pos] = (float)(f + 1u) / (float)tot;
            col_ind[pos] = first_col + (v & colmask);
        }
    };
    const uint32_t hw = tid >> 5, l5 = tid & 31u;  // a half-wave per digit, 16 digits per half-wave and tile
    for (uint32_t tile = t0; tile < t1; ++tile) {
        const uint64_t base = (uint64_t)tile * TILE;
        for (uint32_t i = tid; i < NW * RADIX; i += NT) {
            (&wcnt[0][0])[i] = 0;
            smask[i] = 0ull;
        }
        lds_sync();
        uint32_t wl = w * WCHUNK + lane;
        asm volatile("" : "+v"(wl));
        const uint32_t n_valid = (uint32_t)((E - base < TILE) ? E - base : TILE);
        uint32_t *const wrow = &wcnt[w][0];
        u64 *const wmask = &smask[w * RADIX];
        uint32_t rank[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const bool ok = wl + j * 64 < n_valid;
            const uint32_t d = e[j] >> 24;
            u64 m = 0;
            uint32_t before = 0;
            if (ok) {
                atomicOr(&wmask[d], 1ull << lane);
                m = wmask[d];
                before = wrow[d];
            }
            const uint32_t m_lo = (uint32_t)m, m_hi = (uint32_t)(m >> 32);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u));
            rank[j] = ok ? (before + below) : NONE;
            if (ok && below == 0) {
                wrow[d] = before + (uint32_t)__popc(m_lo) + (uint32_t)__popc(m_hi);
                wmask[d] = 0ull;
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_sync();
        uint32_t dtot = 0;
        uint32_t wc[NW];
        if (tid < RADIX) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                wc[i] = dtot;
                dtot += wcnt[i][tid];
            }
        }
        uint32_t tile_count;
        const uint32_t dstart = block_excl_scan<true>(dtot, stmp, tile_count);
        if (tid < RADIX) {
#pragma unroll
            for (int i = 0; i < NW; ++i) wcnt[i][tid] = dstart + wc[i];
            s_n[tid] = dtot;
        }
        lds_sync();
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            if (rank[j] != NONE) sent[wrow[e[j] >> 24] + rank[j]] = e[j];
            if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (tile + 1 < t1) load_tile(tile + 1);
        lds_sync();
        // four digits per half-wave at a time: their bounds, then their entries, then the gathers of the column totals,
        // then the stores (one group each in the common case; the loop over k takes the longer runs)
        constexpr int DB = 4;
#pragma unroll 1
        for (int it = 0; it < RADIX / (NT / 32); it += DB) {
            uint32_t ls[DB], np[DB], p[DB], ps[DB], a[DB], full[DB];
            uint32_t maxfull = 0;
#pragma unroll
            for (int u = 0; u < DB; ++u) {
                const uint32_t d = hw + (uint32_t)(it + u) * (NT / 32);
                ls[u] = wcnt[0][d];
                np[u] = s_n[d];
                p[u] = cpend[d];
                ps[u] = cpos[d];
            }
#pragma unroll
            for (int u = 0; u < DB; ++u) {
                a[u] = ps[u] & 31u;
                full[u] = (a[u] + p[u] + np[u]) >> 5;  // complete 32-entry groups
                maxfull = full[u] > maxfull ? full[u] : maxfull;
            }
            for (uint32_t k = 0; __any(k < maxfull); ++k) {
                uint32_t v[DB], tot[DB];
                int idx[DB];
#pragma unroll
                for (int u = 0; u < DB; ++u) {
                    const uint32_t d = hw + (uint32_t)(it + u) * (NT / 32);
                    idx[u] = (k < full[u]) ? (int)(32u * k + l5) - (int)a[u] : -1;  // < 0: before the workgroup's first entry
                    v[u] = 0;
                    if (idx[u] >= 0) v[u] = ((uint32_t)idx[u] < p[u]) ? carry[d][idx[u]] : sent[ls[u] + (uint32_t)idx[u] - p[u]];
                }
#pragma unroll
#ifdef KK_FS_NOGATHER  // diagnostic builds (wrong results): what the gather of the column totals costs
                for (int u = 0; u < DB; ++u) tot[u] = (v[u] & colmask) | 1u;
#else
                for (int u = 0; u < DB; ++u) tot[u] = (idx[u] >= 0) ? rec_total[v[u] & colmask] : 1u;
#endif
#pragma unroll
                for (int u = 0; u < DB; ++u)
                    if (idx[u] >= 0) emit(ps[u] + (uint32_t)idx[u], v[u], tot[u]);
            }
#pragma unroll
            for (int u = 0; u < DB; ++u) {  // what stays for the next tile
                const uint32_t d = hw + (uint32_t)(it + u) * (NT / 32);
                const uint32_t rs = full[u] ? 32u * full[u] - a[u] : 0u, rem = p[u] + np[u] - rs;
                // (with a complete group, everything that stays is new: the waiting entries never reach past their group)
                if (l5 < rem && (full[u] || l5 >= p[u])) carry[d][l5] = sent[ls[u] + rs + l5 - p[u]];
                if (l5 == 0) {
                    cpend[d] = rem;
                    cpos[d] = ps[u] + rs;
                }
            }
        }
        lds_sync();
    }
    // the last, incomplete group of every digit
#pragma unroll 1
    for (int it = 0; it < RADIX / (NT / 32); ++it) {
        const uint32_t d = hw + (uint32_t)it * (NT / 32);
        const uint32_t p = cpend[d], ps = cpos[d];
        if (l5 < p) {
            const uint32_t v = carry[d][l5];
            emit(ps + l5, v, rec_total[v & colmask]);
        }
    }
}

// Entries whose count did not fit the count field: find the (row, column) in the finished CSR and write the value.
__global__ void k_fix_escapes(const DDEscape *esc, const uint32_t *scal, uint32_t esc_cap, const uint64_t *row_ind,
                              const uint64_t *col_ind, const uint32_t *rec_total, uint64_t first_col, uint64_t max_nz, float *data) {
    uint32_t n = scal[7];
    if (n > esc_cap) n = esc_cap;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DDEscape x = esc[i];
    uint64_t lo = row_ind[x.key], hi = row_ind[x.key + 1];
    if (hi > max_nz) hi = max_nz;
    const uint64_t want = first_col + x.col;
    while (lo < hi) {  // first entry of the row with column >= want
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (col_ind[mid] < want)
            lo = mid + 1;
        else
            hi = mid;
    }
    if (lo < max_nz && col_ind[lo] == want) data[lo] = (float)x.count / (float)rec_total[x.col];
}

// ------------------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------------------

static std::mutex g_knob_mu;
static Knobs g_knobs;
static bool g_knobs_read = false;
static void read_knobs_locked() {
    Knobs k;
    auto num = [](const char *name) -> unsigned long long {
        const char *e = getenv(name);
        return e ? strtoull(e, nullptr, 10) : 0ull;
    };
    k.y_atomics = getenv("KMERS_Y_ATOMICS") != nullptr;
    if (getenv("KMERS_Y_CAP")) k.y_cap = num("KMERS_Y_CAP") ? num("KMERS_Y_CAP") : 1;
    k.force_wide = getenv("KMERS_FORCE_WIDE") != nullptr;
    if (const char *e = getenv("KMERS_X_PATH")) k.x_path = !strcmp(e, "tables") ? 1 : (!strcmp(e, "rle") ? 2 : 0);
    k.old_pass0 = getenv("KMERS_X_OLD_PASS0") != nullptr;
    k.x_chunk_pos = num("KMERS_X_CHUNK_POS");
    k.verbose = (int)num("KMERS_VERBOSE");
    k.wide14 = getenv("KMERS_X_WIDE14") != nullptr;
    g_knobs = k;
    g_knobs_read = true;
}
const Knobs &knobs() {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    if (!g_knobs_read) read_knobs_locked();
    return g_knobs;
}
int set_knob(const char *name, long long value) {
    if (!name) return -12;
    std::lock_guard<std::mutex> lk(g_knob_mu);
    if (!g_knobs_read) read_knobs_locked();
    const std::string n(name);
    if (n == "reset") read_knobs_locked();
    else if (n == "y_atomics") g_knobs.y_atomics = value != 0;
    else if (n == "y_cap") g_knobs.y_cap = value > 0 ? (unsigned long long)value : 0;
    else if (n == "force_wide") g_knobs.force_wide = value != 0;
    else if (n == "x_path") g_knobs.x_path = (int)value;
    else if (n == "old_pass0") g_knobs.old_pass0 = value != 0;
    else if (n == "x_chunk_pos") g_knobs.x_chunk_pos = value > 0 ? (unsigned long long)value : 0;
    else if (n == "verbose") g_knobs.verbose = (int)value;
    else if (n == "wide14") g_knobs.wide14 = value != 0;
    else return -12;
    return 0;
}

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

bool profile_on() { return g_prof_on; }
void profile_add(const char *name, double ms) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto &p : g_prof)
        if (p.name == name) {
            p.ms += ms;
            p.calls += 1;
            return;
        }
    g_prof.push_back(ProfEntry{name, ms, 1});
}

struct PhaseTimer {
    hipStream_t s;
    bool on;
    std::vector<hipEvent_t> ev;
    std::vector<const char *> label;
    explicit PhaseTimer(hipStream_t s_, bool allowed = true) : s(s_), on(g_prof_on && allowed) { mark(nullptr); }
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

#ifdef KK_STAMPS
int read_stamps(unsigned long long *out16, bool reset) {
    if (const char *e = getenv("KK_ABLATE")) {
        int v = atoi(e);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &v, sizeof v);
    }
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), 16 * sizeof(u64)) != hipSuccess) return -11;
    if (reset) {
        u64 z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z) != hipSuccess) return -11;
    }
    return 0;
}
#else
int read_stamps(unsigned long long *, bool) { return -12; }
#endif

static inline uint32_t cscan_chunks(uint32_t n_tiles) { return (n_tiles + CSCAN_TILES - 1) / CSCAN_TILES; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Digits of the 8-byte-entry path (K >= 9, or K <= 8 with more than 2^24 records), least significant first.  Pass 0
// (k_xscatter0) takes 8, 9 or 10 bits, a later pass (k_scatter_wide) up to 9.
struct WidePlan {
    int n;
    int bits[4];
};
static WidePlan wide_plan(int K, bool classic) {
    const int total = 2 * K;
    WidePlan p{(total + 7) / 8, {8, 8, 8, 8}};
    if (classic) return p;  // rounds 1-2: ceil(2K / 8) passes of 8 bits
    if (K == 9) p = WidePlan{2, {9, 9, 0, 0}};     // config 2's transcriptome: 8.94 -> 7.81 ms
    if (K == 13) p = WidePlan{3, {9, 9, 8, 0}};    // 11.77 -> 10.61 ms
    // K = 14 as 10 + 9 + 9 was built and measured (knob wide14 = 1 selects it; bit-exact): 11.87 ms against 11.98 for
    // four 8-bit passes -- the two 9-bit passes save 1.1 ms, the 10-bit pass 0 (64-byte runs out of the tile, compact
    // staging) gives 0.94 of it back and the 1024-bin scan 0.16: not worth a second code path by default.
    if (K == 14 && knobs().wide14) p = WidePlan{3, {10, 9, 9, 0}};
    return p;
}
static int wide_max_bits(int K) { return K == 14 ? 10 : ((K == 9 || K == 13) ? 9 : 8); }  // (whatever the knobs say)

static bool x_tables_fit(uint64_t n_rec, int K) { return K >= 5 && K <= 8 && n_rec <= DD_MAX_REC; }

struct XLayout {
    uint32_t n_tiles;
    uint32_t dd_rows;        // histogram rows of the small-K path: its tiles + one per workgroup + the all-zero last row
    uint64_t hist_len;       // RADIX * (rows of the larger of the two paths)
    uint32_t scan_blocks;
    uint32_t task_cap;
    uint32_t esc_cap;
    size_t off_entA, off_entB, off_hist, off_blocksums, off_total, off_flags, off_tilecb, off_status, off_scalars, off_regions,
        off_tasks, off_wgrec, off_esc, off_rowcount, bytes;
};

static XLayout x_layout(uint64_t n_pos, uint64_t n_rec, int K) {
    XLayout l;
    const uint64_t emax = n_pos ? n_pos : 1;
    l.n_tiles = (uint32_t)((emax + TILE - 1) / TILE);
    // the small-K table path's buffers exist only where build_x can select it (K and n_rec decide; never a knob: the
    // layout a caller sized its workspace by must not move under it)
    const bool tables_fit = x_tables_fit(n_rec, K);
    l.dd_rows = tables_fit ? (uint32_t)((emax + DD_C - 1) / DD_C) + DD_GRID + 2 : 0;
    // rows of the tile histogram: the partition's tiles (+ 1), the fused emit's region-aligned tiles (at most 256 more, + the
    // all-zero rows behind them), the small-K table path's
    uint32_t rows = l.n_tiles + RADIX + 2;
    if (l.dd_rows > rows) rows = l.dd_rows;
    const uint64_t max_bins = 1ull << wide_max_bits(K);
    l.hist_len = max_bins * rows;
    l.scan_blocks = cscan_chunks(rows);
    l.esc_cap = tables_fit ? (uint32_t)(emax < (1u << 22) ? emax : (1u << 22)) : 0;
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
    l.off_entA = take(emax * 8);
    l.off_entB = take(emax * 8);
    l.off_hist = take(l.hist_len * 4);
    l.off_blocksums = take((size_t)l.scan_blocks * max_bins * 4);
    l.off_total = take((n_rec ? n_rec : 1) * 4);
    l.off_flags = take(((emax + 31) / 32 + 4) * 4);
    l.off_tilecb = take(((size_t)l.n_tiles + 2) * 4);
    l.off_status = take(((size_t)l.n_tiles + 1) * 8);
    l.off_scalars = take(64);
    l.off_regions = take(3 * (RADIX + 1) * 4);  // regions of two passes, tile starts of the fused emit
    l.off_tasks = take((size_t)l.task_cap * sizeof(FillTask));
    l.off_wgrec = take(tables_fit ? (DD_GRID + 1) * 4 : 0);
    l.off_esc = take((size_t)l.esc_cap * sizeof(DDEscape));
    l.off_rowcount = take(tables_fit ? (size_t)65536 * 4 : 0);
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

// hist[n_tiles][RADIX] -> bases in place; *p_total = number of entries.  chunk_sums: cscan_chunks(n_tiles) * RADIX words.
template <int BINS>
static int scan_tilehist_bins(uint32_t *hist, uint32_t n_tiles, uint32_t *chunk_sums, uint32_t *p_total, hipStream_t s,
                              uint32_t *regions, const uint32_t *p_count) {
    const uint32_t nc = cscan_chunks(n_tiles);
    hipLaunchKernelGGL(k_cscan_reduce<BINS>, dim3(nc), dim3(BINS), 0, s, (const uint32_t *)hist, n_tiles, chunk_sums);
    hipLaunchKernelGGL(k_cscan_mid<BINS>, dim3(1), dim3(1024), 0, s, chunk_sums, nc, p_total);
    hipLaunchKernelGGL(k_cscan_apply<BINS>, dim3(nc), dim3(BINS), 0, s, hist, n_tiles, (const uint32_t *)chunk_sums, regions, p_count);
    KK_CHECK(hipGetLastError());
    return 0;
}
// hist rows of 2^bits counts (8, 9 or 10 bits); regions (256 bins only) as before
static int scan_tilehist(uint32_t *hist, uint32_t n_tiles, uint32_t *chunk_sums, uint32_t *p_total, hipStream_t s,
                         uint32_t *regions = nullptr, const uint32_t *p_count = nullptr, int bits = 8) {
    switch (bits) {
        case 8: return scan_tilehist_bins<256>(hist, n_tiles, chunk_sums, p_total, s, regions, p_count);
        case 9: return scan_tilehist_bins<512>(hist, n_tiles, chunk_sums, p_total, s, nullptr, nullptr);
        case 10: return scan_tilehist_bins<1024>(hist, n_tiles, chunk_sums, p_total, s, nullptr, nullptr);
    }
    return -12;
}

// ---- y workspace (K >= 9): entries of one chunk of tiles, the tile histogram and its scan, regions, scalars
constexpr uint64_t Y_CHUNK_MAX_TILES = (1ull << 31) / TILE;  // entry indices stay far below 2^32

struct YLayout {
    uint32_t chunk_tiles;
    uint64_t hist_len;
    uint32_t scan_blocks;
    int n_buf;
    size_t off_entA, off_entB, off_hist, off_blocksums, off_scalars, off_regions, off_ovf, off_bounds, bytes;
    int slice_log2;     // bins per workgroup of the final histogram kernel
    uint32_t slices;    // 4^K >> slice_log2: also the capacity of the overflow queue
};

static YLayout y_layout_for(uint32_t chunk_tiles, int K) {
    YLayout l;
    l.chunk_tiles = chunk_tiles;
    l.hist_len = (uint64_t)RADIX * chunk_tiles;
    l.scan_blocks = cscan_chunks(chunk_tiles);
    l.n_buf = (K >= 13) ? 2 : 1;
    const int low_bits = 2 * K - (K >= 13 ? 16 : 8);  // 10..16
    // one pass: a slice is at most one region (low_bits 10, 12, 14 -> the whole region, 16 -> half of it);
    // two passes: 2^15 bins (2^(15 - low_bits) buckets, or half a bucket at K = 16)
    l.slice_log2 = (K >= 13) ? 15 : (low_bits < 15 ? low_bits : 15);
    l.slices = (uint32_t)((1ull << (2 * K)) >> l.slice_log2);
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o = align_up(o + bytes, 256);
        return at;
    };
    const size_t ent = (size_t)chunk_tiles * TILE * 4;
    l.off_entA = take(ent);
    l.off_entB = take(l.n_buf == 2 ? ent : 0);
    l.off_hist = take(l.hist_len * 4);
    l.off_blocksums = take((size_t)l.scan_blocks * RADIX * 4);
    l.off_scalars = take(64);
    l.off_regions = take(2 * (RADIX + 1) * 4);
    l.off_ovf = take((size_t)l.slices * sizeof(YOverflow));
    l.off_bounds = take(l.n_buf == 2 ? (size_t)65537 * 4 : 0);
    l.bytes = o;
    return l;
}

// Chunks of equal size, the fewest that respect Y_CHUNK_MAX_TILES.
static uint32_t y_default_chunk_tiles(uint64_t n_pos) {
    const uint64_t n_tiles = (n_pos + TILE - 1) / TILE;
    if (n_tiles == 0) return 1;
    const uint64_t n_chunks = (n_tiles + Y_CHUNK_MAX_TILES - 1) / Y_CHUNK_MAX_TILES;
    return (uint32_t)((n_tiles + n_chunks - 1) / n_chunks);
}

size_t y_workspace_bytes(uint64_t n_pos, int K) {
    if (K <= 8 || n_pos == 0) return 0;
    return y_layout_for(y_default_chunk_tiles(n_pos), K).bytes;
}

// ---- the library's own scratch, for callers that cannot allocate (kmers_hip_count_y_static, include/kmers_hip.h):
// a file-scope device array that every GPU the library runs on holds until the process ends.  In host code the
// array's NAME is only a stand-in (passing it to a kernel compiles and then faults on the GPU): the device address
// comes from hipGetSymbolAddress, once per device with that device current, and only that pointer is used below.
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
static std::mutex g_scratch_mu;
static void *g_scratch_ptr[64];  // per device ordinal

int static_scratch(void **ptr, size_t *bytes) {
    int dev = -1;
    KK_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return -12;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        KK_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return -11;
        g_scratch_ptr[dev] = p;
    }
    *ptr = g_scratch_ptr[dev];
    *bytes = STATIC_SCRATCH_BYTES;
    return 0;
}

template <int BPB_LOG2, class ET>
static void launch_y_sorted(uint32_t blocks, hipStream_t s, const ET *ent, const uint32_t *regions, int two_pass, int low_bits,
                            int32_t *counts, uint32_t cap, uint32_t *ovf_count, YOverflow *ovf, uint32_t ovf_cap) {
    hipLaunchKernelGGL((k_count_y_sorted<BPB_LOG2, ET>), dim3(blocks), dim3(YS_NT), 0, s, ent, regions, two_pass, low_bits, counts,
                       cap, ovf_count, ovf, ovf_cap);
    hipLaunchKernelGGL((k_count_y_overflow<BPB_LOG2, ET>), dim3(1024), dim3(YS_NT), 0, s, ent, two_pass, low_bits, counts,
                       (const uint32_t *)ovf_count, (const YOverflow *)ovf, ovf_cap);
}

int count_y(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words, uint64_t n_pos, int K,
            int32_t *d_counts, void *d_ws, size_t ws_bytes, hipStream_t s, bool may_profile) {
    if (n_pos == 0) return 0;
    BasesArgs a{d_bases, d_valid, nullptr, nullptr, bases_words, valid_words, n_pos, K, 0, 0};
    const uint64_t n_tiles = (n_pos + TILE - 1) / TILE;
    PhaseTimer pt(s, may_profile);  // (the per-kernel timers create HIP events: never on the no-allocation entry point)
    if (K <= 8) {
        const uint32_t n_bins = 1u << (2 * K);
        const uint64_t n_words = (n_pos + 15) / 16;
        const uint64_t wg = (n_words + YL_NT - 1) / YL_NT;
        const uint32_t grid = (uint32_t)(wg < 256 ? wg : 256);  // 128 KiB of LDS: one workgroup per CU
        hipLaunchKernelGGL(k_count_y_lds, dim3(grid), dim3(YL_NT), 0, s, a, d_counts, n_words, n_bins);
        pt.mark("k_count_y_lds");
        KK_CHECK(hipGetLastError());
        if (pt.on) {
            KK_CHECK(hipStreamSynchronize(s));
            pt.finish();
        }
        return 0;
    }
    const Knobs kn = knobs();
    if (kn.y_atomics) {  // knob: A-B runs against the plain global-atomics kernel
        const uint32_t grid = (uint32_t)(n_tiles < 2048 ? n_tiles : 2048);
        hipLaunchKernelGGL(k_count_y_global, dim3(grid), dim3(NT), 0, s, a, d_counts, n_tiles);
        pt.mark("k_count_y_global");
        KK_CHECK(hipGetLastError());
        if (pt.on) {
            KK_CHECK(hipStreamSynchronize(s));
            pt.finish();
        }
        return 0;
    }
    // ---- K >= 9: partition by the top 8 / 16 bits of the id, then per-slice LDS histograms
    const bool two_pass = K >= 13;
    const int low_bits = 2 * K - (two_pass ? 16 : 8);          // 10..16
    uint32_t chunk_tiles = y_default_chunk_tiles(n_pos);
    if (!d_ws) return -14;
    if (ws_bytes < y_layout_for(chunk_tiles, K).bytes) {
        // a smaller workspace is fine: more, smaller chunks (bytes are monotone in the chunk size)
        uint32_t lo = 0, hi = chunk_tiles;  // largest chunk that fits
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo + 1) / 2;
            if (y_layout_for(mid, K).bytes <= ws_bytes)
                lo = mid;
            else
                hi = mid - 1;
        }
        if (lo == 0) return -14;
        chunk_tiles = lo;
    }
    const YLayout l = y_layout_for(chunk_tiles, K);
    char *ws = (char *)d_ws;
    uint32_t *entA = (uint32_t *)(ws + l.off_entA), *entB = (uint32_t *)(ws + l.off_entB);
    uint32_t *hist = (uint32_t *)(ws + l.off_hist), *bsum = (uint32_t *)(ws + l.off_blocksums);
    uint32_t *scal = (uint32_t *)(ws + l.off_scalars);  // [0] = E  [4] = dummy total
    uint32_t *regs0 = (uint32_t *)(ws + l.off_regions), *regs1 = regs0 + (RADIX + 1);
    YOverflow *ovf = (YOverflow *)(ws + l.off_ovf);
    const dim3 block(NT);
    const uint32_t slices = l.slices;
    // entry ranges the slices read: the 256 regions, or the slices' own bucket groups (at K = 16: the 65536 buckets)
    const uint64_t n_ranges = two_pass ? (slices < 65536u ? slices : 65536u) : 256u;
    for (uint64_t t0 = 0; t0 < n_tiles; t0 += chunk_tiles) {
        const uint32_t nt = (uint32_t)((n_tiles - t0 < chunk_tiles) ? n_tiles - t0 : chunk_tiles);
        a.tile0 = t0;
        a.low_bits = low_bits;
        KK_CHECK(hipMemsetAsync(scal, 0, 64, s));  // [8] = length of the overflow queue
        // a slice's workgroup counts at most `cap` entries itself: a little over the mean range (the longest workgroup sets
        // the kernel's time when there is one wave of them), at least two overflow chunks
        uint64_t cap64 = (uint64_t)nt * TILE / n_ranges;
        if (K <= 11)
            cap64 /= 2;  // one workgroup per region and CU: half the mean, the rest balanced over the overflow grid (y10: 1.27 -> 0.89 ms per 2^31 positions)
        else
            cap64 += cap64 / 4;  // 1.25 x the mean range (of positions: >= entries): at K >= 12 the queue's atomic flush costs more than it balances
        if (cap64 < 2 * (uint64_t)YS_OVF_CHUNK) cap64 = 2 * (uint64_t)YS_OVF_CHUNK;
        if (kn.y_cap) cap64 = kn.y_cap;  // knob: tests
        const uint32_t cap = (uint32_t)(cap64 > 0xFFFFFFFFull ? 0xFFFFFFFFull : cap64);
        const dim3 grid(nt);
        const dim3 pgrid(nt < (uint32_t)PERSIST_BLOCKS ? nt : (uint32_t)PERSIST_BLOCKS);
        const dim3 hgrid(nt < (uint32_t)KK_HIST_BLOCKS ? nt : (uint32_t)KK_HIST_BLOCKS);
        const uint32_t *c_hist = hist, *c_E = &scal[0];
        // pass 0 tiles: 8192 positions (512 threads) for the 2-byte payloads, 16,384 (1024 threads) for 4-byte entries
        const uint64_t n_words8 = (uint64_t)nt * NT;
        const uint32_t nt0 = (nt + 1) / 2;
        hipLaunchKernelGGL(k_hist_words_y<1024>, dim3(nt0 < (uint32_t)HW_BLOCKS ? nt0 : (uint32_t)HW_BLOCKS), dim3(1024), 0, s, a, hist, n_words8, nt0);
        pt.mark("k_hist_words_y");
        if (int rc = scan_tilehist(hist, nt0, bsum, &scal[0], s, regs0, c_E)) return rc;
        pt.mark("k_scan");
        if (two_pass)
            hipLaunchKernelGGL((k_yscatter<uint32_t, 1024, 1>), dim3(nt0 < (uint32_t)YS32_BLOCKS ? nt0 : (uint32_t)YS32_BLOCKS), dim3(1024), 0, s, a, c_hist, (const uint32_t *)regs0, n_words8, nt0, entA);
        else  // one pass: 2-byte payloads
            hipLaunchKernelGGL((k_yscatter<uint16_t, YS16_NT, 1024 / YS16_NT>), dim3(nt0 < (uint32_t)YS16_BLOCKS ? nt0 : (uint32_t)YS16_BLOCKS), dim3(YS16_NT), 0, s, a, c_hist, (const uint32_t *)regs0, n_words8, nt0, (uint16_t *)entA);
        pt.mark("k_scatter<bases,y>");
        const uint32_t *sorted = entA, *final_regions = regs0;
        if (two_pass) {
            hipLaunchKernelGGL(k_hist_entries<uint32_t>, hgrid, block, 0, s, (const uint32_t *)entA, c_E, 8, hist, nt);
            pt.mark("k_hist_entries");
            if (int rc = scan_tilehist(hist, nt, bsum, &scal[4], s, regs1, c_E)) return rc;
            pt.mark("k_scan");
            // this pass keeps the order of the byte pass 0 sorted on (k_count_y_sorted searches a region by it): mode 2
            // (round 3 built this pass like k_yscatter as well -- 16,384- and 8192-entry tiles, 512 / 1024 threads, known
            // digit counts, one atomic per entry: 26.1 - 32.3 ms per step of y14 against 26.3 ms for this kernel; removed)
            hipLaunchKernelGGL((k_scatter<false, uint32_t, 2>), pgrid, block, 0, s, a, (const uint32_t *)entA, c_E, 8, c_hist, nt, entB,
                               (const uint32_t *)regs0);
            pt.mark("k_scatter<entries>");
            sorted = entB;
            uint32_t *bounds = (uint32_t *)(ws + l.off_bounds);
            hipLaunchKernelGGL(k_bucket_bounds, dim3(65537 / 256 + 1), dim3(256), 0, s, sorted, (const uint32_t *)regs1, bounds);
            final_regions = bounds;
        }
        const uint16_t *sorted16 = (const uint16_t *)entA;
        switch (l.slice_log2) {  // one pass: 10, 12, 14 or 15 (2-byte entries); two passes: 15 (4-byte entries)
            case 10: launch_y_sorted<10>(slices, s, sorted16, final_regions, 0, low_bits, d_counts, cap, &scal[8], ovf, slices); break;
            case 12: launch_y_sorted<12>(slices, s, sorted16, final_regions, 0, low_bits, d_counts, cap, &scal[8], ovf, slices); break;
            case 14: launch_y_sorted<14>(slices, s, sorted16, final_regions, 0, low_bits, d_counts, cap, &scal[8], ovf, slices); break;
            default:
                if (two_pass)
                    launch_y_sorted<15>(slices, s, sorted, final_regions, 1, low_bits, d_counts, cap, &scal[8], ovf, slices);
                else
                    launch_y_sorted<15>(slices, s, sorted16, final_regions, 0, low_bits, d_counts, cap, &scal[8], ovf, slices);
                break;
        }
        pt.mark("k_count_y_sorted");
    }
    KK_CHECK(hipGetLastError());
    if (pt.on) {  // per-kernel timing needs the events completed
        KK_CHECK(hipStreamSynchronize(s));
        pt.finish();
    }
    return 0;
}

// The small-K path (see "X for 5 <= K <= 8" above).  Returns 1 when the build has to be redone on the general path
// (a 16-bit count field wrapped, or more escapes than the queue holds), else 0 / an error code.
static int build_x_tables(const XLayout &l, const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words,
                          const uint64_t *d_rec_start, uint64_t n_pos, uint64_t n_rec, uint64_t first_col, int K, float *d_data,
                          uint64_t *d_col_ind, uint64_t *d_row_ind, uint64_t max_nz, char *ws, uint64_t *nnz_out, hipStream_t s,
                          PhaseTimer &pt) {
    const uint64_t M = 1ull << (2 * K);
    uint32_t *entA = (uint32_t *)(ws + l.off_entA);
    uint32_t *hist = (uint32_t *)(ws + l.off_hist), *bsum = (uint32_t *)(ws + l.off_blocksums);
    uint32_t *total = (uint32_t *)(ws + l.off_total);
    uint32_t *scal = (uint32_t *)(ws + l.off_scalars);  // [0]=nnz [3]=err [4]=dummy [6]=redo [7]=escapes
    uint32_t *regs0 = (uint32_t *)(ws + l.off_regions);
    uint32_t *wgrec = (uint32_t *)(ws + l.off_wgrec);
    DDEscape *esc = (DDEscape *)(ws + l.off_esc);
    uint32_t *rowcount = (uint32_t *)(ws + l.off_rowcount);
    uint32_t colbits = 1;
    while ((1ull << colbits) < n_rec) ++colbits;  // columns 0 .. n_rec - 1 (n_rec <= 2^22: at least 2 bits of count)
    DDArgs a{d_bases, d_valid, bases_words, valid_words, d_rec_start, (uint32_t)n_rec, K, wgrec};
    const uint32_t rows = l.dd_rows;  // the last one stays zero: row r's counts = scanned row r + 1 - scanned row r
    hipLaunchKernelGGL(k_dd_ranges, dim3((DD_GRID + 1 + 255) / 256), dim3(256), 0, s, d_rec_start, (uint32_t)n_rec, n_pos,
                       (uint32_t)DD_GRID, wgrec);
    KK_CHECK(hipMemsetAsync(hist, 0, (size_t)rows * RADIX * 4, s));
    pt.mark("setup");
    hipLaunchKernelGGL(k_dd<false>, dim3(DD_GRID), dim3(DD_NT), 0, s, a, hist, total, (uint32_t *)nullptr, colbits,
                       (DDEscape *)nullptr, 0u, scal);
    pt.mark("k_dd<hist>");
    if (int rc = scan_tilehist(hist, rows, bsum, &scal[0], s, regs0, (const uint32_t *)&scal[0])) return rc;
    pt.mark("k_scan");
    hipLaunchKernelGGL(k_dd<true>, dim3(DD_GRID), dim3(DD_NT), 0, s, a, hist, total, entA, colbits, esc, l.esc_cap, scal);
    pt.mark("k_dd<scatter>");
    const uint32_t nt1 = l.n_tiles + 1;  // one tile beyond the entries: an all-zero last row for k_row_counts
    hipLaunchKernelGGL(k_hist_entries<uint32_t>, dim3(nt1 < 2048u ? nt1 : 2048u), dim3(NT), 0, s, (const uint32_t *)entA,
                       (const uint32_t *)&scal[0], 8, hist, nt1);
    pt.mark("k_hist_entries");
    if (int rc = scan_tilehist(hist, nt1, bsum, &scal[4], s)) return rc;
    pt.mark("k_scan");
    hipLaunchKernelGGL(k_row_counts, dim3(RADIX), dim3(RADIX), 0, s, (const uint32_t *)entA, (const uint32_t *)regs0,
                       (const uint32_t *)hist, (uint32_t)(M >> 8), rowcount);
    hipLaunchKernelGGL(k_row_ptr, dim3(1), dim3(1024), 0, s, (const uint32_t *)rowcount, (uint32_t)M, d_row_ind);
    pt.mark("k_row_ptr");
    const uint32_t pg = l.n_tiles < (uint32_t)PERSIST_BLOCKS ? l.n_tiles : (uint32_t)PERSIST_BLOCKS;
    hipLaunchKernelGGL(k_final_scatter, dim3(pg), dim3(NT), 0, s, (const uint32_t *)entA, (const uint32_t *)&scal[0],
                       (const uint32_t *)hist, (const uint32_t *)total, colbits, first_col, max_nz, d_data, d_col_ind);
    pt.mark("k_final_scatter");
    hipLaunchKernelGGL(k_fix_escapes, dim3((l.esc_cap + 255) / 256), dim3(256), 0, s, (const DDEscape *)esc, (const uint32_t *)scal,
                       l.esc_cap, (const uint64_t *)d_row_ind, (const uint64_t *)d_col_ind, (const uint32_t *)total, first_col,
                       max_nz, d_data);
    pt.mark("k_fix_escapes");
    KK_CHECK(hipGetLastError());
    uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    KK_CHECK(hipMemcpyAsync(h, scal, sizeof h, hipMemcpyDeviceToHost, s));
    KK_CHECK(hipStreamSynchronize(s));
    pt.finish();
    if (h[3]) {
        fprintf(stderr, "libkmers: internal error %u in the small-K X build (3: the two table kernels disagree)\n", h[3]);
        return -11;
    }
    if (h[6]) return 1;  // redo on the general path
    *nnz_out = h[0];
    if ((uint64_t)h[0] > max_nz) return -2;
    return 0;
}

#ifndef XS0_NT_NARROW
#define XS0_NT_NARROW 512
#endif
#ifndef XS0_NT_WIDE
#define XS0_NT_WIDE 512
#endif
#ifndef XS0_BLOCKS
#define XS0_BLOCKS 4096
#endif
static dim3 xs0_grid(uint32_t n_tiles8, int bnt) {
    const uint32_t nbt = (n_tiles8 + (uint32_t)(bnt / NT) - 1) / (uint32_t)(bnt / NT);
    return dim3(nbt < (uint32_t)XS0_BLOCKS ? nbt : (uint32_t)XS0_BLOCKS);
}

// uint64 -> uint32, for the file-level calls' device-to-host copy: column numbers and row pointers of one device call fit
// 32 bits, and PCIe (not HBM) bounds that copy; the host widens them again while it moves them into the caller's arrays.
__global__ void k_narrow64(const uint64_t *in, uint32_t *out, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = (uint32_t)in[i];
}
int narrow64(const uint64_t *d_in, uint32_t *d_out, uint64_t n, hipStream_t s) {
    if (n == 0) return 0;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_narrow64, dim3((uint32_t)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, d_in, d_out, n);
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
    u64 *entA = (u64 *)(ws + l.off_entA), *entB = (u64 *)(ws + l.off_entB);
    uint32_t *hist = (uint32_t *)(ws + l.off_hist), *bsum = (uint32_t *)(ws + l.off_blocksums);
    uint32_t *total = (uint32_t *)(ws + l.off_total);
    uint32_t *flags = (uint32_t *)(ws + l.off_flags), *tilecb = (uint32_t *)(ws + l.off_tilecb);
    u64 *status = (u64 *)(ws + l.off_status);
    uint32_t *scal = (uint32_t *)(ws + l.off_scalars);  // [0]=E [1]=nnz [2]=ntasks [3]=err [4]=dummy [5]=ticket
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

    // 5 <= K <= 8: the per-isoform LDS tables are selected with KMERS_X_PATH=tables.  Measured on config 2 they are not
    // yet the faster of the two paths (6.0 ms against 5.6 ms: DESIGN.md section 4), so the partition path stays the default.
    const Knobs kn = knobs();
    {
        const bool want_tables = kn.x_path == 1 && !kn.force_wide;
        const bool fits = x_tables_fit(n_rec, K);
        if (fits && want_tables) {
            const int rc = build_x_tables(l, d_bases, bases_words, d_valid, valid_words, d_rec_start, n_pos, n_rec, first_col, K,
                                          d_data, d_col_ind, d_row_ind, max_nz, ws, nnz_out, s, pt);
            if (rc != 1) return rc;
            pt.on = false;  // (its timeline was closed by the attempt above)
            if (kn.verbose) fprintf(stderr, "libkmers: a k-mer count left the table's range: X is rebuilt on the general path\n");
            KK_CHECK(hipMemsetAsync(scal, 0, 64, s));
            KK_CHECK(hipMemsetAsync(total, 0, n_rec * 4, s));
        }
    }
    const size_t flag_words = (size_t)((n_pos + 31) / 32 + 4);
    // flags, tile_cb and status lie one after the other in the layout: one memset (tile_cb is written right after)
    KK_CHECK(hipMemsetAsync(flags, 0, (l.off_status - l.off_flags) + ((size_t)l.n_tiles + 1) * 8, s));
    hipLaunchKernelGGL(k_record_tables, dim3((uint32_t)((n_rec + l.n_tiles + 1 + 255) / 256)), dim3(256), 0, s, d_rec_start, n_rec, flags,
                       l.n_tiles, tilecb);
    pt.mark("setup");

    // rec_flags is indexed like valid[] (one bit per position), so it shares valid's word bound
    const uint64_t vw = valid_words < flag_words ? valid_words : flag_words;
    BasesArgs a{d_bases, d_valid, flags, tilecb, bases_words, vw, n_pos, K};
    const dim3 grid(l.n_tiles), block(NT);
    const dim3 pgrid(l.n_tiles < (uint32_t)PERSIST_BLOCKS ? l.n_tiles : (uint32_t)PERSIST_BLOCKS);
    const dim3 hgrid(l.n_tiles < (uint32_t)KK_HIST_BLOCKS ? l.n_tiles : (uint32_t)KK_HIST_BLOCKS);
    const int n_pass = (2 * K + 7) / 8;   // (4-byte entries: one or two passes of 8 bits)
    const bool old_pass0 = kn.old_pass0;  // knob: A-B runs against the position-ordered pass 0

    // K <= 8 and at most 2^24 records: 4-byte entries (digit byte << 24 | column), else 8-byte (k-mer id << 32 | column)
    const bool narrow = (K <= 8) && (n_rec <= (1ull << 24)) && !kn.force_wide;  // knob: tests / A-B runs
    uint32_t *regs0 = (uint32_t *)(ws + l.off_regions), *regs1 = regs0 + (RADIX + 1);
    const uint32_t *c_hist = hist, *c_E = &scal[0];

    // pass 0: packed bases -> instances partitioned by the lowest digit
    const WidePlan plan = wide_plan(K, old_pass0 || kn.x_path == 2);   // (8-byte entries only)
    const int bits0 = narrow ? 8 : plan.bits[0];
    const dim3 hb_grid(l.n_tiles < 4u * HB_BLOCKS ? (l.n_tiles + HB_NW - 1) / HB_NW : (uint32_t)HB_BLOCKS);
    if (bits0 == 8)
        hipLaunchKernelGGL(k_hist_bases<8>, hb_grid, dim3(HB_NT), 0, s, a, hist, l.n_tiles, total);
    else if (bits0 == 9)
        hipLaunchKernelGGL(k_hist_bases<9>, hb_grid, dim3(HB_NT), 0, s, a, hist, l.n_tiles, total);
    else
        hipLaunchKernelGGL(k_hist_bases<10>, hb_grid, dim3(HB_NT), 0, s, a, hist, l.n_tiles, total);
    pt.mark("k_hist_bases");
    if (int rc = scan_tilehist(hist, l.n_tiles, bsum, &scal[0], s, bits0 == 8 ? regs0 : nullptr, c_E, bits0)) return rc;
    pt.mark("k_scan");
    const uint32_t *final_regions = regs0;
    if (narrow) {
        uint32_t *ei = (uint32_t *)entA, *eo = (uint32_t *)entB;
        if (old_pass0)
            hipLaunchKernelGGL((k_scatter<true, uint32_t>), pgrid, block, 0, s, a, (const uint32_t *)nullptr, c_E, 0, c_hist, l.n_tiles,
                               ei, (const uint32_t *)nullptr);
        else
            hipLaunchKernelGGL((k_xscatter0<uint32_t, XS0_NT_NARROW, 8>), xs0_grid(l.n_tiles, XS0_NT_NARROW), dim3(XS0_NT_NARROW), 0, s, a,
                               c_hist, l.n_tiles, ei);
        pt.mark("k_scatter<bases>");
        if (kn.x_path != 2) {  // knob: A-B runs against k_scatter<entries> + k_rle
            // the last pass writes the CSR arrays itself (see "fused emit" above)
            uint32_t *tstart = regs0 + 2 * (RADIX + 1);
            const uint32_t rows = l.n_tiles + RADIX + 2;
            hipLaunchKernelGGL(k_ftiles, dim3(1), dim3(RADIX), 0, s, (const uint32_t *)regs0, tstart);
            hipLaunchKernelGGL(k_heads_hist, dim3(rows < (uint32_t)KK_HIST_BLOCKS ? rows : (uint32_t)KK_HIST_BLOCKS), block, 0, s,
                               (const uint32_t *)ei, (const uint32_t *)regs0, (const uint32_t *)tstart, hist, rows, scal);
            pt.mark("k_heads_hist");
            if (int rc = scan_tilehist(hist, rows, bsum, &scal[7], s)) return rc;
            pt.mark("k_scan");
            hipLaunchKernelGGL(k_row_ptr_fused, dim3((uint32_t)((M + 1 + 255) / 256)), dim3(256), 0, s, c_hist, (const uint32_t *)tstart,
                               (const uint32_t *)&scal[7], M, d_row_ind, scal);
            const uint32_t et = l.n_tiles + RADIX;
            hipLaunchKernelGGL(k_emit, dim3(et < (uint32_t)PERSIST_BLOCKS ? et : (uint32_t)PERSIST_BLOCKS), block, 0, s,
                               (const uint32_t *)ei, (const uint32_t *)regs0, (const uint32_t *)tstart, c_hist, (const uint32_t *)total,
                               first_col, max_nz, d_data, d_col_ind, scal);
            pt.mark("k_emit");
            KK_CHECK(hipGetLastError());
            uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            KK_CHECK(hipMemcpyAsync(h, scal, sizeof h, hipMemcpyDeviceToHost, s));
            KK_CHECK(hipStreamSynchronize(s));
            if (!h[6]) {
                pt.finish();
                *nnz_out = h[1];
                if ((uint64_t)h[1] > max_nz) return -2;
                return 0;
            }
            // a column group longer than the look-back / look-ahead allows (one record with millions of bases): the
            // general path below redoes everything after pass 0
            if (kn.verbose) fprintf(stderr, "libkmers: a record too long for the fused emit: X is finished on the general path\n");
        }
        if (n_pass > 1) {
            hipLaunchKernelGGL(k_hist_entries<uint32_t>, hgrid, block, 0, s, (const uint32_t *)ei, c_E, 8, hist, l.n_tiles);
            pt.mark("k_hist_entries");
            if (int rc = scan_tilehist(hist, l.n_tiles, bsum, &scal[4], s, regs1, c_E)) return rc;
            pt.mark("k_scan");
            hipLaunchKernelGGL((k_scatter<false, uint32_t>), pgrid, block, 0, s, a, (const uint32_t *)ei, c_E, 8, c_hist, l.n_tiles, eo,
                               (const uint32_t *)regs0);
            pt.mark("k_scatter<entries>");
            uint32_t *t = ei;
            ei = eo, eo = t;
            final_regions = regs1;
        }
        hipLaunchKernelGGL((k_rle<uint32_t, RLE_IPT_NARROW, RLE_NT>), dim3((uint32_t)(((uint64_t)l.n_tiles * TILE + RLE_NT * RLE_IPT_NARROW - 1) / (RLE_NT * RLE_IPT_NARROW))), dim3(RLE_NT), 0, s, (const uint32_t *)ei, scal, status, (const uint32_t *)total, first_col, M,
                           max_nz, d_data, d_col_ind, d_row_ind, tasks, l.task_cap, final_regions, n_pass > 1 ? 8 : 0);
    } else {
        u64 *ei = entA, *eo = entB;
        if (old_pass0)
            hipLaunchKernelGGL((k_scatter<true, u64>), pgrid, block, 0, s, a, (const u64 *)nullptr, c_E, 0, c_hist, l.n_tiles, ei,
                               (const uint32_t *)nullptr);
        else if (bits0 == 8)
            hipLaunchKernelGGL((k_xscatter0<u64, XS0_NT_WIDE, 8>), xs0_grid(l.n_tiles, XS0_NT_WIDE), dim3(XS0_NT_WIDE), 0, s, a, c_hist,
                               l.n_tiles, ei);
        else if (bits0 == 9)
            hipLaunchKernelGGL((k_xscatter0<u64, XS0_NT_WIDE, 9>), xs0_grid(l.n_tiles, XS0_NT_WIDE), dim3(XS0_NT_WIDE), 0, s, a, c_hist,
                               l.n_tiles, ei);
        else
            hipLaunchKernelGGL((k_xscatter0<u64, 512, 10>), xs0_grid(l.n_tiles, 512), dim3(512), 0, s, a, c_hist, l.n_tiles, ei);
        pt.mark("k_scatter<bases>");
        int shift = bits0;
        for (int p = 1; p < plan.n; ++p) {
            const int bits = plan.bits[p];
            if (bits == 9)
                hipLaunchKernelGGL((k_hist_entries<u64, 9>), grid, block, 0, s, (const u64 *)ei, c_E, shift, hist, l.n_tiles);
            else
                hipLaunchKernelGGL((k_hist_entries<u64, 8>), grid, block, 0, s, (const u64 *)ei, c_E, shift, hist, l.n_tiles);
            pt.mark("k_hist_entries");
            if (int rc = scan_tilehist(hist, l.n_tiles, bsum, &scal[4], s, nullptr, nullptr, bits)) return rc;
            pt.mark("k_scan");
            if (old_pass0 || kn.x_path == 2)   // (rounds 1-2: 8-bit digits only)
                hipLaunchKernelGGL((k_scatter<false, u64>), pgrid, block, 0, s, a, (const u64 *)ei, c_E, shift, c_hist, l.n_tiles, eo,
                                   (const uint32_t *)nullptr);
            else if (bits == 9)
                hipLaunchKernelGGL(k_scatter_wide<9>, pgrid, block, 0, s, (const u64 *)ei, c_E, shift, bits, c_hist, eo);
            else
                hipLaunchKernelGGL(k_scatter_wide<8>, pgrid, block, 0, s, (const u64 *)ei, c_E, shift, bits, c_hist, eo);
            pt.mark("k_scatter<entries>");
            u64 *t = ei;
            ei = eo, eo = t;
            shift += bits;
        }
        hipLaunchKernelGGL((k_rle<u64, RLE_IPT_WIDE, RLE_NT>), dim3((uint32_t)(((uint64_t)l.n_tiles * TILE + RLE_NT * RLE_IPT_WIDE - 1) / (RLE_NT * RLE_IPT_WIDE))), dim3(RLE_NT), 0, s, (const u64 *)ei, scal, status, (const uint32_t *)total, first_col, M, max_nz,
                           d_data, d_col_ind, d_row_ind, tasks, l.task_cap, (const uint32_t *)nullptr, 0);
    }
    pt.mark("k_rle");
    hipLaunchKernelGGL(k_fill_tasks, dim3(2048), block, 0, s, d_row_ind, (const FillTask *)tasks, (const uint32_t *)&scal[2],
                       l.task_cap);
    pt.mark("k_fill_tasks");
    KK_CHECK(hipGetLastError());
    uint32_t h[4] = {0, 0, 0, 0};
    KK_CHECK(hipMemcpyAsync(h, scal, sizeof h, hipMemcpyDeviceToHost, s));
    KK_CHECK(hipStreamSynchronize(s));
    pt.finish();
    *nnz_out = h[1];
    if (h[3]) {
        fprintf(stderr, "libkmers: internal error %u in the CSR emit (1: row-pointer fill queue overflow, 2: scan timeout)\n", h[3]);
        return -11;
    }
    if ((uint64_t)h[1] > max_nz) return -2;
    return 0;
}

}  // namespace kk
