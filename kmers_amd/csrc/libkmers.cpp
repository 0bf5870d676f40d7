// libkmers.cpp -- the C ABI of libkmers.so: the reference's six entry points (include/libkmers.h) and the
// device-resident stages behind them (include/kmers_hip.h).
//
// Drop-in flow of the four file-level calls:
//   host ingest (ingest.cpp: mmap, all host threads, 2-bit pack into pinned memory)
//     -> one H2D copy of the packed store (3 bits per base)
//     -> HIP kernels (kernels.hip)
//     -> D2H of the result into the caller's numpy buffers.
// There is no CPU compute path: without a GPU these calls return KMERS_ERR_NO_DEVICE.
#include "kmers_hip.h"
#include "kernels.hpp"
#include "store.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>
#include <sys/stat.h>

#include <hip/hip_runtime_api.h>

namespace {

#define API_HIP(expr)                                                                                            \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            fprintf(stderr, "libkmers: HIP error %s at %s:%d (%s)\n", hipGetErrorString(e_), __FILE__, __LINE__, \
                    #expr);                                                                                      \
            return KMERS_ERR_HIP;                                                                                \
        }                                                                                                        \
    } while (0)

// 0..3 for A,C,G,T, -1 otherwise (reference src/libkmers.cpp:199-211)
struct BaseIds {
    signed char t[256];
    BaseIds() {
        memset(t, -1, sizeof t);
        t[(unsigned char)'A'] = 0;
        t[(unsigned char)'C'] = 1;
        t[(unsigned char)'G'] = 2;
        t[(unsigned char)'T'] = 3;
    }
};
const BaseIds g_ids;

struct StoreGuard {
    kmers_store *s = nullptr;
    ~StoreGuard() { kmers_store_free(s); }
};

// device buffers that free themselves
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 256); }
    template <class T>
    T *as() const {
        return (T *)p;
    }
};

// Wall time of the stages of a file-level call: always kept for kmers_last_call_stages() (bench.py's `e2e` record), and
// printed to stderr with KMERS_VERBOSE=2 (the reference prints its phase times to stdout, src/libkmers.cpp:34-41, :86-168;
// nothing is printed by default here).
std::mutex g_stage_mu;
std::string g_last_stages;
struct StageClock {
    bool print;
    std::chrono::steady_clock::time_point t0;
    std::string acc;
    StageClock() : print(kk::knobs().verbose >= 2), t0(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        const auto t1 = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        char line[128];
        snprintf(line, sizeof line, "%s=%.3f;", what, ms);
        acc += line;
        if (print) fprintf(stderr, "libkmers: %-28s %8.3f ms\n", what, ms);
        t0 = t1;
    }
    ~StageClock() {
        std::lock_guard<std::mutex> lk(g_stage_mu);
        g_last_stages = acc;
    }
};

// Device -> caller's (pageable) memory.  hipMemcpy into pageable memory goes through the runtime's small bounce
// buffers at a fraction of the link rate; here 32 MiB chunks land in two pinned buffers by DMA while the host threads
// move the previous chunk to its destination with `sink(dst offset, pinned src, bytes)` (a memcpy, or the "+=" of the
// y calls).  Leaves the stream idle.
// The two pinned chunks and their events are kept for the life of the process (pinning 64 MiB costs milliseconds, and a
// file-level X call makes three of these copies); concurrent callers take turns.
struct PinPool {
    std::mutex mu;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    int dev = -1;
    static constexpr size_t chunk = (size_t)32 << 20;
    int ready() {  // with mu held
        int d = -1;
        if (hipGetDevice(&d) != hipSuccess) return KMERS_ERR_HIP;
        if (pin[0] && d == dev) return KMERS_OK;
        for (int i = 0; i < 2; ++i) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (pin[i]) (void)hipHostFree(pin[i]);
            pin[i] = nullptr, ev[i] = nullptr;
        }
        for (int i = 0; i < 2; ++i)
            if (hipHostMalloc(&pin[i], chunk, hipHostMallocDefault) != hipSuccess ||
                hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                return KMERS_ERR_HIP;
            }
        dev = d;
        return KMERS_OK;
    }
};
PinPool g_pins;

template <class Sink>
int d2h_chunks(const void *d_src, size_t bytes, hipStream_t s, const Sink &sink) {
    if (bytes == 0) return KMERS_OK;
    std::lock_guard<std::mutex> lk(g_pins.mu);
    int rc = g_pins.ready();
    const size_t chunk = PinPool::chunk;
    void *const *pin = g_pins.pin;
    hipEvent_t *ev = g_pins.ev;
    const size_t n_chunks = (bytes + chunk - 1) / chunk;
    auto issue = [&](size_t c) {
        const size_t off = c * chunk, n = std::min(chunk, bytes - off);
        if (hipMemcpyAsync(pin[c & 1], (const char *)d_src + off, n, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipEventRecord(ev[c & 1], s) != hipSuccess)
            rc = KMERS_ERR_HIP;
    };
    if (rc == KMERS_OK) issue(0);
    for (size_t c = 0; c < n_chunks && rc == KMERS_OK; ++c) {
        if (hipEventSynchronize(ev[c & 1]) != hipSuccess) {
            rc = KMERS_ERR_HIP;
            break;
        }
        if (c + 1 < n_chunks) issue(c + 1);  // the next chunk's DMA runs under this chunk's host copy
        const size_t off = c * chunk, n = std::min(chunk, bytes - off);
        sink(off, (const char *)pin[c & 1], n);
    }
    if (rc != KMERS_OK) {
        (void)hipGetLastError();
        fprintf(stderr, "libkmers: HIP error in the device-to-host copy\n");
    }
    (void)hipStreamSynchronize(s);
    return rc;
}

int d2h_copy(void *dst, const void *d_src, size_t bytes, hipStream_t s) {
    return d2h_chunks(d_src, bytes, s, [&](size_t off, const char *src, size_t n) {
        kmers::parallel_for(n, (size_t)1 << 20, [&](size_t lo, size_t hi) { memcpy((char *)dst + off + lo, src + lo, hi - lo); });
    });
}

int select_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return KMERS_ERR_NO_DEVICE;
    }
    if (const char *e = getenv("KMERS_DEVICE")) {
        const int d = atoi(e);
        if (d < 0 || d >= n) return KMERS_ERR_BAD_ARG;
        if (hipSetDevice(d) != hipSuccess) return KMERS_ERR_HIP;
    }
    return KMERS_OK;
}

struct DevStore {
    DevBuf bases, valid, rec_start;
    int upload(const kmers_store *st, bool with_records, hipStream_t s) {
        API_HIP(bases.alloc(st->bases_words * 4));
        API_HIP(valid.alloc(st->valid_words * 4));
        API_HIP(hipMemcpyAsync(bases.p, st->bases, st->bases_words * 4, hipMemcpyHostToDevice, s));
        API_HIP(hipMemcpyAsync(valid.p, st->valid, st->valid_words * 4, hipMemcpyHostToDevice, s));
        if (with_records) {
            API_HIP(rec_start.alloc(st->rec_start.size() * 8));
            API_HIP(hipMemcpyAsync(rec_start.p, st->rec_start.data(), st->rec_start.size() * 8, hipMemcpyHostToDevice, s));
        }
        return KMERS_OK;
    }
};

// First touch of the caller's output arrays on a few threads of their own, from the start of a file-level call to its
// end.  numpy.zeros hands out untouched pages; faulting 4.6 GB in costs 60-100 ms on the GPU box whatever the number of
// threads above four (the kernel zeroes every page: memory-bound, tools/mb_touch.c) -- as much as the PCIe transfer it
// used to be paid inside.  Under the parse and the GPU build (compute-bound) about half of it disappears; started
// only after the parse it competes with the copy for the same memory bandwidth and hides nothing (measured both ways).  A page is touched with an
// atomic OR of zero into one of its bytes: it changes nothing and, being one indivisible read-modify-write, cannot
// undo a store the copy threads make to the same byte at the same time, so the two may overlap freely.  Huge pages are
// asked for first (512 times fewer faults where the system grants them).
int prefault_threads() {
    if (const char *e = getenv("KMERS_PREFAULT_THREADS")) return std::max(0, atoi(e));  // (experiments; 0 = off)
    return std::max(1, kmers::host_threads() / 4);
}

class Prefault {
    struct Range { char *p; size_t bytes; };
    std::vector<Range> ranges_;
    std::vector<std::thread> threads_;
    std::atomic<size_t> next_{0};
    std::atomic<bool> stop_{false};
    static constexpr size_t BLOCK = (size_t)2 << 20;
    size_t blocks_of(const Range &r) const { return (r.bytes + BLOCK - 1) / BLOCK; }
    void work() {
        size_t total = 0;
        for (const Range &r : ranges_) total += blocks_of(r);
        for (;;) {
            size_t b = next_.fetch_add(1, std::memory_order_relaxed);
            if (b >= total || stop_.load(std::memory_order_relaxed)) return;
            size_t ri = 0;
            while (b >= blocks_of(ranges_[ri])) b -= blocks_of(ranges_[ri++]);
            const Range &r = ranges_[ri];
            char *lo = r.p + b * BLOCK, *hi = std::min(r.p + r.bytes, lo + BLOCK);
            const uintptr_t a = ((uintptr_t)lo + BLOCK - 1) & ~(uintptr_t)(BLOCK - 1);
            if (a + BLOCK <= (uintptr_t)r.p + r.bytes) (void)madvise((void *)a, BLOCK, MADV_HUGEPAGE);  // the aligned block that starts in here
            for (char *q = lo; q < hi; q += 4096) (void)__atomic_fetch_or(q, (char)0, __ATOMIC_RELAXED);
            (void)__atomic_fetch_or(hi - 1, (char)0, __ATOMIC_RELAXED);
        }
    }

public:
    Prefault() = default;
    Prefault(const Prefault &) = delete;
    Prefault &operator=(const Prefault &) = delete;
    ~Prefault() { finish(); }
    void add(void *p, size_t bytes) {
        if (p && bytes) ranges_.push_back(Range{(char *)p, bytes});
    }
    void start(int n_threads) {
        if (ranges_.empty()) return;
        for (int i = 0; i < n_threads; ++i) threads_.emplace_back([this] { work(); });
    }
    void finish() {  // after this no thread of ours touches the arrays
        stop_.store(true, std::memory_order_relaxed);
        for (std::thread &t : threads_) t.join();
        threads_.clear();
    }
};

int count_files(int n_files, const char *fnames[], int K, int *counts, int mode) {
    if (K < 1 || K > 16 || !counts || n_files < 0) return KMERS_ERR_BAD_ARG;
    StageClock clk;
    StoreGuard g;
    Prefault pre;
    pre.add(counts, ((size_t)1 << (2 * K)) * 4);
    pre.start(prefault_threads());
    if (int rc = kmers_store_ingest(n_files, fnames, mode, 0, 0, 1, &g.s)) return rc;
    clk.lap("host ingest");
    if (int rc = select_device()) return rc;
    const kmers_store *st = g.s;
    const uint64_t n_pos = kmers_store_n_pos(st);
    if (n_pos == 0) return KMERS_OK;
    const uint64_t M = 1ull << (2 * K);
    hipStream_t s = nullptr;
    DevStore ds;
    if (int rc = ds.upload(st, false, s)) return rc;
    DevBuf dc;
    API_HIP(dc.alloc(M * 4));
    API_HIP(hipMemsetAsync(dc.p, 0, M * 4, s));
    DevBuf d_ws;
    const size_t ws_bytes = kk::y_workspace_bytes(n_pos, K);
    if (ws_bytes) API_HIP(d_ws.alloc(ws_bytes));
    clk.lap("H2D + device buffers");
    if (int rc = kk::count_y(ds.bases.as<uint32_t>(), st->bases_words, ds.valid.as<uint32_t>(), st->valid_words, n_pos, K,
                             dc.as<int32_t>(), d_ws.p, ws_bytes, s))
        return rc;
    clk.lap("histogram (GPU, enqueue)");
    // the reference only ever increments the caller's array (src/libkmers.cpp:213-216): add, do not overwrite
    if (int rc = d2h_chunks(dc.p, M * 4, s, [&](size_t off, const char *src, size_t n) {
            int *dst = counts + off / 4;
            const int32_t *v = (const int32_t *)src;
            kmers::parallel_for(n / 4, (size_t)1 << 18, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) dst[i] += v[i];
            });
        }))
        return rc;
    pre.finish();
    clk.lap("counts += (D2H)");
    return KMERS_OK;
}

constexpr uint64_t X_MAX_POS = 0xFFFF0000ull, X_MAX_REC = 0x7FFFFFFFull;  // one kmers_hip_build_x call (32-bit entry indices)

// One store -> CSR in host arrays (data / col_ind hold min(max_size, windows) entries, row_ind 4^K + 1); columns are
// numbered from the store's first record.
// idx32: row_ind and col_ind are int32 arrays (the caller narrows anyway: scipy); KMERS_ERR_TOO_LARGE when a value would not fit.
int build_store(const kmers_store *st, int K, uint64_t max_size, float *data, uint64_t *row_ind, uint64_t *col_ind, uint64_t *nnz,
                StageClock &clk, Prefault *pre = nullptr, bool idx32 = false) {
    const uint64_t n_pos = kmers_store_n_pos(st), n_rec = kmers_store_n_rec(st);
    if (n_pos > X_MAX_POS || n_rec > X_MAX_REC) return KMERS_ERR_TOO_LARGE;
    const uint64_t M = 1ull << (2 * K);
    const uint64_t windows = kmers_store_n_windows(st, K);   // upper bound on nnz
    const uint64_t cap = std::min<uint64_t>(max_size, windows);
    hipStream_t s = nullptr;
    DevStore ds;
    if (int rc = ds.upload(st, true, s)) return rc;
    DevBuf d_data, d_col, d_row, d_ws;
    const size_t ws_bytes = kk::x_workspace_bytes(n_pos, n_rec, K);
    API_HIP(d_data.alloc(cap * 4));
    API_HIP(d_col.alloc(cap * 8));
    API_HIP(d_row.alloc((M + 1) * 8));
    API_HIP(d_ws.alloc(ws_bytes));
    clk.lap("H2D + device buffers");
    uint64_t nz = 0;
    const int rc = kk::build_x(ds.bases.as<uint32_t>(), st->bases_words, ds.valid.as<uint32_t>(), st->valid_words,
                               ds.rec_start.as<uint64_t>(), n_pos, n_rec, kmers_store_first_rec(st), K, d_data.as<float>(),
                               d_col.as<uint64_t>(), d_row.as<uint64_t>(), cap, d_ws.p, ws_bytes, &nz, s);
    if (rc == KMERS_ERR_MAX_NZ && nz <= max_size) {
        fprintf(stderr, "libkmers: internal error: nnz %llu above the window count %llu\n", (unsigned long long)nz,
                (unsigned long long)windows);
        return KMERS_ERR_HIP;
    }
    if (rc) return rc;
    clk.lap("build X (GPU)");
    // PCIe bounds this stage (config 2: 4.6 GB of CSR): indices travel as 32-bit values -- narrowed on the device into the
    // workspace, which is free again, widened by the host threads as they move each chunk into the caller's uint64
    // arrays -- a third fewer bytes on the link.  (Every value fits: nnz and the column numbers of one call are < 2^32.)
    auto d2h_widen = [&](uint64_t *dst, const uint32_t *d_src32, uint64_t n) {
        return d2h_chunks(d_src32, n * 4, s, [&](size_t off, const char *src, size_t nb) {
            uint64_t *out = dst + off / 4;
            const uint32_t *v = (const uint32_t *)src;
            kmers::parallel_for(nb / 4, (size_t)1 << 18, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) out[i] = v[i];
            });
        });
    };
    // The caller's arrays are usually fresh (numpy.zeros: untouched pages), so this copy also pays their first-touch
    // page faults; asking for huge pages first makes those 512 times fewer where the system allows it.
    auto huge = [](void *p, size_t bytes) {
        const uintptr_t a = ((uintptr_t)p + (2u << 20) - 1) & ~(uintptr_t)((2u << 20) - 1), e = ((uintptr_t)p + bytes) & ~(uintptr_t)((2u << 20) - 1);
        if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
    };
    const size_t isz = idx32 ? 4 : 8;
    if (nz) {
        huge(data, nz * 4);
        huge(col_ind, nz * isz);
    }
    huge(row_ind, (M + 1) * isz);
    const bool cols_fit = kmers_store_first_rec(st) + n_rec <= 0xFFFFFFFFull;
    const size_t col32_bytes = (size_t)((nz * 4 + 255) / 256 * 256), row32_bytes = (size_t)(M + 1) * 4;
    const bool narrow_cols = cols_fit && nz && col32_bytes <= ws_bytes;
    const bool narrow_rows = (narrow_cols ? col32_bytes : 0) + row32_bytes <= ws_bytes;
    if (idx32 && (nz > 0x7FFFFFFFull || kmers_store_first_rec(st) + n_rec > 0x7FFFFFFFull))
        return KMERS_ERR_TOO_LARGE;  // (the 64-bit call takes it)
    uint32_t *d_col32 = d_ws.as<uint32_t>();
    uint32_t *d_row32 = (uint32_t *)((char *)d_ws.p + (narrow_cols ? col32_bytes : 0));
    DevBuf d_n32;  // idx32 with a workspace too small for the narrowed copies (few positions, many rows): a buffer of their own
    bool do_cols = narrow_cols, do_rows = narrow_rows;
    if (idx32 && ((nz && !narrow_cols) || !narrow_rows)) {
        API_HIP(d_n32.alloc(col32_bytes + row32_bytes));
        d_col32 = d_n32.as<uint32_t>();
        d_row32 = (uint32_t *)((char *)d_n32.p + col32_bytes);
        do_cols = nz != 0, do_rows = true;
    }
    if (do_cols)
        if (int e = kk::narrow64(d_col.as<uint64_t>(), d_col32, nz, s)) return e;
    if (do_rows)
        if (int e = kk::narrow64(d_row.as<uint64_t>(), d_row32, M + 1, s)) return e;
    if (idx32) {  // the narrowed arrays as they are (values < 2^31: int32 == uint32)
        if (int e = d2h_copy(row_ind, d_row32, (M + 1) * 4, s)) return e;
        if (nz)
            if (int e = d2h_copy(col_ind, d_col32, nz * 4, s)) return e;
    } else if (narrow_rows) {
        if (int e = d2h_widen(row_ind, d_row32, M + 1)) return e;
    } else {
        if (int e = d2h_copy(row_ind, d_row.p, (M + 1) * 8, s)) return e;
    }
    if (nz) {
        if (idx32) {
        } else if (narrow_cols) {
            if (int e = d2h_widen(col_ind, d_col32, nz)) return e;
        } else {
            if (int e = d2h_copy(col_ind, d_col.p, nz * 8, s)) return e;
        }
        if (int e = d2h_copy(data, d_data.p, nz * 4, s)) return e;
    }
    if (pre) pre->finish();  // (what it has not reached by now the copy has touched itself)
    clk.lap("D2H into the host arrays");
    *nnz = nz;
    return KMERS_OK;
}

struct HostCsr {  // one chunk's matrix while the others are built
    float *data = nullptr;
    uint64_t *col = nullptr, *row = nullptr;
    uint64_t nnz = 0;
    HostCsr() = default;
    HostCsr(const HostCsr &) = delete;
    HostCsr &operator=(const HostCsr &) = delete;
    HostCsr(HostCsr &&o) noexcept : data(o.data), col(o.col), row(o.row), nnz(o.nnz) { o.data = nullptr, o.col = o.row = nullptr; }
    ~HostCsr() {
        free(data);
        free(col);
        free(row);
    }
};

int build_files(int n_files, const char *fnames[], int K, int L, int mode, float *data, uint64_t *row_ind, uint64_t *col_ind,
                uint64_t max_size, uint64_t *nnz, int *n_cols, bool idx32 = false) {
    if (K < 1 || K > 16 || !row_ind || !nnz || !n_cols || n_files < 0 || (max_size && (!data || !col_ind)))
        return KMERS_ERR_BAD_ARG;
    StageClock clk;
    const uint64_t M = 1ull << (2 * K);
    uint64_t n_pos, n_rec_total, n_ranges;
    // the output arrays are faulted in while the files are parsed and the GPU builds: row_ind whole, of data / col_ind
    // what the files can fill at most (every non-zero has a base of its own in them; max_size may be far larger)
    uint64_t file_bytes = 0;
    for (int f = 0; f < n_files; ++f) {
        struct stat sb;
        if (fnames && fnames[f] && stat(fnames[f], &sb) == 0 && sb.st_size > 0) file_bytes += (uint64_t)sb.st_size;
    }
    const uint64_t pre_n = std::min(max_size, file_bytes);
    Prefault pre;
    pre.add(row_ind, (size_t)(M + 1) * (idx32 ? 4 : 8));
    pre.add(col_ind, (size_t)pre_n * (idx32 ? 4 : 8));
    pre.add(data, (size_t)pre_n * 4);
    pre.start(prefault_threads());
    {
        StoreGuard g;
        if (int rc = kmers_store_ingest(n_files, fnames, mode, L, 0, 1, &g.s)) return rc;
        clk.lap("host ingest");
        if (int rc = select_device()) return rc;
        n_pos = kmers_store_n_pos(g.s);
        n_rec_total = kmers_store_n_rec_total(g.s);
        // positions one device call takes; KMERS_X_CHUNK_POS lowers it (tests of the chunked route at small sizes)
        uint64_t chunk_pos = X_MAX_POS;
        if (const unsigned long long v = kk::knobs().x_chunk_pos)
            if (v < X_MAX_POS) chunk_pos = v;
        if (n_pos <= chunk_pos && n_rec_total <= X_MAX_REC) {
            uint64_t nz = 0;
            if (int rc = build_store(g.s, K, max_size, data, row_ind, col_ind, &nz, clk, &pre, idx32)) return rc;
            *nnz = nz;
            *n_cols = (int)n_rec_total;
            return KMERS_OK;
        }
        pre.finish();
        if (idx32) return KMERS_ERR_TOO_LARGE;  // (ranges are merged from 64-bit parts: the 64-bit call takes it)
        // more positions than one device call indexes: contiguous record ranges (the ingest's shards, balanced by
        // positions), one matrix each, interleaved row by row at the end (columns ascend from range to range)
        const uint64_t *rs = kmers_store_rec_start(g.s);
        uint64_t longest = 0;
        for (uint64_t r = 0, n = kmers_store_n_rec(g.s); r < n; ++r) longest = std::max(longest, rs[r + 1] - rs[r]);
        n_ranges = kmers_x_range_count(n_pos, n_rec_total, longest, chunk_pos);
        if (n_ranges == 0) return KMERS_ERR_TOO_LARGE;
    }
    const int n_parts = (int)n_ranges;
    std::vector<HostCsr> parts;
    parts.reserve((size_t)n_parts);
    uint64_t total = 0;
    for (int p = 0; p < n_parts; ++p) {
        StoreGuard g;
        if (int rc = kmers_store_ingest(n_files, fnames, mode, L, p, n_parts, &g.s)) return rc;
        if (kmers_store_n_rec(g.s) == 0) continue;  // (the room left for the longest record makes some ranges empty)
        const uint64_t windows = kmers_store_n_windows(g.s, K);
        parts.emplace_back();
        HostCsr &h = parts.back();
        h.data = (float *)malloc(std::max<uint64_t>(windows, 1) * 4);
        h.col = (uint64_t *)malloc(std::max<uint64_t>(windows, 1) * 8);
        h.row = (uint64_t *)malloc((M + 1) * 8);
        if (!h.data || !h.col || !h.row) return KMERS_ERR_NOMEM;
        if (int rc = build_store(g.s, K, windows, h.data, h.row, h.col, &h.nnz, clk)) return rc;  // (a single record above the cap: TOO_LARGE)
        total += h.nnz;
        if (h.nnz < windows) {  // give the slack back before the next range is built
            if (void *q = realloc(h.data, std::max<uint64_t>(h.nnz, 1) * 4)) h.data = (float *)q;
            if (void *q = realloc(h.col, std::max<uint64_t>(h.nnz, 1) * 8)) h.col = (uint64_t *)q;
        }
    }
    *nnz = total;
    *n_cols = (int)n_rec_total;
    if (total > max_size) return KMERS_ERR_MAX_NZ;
    std::vector<const float *> pd;
    std::vector<const uint64_t *> pr, pc;
    for (const HostCsr &h : parts) pd.push_back(h.data), pr.push_back(h.row), pc.push_back(h.col);
    if (int rc = kmers_csr_merge((int)parts.size(), M, pd.data(), pr.data(), pc.data(), data, row_ind, col_ind)) return rc;
    clk.lap("interleave of the ranges");
    return KMERS_OK;
}

// Every failure of a file-level call says so on stderr: the reference's own wrapper only looks for -1 and -2
// (kmerexpr/libkmers.py:102-103, :173-176), so without this line a code <= -10 would go by unnoticed there.
// KMERS_REFERENCE_CODES=1 folds those codes into -1, the one failure the unchanged wrapper raises on.
int report(int rc, const char *call) {
    if (rc == KMERS_OK) return rc;
    const char *why = "unknown error";
    switch (rc) {
        case KMERS_ERR_OPEN: why = "a FASTA / FASTQ file could not be opened"; break;
        case KMERS_ERR_MAX_NZ: why = "more non-zeros than max_size"; break;
        case KMERS_ERR_NO_DEVICE: why = "no usable HIP device (this library has no CPU fallback); outputs were not written"; break;
        case KMERS_ERR_HIP: why = "a HIP runtime call failed; outputs are incomplete"; break;
        case KMERS_ERR_BAD_ARG: why = "bad argument (K outside 1..16, a null pointer, a negative count, KMERS_DEVICE out of range)"; break;
        case KMERS_ERR_TOO_LARGE: why = "input too large (one record, or one device-level call, of more than 2^32-65536 positions)"; break;
        case KMERS_ERR_WORKSPACE: why = "workspace too small"; break;
        case KMERS_ERR_NOMEM: why = "host allocation failed"; break;
    }
    fprintf(stderr, "libkmers: %s failed with code %d: %s\n", call, rc, why);
    if (rc <= KMERS_ERR_NO_DEVICE) {
        const char *e = getenv("KMERS_REFERENCE_CODES");
        if (e && atoi(e) != 0) return KMERS_ERR_OPEN;
    }
    return rc;
}

}  // namespace

extern "C" {

// A record belongs to range floor(start * n_ranges / n_pos) (ingest.cpp), so a range holds at most
// ceil(n_pos / n_ranges) + longest - 1 positions: the count leaves that much room under the cap.
uint64_t kmers_x_range_count(uint64_t n_pos, uint64_t n_rec_total, uint64_t longest_rec_pos, uint64_t cap_pos) {
    if (cap_pos == 0 || cap_pos > X_MAX_POS) cap_pos = X_MAX_POS;
    if (longest_rec_pos > cap_pos) return 0;
    uint64_t want = n_rec_total / X_MAX_REC + 1;
    if (n_pos > cap_pos) {
        const uint64_t room = cap_pos - longest_rec_pos + 1;   // >= 1
        want = std::max(want, (n_pos + room - 1) / room);
    }
    return want > 4096 ? 0 : want;
}

// ---- the reference's six symbols ----------------------------------------------------------------------

uint32_t kmer_to_id(const char *kmer, int len) {
    // reference src/libkmers.cpp:331-337: id = id*4 + table[byte]; a non-ACGT byte adds the table's -1
    uint32_t id = 0;
    for (int i = 0; i < len; ++i) id = id * 4u + (uint32_t)(int)g_ids.t[(unsigned char)kmer[i]];
    return id;
}

bool valid_kmer(const char *kmer, int len) {
    for (int i = 0; i < len; ++i)
        if (g_ids.t[(unsigned char)kmer[i]] < 0) return false;
    return true;
}

int fasta_count_kmers(int n_files, const char *fnames[], int K, int *total_kmer_counts) {
    return report(count_files(n_files, fnames, K, total_kmer_counts, KMERS_MODE_Y_FASTA), "fasta_count_kmers");
}

int fastq_gz_count_kmers(int n_files, const char *fnames[], int K, int *total_kmer_counts) {
    return report(count_files(n_files, fnames, K, total_kmer_counts, KMERS_MODE_Y_FASTQ), "fastq_gz_count_kmers");
}

int fasta_to_kmers_csr(int n_files, const char *fnames[], int K, int L, float *data, uint64_t *row_ind, uint64_t *col_ind,
                       uint64_t max_size, uint64_t *nnz, int *n_cols) {
    return report(build_files(n_files, fnames, K, L, KMERS_MODE_X, data, row_ind, col_ind, max_size, nnz, n_cols),
                  "fasta_to_kmers_csr");
}

int fasta_to_kmers_csr_cat_subseq(int n_files, const char *fnames[], int K, int L, float *data, uint64_t *row_ind,
                                  uint64_t *col_ind, uint64_t max_size, uint64_t *nnz, int *n_cols) {
    return report(build_files(n_files, fnames, K, L, KMERS_MODE_X_CAT, data, row_ind, col_ind, max_size, nnz, n_cols),
                  "fasta_to_kmers_csr_cat_subseq");
}

int kmers_fasta_to_kmers_csr_i32(int n_files, const char *fnames[], int K, int L, int concatenate_subseq, float *data,
                                 int32_t *row_ind, int32_t *col_ind, uint64_t max_size, uint64_t *nnz, int *n_cols) {
    // (no report(): KMERS_ERR_TOO_LARGE is this call's way of saying "use the 64-bit one", not a failure to announce)
    const int rc = build_files(n_files, fnames, K, L, concatenate_subseq ? KMERS_MODE_X_CAT : KMERS_MODE_X, data,
                               (uint64_t *)row_ind, (uint64_t *)col_ind, max_size, nnz, n_cols, true);
    return rc == KMERS_ERR_TOO_LARGE ? rc : report(rc, "kmers_fasta_to_kmers_csr_i32");
}

// ---- device-resident stages ----------------------------------------------------------------------------

int kmers_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int kmers_hip_store_upload(const kmers_store *st, uint32_t *d_bases, uint32_t *d_valid, uint64_t *d_rec_start, void *stream) {
    if (!st || !d_bases || !d_valid) return KMERS_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    API_HIP(hipMemcpyAsync(d_bases, st->bases, st->bases_words * 4, hipMemcpyHostToDevice, s));
    API_HIP(hipMemcpyAsync(d_valid, st->valid, st->valid_words * 4, hipMemcpyHostToDevice, s));
    if (d_rec_start)
        API_HIP(hipMemcpyAsync(d_rec_start, st->rec_start.data(), st->rec_start.size() * 8, hipMemcpyHostToDevice, s));
    API_HIP(hipStreamSynchronize(s));  // the store may be freed as soon as this returns
    return KMERS_OK;
}

int kmers_hip_count_y(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words, uint64_t n_pos,
                      int K, int32_t *d_counts, void *stream) {
    if (K < 1 || K > 16 || !d_counts || (n_pos && (!d_bases || !d_valid))) return KMERS_ERR_BAD_ARG;
    if (bases_words < (n_pos + 15) / 16 || valid_words < (n_pos + 31) / 32) return KMERS_ERR_BAD_ARG;
    const size_t ws_bytes = kk::y_workspace_bytes(n_pos, K);
    if (!ws_bytes) return kk::count_y(d_bases, bases_words, d_valid, valid_words, n_pos, K, d_counts, nullptr, 0, (hipStream_t)stream);
    // K >= 9 works in a scratch buffer: this call owns one for its duration, so it has to wait for the stream
    DevBuf d_ws;
    API_HIP(d_ws.alloc(ws_bytes));
    const int rc = kk::count_y(d_bases, bases_words, d_valid, valid_words, n_pos, K, d_counts, d_ws.p, ws_bytes, (hipStream_t)stream);
    API_HIP(hipStreamSynchronize((hipStream_t)stream));
    return rc;
}

size_t kmers_hip_y_workspace_bytes(uint64_t n_pos, int K) {
    if (K < 1 || K > 16) return 0;
    return kk::y_workspace_bytes(n_pos, K);
}

int kmers_hip_count_y_ws(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words, uint64_t n_pos,
                         int K, int32_t *d_counts, void *d_workspace, size_t workspace_bytes, void *stream) {
    if (K < 1 || K > 16 || !d_counts || (n_pos && (!d_bases || !d_valid))) return KMERS_ERR_BAD_ARG;
    if (bases_words < (n_pos + 15) / 16 || valid_words < (n_pos + 31) / 32) return KMERS_ERR_BAD_ARG;
    if (K >= 9 && n_pos && !d_workspace) return KMERS_ERR_WORKSPACE;
    return kk::count_y(d_bases, bases_words, d_valid, valid_words, n_pos, K, d_counts, d_workspace, workspace_bytes,
                       (hipStream_t)stream);
}

int kmers_hip_count_y_static(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words,
                             uint64_t n_pos, int K, int32_t *d_counts, void *stream) {
    if (K < 1 || K > 16 || !d_counts || (n_pos && (!d_bases || !d_valid))) return KMERS_ERR_BAD_ARG;
    if (bases_words < (n_pos + 15) / 16 || valid_words < (n_pos + 31) / 32) return KMERS_ERR_BAD_ARG;
    if (n_pos > kk::STATIC_MAX_POS) {
        fprintf(stderr, "libkmers: kmers_hip_count_y_static takes at most %llu positions per call (got %llu): split the "
                        "store or use kmers_hip_count_y_ws\n",
                (unsigned long long)kk::STATIC_MAX_POS, (unsigned long long)n_pos);
        return KMERS_ERR_TOO_LARGE;
    }
    void *ws = nullptr;
    size_t ws_bytes = 0;
    // K <= 8 counts in LDS and needs no scratch at all.  The lookup (hipGetSymbolAddress, first call on a device only) is
    // made for an empty input too: that is the warm-up call of a caller about to capture a graph.
    if (K >= 9)
        if (int rc = kk::static_scratch(&ws, &ws_bytes)) return rc;
    // the same kernels as kmers_hip_count_y_ws, in chunks that fit the static array; no hipMalloc, no events
    return kk::count_y(d_bases, bases_words, d_valid, valid_words, n_pos, K, d_counts, ws, ws_bytes, (hipStream_t)stream,
                       /*may_profile=*/false);
}

size_t kmers_hip_x_workspace_bytes(uint64_t n_pos, uint64_t n_rec, int K) {
    if (K < 1 || K > 16) return 0;
    return kk::x_workspace_bytes(n_pos, n_rec, K);
}

int kmers_hip_build_x(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words,
                      const uint64_t *d_rec_start, uint64_t n_pos, uint64_t n_rec, uint64_t first_col, int K, float *d_data,
                      uint64_t *d_col_ind, uint64_t *d_row_ind, uint64_t max_nz, void *d_workspace, size_t workspace_bytes,
                      uint64_t *nnz_out, void *stream) {
    if (K < 1 || K > 16 || !d_row_ind || !nnz_out || !d_workspace) return KMERS_ERR_BAD_ARG;
    if (n_pos && (!d_bases || !d_valid || !d_rec_start)) return KMERS_ERR_BAD_ARG;
    if (max_nz && (!d_data || !d_col_ind)) return KMERS_ERR_BAD_ARG;
    if (bases_words < (n_pos + 15) / 16 || valid_words < (n_pos + 31) / 32) return KMERS_ERR_BAD_ARG;
    if (n_pos > 0xFFFF0000ull || n_rec > 0x7FFFFFFFull) return KMERS_ERR_TOO_LARGE;
    return kk::build_x(d_bases, bases_words, d_valid, valid_words, d_rec_start, n_pos, n_rec, first_col, K, d_data, d_col_ind,
                       d_row_ind, max_nz, d_workspace, workspace_bytes, nnz_out, (hipStream_t)stream);
}

// ---- per-kernel timing (HIP events on the launch stream) ------------------------------------------------------

void kmers_hip_profile_enable(int on) {
    kk::profile_enable(on != 0);
    if (on) kk::profile_reset();
}

int kmers_hip_profile_count(void) { return (int)kk::profile_snapshot().size(); }

int kmers_hip_profile_get(int i, char *name, size_t name_cap, double *ms_total, uint64_t *calls) {
    const auto snap = kk::profile_snapshot();
    if (i < 0 || (size_t)i >= snap.size() || !name || !name_cap || !ms_total || !calls) return KMERS_ERR_BAD_ARG;
    snprintf(name, name_cap, "%s", snap[(size_t)i].name.c_str());
    *ms_total = snap[(size_t)i].ms;
    *calls = snap[(size_t)i].calls;
    return KMERS_OK;
}

int kmers_last_call_stages(char *buf, size_t cap) {
    if (!buf || !cap) return KMERS_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g_stage_mu);
    snprintf(buf, cap, "%s", g_last_stages.c_str());
    return KMERS_OK;
}

int kmers_hip_debug_knob(const char *name, long long value) { return kk::set_knob(name, value); }

// diagnostic builds only (-DKK_STAMPS): per-phase cycle sums of the scatter kernels; KMERS_ERR_BAD_ARG otherwise
int kmers_hip_debug_stamps(unsigned long long *out16, int reset) { return kk::read_stamps(out16, reset != 0); }

// ---- host merge of per-shard CSRs -------------------------------------------------------------------------

int kmers_csr_merge(int n_parts, uint64_t n_rows, const float *const data[], const uint64_t *const row_ind[],
                    const uint64_t *const col_ind[], float *out_data, uint64_t *out_row_ind, uint64_t *out_col_ind) {
    if (n_parts < 1 || !data || !row_ind || !col_ind || !out_row_ind) return KMERS_ERR_BAD_ARG;
    // out_row_ind[r] = sum over parts of row_ind_g[r]  (each part's pointer is its own exclusive prefix)
    kmers::parallel_for((size_t)n_rows + 1, 1u << 18, [&](size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; ++r) {
            uint64_t s = 0;
            for (int g = 0; g < n_parts; ++g) s += row_ind[g][r];
            out_row_ind[r] = s;
        }
    });
    kmers::parallel_for((size_t)n_rows, 1u << 14, [&](size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; ++r) {
            uint64_t at = out_row_ind[r];
            for (int g = 0; g < n_parts; ++g) {
                const uint64_t a = row_ind[g][r], b = row_ind[g][r + 1];
                if (b > a) {
                    memcpy(out_col_ind + at, col_ind[g] + a, (size_t)(b - a) * 8);
                    memcpy(out_data + at, data[g] + a, (size_t)(b - a) * 4);
                    at += b - a;
                }
            }
        }
    });
    return KMERS_OK;
}

int kmers_csr_place(uint64_t n_rows, const uint64_t *row_ind, const uint64_t *before, const uint64_t *global_row_ind,
                    const float *data, const uint64_t *col_ind, float *out_data, uint64_t *out_col_ind, uint64_t *out_row_ind) {
    if (!row_ind || !global_row_ind || !out_data || !out_col_ind) return KMERS_ERR_BAD_ARG;
    if (out_row_ind)
        kmers::parallel_for((size_t)n_rows + 1, 1u << 18, [&](size_t lo, size_t hi) {
            memcpy(out_row_ind + lo, global_row_ind + lo, (hi - lo) * 8);
        });
    kmers::parallel_for((size_t)n_rows, 1u << 14, [&](size_t lo, size_t hi) {
        for (size_t r = lo; r < hi; ++r) {
            const uint64_t a = row_ind[r], n = row_ind[r + 1] - a;
            if (!n) continue;
            const uint64_t at = global_row_ind[r] + (before ? before[r + 1] - before[r] : 0);
            memcpy(out_col_ind + at, col_ind + a, (size_t)n * 8);
            memcpy(out_data + at, data + a, (size_t)n * 4);
        }
    });
    return KMERS_OK;
}

int kmers_csr_place_rows(uint64_t n, const uint64_t *src_start, const uint32_t *len, const uint64_t *dst_start, const float *data,
                         const uint64_t *col_ind, float *out_data, uint64_t *out_col_ind) {
    if (n && (!src_start || !len || !dst_start || !data || !col_ind || !out_data || !out_col_ind)) return KMERS_ERR_BAD_ARG;
    kmers::parallel_for((size_t)n, 1u << 14, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            memcpy(out_col_ind + dst_start[i], col_ind + src_start[i], (size_t)len[i] * 8);
            memcpy(out_data + dst_start[i], data + src_start[i], (size_t)len[i] * 4);
        }
    });
    return KMERS_OK;
}

}  // extern "C"
