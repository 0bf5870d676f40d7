// ingest.cpp -- host side of the hot path: FASTA / FASTQ(.gz) -> packed sequence store.
//
// Replaces the reference's serial reader and record filtering:
//   get_next_sequence_fasta ........ reference src/libkmers.cpp:302-318
//   get_sequence_fastq_gz .......... reference src/libkmers.cpp:240-300
//   record loops / filters ......... reference src/libkmers.cpp:360-371 (y), :398-421 (cat), :444-457 (X)
// The reference parses on one thread with getline + tellg/seekg; here a file is mmap'ed, record boundaries
// are found by all host threads at once, and every record is packed to 2 bits + 1 validity bit per base
// directly into (pinned) host memory laid out as the kernels read it (include/kmers_hip.h).
//
// Line rules kept bit-for-bit (they decide column numbering and k-mer validity):
//   * the first line of a file is a header whatever it starts with; an empty first line ends the file;
//   * after that, a header is a line whose first byte is '>';
//   * a sequence line loses its trailing ' ' bytes only; '\r', tabs, inner blanks stay (as invalid bases);
//   * a record is skipped when its header contains "PREDICTED"; on the X path also when its length < L
//     (compared as unsigned in the reference, so a negative L skips everything);
//   * on the two X paths a sequence ends at its first NUL byte: the reference hands sequence.data() to a
//     function taking const std::string& (src/libkmers.cpp:233), which re-measures it with strlen.
//
// KMERS_MODE_FASTA_PY: the OTHER FASTA reader of the reference, kmerexpr/fasta.py:19-46 (python text mode), which its
// read simulator uses for the isoforms (kmerexpr/simulate_reads.py:78-92).  Its rules differ and are kept as they are:
//   * a line ends at '\n', "\r\n" or a lone '\r' (open(path, "r") translates them all);
//   * records start only at lines whose first byte is '>'; whatever precedes the first such line is ignored; if the
//     FIRST header is empty after '>' and rstrip(), the file holds no records (fasta.py:32-33);
//   * every sequence line is strip()ped on both sides of python's ASCII whitespace (9-13, 28-32); inner bytes stay;
//   * nothing ends at a NUL; a record is kept when its header does not contain "PREDICTED" and len >= L (signed);
//   * bytes >= 0x80 are refused (KMERS_ERR_BAD_ARG): python would decode them (or raise), one character per sequence.
#include "kmers_hip.h"
#include "store.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <zlib.h>
#include <immintrin.h>

#include <hip/hip_runtime_api.h>

namespace kmers {

int host_threads() {
    const char *e = getenv("KMER_NUM_THREADS");
    int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    return n;
}

void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)> &fn) {
    if (n == 0) return;
    if (grain == 0) grain = 1;
    const size_t n_chunks = (n + grain - 1) / grain;
    const int nt = (int)std::min<size_t>((size_t)host_threads(), n_chunks);
    if (nt <= 1) {
        fn(0, n);
        return;
    }
    std::atomic<size_t> next{0};
    auto body = [&]() {
        for (;;) {
            const size_t c = next.fetch_add(1, std::memory_order_relaxed);
            if (c >= n_chunks) return;
            const size_t lo = c * grain;
            fn(lo, std::min(n, lo + grain));
        }
    };
    std::vector<std::thread> th;
    th.reserve(nt - 1);
    for (int i = 1; i < nt; ++i) th.emplace_back(body);
    body();
    for (auto &t : th) t.join();
}

}  // namespace kmers

namespace {

using kmers::parallel_for;

// ---- base code table (reference src/libkmers.cpp:199-211): 0..3 for A,C,G,T, 4 = invalid
struct CodeTable {
    uint8_t t[256];
    CodeTable() {
        memset(t, 4, sizeof t);
        t[(uint8_t)'A'] = 0;
        t[(uint8_t)'C'] = 1;
        t[(uint8_t)'G'] = 2;
        t[(uint8_t)'T'] = 3;
    }
};
const CodeTable g_codes;

// ---- a read-only view of a whole file
struct FileView {
    const uint8_t *p = nullptr;
    size_t n = 0;
    bool ok = false;
    bool mapped = false;
    ~FileView() {
        if (mapped && p) munmap((void *)p, n);
    }
    bool open(const char *path) {
        int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
            // not a regular file (pipe, directory...): the reference's fstream would fail on a directory
            // and stream a pipe; only regular files are supported here
            ::close(fd);
            return false;
        }
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) {
                ::close(fd);
                return false;
            }
            madvise(m, n, MADV_SEQUENTIAL | MADV_WILLNEED);
            p = (const uint8_t *)m;
            mapped = true;
        }
        ::close(fd);
        ok = true;
        return true;
    }
};

// ---- one FASTA record as found in a file
struct RecInfo {
    size_t seq_off;      // first byte after the header line
    size_t end_off;      // first byte of the next header (or file size)
    uint64_t full_len;   // bases after stripping trailing blanks of every line
    uint64_t nul_at;     // offset (in bases) of the first NUL byte, UINT64_MAX if none
    bool predicted;
};

// ---- a run of source bytes that lands on consecutive store positions
struct Piece {
    const uint8_t *src;  // FASTA: first sequence line of the record; raw: the bytes themselves
    const uint8_t *end;
    uint64_t rec;        // global (pre-shard) record index it belongs to
    uint64_t off;        // offset inside that record
    uint64_t take;       // bases to take
    bool lines;          // true: walk FASTA lines, false: raw bytes
    bool py = false;     // lines follow kmerexpr/fasta.py (universal newlines, strip() on both sides)
};

// A growing array of trivially copyable T without value-initialisation (10^7 read records make std::vector's
// zero-fill and push_back a serial stage of their own).
// The big per-call tables (one RecInfo and one Piece per record: ~200 MB for 2*10^6 reads) come from and go back to a
// small process-wide cache: glibc maps blocks of this size afresh on every malloc and unmaps them on free, and touching
// the new pages cost more than filling the tables.  At most 8 blocks / 1 GiB are kept; a block serves requests down to a
// quarter of its size.
struct HeapCache {
    struct Block { void *p; size_t bytes; };
    std::mutex mu;
    std::vector<Block> blocks;
    static constexpr size_t MIN_BYTES = (size_t)1 << 20, MAX_BLOCKS = 8, MAX_TOTAL = (size_t)1 << 30;
    void *take(size_t need, size_t *got) {
        if (need < MIN_BYTES) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        size_t best = blocks.size();
        for (size_t i = 0; i < blocks.size(); ++i)
            if (blocks[i].bytes >= need && blocks[i].bytes / 4 <= need && (best == blocks.size() || blocks[i].bytes < blocks[best].bytes))
                best = i;
        if (best == blocks.size()) return nullptr;
        const Block b = blocks[best];
        blocks.erase(blocks.begin() + (long)best);
        *got = b.bytes;
        return b.p;
    }
    bool give(void *p, size_t bytes) {  // false: not kept (the caller frees it)
        if (bytes < MIN_BYTES) return false;
        std::lock_guard<std::mutex> lk(mu);
        size_t total = bytes;
        for (const Block &b : blocks) total += b.bytes;
        if (blocks.size() >= MAX_BLOCKS || total > MAX_TOTAL) return false;
        blocks.push_back(Block{p, bytes});
        return true;
    }
};
HeapCache g_heap_cache;

template <class T>
struct RawVec {
    T *p = nullptr;
    size_t n = 0, cap = 0;
    RawVec() = default;
    RawVec(const RawVec &) = delete;
    RawVec &operator=(const RawVec &) = delete;
    ~RawVec() {
        if (p && !g_heap_cache.give(p, cap * sizeof(T))) free(p);
    }
    T *grow(size_t more) {  // room for `more` new elements at the end; returns the first of them (nullptr: no memory)
        if (n + more > cap) {
            size_t nc = std::max(n + more, cap + cap / 2 + 16);
            T *q = nullptr;
            if (!p) {
                size_t got = 0;
                if ((q = (T *)g_heap_cache.take(nc * sizeof(T), &got)) != nullptr) nc = got / sizeof(T);
            }
            if (!q) q = (T *)realloc(p, nc * sizeof(T));
            if (!q) return nullptr;
            p = q;
            cap = nc;
        }
        T *r = p + n;
        n += more;
        return r;
    }
    bool push(const T &v) {  // false: no memory (the element is not stored)
        T *r = grow(1);
        if (!r) return false;
        *r = v;
        return true;
    }
    size_t size() const { return n; }
    const T *begin() const { return p; }
    const T *end() const { return p + n; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
};

// Exclusive prefix sums of f(0..n) in parallel: out[i] = sum_{j<i} f(j), returns the total.  out has n entries.
template <class F>
uint64_t parallel_prefix(size_t n, uint64_t *out, const F &f) {
    if (n == 0) return 0;
    const size_t grain = (size_t)1 << 16;
    const size_t nc = (n + grain - 1) / grain;
    std::vector<uint64_t> part(nc + 1, 0);
    kmers::parallel_for(nc, 1, [&](size_t lo, size_t hi) {
        for (size_t c = lo; c < hi; ++c) {
            uint64_t a = 0;
            for (size_t i = c * grain, e = std::min(n, i + grain); i < e; ++i) a += f(i);
            part[c + 1] = a;
        }
    });
    for (size_t c = 0; c < nc; ++c) part[c + 1] += part[c];
    kmers::parallel_for(nc, 1, [&](size_t lo, size_t hi) {
        for (size_t c = lo; c < hi; ++c) {
            uint64_t a = part[c];
            for (size_t i = c * grain, e = std::min(n, i + grain); i < e; ++i) {
                out[i] = a;
                a += f(i);
            }
        }
    });
    return part[nc];
}

inline const uint8_t *find_nl(const uint8_t *p, const uint8_t *end) {
    const void *r = memchr(p, '\n', (size_t)(end - p));
    return r ? (const uint8_t *)r : end;
}

// ---- kmerexpr/fasta.py line rules
inline bool py_space(uint8_t c) { return (c >= 9 && c <= 13) || (c >= 28 && c <= 32); }   // str.strip() on ASCII text
inline const uint8_t *find_eol_py(const uint8_t *p, const uint8_t *end) {                 // first '\n' or '\r'
    for (; p < end; ++p)
        if (*p == '\n' || *p == '\r') return p;
    return end;
}
// the stripped extent of one line [p, e)
inline void strip_py(const uint8_t *&p, const uint8_t *&e) {
    while (p < e && py_space(*p)) ++p;
    while (e > p && py_space(e[-1])) --e;
}

bool header_has_predicted(const uint8_t *h, size_t n) {
    return n >= 9 && memmem(h, n, "PREDICTED", 9) != nullptr;
}

// Header offsets of a file inside the byte range [lo, hi): offset 0 (unless line 0 is empty, which ends the file at
// once) and every offset that follows a '\n' and holds '>'.
void find_headers(const FileView &f, size_t lo, size_t hi, std::vector<size_t> &hdrs) {
    hdrs.clear();
    if (f.n == 0 || f.p[0] == '\n' || lo >= hi) return;
    const size_t grain = 4u << 20;
    const size_t n_chunks = (hi - lo + grain - 1) / grain;
    std::vector<std::vector<size_t>> part(n_chunks);
    parallel_for(n_chunks, 1, [&](size_t c0, size_t c1) {
        for (size_t c = c0; c < c1; ++c) {
            // a header at offset x is found through the '\n' at x - 1: scan [a - 1, b - 1)
            const size_t xa = lo + c * grain, xb = std::min(hi, xa + grain);
            const uint8_t *a = f.p + (xa ? xa - 1 : 0);
            const uint8_t *b = f.p + (xb - 1);
            auto &out = part[c];
            if (xa == 0) out.push_back(0);          // the first line is a header whatever it starts with
            while (a < b) {
                const uint8_t *nl = (const uint8_t *)memchr(a, '\n', (size_t)(b - a));
                if (!nl) break;
                if (nl[1] == '>') out.push_back((size_t)(nl + 1 - f.p));
                a = nl + 1;
            }
        }
    });
    for (auto &v : part) hdrs.insert(hdrs.end(), v.begin(), v.end());
}

// The first header at or after `from` (f.n if none): where the last record of a byte range ends.
size_t next_header(const FileView &f, size_t from, bool py) {
    if (from >= f.n) return f.n;
    if (from == 0) from = 1;
    const uint8_t *q = f.p + from, *e = f.p + f.n;
    while (q < e) {
        const uint8_t *g = (const uint8_t *)memchr(q, '>', (size_t)(e - q));
        if (!g) break;
        if (g[-1] == '\n' || (py && g[-1] == '\r')) return (size_t)(g - f.p);
        q = g + 1;
    }
    return f.n;
}

// kmerexpr/fasta.py: every '>' at the start of a line (offset 0, or after '\n' or '\r') inside [lo, hi), none if the
// file's first header is empty.  Returns false when [lo, hi) holds a byte >= 0x80.
bool find_headers_py(const FileView &f, size_t lo, size_t hi, std::vector<size_t> &hdrs) {
    hdrs.clear();
    if (f.n == 0 || lo >= hi) return true;
    const size_t grain = 4u << 20;
    const size_t n_chunks = (hi - lo + grain - 1) / grain;
    std::vector<std::vector<size_t>> part(n_chunks);
    std::atomic<bool> high{false};
    parallel_for(n_chunks, 1, [&](size_t c0, size_t c1) {
        for (size_t c = c0; c < c1; ++c) {
            const size_t a = lo + c * grain, b = std::min(hi, a + grain);
            uint8_t any = 0;
            for (size_t i = a; i < b; ++i) any |= f.p[i];
            if (any & 0x80) high.store(true, std::memory_order_relaxed);
            const uint8_t *q = f.p + a, *e = f.p + b;
            while (q < e) {
                const uint8_t *g = (const uint8_t *)memchr(q, '>', (size_t)(e - q));
                if (!g) break;
                const size_t at = (size_t)(g - f.p);
                if (at == 0 || f.p[at - 1] == '\n' || f.p[at - 1] == '\r') part[c].push_back(at);
                q = g + 1;
            }
        }
    });
    if (high.load()) return false;
    // fasta.py:29-33: an empty first header ends the generator before any record -- whichever range it lies in
    const size_t first = (f.p[0] == '>') ? 0 : next_header(f, 0, true);
    if (first < f.n) {
        const uint8_t *h = f.p + first + 1, *e = find_eol_py(h, f.p + f.n);
        while (e > h && py_space(e[-1])) --e;
        if (e == h) return true;
    }
    for (auto &v : part) hdrs.insert(hdrs.end(), v.begin(), v.end());
    return true;
}

bool analyse_records_py(const FileView &f, const std::vector<size_t> &hdrs, size_t tail_end, RawVec<RecInfo> &recs) {
    recs.n = 0;
    if (hdrs.size() && !recs.grow(hdrs.size())) return false;
    parallel_for(hdrs.size(), 256, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint8_t *h = f.p + hdrs[i];
            const uint8_t *end = f.p + ((i + 1 < hdrs.size()) ? hdrs[i + 1] : tail_end);
            const uint8_t *he = find_eol_py(h, end);
            RecInfo r;
            r.predicted = header_has_predicted(h + 1, (size_t)(he - h - 1));
            const uint8_t *p = (he < end) ? he + 1 : end;
            r.seq_off = (size_t)(p - f.p);
            r.end_off = (size_t)(end - f.p);
            r.full_len = 0;
            r.nul_at = UINT64_MAX;
            while (p < end) {
                const uint8_t *e = find_eol_py(p, end);
                const uint8_t *a = p, *b = e;
                strip_py(a, b);
                r.full_len += (uint64_t)(b - a);
                p = (e < end) ? e + 1 : end;
            }
            recs[i] = r;
        }
    });
    return true;
}

bool analyse_records(const FileView &f, const std::vector<size_t> &hdrs, size_t tail_end, bool want_nul, RawVec<RecInfo> &recs) {
    recs.n = 0;
    if (hdrs.size() && !recs.grow(hdrs.size())) return false;
    parallel_for(hdrs.size(), 256, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const uint8_t *h = f.p + hdrs[i];
            const uint8_t *end = f.p + ((i + 1 < hdrs.size()) ? hdrs[i + 1] : tail_end);
            const uint8_t *he = find_nl(h, end);
            RecInfo r;
            r.predicted = header_has_predicted(h, (size_t)(he - h));
            const uint8_t *p = (he < end) ? he + 1 : end;
            r.seq_off = (size_t)(p - f.p);
            r.end_off = (size_t)(end - f.p);
            r.full_len = 0;
            r.nul_at = UINT64_MAX;
            while (p < end) {
                const uint8_t *e = find_nl(p, end);
                size_t ln = (size_t)(e - p);
                while (ln > 0 && p[ln - 1] == ' ') --ln;
                if (want_nul && r.nul_at == UINT64_MAX && ln) {
                    const void *z = memchr(p, 0, ln);
                    if (z) r.nul_at = r.full_len + (uint64_t)((const uint8_t *)z - p);
                }
                r.full_len += ln;
                p = (e < end) ? e + 1 : end;
            }
            recs[i] = r;
        }
    });
    return true;
}

// ---- streaming packer: appends bases at consecutive positions of the store arrays.
// Words completely covered by one piece are written with plain stores; the first and last word a piece
// touches may be shared with a neighbouring piece handled by another thread and are OR-ed in atomically.
const bool g_poison = getenv("KMERS_INGEST_POISON") != nullptr;  // tests: fill the store arrays with 0xA5 before packing
// (KMERS_NO_SIMD: the scalar packer, for same-input comparisons of the two in tests)
const bool g_have_ssse3 = [] {
#if defined(__HIP_DEVICE_COMPILE__)  // (hipcc also runs this file through its device pass)
    return false;
#else
    __builtin_cpu_init();
    return __builtin_cpu_supports("ssse3") && !getenv("KMERS_NO_SIMD");
#endif
}();

class StreamPacker {
  public:
    StreamPacker(uint32_t *bases, uint32_t *valid, uint64_t pos) : B(bases), V(valid), pos_(pos) {
        b_shared_ = (pos & 15) != 0;
        v_shared_ = (pos & 31) != 0;
    }
    void push(const uint8_t *s, size_t n) {
        if (g_have_ssse3)
            push_simd(s, n);
        else
            push_scalar(s, n);
    }

  private:
    // 16 bytes -> the 32-bit word of their 2-bit codes (first base in the top bits, 0 for a byte that is no base) and
    // the 16 validity bits (bit k = byte k is one of ACGT).  The low nibbles of 'A' 'C' 'G' 'T' are 1, 3, 7, 4: one
    // shuffle looks the code up, a second one the character that nibble stands for, and a byte is valid iff it IS that
    // character (the nibbles that stand for nothing expect a byte their own low nibble rules out).
    __attribute__((target("ssse3"))) static void conv16(const uint8_t *p, uint32_t &w, uint32_t &m) {
        const __m128i x = _mm_loadu_si128(reinterpret_cast<const __m128i *>(p));
        const __m128i lo = _mm_and_si128(x, _mm_set1_epi8(0x0F));
        const __m128i lut_code = _mm_setr_epi8(0, 0, 0, 1, 3, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0);
        const __m128i lut_char = _mm_setr_epi8((char)0xFF, 'A', 0, 'C', 'T', 0, 0, 'G', 0, 0, 0, 0, 0, 0, 0, 0);
        const __m128i ok = _mm_cmpeq_epi8(x, _mm_shuffle_epi8(lut_char, lo));
        const __m128i code = _mm_and_si128(_mm_shuffle_epi8(lut_code, lo), ok);
        m = (uint32_t)_mm_movemask_epi8(ok);
        const __m128i t = _mm_maddubs_epi16(code, _mm_set1_epi16(0x0104));       // pairs: c0 * 4 + c1 (0x04 is the even byte)
        const __m128i u = _mm_madd_epi16(t, _mm_set1_epi32(0x00010010));          // fours: t0 * 16 + t1, one byte's worth per dword
        const __m128i g = _mm_shuffle_epi8(u, _mm_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1));
        w = (uint32_t)_mm_cvtsi128_si32(g);
    }
    // r <= 16 bases (the top 2 r bits of w, the low r bits of m) behind what is there
    void append(uint32_t w, uint32_t m, uint32_t r) {
        bacc_ |= (uint64_t)w << (32 - bfill_);
        bfill_ += 2 * r;
        if (bfill_ >= 32) {
            store(B + bword_++, (uint32_t)(bacc_ >> 32), b_shared_);
            b_shared_ = false;
            bacc_ <<= 32;
            bfill_ -= 32;
        }
        vacc_ |= (uint64_t)m << vfill_;
        vfill_ += r;
        if (vfill_ >= 32) {
            store(V + vword_++, (uint32_t)vacc_, v_shared_);
            v_shared_ = false;
            vacc_ >>= 32;
            vfill_ -= 32;
        }
    }
    __attribute__((target("ssse3"))) void push_simd(const uint8_t *s, size_t n) {
        if (!simd_) {  // (the first push decides: a packer never mixes the two forms)
            simd_ = true;
            bword_ = pos_ / 16, bfill_ = 2 * (uint32_t)(pos_ & 15);
            vword_ = pos_ / 32, vfill_ = (uint32_t)(pos_ & 31);
        }
        size_t i = 0;
        uint32_t w, m;
        for (; n - i >= 16; i += 16) {
            conv16(s + i, w, m);
            append(w, m, 16);
        }
        if (i < n) {  // the tail through a zero-padded copy (zero is no base: code 0, validity 0)
            alignas(16) uint8_t buf[16] = {0};
            memcpy(buf, s + i, n - i);
            conv16(buf, w, m);
            append(w, m, (uint32_t)(n - i));
        }
        pos_ += n;
    }
    void push_scalar(const uint8_t *s, size_t n) {
        uint64_t pos = pos_;
        uint32_t bw = bw_, vw = vw_;
        size_t i = 0;
        // up to a word boundary one base at a time, then whole 16-base words without the per-base boundary tests
        for (; i < n && (pos & 15); ++i) {
            const uint32_t c = g_codes.t[s[i]];
            bw |= (c & 3u) << (30 - 2 * (pos & 15));
            vw |= (uint32_t)(c < 4) << (pos & 31);
            ++pos;
            if ((pos & 15) == 0) {
                store(B + (pos - 1) / 16, bw, b_shared_);
                b_shared_ = false;
                bw = 0;
            }
            if ((pos & 31) == 0) {
                store(V + (pos - 1) / 32, vw, v_shared_);
                v_shared_ = false;
                vw = 0;
            }
        }
        for (; n - i >= 16 && (pos & 15) == 0; i += 16) {
            uint32_t w = 0, m = 0;
#pragma GCC unroll 16
            for (int k = 0; k < 16; ++k) {
                const uint32_t c = g_codes.t[s[i + k]];
                w |= (c & 3u) << (30 - 2 * k);
                m |= (uint32_t)(c < 4) << k;
            }
            store(B + pos / 16, w, b_shared_);
            b_shared_ = false;
            vw |= m << (pos & 31);
            pos += 16;
            if ((pos & 31) == 0) {
                store(V + (pos - 1) / 32, vw, v_shared_);
                v_shared_ = false;
                vw = 0;
            }
        }
        for (; i < n; ++i) {
            const uint32_t c = g_codes.t[s[i]];
            const uint32_t ok = c < 4;
            bw |= (c & 3u) << (30 - 2 * (pos & 15));
            vw |= ok << (pos & 31);
            ++pos;
            if ((pos & 15) == 0) {
                store(B + (pos - 1) / 16, bw, b_shared_);
                b_shared_ = false;
                bw = 0;
            }
            if ((pos & 31) == 0) {
                store(V + (pos - 1) / 32, vw, v_shared_);
                v_shared_ = false;
                vw = 0;
            }
        }
        pos_ = pos;
        bw_ = bw;
        vw_ = vw;
    }

  public:
    void finish() {
        if (simd_) {
            if (bfill_) store(B + bword_, (uint32_t)(bacc_ >> 32), true);
            if (vfill_) store(V + vword_, (uint32_t)vacc_, true);
            bacc_ = vacc_ = 0;
            bfill_ = vfill_ = 0;
            simd_ = false;
            b_shared_ = (pos_ & 15) != 0;
            v_shared_ = (pos_ & 31) != 0;
            return;
        }
        if (pos_ & 15) store(B + pos_ / 16, bw_, true);
        if (pos_ & 31) store(V + pos_ / 32, vw_, true);
        bw_ = vw_ = 0;
    }

  private:
    static void store(uint32_t *w, uint32_t v, bool shared) {
        if (shared) {
            if (v) __atomic_fetch_or(w, v, __ATOMIC_RELAXED);
        } else {
            *w = v;
        }
    }
    uint32_t *B, *V;
    uint64_t pos_;
    uint32_t bw_ = 0, vw_ = 0;
    bool b_shared_, v_shared_;
    // the SIMD form's state: bases top-aligned in bacc_ (bfill_ < 32 bits pending), validity from bit 0 of vacc_
    bool simd_ = false;
    uint64_t bacc_ = 0, vacc_ = 0, bword_ = 0, vword_ = 0;
    uint32_t bfill_ = 0, vfill_ = 0;
};

void pack_piece(const Piece &pc, uint64_t dst, uint32_t *bases, uint32_t *valid) {
    if (pc.take == 0) return;
    StreamPacker sp(bases, valid, dst);
    uint64_t left = pc.take;
    if (!pc.lines) {
        sp.push(pc.src, (size_t)left);
    } else if (pc.py) {
        const uint8_t *p = pc.src;
        while (p < pc.end && left) {
            const uint8_t *e = find_eol_py(p, pc.end);
            const uint8_t *a = p, *b = e;
            strip_py(a, b);
            size_t ln = (size_t)(b - a);
            if (ln > left) ln = (size_t)left;
            sp.push(a, ln);
            left -= ln;
            p = (e < pc.end) ? e + 1 : pc.end;
        }
    } else {
        const uint8_t *p = pc.src;
        while (p < pc.end && left) {
            const uint8_t *e = find_nl(p, pc.end);
            size_t ln = (size_t)(e - p);
            while (ln > 0 && p[ln - 1] == ' ') --ln;
            if (ln > left) ln = (size_t)left;
            sp.push(p, ln);
            left -= ln;
            p = (e < pc.end) ? e + 1 : pc.end;
        }
    }
    sp.finish();
}

// ---- FASTQ(.gz): restates get_sequence_fastq_gz (reference src/libkmers.cpp:240-300) as a line state machine.
// Lines 1, 5, 9, ... (0-based) of the inflated stream, '\n'-terminated ones only, joined with nothing in between.
// The reference reads through an 8192-byte window and returns an empty sequence as soon as a line does not fit
// in it, i.e. when any line (terminated or not) is 8192 bytes or longer; so does this.
// The line state machine, fed piece by piece with the inflated stream.
struct FastqLines {
    std::vector<uint8_t> &seq;
    uint64_t line_no = 0;
    size_t line_len = 0;    // bytes of the current line seen so far
    size_t committed = 0;   // seq.size() at the start of the current (uncommitted) sequence line
    bool bad = false;
    explicit FastqLines(std::vector<uint8_t> &s) : seq(s) {}
    void feed(const uint8_t *p, size_t n) {
        const uint8_t *end = p + n;
        while (!bad && p < end) {
            const uint8_t *nl = (const uint8_t *)memchr(p, '\n', (size_t)(end - p));
            const uint8_t *e = nl ? nl : end;
            const size_t m = (size_t)(e - p);
            if (line_len + m >= 8192) {
                bad = true;
                return;
            }
            if ((line_no & 3) == 1) seq.insert(seq.end(), p, e);
            line_len += m;
            if (!nl) return;
            if ((line_no & 3) == 1) committed = seq.size();  // line complete
            ++line_no;
            line_len = 0;
            p = nl + 1;
        }
    }
};

// One gzip member inflated from `src` (zlib, gzip wrapper, CRC checked): true iff it ended cleanly; *used = its bytes.
bool inflate_member(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t *used) {
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return false;
    out.clear();
    out.resize(std::max<size_t>(1u << 16, std::min<size_t>(n * 4, (size_t)1 << 28)));
    size_t in_at = 0, out_at = 0;
    int rc = Z_OK;
    while (rc != Z_STREAM_END) {
        if (zs.avail_in == 0) {
            const size_t take = std::min<size_t>(n - in_at, (size_t)1 << 30);
            if (take == 0) break;  // the input ran out inside the member
            zs.next_in = const_cast<Bytef *>(src + in_at);
            zs.avail_in = (uInt)take;
            in_at += take;
        }
        if (out_at == out.size()) out.resize(out.size() * 2);
        const size_t room = std::min<size_t>(out.size() - out_at, (size_t)1 << 30);
        zs.next_out = out.data() + out_at;
        zs.avail_out = (uInt)room;
        rc = inflate(&zs, Z_NO_FLUSH);
        out_at += room - zs.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END && !(rc == Z_BUF_ERROR && (zs.avail_in == 0 || zs.avail_out == 0))) break;
    }
    const bool ok = rc == Z_STREAM_END;
    *used = in_at - zs.avail_in;
    inflateEnd(&zs);
    out.resize(ok ? out_at : 0);
    return ok;
}

// Multi-member gzip (bgzip output, or .gz files laid end to end: what sequencers and `cat` produce) on all host threads.
// gzip has no index, so members are found by SPECULATION: every offset that looks like a member header (1f 8b 08, reserved
// flag bits zero) is inflated as if a member started there, one candidate per task; then the chain is followed from offset
// 0 -- a member's end must be the next member's start -- and only candidates on the chain count (a look-alike inside
// compressed data fails or is never reached).  Returns false -- and the caller takes the serial gzread path, whose exact
// behaviour on damaged input is the reference's -- unless the chain covers the file exactly with every member intact.
bool inflate_members_parallel(const uint8_t *p, size_t n, std::vector<std::vector<uint8_t>> &parts) {
    std::vector<size_t> cand;
    for (size_t i = 0; i + 10 <= n;) {
        const uint8_t *q = (const uint8_t *)memchr(p + i, 0x1f, n - 10 - i + 1);
        if (!q) break;
        i = (size_t)(q - p);
        if (p[i + 1] == 0x8b && p[i + 2] == 8 && (p[i + 3] & 0xe0) == 0) cand.push_back(i);
        ++i;
    }
    if (cand.size() < 2 || cand[0] != 0 || cand.size() > ((size_t)1 << 20)) return false;
    struct Try {
        std::vector<uint8_t> out;
        size_t used = 0;
        bool ok = false, done = false;
    };
    std::vector<Try> tries(cand.size());
    // candidates are cheap to reject (a look-alike fails within a few bytes), real members cost their size
    parallel_for(cand.size(), 1, [&](size_t lo, size_t hi) {
        for (size_t c = lo; c < hi; ++c) {
            tries[c].ok = inflate_member(p + cand[c], n - cand[c], tries[c].out, &tries[c].used);
            tries[c].done = true;
        }
    });
    parts.clear();
    size_t at = 0;
    while (at < n) {
        const auto it = std::lower_bound(cand.begin(), cand.end(), at);
        if (it == cand.end() || *it != at) return false;  // the stream does not go on with a member here
        Try &t = tries[(size_t)(it - cand.begin())];
        if (!t.ok || t.used == 0) return false;
        parts.push_back(std::move(t.out));
        at += t.used;
    }
    return at == n;
}

bool read_fastq_sequence(const char *path, std::vector<uint8_t> &seq, bool members_in_parallel) {
    seq.clear();
    if (members_in_parallel) {  // (fewer files than threads: the threads go to the members of one file)
        FileView fv;
        std::vector<std::vector<uint8_t>> parts;
        if (fv.open(path) && fv.n >= 20 && fv.p[0] == 0x1f && fv.p[1] == 0x8b && inflate_members_parallel(fv.p, fv.n, parts)) {
            FastqLines fl(seq);
            for (const auto &part : parts) fl.feed(part.data(), part.size());
            if (fl.bad) {
                seq.clear();
                return false;
            }
            seq.resize(fl.committed);  // drop an unterminated tail
            return true;
        }
        seq.clear();
    }
    gzFile in = gzopen(path, "rb");
    if (!in) {
        fprintf(stderr, "Invalid file: %s\n", path);
        return false;
    }
    gzbuffer(in, 1u << 20);
    std::vector<uint8_t> buf(1u << 20);
    FastqLines fl(seq);
    int got;
    while (!fl.bad && (got = gzread(in, buf.data(), (unsigned)buf.size())) > 0) fl.feed(buf.data(), (size_t)got);
    bool bad = fl.bad;
    if (!bad && got < 0) {
        int err;
        fprintf(stderr, "%s\n", gzerror(in, &err));
        bad = true;
    }
    if (gzclose(in) != Z_OK) {
        fprintf(stderr, "failed gzclose\n");
        bad = true;
    }
    if (bad) {
        seq.clear();
        return false;
    }
    seq.resize(fl.committed);  // drop an unterminated tail
    return true;
}

// Pinned store arrays are handed back to a small cache instead of the driver: a fresh hipHostMalloc of config 2's 150 MB
// costs ~6 ms per file-level call, most of it the first touch of its pages (whichever stage writes them first pays).
// At most 4 blocks / 1 GiB are kept; a block serves requests down to half its size.
struct PinCache {
    struct Block { void *p; size_t bytes; };
    std::mutex mu;
    std::vector<Block> free_blocks;
    std::vector<Block> live;  // sizes of the pinned blocks handed out (free_words gets the pointer only)
    static constexpr size_t MAX_BLOCKS = 4, MAX_BYTES = (size_t)1 << 30;
    void *take(size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        size_t best = free_blocks.size();
        for (size_t i = 0; i < free_blocks.size(); ++i)
            if (free_blocks[i].bytes >= bytes && free_blocks[i].bytes / 2 <= bytes &&
                (best == free_blocks.size() || free_blocks[i].bytes < free_blocks[best].bytes))
                best = i;
        if (best == free_blocks.size()) return nullptr;
        const Block b = free_blocks[best];
        free_blocks.erase(free_blocks.begin() + (long)best);
        live.push_back(b);
        return b.p;
    }
    void note(void *p, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        live.push_back(Block{p, bytes});
    }
    bool give(void *p) {  // true: kept; false: the caller frees it
        std::lock_guard<std::mutex> lk(mu);
        size_t at = live.size();
        for (size_t i = 0; i < live.size(); ++i)
            if (live[i].p == p) at = i;
        if (at == live.size()) return false;
        const Block b = live[at];
        live.erase(live.begin() + (long)at);
        size_t total = b.bytes;
        for (const Block &f : free_blocks) total += f.bytes;
        if (free_blocks.size() >= MAX_BLOCKS || total > MAX_BYTES) return false;
        free_blocks.push_back(b);
        return true;
    }
};
PinCache g_pin_cache;

void *alloc_words(size_t words, bool *pinned) {
    void *p = nullptr;
    *pinned = false;
    const size_t bytes = words * sizeof(uint32_t);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
        if ((p = g_pin_cache.take(bytes)) != nullptr) {
            *pinned = true;
            return p;
        }
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess && p) {
            *pinned = true;
            g_pin_cache.note(p, bytes);
            return p;
        }
        (void)hipGetLastError();
        p = nullptr;
    } else {
        (void)hipGetLastError();
    }
    if (posix_memalign(&p, 4096, bytes ? bytes : 4096) != 0) return nullptr;
    return p;
}

void free_words(void *p, bool pinned) {
    if (!p) return;
    if (pinned) {
        if (!g_pin_cache.give(p)) (void)hipHostFree(p);
    } else {
        free(p);
    }
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// by_bytes: the shard is a byte range of the files laid end to end (only its records are looked at; numbering is local)
int ingest(int n_files, const char *fnames[], int mode, int L, int shard, int n_shards, bool by_bytes, kmers_store **out) {
    if (!out) return KMERS_ERR_BAD_ARG;
    *out = nullptr;
    if (n_files < 0 || (n_files > 0 && !fnames) || mode < KMERS_MODE_X || mode > KMERS_MODE_FASTA_PY || n_shards < 1 ||
        shard < 0 || shard >= n_shards)
        return KMERS_ERR_BAD_ARG;

    struct Lap {  // KMERS_VERBOSE=2: stage times on stderr
        bool on;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        Lap() : on(kk::knobs().verbose >= 2) {}
        void operator()(const char *what) {
            if (!on) return;
            const auto t1 = std::chrono::steady_clock::now();
            fprintf(stderr, "libkmers:   ingest %-20s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
            t0 = t1;
        }
    } lap;
    const bool fastq = (mode == KMERS_MODE_Y_FASTQ);
    const bool cat = (mode == KMERS_MODE_X_CAT);
    const bool xmode = (mode == KMERS_MODE_X || cat);
    const bool py = (mode == KMERS_MODE_FASTA_PY);
    if (by_bytes && (fastq || cat)) return KMERS_ERR_BAD_ARG;      // one record per file: shard those by records
    const bool l_skips_all = xmode && L < 0;  // unsigned comparison in the reference

    // Everything that must outlive parsing: the mapped files and the inflated FASTQ sequences.
    std::vector<std::unique_ptr<FileView>> files((size_t)n_files);
    std::vector<std::vector<uint8_t>> fq((size_t)(fastq ? n_files : 0));
    RawVec<uint64_t> rec_len;  // global (pre-shard) record lengths
    RawVec<Piece> pieces;      // in global record order (records without bases have no piece, or one that takes 0)

    if (fastq) {
        // files side by side when there are enough of them to keep the threads busy, else file after file with the
        // members of each inflated side by side (reference src/libkmers.cpp:240-300 is one serial gzread loop per file)
        if (n_files >= kmers::host_threads() || kmers::host_threads() == 1) {
            parallel_for((size_t)n_files, 1, [&](size_t lo, size_t hi) {
                for (size_t f = lo; f < hi; ++f) read_fastq_sequence(fnames[f], fq[f], false);
            });
        } else {
            for (int f = 0; f < n_files; ++f) read_fastq_sequence(fnames[f], fq[(size_t)f], true);
        }
        for (int f = 0; f < n_files; ++f) {
            if (!pieces.push(Piece{fq[f].data(), fq[f].data() + fq[f].size(), (uint64_t)f, 0, fq[f].size(), false}) ||
                !rec_len.push(fq[f].size()))
                return KMERS_ERR_NOMEM;
        }
    } else {
        std::vector<size_t> hdrs;
        RawVec<RecInfo> recs;
        // by_bytes: the files laid end to end, this shard's range of that
        std::vector<uint64_t> file_base((size_t)n_files + 1, 0);
        if (by_bytes) {
            for (int f = 0; f < n_files; ++f) {
                struct stat sb;
                if (stat(fnames[f], &sb) != 0 || !S_ISREG(sb.st_mode)) return KMERS_ERR_OPEN;
                file_base[f + 1] = file_base[f] + (uint64_t)sb.st_size;
            }
        }
        const unsigned __int128 bytes_total = file_base[n_files];
        const uint64_t want_lo = by_bytes ? (uint64_t)(bytes_total * (unsigned)shard / (unsigned)n_shards) : 0;
        const uint64_t want_hi = by_bytes ? (uint64_t)(bytes_total * (unsigned)(shard + 1) / (unsigned)n_shards) : UINT64_MAX;
        for (int f = 0; f < n_files; ++f) {
            files[f].reset(new FileView());
            FileView &fv = *files[f];
            if (by_bytes && (file_base[f] >= want_hi || file_base[f + 1] <= want_lo)) continue;   // not this shard's: not mapped
            if (!fv.open(fnames[f])) {
                if (cat) {  // reference src/libkmers.cpp:402-403: the job just returns, the column stays empty
                    if (!rec_len.push(0)) return KMERS_ERR_NOMEM;
                    continue;
                }
                return KMERS_ERR_OPEN;  // reference src/libkmers.cpp:356-357, 440-441
            }
            if (kk::knobs().verbose) printf("processing %s\n", fnames[f]);
            // this shard's bytes of the file: headers inside [lo, hi) are its records
            const uint64_t fbase = file_base[f];
            const size_t lo = (size_t)(want_lo > fbase ? want_lo - fbase : 0);
            const size_t hi = (size_t)std::min<uint64_t>(fv.n, want_hi > fbase ? want_hi - fbase : 0);
            const size_t tail_end = (hi >= fv.n) ? fv.n : next_header(fv, hi, py);
            if (py) {
                if (!find_headers_py(fv, lo, hi, hdrs)) {
                    fprintf(stderr, "libkmers: %s holds bytes >= 0x80; the kmerexpr/fasta.py rules are restated for ASCII files only\n", fnames[f]);
                    return KMERS_ERR_BAD_ARG;
                }
            } else {
                find_headers(fv, lo, hi, hdrs);
            }
            lap("find headers");
            if (!(py ? analyse_records_py(fv, hdrs, tail_end, recs) : analyse_records(fv, hdrs, tail_end, xmode, recs)))
                return KMERS_ERR_NOMEM;
            lap("analyse records");
            if (cat) {
                uint64_t full = 0, nul = UINT64_MAX;
                for (const RecInfo &r : recs) {
                    if (r.predicted) continue;
                    if (nul == UINT64_MAX && r.nul_at != UINT64_MAX) nul = full + r.nul_at;
                    full += r.full_len;
                }
                uint64_t eff = (l_skips_all || full < (uint64_t)(L < 0 ? 0 : L)) ? 0 : std::min(full, nul);
                const uint64_t rec = rec_len.size();
                uint64_t off = 0;
                for (const RecInfo &r : recs) {
                    if (r.predicted || off >= eff) continue;
                    const uint64_t take = std::min<uint64_t>(r.full_len, eff - off);
                    if (take && !pieces.push(Piece{fv.p + r.seq_off, fv.p + r.end_off, rec, off, take, true})) return KMERS_ERR_NOMEM;
                    off += r.full_len;
                }
                if (!rec_len.push(eff)) return KMERS_ERR_NOMEM;
            } else {
                // kept records get consecutive numbers: count, prefix, fill -- all in parallel
                auto kept = [&](size_t i) -> uint64_t {
                    const RecInfo &r = recs[i];
                    if (r.predicted) return 0;
                    if (xmode && (l_skips_all || r.full_len < (uint64_t)L)) return 0;
                    if (py && L > 0 && r.full_len < (uint64_t)L) return 0;   // len(seq) >= L, signed (simulate_reads.py:88)
                    return 1;
                };
                std::unique_ptr<uint64_t[]> slot(new (std::nothrow) uint64_t[recs.size() + 1]);
                if (!slot) return KMERS_ERR_NOMEM;
                const uint64_t n_kept = parallel_prefix(recs.size(), slot.get(), kept);
                const uint64_t rec0 = rec_len.size();
                uint64_t *rl = rec_len.grow((size_t)n_kept);
                Piece *pc = pieces.grow((size_t)n_kept);
                if (n_kept && (!rl || !pc)) return KMERS_ERR_NOMEM;
                parallel_for(recs.size(), (size_t)1 << 14, [&](size_t lo, size_t hi) {
                    for (size_t i = lo; i < hi; ++i) {
                        if (!kept(i)) continue;
                        const RecInfo &r = recs[i];
                        const uint64_t len = xmode ? std::min(r.full_len, r.nul_at) : r.full_len;
                        const uint64_t k = slot[i];
                        rl[k] = len;
                        pc[k] = Piece{fv.p + r.seq_off, fv.p + r.end_off, rec0 + k, 0, len, true, py};
                    }
                });
            }
        }
    }

    lap("pieces");
    // ---- shard selection: contiguous record ranges with about equal numbers of positions
    const uint64_t n_total = rec_len.size();
    std::unique_ptr<uint64_t[]> gstart(new (std::nothrow) uint64_t[n_total + 1]);  // global start position of every record
    if (!gstart) return KMERS_ERR_NOMEM;
    const uint64_t pos_total = parallel_prefix((size_t)n_total, gstart.get(), [&](size_t r) { return rec_len[r] + 1; });
    gstart[n_total] = pos_total;
    uint64_t r_lo = 0, r_hi = n_total;
    if (n_shards > 1 && !by_bytes) {
        // record r belongs to shard floor(start_pos(r) * n_shards / pos_total)
        auto bound = [&](int s) -> uint64_t {  // first record of shard s
            if (s <= 0) return 0;
            if (s >= n_shards) return n_total;
            const long double target = (long double)pos_total * s / n_shards;
            // first record whose start position is >= target
            uint64_t lo = 0, hi = n_total;
            while (lo < hi) {
                const uint64_t mid = lo + (hi - lo) / 2;
                if ((long double)gstart[mid] >= target)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            return lo;
        };
        r_lo = bound(shard);
        r_hi = bound(shard + 1);
    }

    std::unique_ptr<kmers_store> st(new (std::nothrow) kmers_store());
    if (!st) return KMERS_ERR_NOMEM;
    st->n_rec_total = n_total;
    st->first_rec = r_lo;
    st->rec_start.resize((size_t)(r_hi - r_lo) + 1);
    {
        const uint64_t base = gstart[r_lo];
        uint64_t *rs = st->rec_start.data();
        parallel_for((size_t)(r_hi - r_lo) + 1, (size_t)1 << 16, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) rs[i] = gstart[r_lo + i] - base;
        });
    }
    const uint64_t n_pos = st->rec_start.back();
    st->bases_words = (size_t)((n_pos + 15) / 16) + 4;
    st->valid_words = (size_t)((n_pos + 31) / 32) + 4;
    bool pin_b = false, pin_v = false;
    st->bases = (uint32_t *)alloc_words(st->bases_words, &pin_b);
    st->valid = (uint32_t *)alloc_words(st->valid_words, &pin_v);
    if (!st->bases || !st->valid || pin_b != pin_v) {
        free_words(st->bases, pin_b);
        free_words(st->valid, pin_v);
        st->bases = st->valid = nullptr;
        return KMERS_ERR_NOMEM;
    }
    st->pinned = pin_b;
    lap("record table + alloc");

    // pieces of this shard: records are ascending in `pieces`, find the range by binary search
    const Piece *plo = std::lower_bound(pieces.p, pieces.p + pieces.n, r_lo, [](const Piece &p, uint64_t r) { return p.rec < r; });
    const Piece *phi = std::lower_bound(pieces.p, pieces.p + pieces.n, r_hi, [](const Piece &p, uint64_t r) { return p.rec < r; });
    const size_t np = (size_t)(phi - plo);
    const Piece *pp = np ? plo : nullptr;
    auto piece_dst = [&](size_t i) { return st->rec_start[(size_t)(pp[i].rec - r_lo)] + pp[i].off; };
    // Zero only what the packers do not overwrite: a piece writes the words it covers completely with plain stores and
    // ORs into the first and the last word it touches, so the words from the end of one piece to the start of the next
    // (its own last word, separators, empty records, the next one's first word), the words before the first piece and
    // everything behind the last one must start as zero -- a few words per piece instead of the whole arrays (7 of the
    // 22 ms of config 2's ingest went into the memset).
    {
        uint32_t *const B = st->bases, *const V = st->valid;
        const size_t bw = st->bases_words, vw = st->valid_words;
        if (g_poison) {  // tests: whatever is not written or zeroed below shows up
            memset(B, 0xA5, bw * 4);
            memset(V, 0xA5, vw * 4);
        }
        parallel_for(np + 1, 4096, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                const uint64_t a = i ? piece_dst(i - 1) + pp[i - 1].take : 0;  // first position piece i - 1 does not write
                size_t b0 = (size_t)(a / 16), v0 = (size_t)(a / 32), b1, v1;  // words [b0, b1), [v0, v1)
                if (i == np) {
                    b1 = bw, v1 = vw;
                } else {
                    const uint64_t b = piece_dst(i);
                    b1 = std::min(bw, (size_t)(b / 16) + 1), v1 = std::min(vw, (size_t)(b / 32) + 1);
                }
                if (b0 < b1) memset(B + b0, 0, (b1 - b0) * 4);
                if (v0 < v1) memset(V + v0, 0, (v1 - v0) * 4);
            }
        });
    }
    lap("zero");
    // grain: aim at ~1 MiB of bases per chunk for typical transcript lengths, 1 piece minimum
    size_t grain = 1;
    if (np) {
        const uint64_t avg = std::max<uint64_t>(1, (n_pos) / np);
        grain = (size_t)std::max<uint64_t>(1, (1u << 20) / avg);
    }
    parallel_for(np, grain, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            pack_piece(pp[i], piece_dst(i), st->bases, st->valid);
        }
    });

    lap("pack");
    *out = st.release();
    return KMERS_OK;
}

}  // namespace

extern "C" {

int kmers_store_ingest(int n_files, const char *fnames[], int mode, int L, int shard, int n_shards, kmers_store **out) {
    return ingest(n_files, fnames, mode, L, shard, n_shards, false, out);
}

int kmers_store_ingest_bytes(int n_files, const char *fnames[], int mode, int L, int shard, int n_shards, kmers_store **out) {
    return ingest(n_files, fnames, mode, L, shard, n_shards, true, out);
}

void kmers_store_set_numbering(kmers_store *s, uint64_t first_rec, uint64_t n_rec_total) {
    s->first_rec = first_rec;
    s->n_rec_total = n_rec_total;
}

void kmers_store_free(kmers_store *s) {
    if (!s) return;
    free_words(s->bases, s->pinned);
    free_words(s->valid, s->pinned);
    delete s;
}

uint64_t kmers_store_n_pos(const kmers_store *s) { return s->rec_start.back(); }
uint64_t kmers_store_n_rec(const kmers_store *s) { return s->rec_start.size() - 1; }
uint64_t kmers_store_n_rec_total(const kmers_store *s) { return s->n_rec_total; }
uint64_t kmers_store_first_rec(const kmers_store *s) { return s->first_rec; }
uint64_t kmers_store_n_bases(const kmers_store *s) { return s->rec_start.back() - (s->rec_start.size() - 1); }
uint64_t kmers_store_n_windows(const kmers_store *s, int K) {
    uint64_t w = 0;
    for (size_t r = 0; r + 1 < s->rec_start.size(); ++r) {
        const uint64_t len = s->rec_start[r + 1] - s->rec_start[r] - 1;
        if (K >= 1 && len >= (uint64_t)K) w += len - (uint64_t)K + 1;
    }
    return w;
}
const uint32_t *kmers_store_bases(const kmers_store *s) { return s->bases; }
const uint32_t *kmers_store_valid(const kmers_store *s) { return s->valid; }
const uint64_t *kmers_store_rec_start(const kmers_store *s) { return s->rec_start.data(); }
size_t kmers_store_bases_words(const kmers_store *s) { return s->bases_words; }
size_t kmers_store_valid_words(const kmers_store *s) { return s->valid_words; }

}  // extern "C"
