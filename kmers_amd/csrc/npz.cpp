// npz.cpp -- write X as the uncompressed scipy .npz the reference's transcriptome_to_x saves
// (kmerexpr/transcriptome_reader.py:39: scipy.sparse.save_npz(x_file, xt, compressed=False)), straight from the
// ABI's arrays: SURVEY.md section 8(f) row 2.  A scipy CSR .npz is a stored ZIP of five .npy members, in this order:
//   indices.npy  int32 or int64 [nnz]       (scipy narrows the index arrays to int32 when everything fits)
//   indptr.npy   same dtype     [n_rows+1]
//   format.npy   |S3, 0-d       b"csr"
//   shape.npy    int64 [2]
//   data.npy     float32 [nnz]
// The uint64 index arrays of the ABI are narrowed chunk by chunk on the way out (no full-size copy), member CRCs
// are computed in parallel (zlib crc32 + crc32_combine) and patched into the local headers afterwards.
// ZIP64 structures are always written (numpy does the same for the local headers), so members over 4 GiB need no
// special case.  Host-only code.
#include "kmers_hip.h"
#include "store.hpp"

#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Member {
    std::string name;
    uint64_t offset = 0;  // of the local header
    uint64_t size = 0;    // of the member's bytes (.npy header included)
    uint32_t crc = 0;
    long crc_at = 0;      // file position of the local header's crc field
};

void put16(std::vector<unsigned char> &b, uint32_t v) {
    b.push_back(v & 255);
    b.push_back((v >> 8) & 255);
}
void put32(std::vector<unsigned char> &b, uint32_t v) {
    put16(b, v & 0xFFFF);
    put16(b, v >> 16);
}
void put64(std::vector<unsigned char> &b, uint64_t v) {
    put32(b, (uint32_t)v);
    put32(b, (uint32_t)(v >> 32));
}

// .npy version 1.0 header, padded with blanks so that the data starts at a multiple of 64 bytes
std::string npy_header(const char *descr, const std::string &shape) {
    std::string dict = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': " + shape + ", }";
    size_t total = 10 + dict.size() + 1;
    size_t pad = (64 - total % 64) % 64;
    dict.append(pad, ' ');
    dict.push_back('\n');
    std::string h("\x93NUMPY\x01\x00", 8);
    h.push_back((char)(dict.size() & 255));
    h.push_back((char)(dict.size() >> 8));
    return h + dict;
}

uint32_t crc_parallel(const unsigned char *p, size_t n) {
    if (n == 0) return (uint32_t)crc32(0L, Z_NULL, 0);
    const size_t grain = (size_t)8 << 20;
    const size_t parts = (n + grain - 1) / grain;
    std::vector<uint32_t> crcs(parts);
    kmers::parallel_for(parts, 1, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) {
            const size_t a = i * grain, b = std::min(n, a + grain);
            uLong c = crc32(0L, Z_NULL, 0);
            for (size_t o = a; o < b;) {  // crc32 takes a uInt length
                const size_t m = std::min<size_t>(b - o, (size_t)1 << 30);
                c = crc32(c, p + o, (uInt)m);
                o += m;
            }
            crcs[i] = (uint32_t)c;
        }
    });
    uLong c = crcs[0];
    for (size_t i = 1; i < parts; ++i) {
        const size_t a = i * grain, b = std::min(n, a + grain);
        c = crc32_combine(c, crcs[i], (z_off_t)(b - a));
    }
    return (uint32_t)c;
}

struct Writer {
    FILE *f = nullptr;
    std::vector<Member> members;
    bool ok = true;

    void write(const void *p, size_t n) {
        if (ok && n && fwrite(p, 1, n, f) != n) ok = false;
    }
    // local header with a zero CRC (patched by end_member) and the sizes in the ZIP64 extra field
    void begin_member(const std::string &name, uint64_t size) {
        Member m;
        m.name = name;
        m.offset = (uint64_t)ftello(f);
        m.size = size;
        std::vector<unsigned char> b;
        put32(b, 0x04034b50);
        put16(b, 45);  // version needed: ZIP64
        put16(b, 0);   // flags
        put16(b, 0);   // stored
        put16(b, 0);   // time
        put16(b, 0x21);  // date: 1980-01-01, what numpy writes too
        m.crc_at = (long)b.size();
        put32(b, 0);   // crc, patched later
        put32(b, 0xFFFFFFFFu);
        put32(b, 0xFFFFFFFFu);
        put16(b, (uint32_t)name.size());
        put16(b, 20);
        b.insert(b.end(), name.begin(), name.end());
        put16(b, 0x0001);
        put16(b, 16);
        put64(b, size);
        put64(b, size);
        m.crc_at += (long)m.offset;
        write(b.data(), b.size());
        members.push_back(m);
    }
    void end_member(uint32_t crc) { members.back().crc = crc; }

    void finish() {
        const uint64_t cd_start = (uint64_t)ftello(f);
        std::vector<unsigned char> b;
        for (const Member &m : members) {
            put32(b, 0x02014b50);
            put16(b, 45 | (3 << 8));  // made by: unix
            put16(b, 45);
            put16(b, 0);
            put16(b, 0);
            put16(b, 0);
            put16(b, 0x21);
            put32(b, m.crc);
            put32(b, 0xFFFFFFFFu);
            put32(b, 0xFFFFFFFFu);
            put16(b, (uint32_t)m.name.size());
            put16(b, 28);  // extra
            put16(b, 0);   // comment
            put16(b, 0);   // disk
            put16(b, 0);   // internal attributes
            put32(b, 0600u << 16);
            put32(b, 0xFFFFFFFFu);  // local header offset: in the extra field
            b.insert(b.end(), m.name.begin(), m.name.end());
            put16(b, 0x0001);
            put16(b, 24);
            put64(b, m.size);
            put64(b, m.size);
            put64(b, m.offset);
        }
        const uint64_t cd_size = b.size();
        // ZIP64 end of central directory record + locator, then the classic record pointing at them
        put32(b, 0x06064b50);
        put64(b, 44);
        put16(b, 45);
        put16(b, 45);
        put32(b, 0);
        put32(b, 0);
        put64(b, members.size());
        put64(b, members.size());
        put64(b, cd_size);
        put64(b, cd_start);
        put32(b, 0x07064b50);
        put32(b, 0);
        put64(b, cd_start + cd_size);
        put32(b, 1);
        put32(b, 0x06054b50);
        put16(b, 0);
        put16(b, 0);
        put16(b, 0xFFFF);
        put16(b, 0xFFFF);
        put32(b, 0xFFFFFFFFu);
        put32(b, 0xFFFFFFFFu);
        put16(b, 0);
        write(b.data(), b.size());
        for (const Member &m : members) {  // patch the CRCs of the local headers
            if (!ok) break;
            if (fseeko(f, (off_t)m.crc_at, SEEK_SET) != 0) {
                ok = false;
                break;
            }
            unsigned char c[4] = {(unsigned char)(m.crc & 255), (unsigned char)((m.crc >> 8) & 255),
                                  (unsigned char)((m.crc >> 16) & 255), (unsigned char)(m.crc >> 24)};
            write(c, 4);
        }
    }
};

// One member holding an array: header, then the bytes produced chunk by chunk by fill(dst, first element, count).
template <class Fill>
void array_member(Writer &w, const char *name, const char *descr, uint64_t n, size_t elem, const Fill &fill) {
    const std::string h = npy_header(descr, "(" + std::to_string(n) + ",)");
    w.begin_member(name, h.size() + n * elem);
    uLong crc = crc32(0L, Z_NULL, 0);
    crc = crc32(crc, (const Bytef *)h.data(), (uInt)h.size());
    w.write(h.data(), h.size());
    const uint64_t chunk = ((uint64_t)64 << 20) / elem;
    std::vector<unsigned char> buf((size_t)std::min<uint64_t>(n, chunk) * elem);
    for (uint64_t at = 0; at < n && w.ok; at += chunk) {
        const uint64_t m = std::min(chunk, n - at);
        const unsigned char *src = fill(buf.data(), at, m);
        const uint32_t c = crc_parallel(src, (size_t)(m * elem));
        crc = crc32_combine(crc, c, (z_off_t)(m * elem));
        w.write(src, (size_t)(m * elem));
    }
    w.end_member((uint32_t)crc);
}

}  // namespace

extern "C" int kmers_npz_write_csr(const char *path, const float *data, const uint64_t *col_ind, const uint64_t *row_ind,
                                   uint64_t nnz, uint64_t n_rows, uint64_t n_cols) {
    if (!path || !row_ind || (nnz && (!data || !col_ind))) return KMERS_ERR_BAD_ARG;
    Writer w;
    w.f = fopen(path, "wb");
    if (!w.f) return KMERS_ERR_OPEN;
    // scipy's rule (get_index_dtype with check_contents): int32 when the shape and every index fit
    const bool wide = std::max(std::max(n_rows, n_cols), nnz) > 0x7FFFFFFFull;
    const char *idx = wide ? "<i8" : "<i4";
    auto narrow = [&](const uint64_t *src) {
        return [src, wide](unsigned char *dst, uint64_t at, uint64_t m) -> const unsigned char * {
            if (wide) return reinterpret_cast<const unsigned char *>(src + at);  // int64 == the uint64 bits (values < 2^63)
            int32_t *d = reinterpret_cast<int32_t *>(dst);
            kmers::parallel_for((size_t)m, (size_t)1 << 20, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) d[i] = (int32_t)src[at + i];
            });
            return dst;
        };
    };
    array_member(w, "indices.npy", idx, nnz, wide ? 8 : 4, narrow(col_ind));
    array_member(w, "indptr.npy", idx, n_rows + 1, wide ? 8 : 4, narrow(row_ind));
    {
        const std::string h = npy_header("|S3", "()") + "csr";
        w.begin_member("format.npy", h.size());
        w.write(h.data(), h.size());
        w.end_member((uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)h.data(), (uInt)h.size()));
    }
    {
        const int64_t shape[2] = {(int64_t)n_rows, (int64_t)n_cols};
        array_member(w, "shape.npy", "<i8", 2, 8, [&](unsigned char *, uint64_t, uint64_t) {
            return reinterpret_cast<const unsigned char *>(shape);
        });
    }
    array_member(w, "data.npy", "<f4", nnz, 4, [&](unsigned char *, uint64_t at, uint64_t) {
        return reinterpret_cast<const unsigned char *>(data + at);
    });
    w.finish();
    if (fclose(w.f) != 0) w.ok = false;
    if (!w.ok) {
        remove(path);
        return KMERS_ERR_OPEN;
    }
    return KMERS_OK;
}

extern "C" int kmers_npz_write_csr_i32(const char *path, const float *data, const int32_t *col_ind, const int32_t *row_ind,
                                       uint64_t nnz, uint64_t n_rows, uint64_t n_cols) {
    if (!path || !row_ind || (nnz && (!data || !col_ind))) return KMERS_ERR_BAD_ARG;
    if (std::max(std::max(n_rows, n_cols), nnz) > 0x7FFFFFFFull) return KMERS_ERR_TOO_LARGE;  // scipy would store int64
    Writer w;
    w.f = fopen(path, "wb");
    if (!w.f) return KMERS_ERR_OPEN;
    auto as_is = [](const void *src, size_t elem) {
        return [src, elem](unsigned char *, uint64_t at, uint64_t) { return reinterpret_cast<const unsigned char *>(src) + at * elem; };
    };
    array_member(w, "indices.npy", "<i4", nnz, 4, as_is(col_ind, 4));
    array_member(w, "indptr.npy", "<i4", n_rows + 1, 4, as_is(row_ind, 4));
    {
        const std::string h = npy_header("|S3", "()") + "csr";
        w.begin_member("format.npy", h.size());
        w.write(h.data(), h.size());
        w.end_member((uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)h.data(), (uInt)h.size()));
    }
    {
        const int64_t shape[2] = {(int64_t)n_rows, (int64_t)n_cols};
        array_member(w, "shape.npy", "<i8", 2, 8, [&](unsigned char *, uint64_t, uint64_t) {
            return reinterpret_cast<const unsigned char *>(shape);
        });
    }
    array_member(w, "data.npy", "<f4", nnz, 4, as_is(data, 4));
    w.finish();
    if (fclose(w.f) != 0) w.ok = false;
    if (!w.ok) {
        remove(path);
        return KMERS_ERR_OPEN;
    }
    return KMERS_OK;
}
