/*
 * kmers_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * A plain-C, single-threaded CPU restatement of the k-mer shredding / counting hot path of the
 * reference library (bob-carpenter/kmers, src/libkmers.cpp).  It exists so that the HIP path in
 * kmers_amd/csrc can be checked bit-for-bit on arbitrary seeded inputs, on machines where the
 * reference sources are not present (the GPU box).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product library never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against
 *   (a) the known answers of the reference's own tests (test/test_transcriptome_reader.py:12-49), and
 *   (b) outputs of the unmodified reference library compiled in the build container
 *       (oracle/Makefile target `ref`, fixtures written by oracle/make_golden.py into tests/golden/).
 *
 * What is restated, and where it lives in the reference:
 *   base code table A,C,G,T -> 0..3, everything else invalid .... libkmers.cpp:199-211
 *   k-mer id = big-endian base-4 number of the K bases ............ libkmers.cpp:331-337
 *   a window is counted iff all K bytes are valid ................. libkmers.cpp:339-345
 *   per-isoform multiset count, value = float(count)/total ........ libkmers.cpp:218-238
 *   FASTA record reader (line rules, trailing-space strip) ........ libkmers.cpp:302-318
 *   column numbering / PREDICTED / L filters ...................... libkmers.cpp:444-457
 *   concatenate-subsequence variant (one column per file) ......... libkmers.cpp:394-430
 *   COO -> CSR, rows ascending, columns ascending inside a row .... libkmers.cpp:77-169
 *   y from FASTA reads (no L filter) .............................. libkmers.cpp:320-328, 347-376
 *   y from FASTQ(.gz): line 2 of every 4, joined without separator  libkmers.cpp:240-300, 378-392
 *
 * The algorithms are deliberately different from the reference's (rolling encoder + sort/run-length
 * instead of per-window re-encoding + unordered_map; counting sort instead of the threaded merge), so
 * an agreement between the two is evidence about the semantics, not about shared code.
 *
 * One intentional divergence: a record shorter than K-1 bases makes the
 * reference read out of bounds (unsigned underflow at libkmers.cpp:221-222, :321).  Here such a record
 * has zero k-mers; on the X path it still occupies a column.
 */
#include <stdint.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ base codes */

static int8_t g_code[256];
static int g_code_ready = 0;

static void code_init(void) {
    if (g_code_ready) return;
    memset(g_code, -1, sizeof g_code);
    g_code['A'] = 0;
    g_code['C'] = 1;
    g_code['G'] = 2;
    g_code['T'] = 3;
    g_code_ready = 1;
}

/* libkmers.cpp:331-337.  For a byte that is not A/C/G/T the reference adds the table's -1 (as int, wrapped
 * to uint32_t); the same is done here for bytes < 0x80.  Bytes >= 0x80 index the reference's table out of
 * bounds (signed char) and have no defined answer.  All internal callers check validity first. */
uint32_t oracle_kmer_to_id(const char *kmer, int len) {
    code_init();
    uint32_t id = 0;
    for (int i = 0; i < len; ++i) id = id * 4u + (uint32_t)(int)g_code[(uint8_t)kmer[i]];
    return id;
}

/* libkmers.cpp:339-345 */
bool oracle_valid_kmer(const char *kmer, int len) {
    code_init();
    for (int i = 0; i < len; ++i)
        if (g_code[(uint8_t)kmer[i]] < 0) return false;
    return true;
}

/* ------------------------------------------------------------------ small containers */

typedef struct {
    char *p;
    size_t n, cap;
} bytes_t;

static void bytes_reserve(bytes_t *b, size_t need) {
    if (need <= b->cap) return;
    size_t c = b->cap ? b->cap : 4096;
    while (c < need) c *= 2;
    b->p = (char *)realloc(b->p, c);
    if (!b->p) abort();
    b->cap = c;
}
static void bytes_append(bytes_t *b, const char *s, size_t n) {
    bytes_reserve(b, b->n + n + 1);
    memcpy(b->p + b->n, s, n);
    b->n += n;
}

typedef struct {
    uint32_t row, col;
    float val;
} coo_t;
typedef struct {
    coo_t *p;
    size_t n, cap;
} coo_list;

static void coo_push(coo_list *l, uint32_t row, uint32_t col, float val) {
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 1024;
        l->p = (coo_t *)realloc(l->p, l->cap * sizeof(coo_t));
        if (!l->p) abort();
    }
    l->p[l->n].row = row;
    l->p[l->n].col = col;
    l->p[l->n].val = val;
    l->n++;
}

/* ------------------------------------------------------------------ window enumeration */

/* Calls emit(ctx, id) once per window of `seq` whose K bytes are all valid, in position order.
 * Rolling form of valid_kmer + kmer_to_id applied at every offset (libkmers.cpp:222-227, 321-326). */
typedef void (*emit_fn)(void *ctx, uint32_t id);

static void shred(const char *seq, size_t len, int K, emit_fn emit, void *ctx) {
    code_init();
    if (K < 1 || K > 16 || len < (size_t)K) return;
    const uint32_t mask = (K == 16) ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
    uint32_t id = 0;
    size_t run = 0; /* number of consecutive valid bytes ending here */
    for (size_t i = 0; i < len; ++i) {
        int c = g_code[(uint8_t)seq[i]];
        if (c < 0) {
            run = 0;
            id = 0;
            continue;
        }
        id = ((id << 2) | (uint32_t)c) & mask;
        if (++run >= (size_t)K) emit(ctx, id);
    }
}

typedef struct {
    uint32_t *p;
    size_t n, cap;
} idlist;
static void idlist_emit(void *ctx, uint32_t id) {
    idlist *l = (idlist *)ctx;
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 1024;
        l->p = (uint32_t *)realloc(l->p, l->cap * sizeof(uint32_t));
        if (!l->p) abort();
    }
    l->p[l->n++] = id;
}
static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* One column of X: distinct ids with float(count)/total (libkmers.cpp:232-238).
 * fill_indices_coo hands `sequence.data()` (a const char*) to collect_kmers(const std::string&), which builds a
 * new string from a C string: on the X path the sequence is therefore cut at its first NUL byte (after the L
 * filter, which saw the full length).  The y path passes the std::string itself and is not cut. */
static void column_of(const char *seq, size_t len, int K, uint32_t col, idlist *scratch, coo_list *out) {
    const char *nul = len ? (const char *)memchr(seq, 0, len) : NULL;
    if (nul) len = (size_t)(nul - seq);
    scratch->n = 0;
    shred(seq, len, K, idlist_emit, scratch);
    const size_t total = scratch->n;
    if (!total) return;
    qsort(scratch->p, total, sizeof(uint32_t), cmp_u32);
    const uint32_t total32 = (uint32_t)total;
    size_t i = 0;
    while (i < total) {
        size_t j = i + 1;
        while (j < total && scratch->p[j] == scratch->p[i]) ++j;
        const uint32_t count = (uint32_t)(j - i);
        /* float(count) / total_count with total_count uint32_t: both operands become float first */
        const float v = (float)count / (float)total32;
        coo_push(out, scratch->p[i], col, v);
        i = j;
    }
}

static void count_emit(void *ctx, uint32_t id) {
    int *counts = (int *)ctx;
    /* libkmers.cpp:213-216 indexes with `int`; ids >= 2^31 (only K=16) are out of the documented range */
    counts[id] += 1;
}

/* ------------------------------------------------------------------ FASTA reader */

static int slurp(const char *path, bytes_t *out) {
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    char buf[1 << 16];
    size_t got;
    out->n = 0;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) bytes_append(out, buf, got);
    fclose(f);
    bytes_reserve(out, out->n + 1);
    return 0;
}

/* Restates get_next_sequence_fasta (libkmers.cpp:302-318) over an in-memory copy of the file:
 *   - the line at the read position is the header, whatever it starts with; an empty header ends the file
 *     (libkmers.cpp:363-364, 447-448);
 *   - following lines up to (not including) the next line whose first byte is '>' are sequence lines;
 *   - only trailing ' ' bytes are stripped from a sequence line ('\r', tabs, interior blanks stay and are
 *     simply invalid bases); blank lines contribute nothing;
 *   - a last line without '\n' is a line like any other.
 * Returns 1 and fills header/sequence when a record was read, 0 at end of file. */
typedef struct {
    const char *buf;
    size_t n, pos;
} fasta_cursor;

static size_t line_end(const fasta_cursor *c, size_t from) {
    const char *nl = (const char *)memchr(c->buf + from, '\n', c->n - from);
    return nl ? (size_t)(nl - c->buf) : c->n;
}

static int next_record(fasta_cursor *c, const char **hdr, size_t *hlen, bytes_t *seq) {
    if (c->pos >= c->n) return 0;
    size_t e = line_end(c, c->pos);
    *hdr = c->buf + c->pos;
    *hlen = e - c->pos;
    if (*hlen == 0) return 0;
    c->pos = (e < c->n) ? e + 1 : c->n;
    seq->n = 0;
    while (c->pos < c->n) {
        e = line_end(c, c->pos);
        const char *l = c->buf + c->pos;
        size_t ln = e - c->pos;
        if (ln > 0 && l[0] == '>') break;
        while (ln > 0 && l[ln - 1] == ' ') --ln;
        bytes_append(seq, l, ln);
        c->pos = (e < c->n) ? e + 1 : c->n;
    }
    return 1;
}

static bool has_predicted(const char *hdr, size_t hlen) {
    static const char tag[] = "PREDICTED";
    const size_t tl = sizeof tag - 1;
    if (hlen < tl) return false;
    for (size_t i = 0; i + tl <= hlen; ++i)
        if (memcmp(hdr + i, tag, tl) == 0) return true;
    return false;
}

/* ------------------------------------------------------------------ COO -> CSR */

/* Rows ascending, columns ascending inside a row (what ThreadPoolLocalData::to_csr produces,
 * libkmers.cpp:77-169).  `coo` is in column-major generation order, so a stable counting sort by row
 * keeps columns ascending. */
static void coo_to_csr(const coo_list *coo, uint64_t M, float *data, uint64_t *row_ind, uint64_t *col_ind) {
    for (uint64_t r = 0; r <= M; ++r) row_ind[r] = 0;
    for (size_t i = 0; i < coo->n; ++i) row_ind[(uint64_t)coo->p[i].row + 1]++;
    for (uint64_t r = 0; r < M; ++r) row_ind[r + 1] += row_ind[r];
    /* fill using row_ind[r] as a cursor, then shift back */
    for (size_t i = 0; i < coo->n; ++i) {
        uint64_t at = row_ind[coo->p[i].row]++;
        col_ind[at] = coo->p[i].col;
        data[at] = coo->p[i].val;
    }
    for (uint64_t r = M; r > 0; --r) row_ind[r] = row_ind[r - 1];
    row_ind[0] = 0;
}

static uint64_t pow4(int K) { return (uint64_t)1 << (2 * K); }

/* ------------------------------------------------------------------ exported: X path */

static int build_x(int n_files, const char *fnames[], int K, int L, int cat, float *data, uint64_t *row_ind,
                   uint64_t *col_ind, uint64_t max_size, uint64_t *nnz, int *n_cols) {
    if (K < 1 || K > 16) return -3;
    coo_list coo = {0, 0, 0};
    idlist scratch = {0, 0, 0};
    bytes_t file = {0, 0, 0}, seq = {0, 0, 0}, joined = {0, 0, 0};
    uint32_t col = 0;
    int rc = 0;
    for (int f = 0; f < n_files; ++f) {
        if (slurp(fnames[f], &file) != 0) {
            if (cat) { /* libkmers.cpp:402-403: silently an empty column */
                ++col;
                continue;
            }
            rc = -1; /* libkmers.cpp:440-441 */
            goto done;
        }
        fasta_cursor cur = {file.p, file.n, 0};
        const char *hdr;
        size_t hlen;
        joined.n = 0;
        while (next_record(&cur, &hdr, &hlen, &seq)) {
            if (has_predicted(hdr, hlen)) continue;
            if (cat) {
                bytes_append(&joined, seq.p, seq.n); /* libkmers.cpp:414 */
                continue;
            }
            /* libkmers.cpp:451: `sequence.length() < L` compares as unsigned, so a negative L skips all */
            if (L < 0 || seq.n < (size_t)L) continue;
            column_of(seq.p, seq.n, K, col, &scratch, &coo);
            ++col;
        }
        if (cat) {
            if (!(L < 0 || joined.n < (size_t)L)) column_of(joined.p, joined.n, K, col, &scratch, &coo);
            ++col; /* libkmers.cpp:428: n_cols = n_files */
        }
    }
    if ((uint64_t)coo.n > max_size) { /* the -2 the reference reserves but never raises (libkmers.py:175) */
        rc = -2;
        goto done;
    }
    coo_to_csr(&coo, pow4(K), data, row_ind, col_ind);
    *nnz = coo.n;
    *n_cols = (int)col;
done:
    free(coo.p);
    free(scratch.p);
    free(file.p);
    free(seq.p);
    free(joined.p);
    return rc;
}

int oracle_fasta_to_kmers_csr(int n_files, const char *fnames[], int K, int L, float *data, uint64_t *row_ind,
                              uint64_t *col_ind, uint64_t max_size, uint64_t *nnz, int *n_cols) {
    return build_x(n_files, fnames, K, L, 0, data, row_ind, col_ind, max_size, nnz, n_cols);
}

int oracle_fasta_to_kmers_csr_cat_subseq(int n_files, const char *fnames[], int K, int L, float *data,
                                         uint64_t *row_ind, uint64_t *col_ind, uint64_t max_size, uint64_t *nnz,
                                         int *n_cols) {
    return build_x(n_files, fnames, K, L, 1, data, row_ind, col_ind, max_size, nnz, n_cols);
}

/* ------------------------------------------------------------------ exported: y path */

/* libkmers.cpp:347-376.  Adds into the caller's (zero-initialised) histogram. */
int oracle_fasta_count_kmers(int n_files, const char *fnames[], int K, int *total_kmer_counts) {
    if (K < 1 || K > 16) return -3;
    bytes_t file = {0, 0, 0}, seq = {0, 0, 0};
    int rc = 0;
    for (int f = 0; f < n_files; ++f) {
        if (slurp(fnames[f], &file) != 0) {
            rc = -1;
            break;
        }
        fasta_cursor cur = {file.p, file.n, 0};
        const char *hdr;
        size_t hlen;
        while (next_record(&cur, &hdr, &hlen, &seq)) {
            if (has_predicted(hdr, hlen)) continue;
            shred(seq.p, seq.n, K, count_emit, total_kmer_counts);
        }
    }
    free(file.p);
    free(seq.p);
    return rc;
}

/* get_sequence_fastq_gz (libkmers.cpp:240-300): gunzip (or read as-is when not gzip: zlib's transparent
 * mode), split at '\n', keep lines 1, 5, 9, ... (0-based), join them with nothing in between.  Only
 * '\n'-terminated lines count; an unterminated tail is dropped.  The reference works through an 8192-byte
 * window and gives up (empty result for the whole file) as soon as a line does not fit, i.e. when any
 * line, terminated or not, is 8192 bytes or longer.  Returns 0 and fills `out`; `out` is empty on error. */
static void fastq_sequence(const char *path, bytes_t *out) {
    out->n = 0;
    gzFile in = gzopen(path, "rb");
    if (!in) return;
    bytes_t all = {0, 0, 0};
    char buf[1 << 16];
    int got;
    while ((got = gzread(in, buf, sizeof buf)) > 0) bytes_append(&all, buf, (size_t)got);
    int bad = (got < 0);
    if (gzclose(in) != Z_OK) bad = 1;
    if (!bad) {
        size_t pos = 0, line_no = 0;
        while (pos < all.n) {
            const char *nl = (const char *)memchr(all.p + pos, '\n', all.n - pos);
            size_t e = nl ? (size_t)(nl - all.p) : all.n;
            if (e - pos >= 8192) {
                bad = 1;
                break;
            }
            if (!nl) break;
            if ((line_no & 3) == 1) bytes_append(out, all.p + pos, e - pos);
            ++line_no;
            pos = e + 1;
        }
    }
    if (bad) out->n = 0;
    free(all.p);
}

/* libkmers.cpp:378-392 */
int oracle_fastq_gz_count_kmers(int n_files, const char *fnames[], int K, int *total_kmer_counts) {
    if (K < 1 || K > 16) return -3;
    bytes_t seq = {0, 0, 0};
    for (int f = 0; f < n_files; ++f) {
        fastq_sequence(fnames[f], &seq);
        shred(seq.p, seq.n, K, count_emit, total_kmer_counts);
    }
    free(seq.p);
    return 0;
}

/* ------------------------------------------------------------------ exported: helpers for tests */

/* Number of window positions, valid or not: sum over kept records of max(len-K+1, 0).  This is the "k-mer"
 * of the throughput metric.  mode 0: X rules (PREDICTED + L filters), mode 1: y rules. */
int64_t oracle_fasta_window_count(int n_files, const char *fnames[], int K, int L, int mode) {
    bytes_t file = {0, 0, 0}, seq = {0, 0, 0};
    int64_t total = 0;
    for (int f = 0; f < n_files; ++f) {
        if (slurp(fnames[f], &file) != 0) {
            total = -1;
            break;
        }
        fasta_cursor cur = {file.p, file.n, 0};
        const char *hdr;
        size_t hlen;
        while (next_record(&cur, &hdr, &hlen, &seq)) {
            if (has_predicted(hdr, hlen)) continue;
            if (mode == 0 && (L < 0 || seq.n < (size_t)L)) continue;
            if (seq.n >= (size_t)K) total += (int64_t)(seq.n - (size_t)K + 1);
        }
    }
    free(file.p);
    free(seq.p);
    return total;
}
