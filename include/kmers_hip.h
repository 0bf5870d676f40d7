/*
 * kmers_hip.h -- the extended C ABI of libkmers.so: the stages that sit behind the six drop-in calls of
 * libkmers.h, exposed so that callers can keep data resident in HBM (bench.py, the multi-GPU layer in
 * kmers_amd/distributed.py, pipelines that feed X straight into a device-side solver).
 *
 * Plain pointers and sizes only.  "d_" pointers are device (HBM) addresses on the current HIP device;
 * everything else is host memory.  `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *
 *   host ingest   kmers_store_*      FASTA / FASTQ(.gz) -> 2-bit packed bases + validity mask + record table
 *                                    (replaces the reference's serial parser, src/libkmers.cpp:240-318, and
 *                                    the record filtering / numbering of :360-371, :398-421, :444-457)
 *   device y      kmers_hip_count_y  packed store -> dense int32 histogram   (replaces count_kmers :320-328)
 *   device X      kmers_hip_build_x  packed store -> CSR (data,row_ind,col_ind)
 *                                    (replaces collect_kmers/fill_indices_coo :218-238 and to_csr :77-169)
 *   device model  kmers_hip_model_*  X and y in HBM -> logp_grad's two sparse products (kmerexpr/utils.py:76-108)
 *   device reads  kmers_hip_simulate_reads  transcriptome store + (isoform, start) per read -> reads store
 *                                    (kmerexpr/simulate_reads.py:111-121)
 */
#ifndef KMERS_HIP_H
#define KMERS_HIP_H

#include "libkmers.h"
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes beyond the reference's 0 / -1 / -2 ---- */
#define KMERS_OK 0
#define KMERS_ERR_OPEN (-1)          /* FASTA file could not be opened (reference code)            */
#define KMERS_ERR_MAX_NZ (-2)        /* nnz > max_size (reference wrapper's reserved code)         */
#define KMERS_ERR_NO_DEVICE (-10)    /* no usable HIP device: this library has no CPU fallback     */
#define KMERS_ERR_HIP (-11)          /* a HIP runtime call failed (message on stderr)              */
#define KMERS_ERR_BAD_ARG (-12)      /* K outside 1..16, null pointer, bad mode/shard ...          */
#define KMERS_ERR_TOO_LARGE (-13)    /* more than 2^32-65536 sequence positions in one device call     */
#define KMERS_ERR_WORKSPACE (-14)    /* workspace smaller than kmers_hip_x_workspace_bytes()       */
#define KMERS_ERR_NOMEM (-15)        /* host allocation failed                                     */

/* ---- host ingest ------------------------------------------------------------------------------- */

/* What the records of a store are (which reference entry point's parsing rules apply). */
#define KMERS_MODE_X 0        /* fasta_to_kmers_csr:            one record per kept FASTA record (PREDICTED, L) */
#define KMERS_MODE_X_CAT 1    /* fasta_to_kmers_csr_cat_subseq: one record per file                              */
#define KMERS_MODE_Y_FASTA 2  /* fasta_count_kmers:             one record per non-PREDICTED FASTA record        */
#define KMERS_MODE_Y_FASTQ 3  /* fastq_gz_count_kmers:          one record per file (reads joined)               */
#define KMERS_MODE_FASTA_PY 4 /* the isoforms as kmerexpr/fasta.py reads them for simulate_reads (python text-mode lines,
                                 strip() on both sides, records only at '>' lines, PREDICTED and len >= L filters;
                                 reference kmerexpr/fasta.py:19-46, simulate_reads.py:78-92); ASCII files only */

/* A packed sequence store in host memory (pinned when a HIP device is present).
 *
 * Layout (the same arrays are what the device entry points take):
 *   position space: record r occupies positions [rec_start[r], rec_start[r+1] - 1) followed by ONE separator
 *                   position whose validity bit is 0, so no k-mer window can span two records;
 *                   n_pos = rec_start[n_rec].
 *   bases[i / 16]   2-bit codes A,C,G,T = 0..3, position i in bits [31-2(i%16) : 30-2(i%16)] (big-endian in
 *                   the word, so a K-window read as a bit-field IS the reference's k-mer id).  Invalid
 *                   positions hold code 0.
 *   valid[i / 32]   bit (i % 32) = 1 iff the byte at position i was one of A,C,G,T.
 *   rec_start       uint64[n_rec + 1].
 * Both word arrays are zero-padded with 4 extra words. */
typedef struct kmers_store kmers_store;

/* Parse `n_files` files under the rules of `mode` with `KMER_NUM_THREADS` host threads (default: all cores,
 * at most 64) and pack shard `shard` of `n_shards` (contiguous record ranges balanced by positions; pass
 * 0, 1 for everything).  L is used by the two X modes and KMERS_MODE_FASTA_PY.  Returns 0 or KMERS_ERR_OPEN / _BAD_ARG / _NOMEM. */
int kmers_store_ingest(int n_files, const char *fnames[], int mode, int L, int shard, int n_shards,
                       kmers_store **out);
/* The same for multi-process ingest where no rank looks at the whole input: shard `shard` is the records whose header
 * lies in bytes [T shard / n_shards, T (shard + 1) / n_shards) of the files laid end to end (T bytes), and only those
 * bytes are scanned (plus the tail of the range's last record).  Record numbers are local (first_rec 0, n_rec_total =
 * n_rec) until kmers_store_set_numbering() is given the sum of the earlier shards' n_rec and the total (one small
 * all-gather, kmers_amd.distributed.ingest_sharded).  The concatenation of the shards in order is the whole list, so
 * column numbers are the single-process ones.  KMERS_MODE_X, _Y_FASTA and _FASTA_PY (the other two modes have one
 * record per file: shard those with kmers_store_ingest). */
int kmers_store_ingest_bytes(int n_files, const char *fnames[], int mode, int L, int shard, int n_shards,
                             kmers_store **out);
void kmers_store_set_numbering(kmers_store *s, uint64_t first_rec, uint64_t n_rec_total);
void kmers_store_free(kmers_store *s);

uint64_t kmers_store_n_pos(const kmers_store *s);         /* positions in this shard (bases + separators)      */
uint64_t kmers_store_n_rec(const kmers_store *s);         /* records in this shard                             */
uint64_t kmers_store_n_rec_total(const kmers_store *s);   /* records over all shards (X: the matrix's n_cols)  */
uint64_t kmers_store_first_rec(const kmers_store *s);     /* global index of this shard's first record         */
uint64_t kmers_store_n_bases(const kmers_store *s);       /* bases in this shard (n_pos - n_rec)               */
uint64_t kmers_store_n_windows(const kmers_store *s, int K); /* sum over records of max(len-K+1,0): the metric's k-mers */
const uint32_t *kmers_store_bases(const kmers_store *s);
const uint32_t *kmers_store_valid(const kmers_store *s);
const uint64_t *kmers_store_rec_start(const kmers_store *s);
size_t kmers_store_bases_words(const kmers_store *s);     /* length of bases[] incl. padding                   */
size_t kmers_store_valid_words(const kmers_store *s);     /* length of valid[] incl. padding                   */

/* ---- device entry points -------------------------------------------------------------------------- */

/* Number of usable HIP devices (0 when there is none; never fails). */
int kmers_hip_device_count(void);

/* Copy a host store into caller-allocated device arrays (d_bases u32[bases_words], d_valid u32[valid_words],
 * d_rec_start u64[n_rec + 1] or NULL) straight from the store's pinned memory: the H2D step of the file-level calls
 * (kmers_amd/csrc/libkmers.cpp, DevStore::upload), for callers that keep the store resident.  Synchronises `stream`. */
int kmers_hip_store_upload(const kmers_store *s, uint32_t *d_bases, uint32_t *d_valid, uint64_t *d_rec_start,
                           void *stream);

/* y: for every position p with valid[p .. p+K) all set, counts[id(p)] += 1.
 * d_counts is int32[4^K] in HBM and is ADDED to (the caller zeroes it), like the reference's histogram.
 * K <= 8: LDS-privatised histograms, asynchronous on `stream`.  K >= 9: the k-mers are partitioned by their top
 * 8 or 16 bits in a scratch buffer and counted slice by slice; this call allocates that buffer itself and therefore
 * synchronises `stream` before returning -- use kmers_hip_count_y_ws to bring the buffer and stay asynchronous.
 * Calls that add into the same d_counts must not overlap (K >= 9 adds with plain read-modify-writes). */
int kmers_hip_count_y(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words,
                      uint64_t n_pos, int K, int32_t *d_counts, void *stream);

/* Recommended scratch bytes for kmers_hip_count_y_ws (0 for K <= 8): 4 bytes per position (8 for K >= 13) plus
 * the tile histogram, for chunks of at most 2^31 positions. */
size_t kmers_hip_y_workspace_bytes(uint64_t n_pos, int K);

/* kmers_hip_count_y with a caller-owned scratch buffer; asynchronous on `stream`, no allocation.  Any size down to
 * one 8192-position tile's worth is accepted (KMERS_ERR_WORKSPACE below that): a smaller buffer means more,
 * smaller chunks.  No counterpart in the reference. */
int kmers_hip_count_y_ws(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words,
                         uint64_t n_pos, int K, int32_t *d_counts, void *d_workspace, size_t workspace_bytes,
                         void *stream);

/* kmers_hip_count_y for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same kernels and same result as kmers_hip_count_y_ws, but
 * the scratch of the K >= 9 path is a 32 MiB array that lives inside the library's code object on every GPU it runs on
 * (looked up once per device on the first K >= 9 call there: hipGetSymbolAddress, a runtime call that is not
 * documented as capture-safe -- before capturing, make one call on that device outside the capture; n_pos = 0
 * suffices and launches nothing); the store is worked through in chunks that fit it.  The call
 * allocates nothing (no hipMalloc / hipFree / events), is asynchronous on `stream`, and takes at most 2^28 positions
 * (KMERS_ERR_TOO_LARGE with a message above that).  All calls on one GPU share the one array: calls that overlap
 * there (two streams, two threads) race -- serialise them. */
#define KMERS_STATIC_MAX_POS (1ull << 28)
int kmers_hip_count_y_static(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words,
                             uint64_t n_pos, int K, int32_t *d_counts, void *stream);

/* Bytes of scratch HBM kmers_hip_build_x needs for a store of n_pos positions / n_rec records. */
size_t kmers_hip_x_workspace_bytes(uint64_t n_pos, uint64_t n_rec, int K);

/* X: CSR of P(k-mer | record) for the store, columns numbered first_col, first_col+1, ...
 *   d_data f32[max_nz], d_col_ind u64[max_nz], d_row_ind u64[4^K + 1]  (all in HBM, written, not accumulated)
 *   *nnz_out (host) receives the number of non-zeros.  Returns KMERS_ERR_MAX_NZ (outputs untouched beyond
 *   row_ind) when nnz > max_nz.  Synchronises `stream` before returning (it has to read nnz back). */
int kmers_hip_build_x(const uint32_t *d_bases, size_t bases_words, const uint32_t *d_valid, size_t valid_words,
                      const uint64_t *d_rec_start, uint64_t n_pos, uint64_t n_rec, uint64_t first_col, int K,
                      float *d_data, uint64_t *d_col_ind, uint64_t *d_row_ind, uint64_t max_nz, void *d_workspace,
                      size_t workspace_bytes, uint64_t *nnz_out, void *stream);

/* ---- per-kernel timing ----------------------------------------------------------------------------------- */

/* While enabled, kmers_hip_build_x brackets every kernel with hipEvents on its launch stream and accumulates the
 * elapsed time per kernel name (enabling also clears the table).  kmers_hip_profile_get returns entry i:
 * kernel name, total milliseconds and number of launches.  No counterpart in the reference (its Timer,
 * src/libkmers.cpp:34-41, prints phase times to stdout). */
void kmers_hip_profile_enable(int on);
int kmers_hip_profile_count(void);
int kmers_hip_profile_get(int i, char *name, size_t name_cap, double *ms_total, uint64_t *calls);

/* Wall time of the stages of the most recent file-level call (fasta_to_kmers_csr, fasta_count_kmers, ...) of this
 * process, as "stage name=milliseconds;" pairs in call order: host ingest, H2D + device buffers, the GPU stage, D2H into
 * the caller's arrays.  The counterpart of the reference's Timer printouts (src/libkmers.cpp:34-41, :93, :123, :166);
 * KMERS_VERBOSE=2 also prints them to stderr. */
int kmers_last_call_stages(char *buf, size_t cap);

/* Path switches for tests and same-run A/B measurements (process-wide; never needed in production).  The library reads
 * the environment variables of the same meaning ONCE, at its first use in the process (KMERS_Y_ATOMICS, KMERS_Y_CAP,
 * KMERS_FORCE_WIDE, KMERS_X_PATH=tables|rle, KMERS_X_OLD_PASS0, KMERS_X_CHUNK_POS, KMERS_VERBOSE); afterwards only this call
 * changes them: name = "y_atomics", "y_cap", "force_wide", "x_path" (0 default, 1 tables, 2 rle), "old_pass0",
 * "x_chunk_pos", "verbose", "wide14" (KMERS_X_WIDE14), or "reset" (re-read the environment).  Returns 0, or KMERS_ERR_BAD_ARG for an unknown name.
 * No counterpart in the reference. */
int kmers_hip_debug_knob(const char *name, long long value);

/* Diagnostic builds only (`python -m kmers_amd.build --stamps`, -DKK_STAMPS): cycle sums of the phases of the two
 * partition kernels (s_memtime, wave 0 of every workgroup).  The product build returns KMERS_ERR_BAD_ARG. */
int kmers_hip_debug_stamps(unsigned long long *out16, int reset);

/* fasta_to_kmers_csr / fasta_to_kmers_csr_cat_subseq (include/libkmers.h) with int32 row_ind[4^K + 1] and col_ind[max_size]:
 * for callers that narrow the indices anyway.  The reference's transcriptome_to_x builds a scipy csr_matrix from the
 * uint64 arrays (kmerexpr/transcriptome_reader.py:33-37), and scipy stores int32 indices whenever they fit -- a serial
 * 3 GB conversion for config 2, several times the whole GPU call; kmers_amd/transcriptome_reader.py uses this entry
 * point instead.  The indices leave the device as 32-bit values in any case, so this also skips the widening and a
 * third of the host memory written.  Same arguments, results and return codes as the two reference calls otherwise;
 * KMERS_ERR_TOO_LARGE (without a message) when nnz or the column count is above 2^31 - 1 or the input needs the
 * record-range route: the caller then uses the 64-bit call.  No counterpart in the reference. */
int kmers_fasta_to_kmers_csr_i32(int n_files, const char *fnames[], int K, int L, int concatenate_subseq, float *data,
                                 int32_t *row_ind, int32_t *col_ind, uint64_t max_size, uint64_t *nnz, int *n_cols);

/* How many contiguous record ranges the file-level X calls cut an input into when it is above one device call's
 * limit (cap_pos positions; 0 = the built-in 2^32 - 65536): a record goes to range floor(start * n / n_pos), so a
 * range holds at most ceil(n_pos / n) + longest_rec_pos - 1 positions and n is chosen to keep that under the cap.
 * 0 when one record alone is above the cap or more than 4096 ranges would be needed (the calls then return
 * KMERS_ERR_TOO_LARGE).  Pure host arithmetic; no counterpart in the reference (it has no size limit of this kind). */
uint64_t kmers_x_range_count(uint64_t n_pos, uint64_t n_rec_total, uint64_t longest_rec_pos, uint64_t cap_pos);

/* ---- host-side merge of per-shard CSRs (multi-GPU X) --------------------------- */

/* Interleave n_parts CSR matrices that cover disjoint ascending column ranges (part g's columns all below
 * part g+1's) into one CSR: out_row_ind[r] = sum_g row_ind_g[r]; inside a row, part 0's entries first.
 * All arrays host memory; out_* sized sum(nnz_g) / n_rows+1. */
int kmers_csr_merge(int n_parts, uint64_t n_rows, const float *const data[], const uint64_t *const row_ind[],
                    const uint64_t *const col_ind[], float *out_data, uint64_t *out_row_ind, uint64_t *out_col_ind);

/* One rank's part of the same interleave when every rank writes into ONE shared output (multi-process, POSIX shared
 * memory): this rank's CSR (row_ind u64[n_rows + 1], data, col_ind) goes to
 *   out[global_row_ind[r] + (before[r + 1] - before[r]) ...]  for every row r,
 * where global_row_ind = sum over ranks of row_ind and before = sum of row_ind over the ranks in front of this one (both
 * u64[n_rows + 1]; kmers_amd.distributed.concat_x_shards gets them from one all-reduce and a prefix over the ranks).
 * before may be NULL: global_row_ind[r] is then taken as the finished destination of row r (the caller has added the
 * earlier ranks' counts itself, e.g. on the device) and out_row_ind must be NULL.
 * out_row_ind, when not NULL, receives global_row_ind (rank 0).  Host memory; all host threads. */
int kmers_csr_place(uint64_t n_rows, const uint64_t *row_ind, const uint64_t *before, const uint64_t *global_row_ind,
                    const float *data, const uint64_t *col_ind, float *out_data, uint64_t *out_col_ind,
                    uint64_t *out_row_ind);

/* The same placement from a list of this rank's NON-EMPTY rows only (a column shard at K = 14 fills a sixth of its 2^28
 * rows): row i of the list is len[i] entries from data / col_ind [src_start[i] ...) to out [dst_start[i] ...).  The list
 * is made where the row pointers are (on the device), so only it crosses PCIe. */
int kmers_csr_place_rows(uint64_t n, const uint64_t *src_start, const uint32_t *len, const uint64_t *dst_start, const float *data,
                         const uint64_t *col_ind, float *out_data, uint64_t *out_col_ind);

/* ---- X as the reference's .npz --------------------------------------------- */

/* Write the CSR arrays of the ABI as the uncompressed scipy .npz that the reference's transcriptome_to_x saves
 * (kmerexpr/transcriptome_reader.py:39, scipy.sparse.save_npz(x_file, xt, compressed=False)): members indices, indptr
 * (int32 when n_rows, n_cols and nnz all fit, else int64 -- scipy's rule), format, shape, data; stored ZIP with
 * ZIP64 records.  The uint64 indices are narrowed chunk by chunk on the way out and the CRCs computed on all host
 * threads, so no second copy of X is made.  Host memory in, file out.  Returns 0, KMERS_ERR_OPEN (cannot create or
 * write the file) or KMERS_ERR_BAD_ARG. */
int kmers_npz_write_csr(const char *path, const float *data, const uint64_t *col_ind, const uint64_t *row_ind,
                        uint64_t nnz, uint64_t n_rows, uint64_t n_cols);

/* The same file from int32 index arrays (what kmers_fasta_to_kmers_csr_i32 fills); KMERS_ERR_TOO_LARGE when the shape or
 * nnz is above 2^31 - 1 (scipy would then store int64: use kmers_npz_write_csr). */
int kmers_npz_write_csr_i32(const char *path, const float *data, const int32_t *col_ind, const int32_t *row_ind,
                            uint64_t nnz, uint64_t n_rows, uint64_t n_cols);

/* ---- the step after the path: logp_grad's two sparse products on the resident X ---- */

/* Device-side state of the reference's multinomial_simplex_model (kmerexpr/multinomial_simplex_model.py:56-62):
 * ymask = y.nonzero(), xnnz = x[ymask] (a CSR with 32-bit indices, plus a transposed copy sorted by (row slab,
 * column) for the transposed product; KMERS_MODEL_SLAB_LOG2 overrides the 2^19 rows per slab),
 * ynnz = y[ymask] / sum(y) (float64).  Everything lives in HBM until kmers_hip_model_free. */
typedef struct kmers_model kmers_model;

/* Build the model from X as kmers_hip_build_x left it in HBM (d_data f32[nnz], d_col_ind u64[nnz],
 * d_row_ind u64[n_rows + 1]) and the histogram d_y int32[n_rows] of kmers_hip_count_y.  Allocates (library-owned
 * device memory, about 16 bytes per non-zero of xnnz plus 20 per masked row; twice that while it is being built)
 * and synchronises `stream`.  Limits: n_cols, masked rows and non-zeros of xnnz below 2^32 (KMERS_ERR_TOO_LARGE);
 * y must have a non-zero and every column index of the masked rows must lie in the model's range
 * (KMERS_ERR_BAD_ARG). */
int kmers_hip_model_create(const float *d_data, const uint64_t *d_col_ind, const uint64_t *d_row_ind, uint64_t n_rows,
                           uint64_t n_cols, const int32_t *d_y, void *stream, kmers_model **out);
void kmers_hip_model_free(kmers_model *m);

/* The same for one column shard of X (multi-GPU fit, one process per GPU): this GPU's X holds the columns
 * [first_col, first_col + n_cols) under their global numbers, as kmers_hip_build_x writes them for a shard; d_y is
 * the histogram of ALL reads (after the all-reduce), so every rank masks the same rows.  theta and grad of this
 * model are the shard's n_cols entries. */
int kmers_hip_model_create_shard(const float *d_data, const uint64_t *d_col_ind, const uint64_t *d_row_ind,
                                 uint64_t n_rows, uint64_t first_col, uint64_t n_cols, const int32_t *d_y, void *stream,
                                 kmers_model **out);
uint64_t kmers_hip_model_n_masked(const kmers_model *m);   /* rows of xnnz = number of non-zeros of y          */
uint64_t kmers_hip_model_nnz(const kmers_model *m);        /* non-zeros of xnnz                                */
int64_t kmers_hip_model_sum_y(const kmers_model *m);       /* the reference's self.N                           */
const uint32_t *kmers_hip_model_rows(const kmers_model *m);   /* device: ymask, u32[n_masked], ascending        */
const float *kmers_hip_model_xtheta(const kmers_model *m);    /* device: xnnz.theta of the last call, f32[n_masked] */

/* Copy rows (what = 0, uint32[n_masked]) or xtheta (what = 1, float32[n_masked]) to host memory; synchronous. */
int kmers_hip_model_read(const kmers_model *m, int what, void *host_dst);

/* The data terms of logp_grad (kmerexpr/utils.py:76-108) for theta = d_theta f32[n_cols] (device):
 *   xtheta = xnnz.theta (:92), *val_out = sum ynnz*log(xtheta) (:93, host double),
 *   unless nograd: d_grad f32[n_cols] (device) = (ynnz / xtheta).xnnz (:102-103).
 * Sums are accumulated in float64 and rounded to float32 once (the reference's MKL float32 products round per
 * operation), in a fixed order: the same inputs give the same bits on every call.  The (beta - 1) terms (:94-96,
 * :104-106) are O(n_cols) arithmetic on the caller's theta and are not computed here.
 * Synchronises `stream` (val is read back).  Calls on one model must not overlap, and the model's device must be
 * the current one (KMERS_ERR_BAD_ARG otherwise). */
int kmers_hip_model_logp_grad(kmers_model *m, const float *d_theta, int nograd, double *val_out, float *d_grad,
                              void *stream);

/* ---- simulated reads straight into a packed store ------------------------------------ */

/* The copying step of the reference's simulate_reads (kmerexpr/simulate_reads.py:111-121: read n = L bases of isoform
 * iso[n] from start[n]) on the device: the transcriptome store (kmers_store_ingest, KMERS_MODE_FASTA_PY with the same L, so
 * that its records are the reference's isoform list) in, a reads store out -- read n at positions
 * [n (L + 1), n (L + 1) + L) plus one separator, the layout kmers_hip_count_y takes with n_pos = n_reads (L + 1).
 * Output arrays: at least n_pos / 16 + 4 and n_pos / 32 + 4 words (KMERS_ERR_WORKSPACE), fully written (padding
 * zeroed).  An isoform index >= n_rec or a start with start + L > the record's length gives KMERS_ERR_BAD_ARG (that
 * read is left empty, nothing is read out of bounds).  Synchronises `stream`. */
int kmers_hip_simulate_reads(const uint32_t *d_bases, const uint32_t *d_valid, const uint64_t *d_rec_start,
                             uint64_t n_rec, const uint32_t *d_iso, const uint32_t *d_start, uint64_t n_reads, uint32_t L,
                             uint32_t *d_out_bases, size_t out_bases_words, uint32_t *d_out_valid,
                             size_t out_valid_words, void *stream);

/* Host: the reference's per-read `start = np.random.choice(len(seq) - L + 1)` (simulate_reads.py:116) replayed over a
 * block of raw MT19937 32-bit outputs (np.random.randint(0, 2**32, dtype=uint32) on the same legacy stream): numpy's
 * scalar choice(n) masks successive outputs with the smallest 2^k - 1 >= n - 1 and rejects values above n - 1; n == 1
 * draws nothing.  n_choices[i] = len_i - L + 1.  Stops early when the block runs out: *reads_done reads are final,
 * *raw_used outputs were consumed by them (the caller continues with the unused outputs followed by a new block). */
int kmers_sim_starts(const uint32_t *raw, uint64_t n_raw, const uint32_t *n_choices, uint64_t n, uint32_t *start_out,
                     uint64_t *reads_done, uint64_t *raw_used);

/* logp_grad in two steps, for column shards: d_xtheta f32[n_masked] (caller's buffer) = this shard's xnnz.theta
 * (asynchronous); the caller sums the shards' buffers (one all-reduce over RCCL: the path's only exchange step);
 * then val, and this shard's gradient entries, from the summed xtheta (16-byte aligned; synchronises `stream`).
 * kmers_hip_model_logp_grad is the two steps on the model's own buffer. */
int kmers_hip_model_xtheta_into(kmers_model *m, const float *d_theta, float *d_xtheta, void *stream);
int kmers_hip_model_grad_from(kmers_model *m, const float *d_xtheta, int nograd, double *val_out, float *d_grad,
                              void *stream);

#ifdef __cplusplus
}
#endif
#endif
