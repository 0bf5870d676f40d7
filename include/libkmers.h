/*
 * libkmers.h -- the drop-in C ABI of libkmers.so (MI355X / gfx950 build).
 *
 * These six entry points are exactly what the reference's ctypes layer binds
 * (reference kmerexpr/libkmers.py:37-74) and what the reference header declares
 * (reference src/libkmers.h:13-20): same names, same parameter lists, same ownership rules
 * (the caller allocates every buffer; the library never retains a pointer), same return codes.
 * Behind them the work is done by hand-written HIP kernels (kmers_amd/csrc/kernels.hip); there is
 * no CPU fallback: without a usable GPU the four file-level calls return KMERS_ERR_NO_DEVICE.
 *
 * Return codes (reference kmerexpr/libkmers.py:102-103, 173-176):
 *    0  success
 *   -1  a FASTA file could not be opened           (reference src/libkmers.cpp:356-357, 440-441)
 *   -2  more than max_size non-zeros               (reserved by the reference wrapper, libkmers.py:175-176;
 *                                                    the reference library never checks -- this one does)
 *   <= -10  errors the reference cannot have: see kmers_hip.h (no device, HIP failure, bad K, too large)
 */
#ifndef LIBKMERS_H
#define LIBKMERS_H

#ifdef __cplusplus
#include <cstdint>
extern "C" {
#else
#include <stdbool.h>
#include <stdint.h>
#endif

/* replaces reference src/libkmers.cpp:331-337 (header src/libkmers.h:13).  Host function.
 * id = big-endian base-4 number of kmer[0..len), A,C,G,T = 0,1,2,3. */
uint32_t kmer_to_id(const char *kmer, int len);

/* replaces reference src/libkmers.cpp:339-345 (src/libkmers.h:14).  Host function.
 * true iff every byte of kmer[0..len) is one of 'A','C','G','T' (upper case only). */
bool valid_kmer(const char *kmer, int len);

/* replaces reference src/libkmers.cpp:378-392 (src/libkmers.h:15).
 * y from FASTQ / FASTQ.gz files: line 2 of every 4-line record, the reads of one file joined without a
 * separator (k-mers spanning two reads ARE counted, as in the reference).  ADDS into
 * total_kmer_counts[0 .. 4^K), which the caller zero-initialises (libkmers.py:116). */
int fastq_gz_count_kmers(int n_files, const char *fnames[], int K, int *total_kmer_counts);

/* replaces reference src/libkmers.cpp:347-376 (src/libkmers.h:16).
 * y from FASTA reads: every record whose header does not contain "PREDICTED" is shredded on its own
 * (no k-mer spans two records).  ADDS into total_kmer_counts[0 .. 4^K). */
int fasta_count_kmers(int n_files, const char *fnames[], int K, int *total_kmer_counts);

/* replaces reference src/libkmers.cpp:394-430 (src/libkmers.h:17-18).
 * X with one column per FILE: the non-PREDICTED records of a file are concatenated (k-mers spanning the
 * joins are counted); a file that cannot be opened, or whose concatenation is shorter than L, gives an
 * empty column; *n_cols = n_files. */
int fasta_to_kmers_csr_cat_subseq(int n_files, const char *fnames[], int K, int L, float *data, uint64_t *row_ind,
                                  uint64_t *col_ind, uint64_t max_size, uint64_t *nnz, int *n_cols);

/* replaces reference src/libkmers.cpp:432-469 (src/libkmers.h:19-20).
 * X = P(k-mer | isoform) as CSR with 4^K rows: row_ind[0 .. 4^K] (row pointer), and the first *nnz entries
 * of col_ind / data (columns ascending inside a row, float32 value = float(count)/float(total)).
 * One column per record whose header does not contain "PREDICTED" and whose length is >= L, numbered
 * in file order across all files. */
int fasta_to_kmers_csr(int n_files, const char *fnames[], int K, int L, float *data, uint64_t *row_ind,
                       uint64_t *col_ind, uint64_t max_size, uint64_t *nnz, int *n_cols);

#ifdef __cplusplus
}
#endif
#endif
