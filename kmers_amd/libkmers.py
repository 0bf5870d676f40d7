"""The ctypes boundary of the hot path -- same functions, arguments, return values and exceptions as the
reference's kmerexpr/libkmers.py (reference file:line in each docstring), bound to the HIP build of libkmers.so.

    fasta_count_kmers(fasta_files, K)                      -> int32[4**K]
    fastq_gz_count_kmers(fastq_files, K)                   -> int32[4**K]
    fasta_to_kmers_csr(fasta_files, K, max_nz=0, L=0,
                       concatenate_subseq=False)           -> (data f32[nnz], row_ind u64[4**K+1], col_ind u64[nnz], n_cols)
"""
import os
from ctypes import POINTER, byref, c_float, c_int, c_uint64

import numpy as np

from . import _lib


def _as_list(files):
    return [files] if isinstance(files, str) else list(files)


def _get_total_size(files):
    """reference kmerexpr/libkmers.py:77-83"""
    tot_size = 0
    for file in files:
        if not os.path.isfile(file):
            raise FileNotFoundError("Invalid FASTA path provided")
        tot_size += os.stat(file).st_size
    return tot_size


def _raise(failed, open_msg):
    if failed == 0:
        return
    if failed == -1:
        raise RuntimeError(open_msg)
    _lib.check(failed)


def fasta_count_kmers(fasta_files, K):
    """y from FASTA reads (reference kmerexpr/libkmers.py:86-105).  Returns int32 counts of length 4**K."""
    fasta_files = _as_list(fasta_files)
    for file in fasta_files:
        if not os.path.isfile(file):
            raise FileNotFoundError("Invalid FASTA path provided")
    counts = np.zeros(4 ** K, dtype=np.int32)
    failed = _lib.handle().fasta_count_kmers(len(fasta_files), _lib.cfiles(fasta_files), K,
                                             counts.ctypes.data_as(POINTER(c_int)))
    _raise(failed, "Failed to open FASTA file")
    return counts


def fastq_gz_count_kmers(fastq_files, K):
    """y from FASTQ(.gz) reads (reference kmerexpr/libkmers.py:108-127)."""
    fastq_files = _as_list(fastq_files)
    for file in fastq_files:
        if not os.path.isfile(file):
            raise FileNotFoundError("Invalid fastq.gz path provided")
    counts = np.zeros(4 ** K, dtype=np.int32)
    failed = _lib.handle().fastq_gz_count_kmers(len(fastq_files), _lib.cfiles(fastq_files), K,
                                                counts.ctypes.data_as(POINTER(c_int)))
    _raise(failed, "Failed to open FASTQ file")
    return counts


def fasta_to_kmers_csr_i32(fasta_files, K, max_nz=0, L=0, concatenate_subseq=False):
    """fasta_to_kmers_csr with int32 row_ind / col_ind (kmers_fasta_to_kmers_csr_i32, include/kmers_hip.h): what
    transcriptome_to_x hands to scipy, which would narrow uint64 indices to int32 itself.  None when the indices do not
    fit int32 or the input is above one device call (the caller then takes fasta_to_kmers_csr).  No counterpart in the
    reference."""
    fasta_files = _as_list(fasta_files)
    if max_nz == 0:
        max_nz = _get_total_size(fasta_files)
    M = 4 ** K
    if max(M + 1, max_nz) > 0x7FFFFFFF:
        return None
    data = np.zeros(max_nz, dtype=np.float32)
    col_ind = np.zeros(max_nz, dtype=np.int32)
    row_ind = np.zeros(M + 1, dtype=np.int32)
    pos = c_uint64(0)
    n_cols = c_int(0)
    failed = _lib.handle().kmers_fasta_to_kmers_csr_i32(len(fasta_files), _lib.cfiles(fasta_files), K, L, int(bool(concatenate_subseq)),
                                                        data.ctypes.data, row_ind.ctypes.data, col_ind.ctypes.data, max_nz,
                                                        byref(pos), byref(n_cols))
    if failed == -13:       # KMERS_ERR_TOO_LARGE: not an error here
        return None
    _raise(failed, "Failed to open FASTA file")
    pos = pos.value
    return data[0:pos], row_ind, col_ind[0:pos], n_cols.value


def fasta_to_kmers_csr(fasta_files, K, max_nz=0, L=0, concatenate_subseq=False):
    """CSR arrays of X (reference kmerexpr/libkmers.py:130-184): (data, row_ind, col_ind, n_cols).

    max_nz defaults to the total size of the input files in bytes, an upper bound on the number of k-mers."""
    fasta_files = _as_list(fasta_files)
    tot_size = _get_total_size(fasta_files)
    if max_nz == 0:
        max_nz = tot_size
    M = 4 ** K
    data = np.zeros(max_nz, dtype=np.float32)
    col_ind = np.zeros(max_nz, dtype=np.uint64)
    row_ind = np.zeros(M + 1, dtype=np.uint64)
    pos = c_uint64(0)
    n_cols = c_int(0)
    lib = _lib.handle()
    fn = lib.fasta_to_kmers_csr_cat_subseq if concatenate_subseq else lib.fasta_to_kmers_csr
    failed = fn(len(fasta_files), _lib.cfiles(fasta_files), K, L, data.ctypes.data_as(POINTER(c_float)),
                row_ind.ctypes.data_as(POINTER(c_uint64)), col_ind.ctypes.data_as(POINTER(c_uint64)), max_nz,
                byref(pos), byref(n_cols))
    _raise(failed, "Failed to open FASTA file")
    pos = pos.value
    return data[0:pos], row_ind, col_ind[0:pos], n_cols.value
