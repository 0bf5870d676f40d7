"""`transcriptome_to_x` / `load_xy` with the reference's signatures and .npz output
(reference kmerexpr/transcriptome_reader.py:6-41), on the HIP build of libkmers."""
import numpy as np
from scipy.sparse import csr_matrix, load_npz

from . import _lib
from .libkmers import fasta_to_kmers_csr, fasta_to_kmers_csr_i32


def save_x_npz(x_file, data, row_ind, col_ind, n_rows, n_cols):
    """The file scipy.sparse.save_npz(x_file, csr_matrix((data, col_ind, row_ind)), compressed=False) would hold
    (same members, dtypes and values; reference kmerexpr/transcriptome_reader.py:39), written by the library
    straight from the ABI's uint64 / float32 arrays (kmers_npz_write_csr)."""
    x_file = str(x_file)
    if not x_file.endswith(".npz"):
        x_file += ".npz"           # what scipy's save_npz does
    data = np.ascontiguousarray(data, dtype=np.float32)
    if row_ind.dtype == np.int32 and col_ind.dtype == np.int32:      # the arrays of fasta_to_kmers_csr_i32, as they are
        row_ind, col_ind = np.ascontiguousarray(row_ind), np.ascontiguousarray(col_ind)
        write = _lib.handle().kmers_npz_write_csr_i32
    else:
        row_ind = np.ascontiguousarray(row_ind, dtype=np.uint64)
        col_ind = np.ascontiguousarray(col_ind, dtype=np.uint64)
        write = _lib.handle().kmers_npz_write_csr
    assert row_ind.shape[0] == n_rows + 1 and col_ind.shape[0] == data.shape[0]
    rc = write(x_file.encode("UTF-8"), data.ctypes.data, col_ind.ctypes.data, row_ind.ctypes.data, data.shape[0], n_rows, n_cols)
    if rc == -1:
        raise OSError(f"cannot write {x_file}")
    _lib.check(rc, "kmers_npz_write_csr")


def load_xy(x_file, y_file):
    """reference kmerexpr/transcriptome_reader.py:6-16"""
    x = x_file if isinstance(x_file, csr_matrix) else load_npz(x_file)
    y = y_file if isinstance(y_file, np.ndarray) else load_npz(y_file).toarray().squeeze()
    return x, y


def transcriptome_to_x(K, fasta_files, x_file="", L=0, max_nz=0, float_t=np.float32, concatenate_subseq=False):
    """X = P(k-mer | isoform) as a scipy CSR matrix of shape (4**K, n_cols), float32
    (reference kmerexpr/transcriptome_reader.py:19-41).  If x_file is given the matrix is also saved there as an
    uncompressed scipy .npz, exactly as the reference does."""
    if float_t != np.float32:
        raise ValueError("Only float_t=np.float32 currently supported for fast reader")
    M = 4 ** K
    # scipy keeps int32 indices whenever they fit: ask the library for them (they leave the device as 32-bit values
    # anyway) instead of letting csr_matrix() narrow 3 GB of uint64 on one core; the uint64 call is the fallback
    res = fasta_to_kmers_csr_i32(fasta_files, K, max_nz, L=L, concatenate_subseq=concatenate_subseq)
    if res is None:
        res = fasta_to_kmers_csr(fasta_files, K, max_nz, L=L, concatenate_subseq=concatenate_subseq)
    data, row_ind, col_ind, n_cols = res
    if x_file:
        save_x_npz(x_file, data, row_ind, col_ind, M, n_cols)
    return csr_matrix((data, col_ind, row_ind), shape=(M, n_cols), dtype=float_t)
