"""kmers_npz_write_csr against scipy itself: the file must load with
scipy.sparse.load_npz into the matrix the reference's transcriptome_to_x would have saved
(kmerexpr/transcriptome_reader.py:36-39), and hold the same members with the same dtypes, shapes and values as
scipy.sparse.save_npz(..., compressed=False) writes for it.  Host-only: runs without a GPU."""
import os
import zipfile

import numpy as np
import pytest
from scipy.sparse import csr_matrix, load_npz, save_npz

from kmers_amd import transcriptome_reader as tr
from tests import kmtest

X_CASES = [c for c in kmtest.manifest() if c["kind"] == "x"]


def _golden_csr(case):
    g = kmtest.golden(case)
    M = 4 ** case["K"]
    rows = g["rows"].astype(np.int64)
    row_ind = np.zeros(M + 1, np.uint64)
    np.cumsum(np.bincount(rows, minlength=M), out=row_ind[1:].view(np.int64))
    return g["data"].astype(np.float32), row_ind, g["cols"].astype(np.uint64), M, int(g["n_cols"])


@pytest.mark.parametrize("case", X_CASES[::4], ids=lambda c: c["name"])
def test_npz_matches_scipy_save(tmp_path, case):
    data, row_ind, col_ind, M, n_cols = _golden_csr(case)
    ours, theirs = str(tmp_path / "ours.npz"), str(tmp_path / "theirs.npz")
    tr.save_x_npz(ours, data, row_ind, col_ind, M, n_cols)
    want = csr_matrix((data, col_ind, row_ind), shape=(M, n_cols), dtype=np.float32)
    save_npz(theirs, want, compressed=False)
    a, b = np.load(ours), np.load(theirs)
    assert a.files == b.files == ["indices", "indptr", "format", "shape", "data"]
    for k in b.files:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
        assert np.array_equal(a[k], b[k]), k
    got = load_npz(ours)
    assert got.shape == want.shape and got.dtype == np.float32 and got.format == "csr"
    assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
    assert np.array_equal(got.data.view(np.uint32), want.data.view(np.uint32))
    with zipfile.ZipFile(ours) as z:
        assert z.testzip() is None                      # every member's CRC checks
        assert all(i.compress_type == zipfile.ZIP_STORED for i in z.infolist())


def test_npz_empty_and_suffix(tmp_path):
    M = 16
    p = str(tmp_path / "empty")                         # scipy appends .npz
    tr.save_x_npz(p, np.zeros(0, np.float32), np.zeros(M + 1, np.uint64), np.zeros(0, np.uint64), M, 3)
    got = load_npz(p + ".npz")
    assert got.shape == (M, 3) and got.nnz == 0
    with pytest.raises(OSError):
        tr.save_x_npz(str(tmp_path / "no" / "such" / "dir.npz"), np.zeros(0, np.float32), np.zeros(M + 1, np.uint64),
                      np.zeros(0, np.uint64), M, 3)


def test_npz_many_chunks_and_wide_indices(tmp_path):
    """More than one 64 MiB write chunk, and the int64 case (a column count beyond int32 forces wide indices,
    exactly as in scipy)."""
    rng = np.random.default_rng(7)
    M, nnz = 4 ** 6, 20_000_000
    counts = rng.multinomial(nnz, np.full(M, 1.0 / M))
    row_ind = np.zeros(M + 1, np.uint64)
    np.cumsum(counts, out=row_ind[1:].view(np.int64))
    col_ind = rng.integers(0, 1 << 20, nnz).astype(np.uint64)
    data = rng.random(nnz, dtype=np.float32)
    p = str(tmp_path / "big.npz")
    tr.save_x_npz(p, data, row_ind, col_ind, M, 1 << 20)
    a = np.load(p)
    assert a["indices"].dtype == np.int32 and np.array_equal(a["indices"], col_ind.astype(np.int32))
    assert np.array_equal(a["indptr"], row_ind.astype(np.int32)) and np.array_equal(a["data"], data)
    with zipfile.ZipFile(p) as z:
        assert z.testzip() is None
    n_cols = (1 << 31) + 5
    col_ind[-1] = n_cols - 1
    tr.save_x_npz(p, data[:1000], np.minimum(row_ind, 1000), col_ind[-1000:], M, n_cols)
    got = load_npz(p)
    assert got.indices.dtype == np.int64 and got.indptr.dtype == np.int64 and got.shape == (M, n_cols)
    assert int(got.indices[-1]) == n_cols - 1
    want = csr_matrix((data[:1000], col_ind[-1000:], np.minimum(row_ind, 1000)), shape=(M, n_cols))
    assert want.indices.dtype == np.int64                # scipy's own choice for this shape
    assert (got != want).nnz == 0


@pytest.mark.parametrize("case", X_CASES[1::5], ids=lambda c: c["name"])
def test_npz_from_int32_indices_is_the_same_file(tmp_path, case):
    """kmers_npz_write_csr_i32 (the arrays of kmers_fasta_to_kmers_csr_i32, which transcriptome_to_x uses) writes the
    file kmers_npz_write_csr writes from the uint64 arrays, byte for byte; and it refuses what scipy would store as int64."""
    from kmers_amd import _lib
    data, row_ind, col_ind, M, n_cols = _golden_csr(case)
    a, b = str(tmp_path / "a.npz"), str(tmp_path / "b.npz")
    tr.save_x_npz(a, data, row_ind, col_ind, M, n_cols)
    tr.save_x_npz(b, data, row_ind.astype(np.int32), col_ind.astype(np.int32), M, n_cols)
    assert open(a, "rb").read() == open(b, "rb").read()
    z = np.zeros(2, np.int32)
    rc = _lib.handle().kmers_npz_write_csr_i32(str(tmp_path / "c.npz").encode(), None, None, z.ctypes.data, 0, 1, 2 ** 31)
    assert rc == -13 and not os.path.exists(tmp_path / "c.npz")
