"""The Python surface (kmers_amd.transcriptome_reader / kmers_amd.libkmers) mirrors the reference's: these are the
reference's own tests (test/test_transcriptome_reader.py:12-64) run against the HIP build, plus the wrapper's
error behaviour (kmerexpr/libkmers.py:80-81, 102-103, 173-176) and the device-resident API."""
import os

import numpy as np
import pytest
from scipy.sparse import load_npz

from kmers_amd import libkmers
from kmers_amd import transcriptome_reader as tr
from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store
from tests import kmtest

pytestmark = pytest.mark.gpu
DATA_PATH = kmtest.INP


def test_k1(tmp_path):
    ISO_FILE = os.path.join(DATA_PATH, "ref_test1.fsa")
    X_FILE = str(tmp_path / "x1_csr.npz")
    tr.transcriptome_to_x(1, ISO_FILE, X_FILE)
    x = load_npz(X_FILE)
    M, T = x.get_shape()
    assert M == 4 ** 1 and T == 2
    assert x.count_nonzero() == 8
    assert x.dtype == np.float32
    for i in range(4):
        assert x[i, 0] == 0.25
    assert x[0, 1] == pytest.approx(0.40, 1e-6)
    for i in (1, 2, 3):
        assert x[i, 1] == pytest.approx(0.20, 1e-6)


def test_k2(tmp_path):
    ISO_FILE = os.path.join(DATA_PATH, "ref_test2.fsa")
    X_FILE = str(tmp_path / "x2_csr.npz")
    tr.transcriptome_to_x(2, ISO_FILE, X_FILE)
    x = load_npz(X_FILE)
    M, T = x.get_shape()
    assert M == 4 ** 2 and T == 3
    assert x.count_nonzero() == 7
    for r, c in ((0, 0), (1, 0), (5, 0), (6, 1), (10, 1)):
        assert x[r, c] == pytest.approx(1.0 / 3.0, 1e-6)
    assert x[15, 2] == 1.0


def test_simplex():
    x = tr.transcriptome_to_x(2, os.path.join(DATA_PATH, "ref_test2.fsa"))
    M, T = x.get_shape()
    rng = np.random.default_rng()
    theta = np.abs(rng.standard_normal(T, dtype=np.float32))
    theta /= np.sum(theta)
    assert sum(x.dot(theta)) == pytest.approx(1.0, 1e-5)


def test_cfg1_simple_example_path():
    """BASELINE.json configs[0]: data/test5.fsa, K=8, L=14 as scripts/simple_example.py:24-27 builds X and y."""
    ISO = os.path.join(DATA_PATH, "ref_data_test5.fsa")
    x = tr.transcriptome_to_x(8, ISO, L=14)
    assert x.shape == (65536, 5000) and x.nnz == 31621
    assert len(np.unique(x.indices)) == 5000 and np.count_nonzero(np.diff(x.indptr)) == 32
    assert set(np.unique(x.data).tolist()) == {np.float32(1) / np.float32(7), np.float32(2) / np.float32(7)}
    y = libkmers.fasta_count_kmers(ISO, 8)
    assert y.dtype == np.int32 and int(y.sum()) == 35000 and np.count_nonzero(y) == 32 and int(y.max()) == 6048
    x2, y2 = tr.load_xy(x, y)
    assert x2 is x and y2 is y


def test_wrapper_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        tr.transcriptome_to_x(2, str(tmp_path / "nope.fsa"))
    with pytest.raises(FileNotFoundError):
        libkmers.fasta_count_kmers(str(tmp_path / "nope.fsa"), 2)
    with pytest.raises(FileNotFoundError):
        libkmers.fastq_gz_count_kmers([str(tmp_path / "nope.fq.gz")], 2)
    with pytest.raises(ValueError):
        tr.transcriptome_to_x(2, os.path.join(DATA_PATH, "ref_test2.fsa"), float_t=np.float64)
    with pytest.raises(RuntimeError, match="insufficient max_nz"):
        tr.transcriptome_to_x(2, os.path.join(DATA_PATH, "ref_test4.fsa"), max_nz=3)


def test_fastq_wrapper():
    y = libkmers.fastq_gz_count_kmers(os.path.join(DATA_PATH, "reads.fastq.gz"), 3)
    g = np.load(os.path.join(kmtest.GOLD, "fastq_gz_K3_y.npz"))
    assert np.array_equal(np.flatnonzero(y), g["idx"]) and np.array_equal(y[g["idx"]], g["cnt"])


def test_device_resident_api_matches_drop_in(product):
    """kmers_hip_build_x / kmers_hip_count_y on a store kept in HBM == the file-level calls; repeated builds
    reuse buffers; first_col offsets the column numbers of a shard."""
    import torch
    iso = os.path.join(DATA_PATH, "syn_small.fsa")
    for K in (8, 12):
        rc, data, row, col, n_cols = product.x([iso], K)
        st = Store(iso, MODE_X)
        ds = st.to_device()
        buf = ds.x_buffers(K, st.n_windows(K))
        for _ in range(2):
            nnz = ds.build_x(buf)
            torch.cuda.synchronize()
            assert nnz == data.shape[0]
            assert np.array_equal(buf["row_ind"].cpu().numpy().view(np.uint64), row)
            assert np.array_equal(buf["col_ind"][:nnz].cpu().numpy().view(np.uint64), col)
            assert np.array_equal(buf["data"][:nnz].cpu().numpy().view(np.uint32), data.view(np.uint32))
        rc, y = product.y([iso], K)
        sy = Store(iso, MODE_Y_FASTA).to_device()
        c = sy.count_y(K)
        c = sy.count_y(K, c)          # accumulates
        assert np.array_equal(c.cpu().numpy(), 2 * y)


def test_shards_merge_to_whole(product):
    """per-shard CSRs (disjoint ascending column ranges) interleave into the whole matrix."""
    from kmers_amd.distributed import merge_csr_parts
    iso = os.path.join(DATA_PATH, "syn_small.fsa")
    K = 8
    rc, data, row, col, n_cols = product.x([iso], K)
    parts = []
    for s in range(3):
        st = Store(iso, MODE_X, shard=s, n_shards=3)
        ds = st.to_device()
        buf = ds.x_buffers(K, st.n_windows(K))
        nnz = ds.build_x(buf)
        parts.append((buf["data"][:nnz].cpu().numpy(), buf["row_ind"].cpu().numpy().view(np.uint64),
                      buf["col_ind"][:nnz].cpu().numpy().view(np.uint64)))
    d, r, c = merge_csr_parts(parts, 4 ** K)
    assert np.array_equal(r, row) and np.array_equal(c, col) and np.array_equal(d.view(np.uint32), data.view(np.uint32))


def test_caller_arrays_keep_what_the_call_does_not_write(product, tmp_path):
    """The file-level calls fault the caller's arrays in on threads of their own (Prefault, libkmers.cpp) while they
    parse and build: whatever the arrays hold must survive -- X entries at and beyond nnz keep the caller's values, and y
    is ADDED to counts that are not zero (reference src/libkmers.cpp:213-216)."""
    import ctypes as C
    from kmers_amd import _lib, synth
    iso = str(tmp_path / "t.fsa")
    synth.synth_transcriptome(iso, n_records=400, total_bases=900_000, seed=21)
    K = 7
    rc, data, row, col, n_cols = product.x([iso], K)
    assert rc == 0
    nnz = data.shape[0]
    max_nz = os.path.getsize(iso)
    lib = _lib.handle()
    rng = np.random.default_rng(3)
    d2 = rng.random(max_nz, dtype=np.float32) + 2.0
    c2 = rng.integers(1 << 40, 1 << 41, size=max_nz, dtype=np.uint64)
    r2 = np.full(4 ** K + 1, 77, np.uint64)
    keep_d, keep_c = d2.copy(), c2.copy()
    n, ncols = C.c_uint64(0), C.c_int(0)
    for _ in range(3):      # (the touching threads race with the copy: more than one attempt at catching a lost store)
        assert lib.fasta_to_kmers_csr(1, _lib.cfiles([iso]), K, 0, d2.ctypes.data_as(C.POINTER(C.c_float)),
                                      r2.ctypes.data_as(C.POINTER(C.c_uint64)), c2.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      max_nz, C.byref(n), C.byref(ncols)) == 0
        assert n.value == nnz and ncols.value == n_cols
        assert np.array_equal(d2[:nnz].view(np.uint32), data.view(np.uint32)) and np.array_equal(c2[:nnz], col)
        assert np.array_equal(r2, row)
        assert np.array_equal(d2[nnz:], keep_d[nnz:]) and np.array_equal(c2[nnz:], keep_c[nnz:])
    for Ky in (8, 12):
        rc, y = product.y([iso], Ky)
        assert rc == 0
        start = rng.integers(-5, 1000, size=4 ** Ky).astype(np.int32)
        acc = start.copy()
        for rep in (1, 2):
            rc, _ = product.y([iso], Ky, into=acc)
            assert rc == 0 and np.array_equal(acc, start + rep * y)


def test_stage_record_of_the_file_level_calls(product, tmp_path):
    """kmers_last_call_stages after fasta_to_kmers_csr / fasta_count_kmers: the four stages, in call order."""
    from kmers_amd import _lib, synth
    iso = str(tmp_path / "t.fsa")
    synth.synth_transcriptome(iso, n_records=50, total_bases=60_000, seed=2)
    assert product.x([iso], 6)[0] == 0
    assert list(_lib.last_call_stages()) == ["host ingest", "H2D + device buffers", "build X (GPU)", "D2H into the host arrays"]
    assert product.y([iso], 10)[0] == 0
    st = _lib.last_call_stages()
    assert list(st) == ["host ingest", "H2D + device buffers", "histogram (GPU, enqueue)", "counts += (D2H)"]
    assert all(v >= 0.0 for v in st.values())


@pytest.mark.parametrize("K,L,cat", [(8, 0, False), (12, 0, False), (6, 1500, True)])
def test_transcriptome_to_x_takes_int32_indices_from_the_library(product, tmp_path, K, L, cat):
    """transcriptome_to_x asks for int32 indices (kmers_fasta_to_kmers_csr_i32) instead of letting scipy narrow the ABI's
    uint64 arrays: the matrix and the .npz must be what the uint64 route gives (reference
    kmerexpr/transcriptome_reader.py:33-39: csr_matrix of the ABI arrays, save_npz uncompressed)."""
    from scipy.sparse import csr_matrix, save_npz
    from kmers_amd import synth
    iso, iso2 = str(tmp_path / "t.fsa"), str(tmp_path / "u.fsa")
    synth.synth_transcriptome(iso, n_records=300, total_bases=500_000, seed=4)
    synth.synth_transcriptome(iso2, n_records=20, total_bases=30_000, seed=5)
    files = [iso, iso2]
    res = libkmers.fasta_to_kmers_csr_i32(files, K, L=L, concatenate_subseq=cat)
    assert res is not None and res[1].dtype == np.int32 and res[2].dtype == np.int32
    rc, data, row, col, n_cols = product.x(files, K, L, cat)
    assert rc == 0 and res[3] == n_cols
    assert np.array_equal(res[0].view(np.uint32), data.view(np.uint32))
    assert np.array_equal(res[1], row.astype(np.int32)) and np.array_equal(res[2], col.astype(np.int32))
    ours = str(tmp_path / "x.npz")
    x = tr.transcriptome_to_x(K, files, ours, L=L, concatenate_subseq=cat)
    want = csr_matrix((data, col, row), shape=(4 ** K, n_cols), dtype=np.float32)
    assert x.indices.dtype == want.indices.dtype == np.int32 and x.shape == want.shape
    assert np.array_equal(x.indptr, want.indptr) and np.array_equal(x.indices, want.indices)
    assert np.array_equal(x.data.view(np.uint32), want.data.view(np.uint32))
    theirs = str(tmp_path / "theirs.npz")
    save_npz(theirs, want, compressed=False)
    a, b = np.load(ours), np.load(theirs)
    for k in b.files:
        assert a[k].dtype == b[k].dtype and np.array_equal(a[k], b[k]), k
