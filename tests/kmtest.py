"""Shared helpers for the test-suite: ctypes bindings of the oracle and of the product C-ABI library,
golden-fixture access, and CSR <-> triple conversion.  Test infrastructure only."""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
INP = os.path.join(GOLD, "inputs")
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libkmers_oracle.so")
PRODUCT_SO = os.path.join(ROOT, "kmers_amd", "lib", "libkmers.so")

X_ARGS = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64),
          C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
Y_ARGS = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int)]


def cfiles(paths):
    arr = (C.c_char_p * len(paths))()
    arr[:] = [p.encode() for p in paths]
    return arr


class Lib:
    """Uniform view of a library exporting the six reference symbols under an optional prefix."""

    def __init__(self, path, prefix="", hip=True):
        if not prefix and hip:                     # hip=False: a CPU library (the compiled reference)
            from kmers_amd import _lib
            _lib._share_hip_runtime_with_torch()   # one HIP runtime per process (see kmers_amd/_lib.py)
        self.lib = C.CDLL(path)
        self.prefix = prefix
        for name in ("fasta_to_kmers_csr", "fasta_to_kmers_csr_cat_subseq"):
            f = getattr(self.lib, prefix + name)
            f.restype = C.c_int
            f.argtypes = X_ARGS
        for name in ("fasta_count_kmers", "fastq_gz_count_kmers"):
            f = getattr(self.lib, prefix + name)
            f.restype = C.c_int
            f.argtypes = Y_ARGS
        f = getattr(self.lib, prefix + "kmer_to_id")
        f.restype = C.c_uint32
        f.argtypes = [C.c_char_p, C.c_int]
        f = getattr(self.lib, prefix + "valid_kmer")
        f.restype = C.c_bool
        f.argtypes = [C.c_char_p, C.c_int]

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def x(self, paths, K, L=0, cat=False, max_nz=None):
        """-> (rc, data[:nnz], row_ind, col_ind[:nnz], n_cols)"""
        M = 4 ** K
        if max_nz is None:
            max_nz = sum(os.stat(p).st_size for p in paths if os.path.isfile(p)) + 16
        data = np.zeros(max_nz, np.float32)
        col = np.zeros(max_nz, np.uint64)
        row = np.zeros(M + 1, np.uint64)
        nnz = C.c_uint64(0)
        ncols = C.c_int(0)
        fn = self.fn("fasta_to_kmers_csr_cat_subseq" if cat else "fasta_to_kmers_csr")
        rc = fn(len(paths), cfiles(paths), K, L, data.ctypes.data_as(C.POINTER(C.c_float)),
                row.ctypes.data_as(C.POINTER(C.c_uint64)), col.ctypes.data_as(C.POINTER(C.c_uint64)), max_nz,
                C.byref(nnz), C.byref(ncols))
        n = nnz.value
        return rc, data[:n], row, col[:n], ncols.value

    def y(self, paths, K, fastq=False, into=None):
        M = 4 ** K
        y = np.zeros(M, np.int32) if into is None else into
        fn = self.fn("fastq_gz_count_kmers" if fastq else "fasta_count_kmers")
        rc = fn(len(paths), cfiles(paths), K, y.ctypes.data_as(C.POINTER(C.c_int)))
        return rc, y


def load_oracle():
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ROOT, "oracle", "kmers_oracle.c")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    return Lib(ORACLE_SO, "oracle_")


def load_product():
    if not os.path.exists(PRODUCT_SO):
        raise RuntimeError(f"{PRODUCT_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    return Lib(PRODUCT_SO, "")


def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


def case_paths(case):
    return [os.path.join(INP, f) for f in case["files"]]


def golden(case):
    return np.load(os.path.join(GOLD, case["name"] + ".npz"))


def csr_rows(row_ind):
    """Expand a CSR row pointer to one row id per stored entry (uint32)."""
    cnt = np.diff(row_ind)
    nzr = np.flatnonzero(cnt)
    return np.repeat(nzr.astype(np.uint32), cnt[nzr].astype(np.int64))


def assert_x_equal_golden(case, rc, data, row_ind, col_ind, n_cols):
    g = golden(case)
    assert rc == 0
    assert n_cols == int(g["n_cols"])
    assert data.shape[0] == int(g["nnz"])
    assert int(row_ind[0]) == 0 and int(row_ind[-1]) == int(g["nnz"])
    assert np.array_equal(csr_rows(row_ind), g["rows"])
    assert np.array_equal(col_ind.astype(np.uint32), g["cols"])
    # float32 values must be bit-identical
    assert np.array_equal(data.view(np.uint32), g["data"].view(np.uint32))


def assert_y_equal_golden(case, rc, y):
    g = golden(case)
    assert rc == 0
    idx = np.flatnonzero(y).astype(np.uint32)
    assert np.array_equal(idx, g["idx"])
    assert np.array_equal(y[idx], g["cnt"])


def assert_x_equal(a, b):
    """a, b: (rc, data, row_ind, col_ind, n_cols) tuples from Lib.x"""
    assert a[0] == b[0]
    assert a[4] == b[4]
    assert np.array_equal(a[2], b[2])
    assert np.array_equal(a[3], b[3])
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


# ---- inputs of the fasta.py-rule reader (kmerexpr/fasta.py:19-46), shared by oracle/make_golden_py.py and the tests

def awkward_fasta(path, seed=0):
    """A FASTA on which the reference's two readers differ (src/libkmers.cpp:302-318 vs kmerexpr/fasta.py:19-46): text
    before the first header, CRLF and lone-CR line ends, blank and whitespace-only lines, leading / trailing / inner
    blanks of every kind python strips, an empty later header, a NUL inside a sequence."""
    rng = np.random.default_rng(seed)
    rnd = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n))
    b = bytearray()
    b += b"some text before the first header\nACGTACGTACGT\n\n"
    b += b">r1 crlf lines\r\n" + rnd(70) + b"\r\n" + rnd(70) + b"\r\n  " + rnd(33) + b" \t\r\n\r\n"
    b += b">r2 PREDICTED: dropped\n" + rnd(300) + b"\n"
    b += b">r3 lone cr\r" + rnd(120) + b"\r" + rnd(55) + b"\r"
    b += b">r4 header ends with a lone cr\r" + rnd(90) + b"\n\n   \n\t\n" + rnd(90) + b"\n"
    b += b">r5 inner blanks\n\t" + rnd(40) + b" " + rnd(40) + b"\x0b" + rnd(30) + b"\x0c\n \x1c\x1d" + rnd(64) + b"\x1f\n"
    b += b">\n" + rnd(200) + b"\n"                                     # an empty header that is not the first one
    b += b">r7 nul inside\n" + rnd(50) + b"\x00" + rnd(150) + b"\n"
    b += b">r8 short\nACGT\n"
    b += b">r9 N and lower case\n" + rnd(60) + b"NNNnnacgt" + rnd(120) + b"   \n" + rnd(30)     # no newline at the end
    with open(path, "wb") as f:
        f.write(bytes(b))
    return path


def fuzz_fasta_bytes(seed):
    """Random text from an alphabet of bases, '>', every line end and every blank python strips."""
    rng = np.random.default_rng(900 + seed)
    alphabet = [b"A", b"C", b"G", b"T", b"N", b"a", b">", b"\n", b"\r", b"\r\n", b" ", b"\t", b"\x0b", b"\x0c", b"\x1c", b"\x1f",
                b"\x00", b"PREDICTED", b">h", b"\n>rec ", b"ACGTACGTAC"]
    weights = np.array([8, 8, 8, 8, 1, 1, 0.3, 2, 0.7, 0.7, 1, 0.5, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.5, 1.5, 6], float)
    text = b"".join(alphabet[i] for i in rng.choice(len(alphabet), int(rng.integers(0, 3000)), p=weights / weights.sum()))
    if rng.random() < 0.5:
        text = b">first\n" + text
    return text


FIRST_HEADER_CASES = (b">\nACGT\n>b\nACGT\n", b">  \t \r\nACGT\n>b\nACGT\n", b"", b"no header at all\nACGT\n",
                      b"x>a\nACGT\n", b"\n\n>a\nAC\n")


def write_fasta_py_inputs(directory):
    """tests/golden/inputs/py_*.fsa: the files oracle/make_golden_py.py runs the reference's python reader on."""
    made = []
    for seed in (0, 3):
        made.append(awkward_fasta(os.path.join(directory, f"py_awkward_s{seed}.fsa"), seed))
    for seed in range(12):
        p = os.path.join(directory, f"py_fuzz_{seed:02d}.fsa")
        with open(p, "wb") as f:
            f.write(fuzz_fasta_bytes(seed))
        made.append(p)
    for i, text in enumerate(FIRST_HEADER_CASES):
        p = os.path.join(directory, f"py_first_{i}.fsa")
        with open(p, "wb") as f:
            f.write(text)
        made.append(p)
    return made


# ---- the oracle as the per-rank engine of the multi-process CPU tests (tests/dist_worker.py, dist_model_worker.py): the
# workers patch kmers_amd.store.Store.to_device with this, so that the product's sharded flows (ingest shards,
# collectives, CSR placement) run unchanged on a box without a GPU.  Test infrastructure only.

class OracleDeviceStore:
    def __init__(self, store, oracle, tmpdir, tag):
        self.st, self.oracle = store, oracle
        self.n_pos, self.n_rec, self.first_rec, self.n_rec_total = store.n_pos, store.n_rec, store.first_rec, store.n_rec_total
        self.path = os.path.join(tmpdir, f"shard_{tag}_{id(store)}.fsa")
        with open(self.path, "wb") as f:
            for i, q in enumerate(store.sequences()):
                f.write(b">r%d\n" % i + q + b"\n")

    def count_y(self, K, counts=None, stream=None, workspace=None):
        import torch
        rc, y = self.oracle.y([self.path], K)
        assert rc == 0
        return torch.from_numpy(y.copy())

    def x_buffers(self, K, max_nz):
        return dict(K=K, max_nz=max_nz)

    def build_x(self, buf, stream=None):
        import torch
        rc, data, row, col, n_cols = self.oracle.x([self.path], buf["K"], max_nz=self.n_pos + 16)
        assert rc == 0 and n_cols == self.n_rec
        buf["data"] = torch.from_numpy(np.ascontiguousarray(data, np.float32))
        buf["col_ind"] = torch.from_numpy((col + np.uint64(self.first_rec)).view(np.int64).copy())
        buf["row_ind"] = torch.from_numpy(np.ascontiguousarray(row, np.uint64).view(np.int64).copy())
        return int(data.shape[0])


def patch_oracle_engine(tmpdir, tag):
    """Make Store.to_device return the oracle-backed stand-in (CPU workers only)."""
    from kmers_amd import store as _store
    oracle = load_oracle()
    _store.Store.to_device = lambda self, device="cuda": OracleDeviceStore(self, oracle, tmpdir, tag)
    return oracle
