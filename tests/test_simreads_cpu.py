"""kmers_amd.simulate_reads.sample_reads (numpy's legacy calls + the C replay of the per-read starts) against the
restated reference simulator (oracle/reads_oracle.py): identical psi, theta, isoforms and starts."""
import os

import numpy as np
import pytest

from kmers_amd import synth
from oracle import reads_oracle as orc
from tests import kmtest


@pytest.fixture(scope="module")
def iso(tmp_path_factory):
    d = tmp_path_factory.mktemp("sim")
    p = str(d / "iso.fsa")
    synth.synth_transcriptome(p, n_records=300, total_bases=300_000, seed=4)
    with open(p, "a") as f:                                  # records the filters drop, and one with a single start
        f.write(">x PREDICTED: dropped\n" + "ACGT" * 100 + "\n>short\nACGTACGT\n>exact\n" + "ACGTTGCA" * 10 + "\n")
    return p


@pytest.mark.parametrize("N,L,alpha", [(2000, 80, 0.1), (500, 100, 1.0), (3000, 14, 0.5)])
def test_sampler_equals_reference_calls(iso, tmp_path, N, L, alpha):
    from kmers_amd.simulate_reads import sample_reads
    from kmers_amd.store import MODE_FASTA_PY, Store
    want = orc.simulate_reads(N, L, alpha, str(tmp_path / "reads.fsa"), iso)
    st = Store(iso, MODE_FASTA_PY, L=L)
    got = sample_reads(np.diff(st.rec_start.astype(np.int64)) - 1, N, L, alpha)
    for k in ("psi", "theta_true", "theta_sampled", "lengths", "iso_sampled"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["starts"].astype(np.int64), want["starts"])
    if L == 80:
        assert (want["lengths"] == 1).any()                 # the 80-base record: choice(1) draws nothing


awkward_fasta = kmtest.awkward_fasta


def _norm(text):
    return bytes(c if c in b"ACGT" else ord("N") for c in text.encode("latin-1"))


@pytest.mark.parametrize("L", [0, 50, 100, 250, -3])
def test_isoform_list_follows_fasta_py(tmp_path, L):
    """Store(MODE_FASTA_PY) == the records of the restated kmerexpr/fasta.py reader + simulate_reads' filters."""
    from kmers_amd.store import MODE_FASTA_PY, MODE_X, Store
    p = awkward_fasta(str(tmp_path / "awk.fsa"))
    want = [_norm(q) for _, q in orc.kept_isoforms(p, L)]
    st = Store(p, MODE_FASTA_PY, L=L)
    assert st.sequences() == want
    if L == 0:
        assert len(want) == 8 and len(want[0]) == 173
        cpp = Store(p, MODE_X, L=0).sequences()              # the C++ reader's view of the same file is another list
        assert cpp != want and cpp[0].startswith(b"ACGTACGTACGT")
    for shards in (2, 3):
        parts = [Store(p, MODE_FASTA_PY, L=L, shard=r, n_shards=shards) for r in range(shards)]
        assert sum((q.sequences() for q in parts), []) == want


@pytest.mark.parametrize("seed", range(12))
def test_isoform_list_fuzz(tmp_path, seed):
    """Random files from an alphabet of bases, '>' , every line end and every blank python strips: the store under the
    fasta.py rules == the restated fasta.py reader, whole and in byte-range shards."""
    from kmers_amd.store import MODE_FASTA_PY, Store
    text = kmtest.fuzz_fasta_bytes(seed)
    p = tmp_path / "f.fsa"
    p.write_bytes(text)
    for L in (0, 7, 40):
        want = [_norm(q) for _, q in orc.kept_isoforms(str(p), L)]
        assert Store(str(p), MODE_FASTA_PY, L=L).sequences() == want, (seed, L)
        for n in (2, 5):
            got = sum((Store(str(p), MODE_FASTA_PY, L=L, shard=r, n_shards=n, by_bytes=True).sequences() for r in range(n)), [])
            assert got == want, (seed, L, n)


def test_first_header_empty_means_no_records(tmp_path):
    from kmers_amd.store import MODE_FASTA_PY, Store
    for text in (b">\nACGT\n>b\nACGT\n", b">  \t \r\nACGT\n>b\nACGT\n", b"", b"no header at all\nACGT\n", b"x>a\nACGT\n"):
        p = tmp_path / "e.fsa"
        p.write_bytes(text)
        assert orc.kept_isoforms(str(p), 0) == []
        assert Store(str(p), MODE_FASTA_PY).n_rec == 0
    p.write_bytes(b"\n\n>a\nAC\n")                      # the C++ reader stops at an empty first line; fasta.py does not
    assert Store(str(p), MODE_FASTA_PY).sequences() == [b"AC"] == [_norm(q) for _, q in orc.kept_isoforms(str(p), 0)]


def test_non_ascii_is_refused(tmp_path, capfd):
    from kmers_amd.store import MODE_FASTA_PY, MODE_X, Store
    p = tmp_path / "u.fsa"
    p.write_bytes(">a caf\xc3\xa9\nACGT\n".encode("latin-1"))
    with pytest.raises(Exception):
        Store(str(p), MODE_FASTA_PY)
    assert "0x80" in capfd.readouterr().err
    assert Store(str(p), MODE_X).n_rec == 1                 # the C++ reader's rules take any byte


def test_sampler_on_the_awkward_file(tmp_path):
    from kmers_amd.simulate_reads import sample_reads
    from kmers_amd.store import MODE_FASTA_PY, Store
    p = awkward_fasta(str(tmp_path / "awk.fsa"), seed=3)
    N, L = 700, 100
    want = orc.simulate_reads(N, L, 0.4, str(tmp_path / "reads.fsa"), p)
    st = Store(p, MODE_FASTA_PY, L=L)
    got = sample_reads(np.diff(st.rec_start.astype(np.int64)) - 1, N, L, 0.4)
    for k in ("psi", "theta_true", "theta_sampled", "lengths", "iso_sampled"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["starts"].astype(np.int64), want["starts"])


def test_starts_replay_resumes_across_blocks():
    """kmers_sim_starts stops when its block of raw outputs runs out inside a read; the caller continues with the
    unused outputs and a new block, and the result does not depend on the block size."""
    import ctypes as C
    from kmers_amd import _lib
    lib = _lib.handle()
    rng = np.random.RandomState(7)
    n = 5000
    n_choices = rng.randint(1, 3000, size=n).astype(np.uint32)
    raw = rng.randint(0, 2 ** 32, size=4 * n, dtype=np.uint32)
    def run(block):
        out = np.empty(n, np.uint32)
        done = used = 0
        while done < n:
            chunk = raw[used:used + block]
            rd, ru = C.c_uint64(0), C.c_uint64(0)
            assert lib.kmers_sim_starts(chunk.ctypes.data, chunk.size, n_choices[done:].ctypes.data, n - done,
                                        out[done:].ctypes.data, C.byref(rd), C.byref(ru)) == 0
            assert rd.value > 0 or chunk.size < 64
            done += int(rd.value)
            used += int(ru.value)
            if rd.value == 0:
                block *= 2
        return out, used
    a, ua = run(4 * n)
    b, ub = run(97)
    assert np.array_equal(a, b) and ua == ub
    assert (a < n_choices).all() and (a[n_choices == 1] == 0).all()
