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
    from kmers_amd.store import MODE_X, Store
    want = orc.simulate_reads(N, L, alpha, str(tmp_path / "reads.fsa"), iso)
    st = Store(iso, MODE_X, L=L)
    got = sample_reads(np.diff(st.rec_start.astype(np.int64)) - 1, N, L, alpha)
    for k in ("psi", "theta_true", "theta_sampled", "lengths", "iso_sampled"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["starts"].astype(np.int64), want["starts"])
    if L == 80:
        assert (want["lengths"] == 1).any()                 # the 80-base record: choice(1) draws nothing


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
