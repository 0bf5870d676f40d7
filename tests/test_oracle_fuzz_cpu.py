"""The oracle against the compiled reference (oracle/_ref, built by `make -C oracle ref` where /root/reference exists)
on the fuzz inputs of tests/test_fuzz_gpu.py, so that the GPU fuzz test's checker is itself checked beyond the golden
fixtures.  The reference reads out of bounds on records shorter than K-1: every case runs
in a forked child and a child that dies is skipped -- the oracle's definition for those inputs has no reference
answer.  Skipped entirely where oracle/_ref is absent (the GPU box)."""
import multiprocessing as mp
import os

import numpy as np
import pytest

from tests import kmtest
from tests.test_fuzz_gpu import _fastq, _file

REF_SO = os.path.join(kmtest.ROOT, "oracle", "_ref", "libkmers_ref.so")
pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built here")


def _child(conn, files, K, L, cat, fastq):
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)                                  # the reference prints timings
    ref = kmtest.Lib(REF_SO, "", hip=False)
    out = {}
    if not fastq:
        out["x"] = ref.x(files, K, L, cat)
    out["y"] = ref.y(files, K, fastq=fastq)
    conn.send(out)
    conn.close()
    os._exit(0)                                          # no pytest / interpreter teardown in the child


def _reference(files, K, L=0, cat=False, fastq=False):
    ctx = mp.get_context("fork")
    a, b = ctx.Pipe(duplex=False)
    p = ctx.Process(target=_child, args=(b, files, K, L, cat, fastq))
    p.start()
    b.close()
    try:
        out = a.recv() if a.poll(60) else None
    except EOFError:                                      # the child died before answering
        out = None
    p.join(10)
    if p.is_alive():
        p.kill()
    return out if p.exitcode == 0 else None


@pytest.mark.parametrize("seed", range(60))
def test_oracle_equals_reference_fasta(oracle, tmp_path, seed):
    rng = np.random.default_rng(1000 + seed)              # test_fuzz_gpu's generator, records of >= 12 bases
    files = []
    for i in range(int(rng.integers(1, 4))):
        p = str(tmp_path / f"f{i}.fsa")
        _file(rng, p, min_len=12)
        files.append(p)
    K = int(rng.integers(1, 11))
    L = int(rng.choice([0, 0, 1, K - 1, K, K + 3, 60]))
    cat = bool(rng.random() < 0.3)
    want = _reference(files, K, L, cat)
    if want is None:
        pytest.skip("the reference itself crashes on this input (record shorter than K-1)")
    kmtest.assert_x_equal(oracle.x(files, K, L, cat), want["x"])
    rc, y = oracle.y(files, K)
    assert rc == want["y"][0] == 0 and np.array_equal(y, want["y"][1])


@pytest.mark.parametrize("seed", range(15))
def test_oracle_equals_reference_fastq(oracle, tmp_path, seed):
    rng = np.random.default_rng(5000 + seed)
    files = []
    for i in range(int(rng.integers(1, 3))):
        p = str(tmp_path / f"q{i}.fastq")
        _fastq(rng, p)
        files.append(p)
    K = int(rng.integers(1, 10))
    want = _reference(files, K, fastq=True)
    if want is None:
        pytest.skip("the reference itself crashes on this input")
    rc, y = oracle.y(files, K, fastq=True)
    assert rc == want["y"][0] == 0 and np.array_equal(y, want["y"][1])
