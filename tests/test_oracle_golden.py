"""The oracle (oracle/kmers_oracle.c) against the fixtures produced by the unmodified reference library
(tests/golden/, written by oracle/make_golden.py) and against the known answers of the reference's own
tests (reference test/test_transcriptome_reader.py:12-64).  CPU only."""
import json
import os

import numpy as np
import pytest

from tests import kmtest

CASES = kmtest.manifest()


@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] == "x"], ids=lambda c: c["name"])
def test_oracle_x_matches_reference(oracle, case):
    out = oracle.x(kmtest.case_paths(case), case["K"], case["L"], case["cat"])
    kmtest.assert_x_equal_golden(case, *out)


@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] != "x"], ids=lambda c: c["name"])
def test_oracle_y_matches_reference(oracle, case):
    rc, y = oracle.y(kmtest.case_paths(case), case["K"], fastq=(case["kind"] == "y_fastq"))
    kmtest.assert_y_equal_golden(case, rc, y)


def test_oracle_scalars(oracle):
    with open(os.path.join(kmtest.GOLD, "scalars.json")) as f:
        scal = json.load(f)
    for w, (kid, ok) in scal.items():
        assert oracle.fn("valid_kmer")(w.encode(), len(w)) == ok
        assert oracle.fn("kmer_to_id")(w.encode(), len(w)) == kid


# ---- the reference's own assertions, restated on the oracle (test_transcriptome_reader.py) -------------

def _dense(out, K):
    from scipy.sparse import csr_matrix
    rc, data, row, col, n_cols = out
    assert rc == 0
    return csr_matrix((data, col.astype(np.int64), row.astype(np.int64)), shape=(4 ** K, n_cols)).toarray()


def test_k1_known_answer(oracle):      # reference test_k1
    x = _dense(oracle.x([os.path.join(kmtest.INP, "ref_test1.fsa")], 1), 1)
    assert x.shape == (4, 2)
    assert np.count_nonzero(x) == 8
    assert all(x[i, 0] == 0.25 for i in range(4))
    assert x[0, 1] == pytest.approx(0.40, 1e-6)
    assert all(x[i, 1] == pytest.approx(0.20, 1e-6) for i in (1, 2, 3))


def test_k2_known_answer(oracle):      # reference test_k2
    x = _dense(oracle.x([os.path.join(kmtest.INP, "ref_test2.fsa")], 2), 2)
    assert x.shape == (16, 3)
    assert np.count_nonzero(x) == 7
    for r, c in ((0, 0), (1, 0), (5, 0), (6, 1), (10, 1)):
        assert x[r, c] == pytest.approx(1.0 / 3.0, 1e-6)
    assert x[15, 2] == 1.0


def test_simplex_property(oracle):     # reference test_simplex
    x = _dense(oracle.x([os.path.join(kmtest.INP, "ref_test2.fsa")], 2), 2)
    rng = np.random.default_rng(0)
    theta = np.abs(rng.standard_normal(x.shape[1]).astype(np.float32))
    theta /= theta.sum()
    assert float((x @ theta).sum()) == pytest.approx(1.0, 1e-5)


def test_oracle_errors(oracle, tmp_path):
    missing = str(tmp_path / "nope.fsa")
    assert oracle.x([missing], 2)[0] == -1
    assert oracle.y([missing], 2)[0] == -1
    # concatenate mode silently keeps an empty column for an unopenable file (libkmers.cpp:402-403, :428)
    rc, data, row, col, n_cols = oracle.x([missing, os.path.join(kmtest.INP, "edge_two_a.fsa")], 2, cat=True)
    assert rc == 0 and n_cols == 2 and set(col.tolist()) == {1}
    # max_nz too small -> -2
    assert oracle.x([os.path.join(kmtest.INP, "ref_test4.fsa")], 2, max_nz=3)[0] == -2


def test_oracle_short_records_defined(oracle, tmp_path):
    """Records shorter than K-1 (no reference answer: it reads out of bounds): zero k-mers, column kept."""
    p = tmp_path / "short.fsa"
    p.write_text(">a\nAC\n>b\n\n>c\nACGTACGT\n>d\nA\n")
    rc, data, row, col, n_cols = oracle.x([str(p)], 5)
    assert rc == 0 and n_cols == 4
    assert set(col.tolist()) == {2} and data.shape[0] == 4
    rc, y = oracle.y([str(p)], 5)
    assert rc == 0 and int(y.sum()) == 4
