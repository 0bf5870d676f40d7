"""The checker of tests/test_model_gpu.py checked: oracle/logp_grad.py (float64 restatement of the reference's
logp_grad_ref, kmerexpr/utils.py:46-73) on the reference's own known-answer fixture and against finite differences."""
import os

import numpy as np
from scipy.sparse import csr_matrix

from oracle import logp_grad as orc
from tests import kmtest


def x_from_oracle(oracle, path, K):
    rc, data, row_ind, col_ind, n_cols = oracle.x([path], K)
    assert rc == 0
    return csr_matrix((data, col_ind.astype(np.int64), row_ind.astype(np.int64)), shape=(4 ** K, n_cols))


def test_setup_and_known_answer(oracle):
    """test1.fsa at K=1 (reference test/test_transcriptome_reader.py:12-30): columns .25 x 4 and .4 .2 .2 .2."""
    x = x_from_oracle(oracle, os.path.join(kmtest.INP, "ref_test1.fsa"), 1)
    y = np.array([3, 0, 1, 4], np.int32)
    rows, xnnz, ynnz = orc.model_setup(x, y)
    assert rows.tolist() == [0, 2, 3] and xnnz.shape == (3, 2) and np.allclose(ynnz, [3 / 8, 1 / 8, 4 / 8])
    theta = np.array([0.25, 0.75])
    val, grad, xt = orc.logp_grad_ref(theta, 1.0, xnnz, ynnz, np.ones(2))
    col = np.asarray(x.todense(), np.float64)
    want_xt = col[[0, 2, 3]] @ theta
    assert np.allclose(xt, want_xt) and np.isclose(val, ynnz @ np.log(want_xt))
    assert np.allclose(grad, (ynnz / want_xt) @ col[[0, 2, 3]])


def test_gradient_matches_finite_differences(oracle):
    x = x_from_oracle(oracle, os.path.join(kmtest.INP, "ref_test4.fsa"), 2)
    rng = np.random.default_rng(1)
    y = rng.integers(0, 5, x.shape[0]).astype(np.int32)
    y[np.asarray(x.sum(axis=1)).ravel() == 0] = 0          # keep log finite
    rows, xnnz, ynnz = orc.model_setup(x, y)
    T = x.shape[1]
    lengths = rng.integers(10, 50, T).astype(np.float64)
    theta = rng.dirichlet(np.ones(T)).astype(np.float32).astype(np.float64)
    for beta in (1.0, 1.7):
        val, grad, _ = orc.logp_grad_ref(theta, beta, xnnz, ynnz, lengths)
        assert np.isfinite(val)
        for j in range(T):
            e = np.zeros(T)
            e[j] = 2.0 ** -9                                  # exactly representable steps survive the float32 cast
            fd = (orc.logp_grad_ref(theta + e, beta, xnnz, ynnz, lengths, nograd=True) -
                  orc.logp_grad_ref(theta - e, beta, xnnz, ynnz, lengths, nograd=True)) / (2 * e[j])
            assert abs(fd - grad[j]) < 2e-3 * max(1.0, abs(grad[j]))


def test_model_has_no_cpu_fallback(oracle):
    """Without a GPU the model refuses to be built (no silent scipy path behind the reference's class name)."""
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from kmers_amd.multinomial_simplex_model import multinomial_simplex_model
    x = x_from_oracle(oracle, os.path.join(kmtest.INP, "ref_test1.fsa"), 1)
    with pytest.raises((RuntimeError, AssertionError)):
        multinomial_simplex_model(x, np.array([3, 0, 1, 4], np.int32))
