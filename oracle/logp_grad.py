"""TEST INFRASTRUCTURE ONLY -- the checker for kmers_hip_model_* (SURVEY.md section 8f row 4); never the product.

numpy / scipy float64 restatement of the reference's model set-up and logp_grad:
  * model_setup:   kmerexpr/multinomial_simplex_model.py:56-62  (ymask, xnnz = x[ymask], ynnz = y[ymask] / sum(y))
  * logp_grad_ref: kmerexpr/utils.py:46-73 (the reference's own plain numpy statement of utils.py:76-108), with
                   theta cast to float32 first as utils.py:84 does.
PARITY UNPINNED by reference fixtures: the reference's tests for this function (test/test_multinomial_simplex_model.py)
cannot run as shipped (missing test/conftest.py, MKL and numba absent here), and kmerexpr.utils
cannot be imported here for the same reason (its module-level `import numba` / `sparse_dot_mkl` raise ImportError).
The tests therefore compare against this float64 restatement with a stated tolerance and check the properties the
reference's tests name: sum(X . theta) == 1 on the simplex (test_transcriptome_reader.py:52-64) and the gradient
against finite differences (test_multinomial_simplex_model.py's check_gradient)."""
import numpy as np


def model_setup(x, y):
    ymask = y.nonzero()
    xnnz = x.tocsr()[ymask].astype(np.float64)
    ynnz = y[ymask]
    ynnz = ynnz / np.sum(ynnz)
    return ymask[0], xnnz, ynnz


def logp_grad_ref(theta, beta, xnnz, ynnz, lengths, nograd=False):
    theta = np.asarray(theta).astype(np.float32).astype(np.float64)
    mask = theta > 0
    thetamask = theta[mask]
    # the reference keeps xtheta, then ynnz / xtheta, in its float32 scratch array (utils.py:87-92,101-102)
    xthetannz = xnnz.dot(theta).astype(np.float32).astype(np.float64)
    with np.errstate(divide="ignore"):
        val = ynnz.dot(np.log(xthetannz))
    val += (beta - 1.0) * np.sum(np.log(thetamask / lengths[mask]))
    val -= (beta - 1.0) * np.log(np.sum(thetamask / lengths[mask]))
    if nograd:
        return val
    with np.errstate(divide="ignore", over="ignore"):
        yxTtheta = (ynnz / xthetannz).astype(np.float32).astype(np.float64)
    gradient = yxTtheta @ xnnz
    gradient[mask] += (beta - 1.0) / thetamask
    gradient[mask] -= (beta - 1.0) / (np.sum(thetamask / lengths[mask]) * lengths[mask])
    return val, gradient, xthetannz
