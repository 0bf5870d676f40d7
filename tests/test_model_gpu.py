"""logp_grad's two sparse products on the resident X (kmers_hip_model_* of
include/kmers_hip.h) against the float64 restatement of the reference's logp_grad_ref (oracle/logp_grad.py,
reference kmerexpr/utils.py:46-73).

Tolerance (floating point, so stated here): the kernels accumulate in float64 and round to float32 once, the oracle
works in float64 on the same float32 inputs (rounding xtheta and w to float32 where the reference stores them in its
float32 scratch array), so xtheta agrees to one float32 rounding (rtol 2e-7), the gradient to
the roundings of xtheta, w and the result (rtol 1e-6), and val = sum ynnz*log(xtheta) with sum ynnz = 1 to
atol 5e-7.  Parity is not pinned by reference fixtures (none exist for this function, see oracle/logp_grad.py); the
properties the reference's tests name are checked as well.  Needs a real MI355X: run with -m gpu."""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

from kmers_amd import synth
from oracle import logp_grad as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def syn(tmp_path_factory):
    d = tmp_path_factory.mktemp("model")
    iso = str(d / "iso.fsa")
    info = synth.synth_transcriptome(iso, n_records=3000, total_bases=3_000_000, seed=11, return_sequences=True)
    reads = str(d / "reads.fsa")
    synth.synth_reads(reads, info["concat"], info["offsets"], 30000, read_len=100, seed=5)
    other = str(d / "other.fsa")
    synth.synth_transcriptome(other, n_records=40, total_bases=40_000, seed=99)
    return dict(iso=iso, reads=reads, other=other)


def xy(syn, K, reads="reads"):
    from kmers_amd.libkmers import fasta_count_kmers, fasta_to_kmers_csr
    data, row_ind, col_ind, n_cols = fasta_to_kmers_csr(syn["iso"], K)
    x = csr_matrix((data, col_ind.astype(np.int64), row_ind.astype(np.int64)), shape=(4 ** K, n_cols), dtype=np.float32)
    return x, fasta_count_kmers(syn[reads], K)


def thetas(T, seed):
    rng = np.random.default_rng(seed)
    yield np.ones(T) / T
    yield rng.dirichlet(np.full(T, 0.5))
    t = rng.dirichlet(np.full(T, 1.0))
    t[rng.random(T) < 0.1] = 0.0            # zeros: masked out of the beta terms, plain zeros in the products
    yield t / t.sum()


def check(model, x, y, theta, beta=1.0, lengths=None):
    rows, xnnz, ynnz = orc.model_setup(x, y)
    lengths = np.ones(x.shape[1]) if lengths is None else lengths
    want_val, want_grad, want_xt = orc.logp_grad_ref(theta, beta, xnnz, ynnz, lengths)
    val, grad = model.logp_grad(theta)
    dm = model.device_model
    assert np.array_equal(dm.rows(), rows)
    assert dm.n_masked == rows.shape[0] and dm.nnz == xnnz.nnz and dm.sum_y == int(y.sum())
    np.testing.assert_allclose(dm.xtheta(), want_xt, rtol=2e-7, atol=1e-45)    # atol: float32 subnormals
    assert grad.dtype == np.float32 and grad.shape == (x.shape[1],)
    if np.isfinite(want_val):
        np.testing.assert_allclose(val, want_val, rtol=0, atol=5e-7)
        np.testing.assert_allclose(grad, want_grad, rtol=1e-6, atol=1e-30)
    else:                                    # a read k-mer with probability 0 under theta: log(0), as in the reference
        assert val == want_val
        fin = np.isfinite(want_grad)
        assert np.array_equal(np.isfinite(grad), fin)
        np.testing.assert_allclose(grad[fin], want_grad[fin], rtol=1e-6, atol=1e-30)
    assert model.logp_grad(theta, nograd=True) == val
    val2, grad2 = model.logp_grad(theta)     # fixed summation order: the same bits on every call
    assert val2 == val and np.array_equal(grad2.view(np.uint32), grad.view(np.uint32))
    return val, grad


@pytest.mark.parametrize("K", [1, 3, 6, 8, 11, 12, 14])
def test_logp_grad_vs_oracle(syn, K):
    from kmers_amd.multinomial_simplex_model import multinomial_simplex_model
    x, y = xy(syn, K)
    model = multinomial_simplex_model(x, y)
    assert model.M == 4 ** K and model.T == x.shape[1] and model.N == y.sum()
    for theta in thetas(model.T, K):
        check(model, x, y, theta)


@pytest.mark.parametrize("K,slab_log2", [(6, 8), (11, 8), (14, 12), (14, 16)])
def test_row_slabs_of_the_transposed_copy(syn, K, slab_log2, monkeypatch):
    """With more masked rows than an L2 holds the transposed copy is sorted by (row slab, column) and the gradient is
    the sum of per-slab partials; KMERS_MODEL_SLAB_LOG2 forces small slabs (2^8 rows: hundreds of slabs) here."""
    from kmers_amd.multinomial_simplex_model import multinomial_simplex_model
    x, y = xy(syn, K)
    monkeypatch.setenv("KMERS_MODEL_SLAB_LOG2", str(slab_log2))
    model = multinomial_simplex_model(x, y)
    assert model.device_model.n_masked > 4 << slab_log2
    monkeypatch.setenv("KMERS_MODEL_SLAB_LOG2", "31")
    whole = multinomial_simplex_model(x, y)
    for theta in thetas(model.T, K):
        val, grad = check(model, x, y, theta)
        val1, grad1 = whole.logp_grad(theta)
        assert val1 == val
        np.testing.assert_allclose(grad, grad1, rtol=2e-7, atol=1e-30)   # float64 partials: one float32 rounding apart


def test_beta_terms_and_lengths(syn):
    from kmers_amd.multinomial_simplex_model import multinomial_simplex_model
    x, y = xy(syn, 7)
    lengths = np.random.default_rng(3).integers(300, 5000, x.shape[1]).astype(np.float64)
    model = multinomial_simplex_model(x, y, beta=1.5, lengths=lengths)
    for theta in thetas(model.T, 1):
        check(model, x, y, theta, beta=1.5, lengths=lengths)


def test_starts_and_fit_dispatch(syn):
    """initialize_iterates_Xy / _lsq against the reference's formulas on the host copies (multinomial_simplex_model.py:
    97-115), and fit() handing logp_grad to a solver (a stand-in module: kmerexpr is not installed here)."""
    import sys
    import types
    from scipy.sparse.linalg import lsqr
    from kmers_amd.multinomial_simplex_model import multinomial_simplex_model
    x, y = xy(syn, 6)
    model = multinomial_simplex_model(x, y)
    ymask, xnnz, ynnz = orc.model_setup(x, y)
    want = np.asarray(ynnz @ xnnz).ravel()
    np.testing.assert_allclose(model.initialize_iterates_Xy(), want / want.sum(), rtol=1e-5, atol=1e-12)
    t = lsqr(xnnz.astype(np.float32), ynnz / model.N, iter_lim=50)[0]
    t[t <= 0] = np.min(t[t > 0]) / 10
    np.testing.assert_allclose(model.initialize_iterates_lsq(iterations=50), t / t.sum(), rtol=1e-4, atol=1e-10)
    seen = {}
    def solver(fn, theta0, **kw):
        seen.update(kw, start=theta0, val=fn(theta0, nograd=True))
        return {"x": theta0}
    pkg, eg, mgm = types.ModuleType("kmerexpr"), types.ModuleType("kmerexpr.exp_grad_solver"), types.ModuleType("kmerexpr.mg")
    eg.exp_grad_solver, mgm.mg = solver, (lambda fn, theta0, tol, max_iter: solver(fn, theta0, tol=tol, max_iter=max_iter))
    old = {k: sys.modules.get(k) for k in ("kmerexpr", "kmerexpr.exp_grad_solver", "kmerexpr.mg")}
    sys.modules.update({"kmerexpr": pkg, "kmerexpr.exp_grad_solver": eg, "kmerexpr.mg": mgm})
    try:
        params = types.SimpleNamespace(init_iterates="Xy", lrs="warmstart")
        model.fit(params, n_iters=7, opt_method="exp_grad")
        assert seen["n_iters"] == 7 and seen["lrs"] == "warmstart" and np.isfinite(seen["val"])
        np.testing.assert_allclose(seen["start"], want / want.sum(), rtol=1e-5, atol=1e-12)
        model.fit(types.SimpleNamespace(init_iterates="uniform"), n_iters=3)
        assert seen["max_iter"] == 3 and np.allclose(seen["start"], 1.0 / model.T)
    finally:
        for k, v in old.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_columns_are_stochastic(syn):
    """The reference's test_simplex (test/test_transcriptome_reader.py:52-64): sum(X . theta) == 1 on the simplex.
    At K=3 every row has a read, so xnnz is all of X."""
    from kmers_amd.multinomial_simplex_model import multinomial_simplex_model
    x, y = xy(syn, 3)
    assert np.all(y > 0)
    model = multinomial_simplex_model(x, y)
    for theta in thetas(model.T, 4):
        model.logp_grad(theta, nograd=True)
        assert abs(float(model.device_model.xtheta().astype(np.float64).sum()) - 1.0) < 1e-6


def test_gradient_against_finite_differences(syn):
    """What the reference's check_gradient does (test/test_multinomial_simplex_model.py): directional derivatives of
    val against grad, here with val from the GPU as well."""
    from kmers_amd.multinomial_simplex_model import multinomial_simplex_model
    x, y = xy(syn, 8)
    model = multinomial_simplex_model(x, y)
    rng = np.random.default_rng(0)
    theta = rng.dirichlet(np.full(model.T, 1.0)) + 1e-4
    theta = (theta / theta.sum()).astype(np.float32).astype(np.float64)
    val, grad = model.logp_grad(theta)
    rows, xnnz, ynnz = orc.model_setup(x, y)
    for _ in range(3):
        d = rng.standard_normal(model.T) * theta
        eps = 1e-3
        fp = orc.logp_grad_ref(theta * (1 + 0) + eps * d, 1.0, xnnz, ynnz, model.lengths, nograd=True)
        fm = orc.logp_grad_ref(theta - eps * d, 1.0, xnnz, ynnz, model.lengths, nograd=True)
        # float32 rounding of theta +- eps d inside logp_grad_ref limits the agreement
        assert abs((fp - fm) / (2 * eps) - float(grad.astype(np.float64) @ d)) < 2e-3 * max(1.0, abs(float(grad @ d)))


def test_reads_outside_the_transcriptome(syn):
    """Read k-mers that no isoform holds are empty rows of xnnz: xtheta = 0 there, val = -inf and w = inf, exactly
    what the reference computes; plus empty rows in front of / behind all non-zeros."""
    from kmers_amd.multinomial_simplex_model import multinomial_simplex_model
    x, y = xy(syn, 12, reads="other")
    rows, xnnz, ynnz = orc.model_setup(x, y)
    assert (np.diff(xnnz.indptr) == 0).sum() > 100
    model = multinomial_simplex_model(x, y)
    val, grad = check(model, x, y, np.ones(model.T) / model.T)
    assert val == -np.inf


def test_device_resident_route(syn):
    """X and y never leave HBM: DeviceStore.build_x / count_y buffers straight into the model."""
    import torch
    from kmers_amd.model import DeviceModel
    from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store
    K = 10
    xs = Store(syn["iso"], MODE_X).to_device()
    buf = xs.x_buffers(K, xs.n_pos)
    nnz = xs.build_x(buf)
    counts = Store(syn["reads"], MODE_Y_FASTA).to_device().count_y(K)
    dm = DeviceModel(buf["data"], buf["col_ind"], buf["row_ind"], xs.n_rec_total, counts)
    x, y = xy(syn, K)
    assert nnz == x.nnz
    rows, xnnz, ynnz = orc.model_setup(x, y)
    theta = np.random.default_rng(2).dirichlet(np.full(x.shape[1], 0.5)).astype(np.float32)
    want_val, want_grad, want_xt = orc.logp_grad_ref(theta, 1.0, xnnz, ynnz, np.ones(x.shape[1]))
    val, grad = dm.logp_grad(torch.from_numpy(theta).cuda())
    np.testing.assert_allclose(dm.xtheta(), want_xt, rtol=2e-7)
    np.testing.assert_allclose(val, want_val, atol=5e-7)
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, rtol=1e-6, atol=1e-30)


def test_k15_device_route(syn):
    """K=15, the reference drivers' own setting (run_single.py): 2^30 rows, X and y built and kept on the device, the
    model made from the device buffers; the oracle works on the downloaded arrays."""
    import torch
    from kmers_amd.model import DeviceModel
    from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store
    K = 15
    xs = Store(syn["iso"], MODE_X).to_device()
    buf = xs.x_buffers(K, xs.n_pos)
    nnz = xs.build_x(buf)
    buf["workspace"] = None
    counts = Store(syn["reads"], MODE_Y_FASTA).to_device().count_y(K)
    T = xs.n_rec_total
    dm = DeviceModel(buf["data"], buf["col_ind"], buf["row_ind"], T, counts)
    y = counts.cpu().numpy()
    rows = np.flatnonzero(y)
    assert np.array_equal(dm.rows(), rows) and dm.sum_y == int(y.sum(dtype=np.int64))
    row_ind = buf["row_ind"].cpu().numpy()
    data, col = buf["data"][:nnz].cpu().numpy(), buf["col_ind"][:nnz].cpu().numpy()
    lens = row_ind[rows + 1] - row_ind[rows]
    indptr = np.concatenate([[0], np.cumsum(lens)])
    take = np.repeat(row_ind[rows] - indptr[:-1], lens) + np.arange(indptr[-1])
    xnnz = csr_matrix((data[take].astype(np.float64), col[take], indptr), shape=(rows.shape[0], T))
    assert dm.nnz == xnnz.nnz
    ynnz = y[rows] / y.sum(dtype=np.int64)
    theta = np.random.default_rng(7).dirichlet(np.full(T, 0.5)).astype(np.float32)
    want_val, want_grad, want_xt = orc.logp_grad_ref(theta, 1.0, xnnz, ynnz, np.ones(T))
    val, grad = dm.logp_grad(torch.from_numpy(theta).cuda())
    np.testing.assert_allclose(dm.xtheta(), want_xt, rtol=2e-7, atol=1e-45)
    np.testing.assert_allclose(val, want_val, atol=5e-7)
    np.testing.assert_allclose(grad.cpu().numpy(), want_grad, rtol=1e-6, atol=1e-30)


def test_all_zero_y_is_rejected(syn):
    from kmers_amd.multinomial_simplex_model import multinomial_simplex_model
    x, y = xy(syn, 4)
    with pytest.raises(RuntimeError, match="bad argument"):
        multinomial_simplex_model(x, np.zeros_like(y))


def test_columns_outside_the_model_are_rejected(syn):
    """A column index beyond n_cols (or a wrong first_col for a shard) would make the products gather past theta:
    the model is refused instead."""
    import torch
    from kmers_amd.model import DeviceModel
    x, y = xy(syn, 5)
    up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
    args = (up(x.data, np.float32), up(x.indices, np.int64), up(x.indptr, np.int64))
    DeviceModel(*args, x.shape[1], up(y, np.int32))
    with pytest.raises(RuntimeError, match="bad argument"):
        DeviceModel(*args, x.shape[1] - 1, up(y, np.int32))
    with pytest.raises(RuntimeError, match="bad argument"):
        DeviceModel(*args, x.shape[1], up(y, np.int32), first_col=1)


def test_sharded_model_rehearsal_two_ranks_one_gpu(oracle, tmp_path):
    """kmers_amd.distributed.ShardedModel with the real HIP model on two ranks that share this box's GPU (gloo for the
    collective): the multi-GPU control flow of BASELINE.json configs[4]."""
    import os
    from tests import kmtest
    from tests.test_dist_cpu import _run_model_workers, check_sharded_model
    iso = os.path.join(kmtest.INP, "syn_small.fsa")
    res = _run_model_workers("gpu", 2, iso, iso, 10, str(tmp_path / "m"))
    check_sharded_model(oracle, res, iso, iso, 10)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_random_matrices(seed, monkeypatch):
    """Random CSR matrices of many shapes (rows of thousands of non-zeros that span tiles, runs of empty rows in
    front, inside and behind, single-entry rows, empty columns; 256- and 1024-thread products; with and without row
    slabs) against the oracle."""
    from kmers_amd.model import DeviceModel
    import torch
    rng = np.random.default_rng(1000 + seed)
    n_rows = int(rng.choice([7, 300, 5000, 40000]))
    n_cols = int(rng.choice([1, 9, 700, 6000]))
    kind = seed % 4
    if kind == 0:        # long rows
        lens = rng.integers(0, n_cols + 1, n_rows)
    elif kind == 1:      # short rows, many empty
        lens = np.minimum(rng.geometric(0.5, n_rows) - 1, n_cols)
    elif kind == 2:      # a few very long rows among empty ones
        lens = np.zeros(n_rows, np.int64)
        hot = rng.choice(n_rows, size=max(1, n_rows // 50), replace=False)
        lens[hot] = n_cols
    else:                # power law
        lens = np.minimum((rng.pareto(1.0, n_rows) * 2).astype(np.int64), n_cols)
    lens[: n_rows // 10] = 0                                  # empty rows in front
    if seed % 3 == 0:
        lens[-(n_rows // 7 + 1):] = 0                          # and behind
    lens = lens[: max(1, min(n_rows, int(3e6 // max(1, lens.max()))))] if lens.max() * n_rows > 3e6 else lens
    n_rows = lens.shape[0]
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    indices = np.concatenate([np.sort(rng.choice(n_cols, size=int(l), replace=False)) for l in lens] + [np.zeros(0, np.int64)])
    data = (rng.random(indptr[-1]) + 1e-3).astype(np.float32)
    x = csr_matrix((data, indices.astype(np.int64), indptr), shape=(n_rows, n_cols))
    y = (rng.integers(0, 4, n_rows) * (rng.random(n_rows) < 0.8)).astype(np.int32)
    if not y.any():
        y[rng.integers(n_rows)] = 3
    if seed % 2:
        monkeypatch.setenv("KMERS_MODEL_SLAB_LOG2", "8")
    rows, xnnz, ynnz = orc.model_setup(x, y)
    dm = DeviceModel.from_host(x, y)
    assert np.array_equal(dm.rows(), rows) and dm.nnz == xnnz.nnz
    theta = rng.dirichlet(np.ones(n_cols)).astype(np.float32)
    want_val, want_grad, want_xt = orc.logp_grad_ref(theta, 1.0, xnnz, ynnz, np.ones(n_cols))
    val, grad = dm.logp_grad(torch.from_numpy(theta).cuda())
    np.testing.assert_allclose(dm.xtheta(), want_xt, rtol=2e-7, atol=1e-45)
    grad = grad.cpu().numpy()
    if np.isfinite(want_val):
        np.testing.assert_allclose(val, want_val, rtol=1e-12, atol=5e-7)
        np.testing.assert_allclose(grad, want_grad, rtol=1e-6, atol=1e-30)
    else:
        assert val == want_val
        fin = np.isfinite(want_grad)
        assert np.array_equal(np.isfinite(grad), fin)
        np.testing.assert_allclose(grad[fin], want_grad[fin], rtol=1e-6, atol=1e-30)
