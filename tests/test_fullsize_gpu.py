"""BASELINE.json's full sizes (config 2: ~170 k isoforms / ~400 Mbases; 10 M reads of 100 bp), where no oracle run
finishes in seconds: the HIP path is checked through properties that do not depend on the size, and through
independent code paths that must agree bit for bit:

  * X at K=8: 4-byte-entry path == 8-byte-entry path (knob force_wide); rows / columns ordered as the reference's CSR;
    every non-empty column sums to 1; row k is non-empty exactly when the k-mer occurs (y of the same file);
  * X at K=14 (four passes, 2^28 rows): same structural checks, and the whole == the interleave of two column shards;
  * y at K=8 (LDS tables), 12 (one partition pass, 2-byte payloads), 14 (two passes) == the plain global-atomics
    kernel; totals == the number of valid windows.
  * y at K=12 on BASELINE.json configs[3] as written: 10^8 simulated 100 bp reads (kmers_amd.simulate_reads, the reads
    made in HBM), 1.01e10 positions = five chunks of the partition path == the plain global-atomics kernel; total ==
    the number of valid windows; the K=14 histogram of the same reads folded to K=12 bounds it from below;
  * configs[4] on one GPU at full size: X at K=14 and y at K=14 of 10 M reads feed the model; sum(X . theta) over all
    rows is 1 for any theta on the simplex (columns of X sum to 1), logp_grad agrees with a float64 restatement on a
    sample of rows / columns taken from the resident arrays, and ten exponentiated-gradient steps raise the likelihood strictly;
  * X at K=15 (2^30 rows, 8.6 GB of row pointer): the structural checks, and rows folded by 4 == the K=14 row counts
    of the windows that extend by one base.
Inputs are the seeded synthetic files bench.py uses (kmers_amd/synth.py)."""
import os

import numpy as np
import pytest

from kmers_amd import synth
from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full(tmp_path_factory):
    d = tmp_path_factory.mktemp("full")
    iso, reads = str(d / "tx.fsa"), str(d / "reads.fsa")
    info = synth.synth_transcriptome(iso, n_records=170_000, total_bases=400_000_000, seed=20240521,
                                     return_sequences=True)
    synth.synth_reads(reads, info["concat"], info["offsets"], 10_000_000, read_len=100, seed=42)
    del info
    yield dict(iso=iso, reads=reads)
    os.remove(iso)
    os.remove(reads)


def _csr_checks(torch, buf, nnz, n_rows, n_cols):
    row = buf["row_ind"]
    col = buf["col_ind"][:nnz]
    data = buf["data"][:nnz]
    assert int(row[0]) == 0 and int(row[-1]) == nnz
    assert bool((row[1:] >= row[:-1]).all())
    assert int(col.min()) >= 0 and int(col.max()) < n_cols
    # columns strictly ascending inside a row: a descent may only happen where a new row starts
    starts = torch.zeros(nnz + 1, dtype=torch.bool, device=row.device)
    starts[row[:-1][row[:-1] < row[1:]]] = True
    bad = (col[1:] <= col[:-1]) & ~starts[1:nnz]
    assert not bool(bad.any())
    assert bool((data > 0).all()) and bool((data <= 1).all())
    sums = torch.zeros(n_cols, dtype=torch.float64, device=row.device).index_add_(0, col, data.double())
    live = sums > 0
    assert float((sums[live] - 1).abs().max()) < 1e-4          # float32 values: ~2400 terms of relative error 6e-8
    return live


def test_x8_full_size(full, knob):
    import torch
    K, M = 8, 4 ** 8
    st = Store(full["iso"], MODE_X)
    assert st.n_bases > 390_000_000 and st.n_rec > 160_000
    ds = st.to_device()
    buf = ds.x_buffers(K, st.n_windows(K))
    nnz = ds.build_x(buf)
    assert 3.0e8 < nnz <= st.n_windows(K)
    live = _csr_checks(torch, buf, nnz, M, st.n_rec)
    assert int(live.sum()) > 0.99 * st.n_rec
    keep = {k: buf[k][:nnz].clone() if k != "row_ind" else buf[k].clone() for k in ("data", "col_ind", "row_ind")}
    knob("force_wide")                                          # the 8-byte-entry path on the same input
    assert ds.build_x(buf) == nnz
    for k in ("data", "col_ind"):
        assert torch.equal(buf[k][:nnz], keep[k]), k
    assert torch.equal(buf["row_ind"], keep["row_ind"])
    # the k-mers that occur are the non-empty rows (same file, same record filter on the y side)
    y = Store(full["iso"], MODE_Y_FASTA).to_device().count_y(K)
    assert torch.equal(y > 0, keep["row_ind"][1:] > keep["row_ind"][:-1])


def test_x14_full_size_and_shards(full):
    import torch
    from kmers_amd.distributed import merge_csr_parts
    K, M = 14, 4 ** 14
    st = Store(full["iso"], MODE_X)
    ds = st.to_device()
    buf = ds.x_buffers(K, st.n_windows(K))
    nnz = ds.build_x(buf)
    _csr_checks(torch, buf, nnz, M, st.n_rec)
    # columns [0, n/2) and [n/2, n) built separately: their per-row counts add up to the whole's row pointer and the
    # whole's entries are the shards' (the full interleave is checked at small sizes in test_api_gpu / test_dist_cpu)
    row_sum = torch.zeros(M + 1, dtype=torch.int64, device=ds.device)
    total = 0
    for s in range(2):
        sh = Store(full["iso"], MODE_X, shard=s, n_shards=2)
        dsh = sh.to_device()
        b = dsh.x_buffers(K, sh.n_windows(K))
        n = dsh.build_x(b)
        total += n
        row_sum += b["row_ind"]
        lo, hi = sh.first_rec, sh.first_rec + sh.n_rec
        assert int(b["col_ind"][:n].min()) >= lo and int(b["col_ind"][:n].max()) < hi
        inside = (buf["col_ind"][:nnz] >= lo) & (buf["col_ind"][:nnz] < hi)
        assert torch.equal(buf["col_ind"][:nnz][inside], b["col_ind"][:n])
        assert torch.equal(buf["data"][:nnz][inside], b["data"][:n])
        del b, dsh
    assert total == nnz and torch.equal(row_sum, buf["row_ind"])


@pytest.mark.parametrize("K", [8, 12, 14])
def test_y_full_size(full, K, knob):
    import torch
    st = Store(full["reads"], MODE_Y_FASTA)
    assert st.n_rec == 10_000_000
    ds = st.to_device()
    y = ds.count_y(K)
    total = int(y.sum(dtype=torch.int64))
    assert total <= st.n_windows(K) and total > 0.99 * st.n_windows(K)      # a few windows cover an N
    if K <= 8:
        # K <= 8 has one kernel: check it against the K=12 histogram folded down (the first 8 bases of every valid
        # 12-mer are a valid 8-mer; the 4 last 8-mers of a read start no 12-mer)
        y12 = ds.count_y(12).view(4 ** 8, -1).sum(1, dtype=torch.int64)
        extra = int((y.long() - y12).sum())                 # 4 per read, plus 8-mers whose 12-mer runs into an N
        assert bool((y.long() >= y12).all()) and 4 * st.n_rec <= extra <= 4 * st.n_rec + total // 1000
    else:
        knob("y_atomics")
        assert torch.equal(ds.count_y(K), y)


def test_y12_one_hundred_million_simulated_reads(full, knob):
    """BASELINE.json configs[3] at its stated size."""
    import torch
    from kmers_amd.simulate_reads import simulate_reads_device
    N, L = 100_000_000, 100
    ds, info = simulate_reads_device(N, L, 0.1, full["iso"])
    assert ds.n_rec == N and ds.n_pos == N * (L + 1) and ds.n_pos > 2 * (1 << 31)        # at least three chunks
    y = ds.count_y(12)
    total = int(y.sum(dtype=torch.int64))
    windows = N * (L - 12 + 1)
    assert 0.99 * windows < total <= windows
    y14 = ds.count_y(14)
    t14 = int(y14.sum(dtype=torch.int64))
    folded = y14.view(4 ** 12, 16).sum(1, dtype=torch.int64)      # every valid 14-mer starts with a valid 12-mer
    extra = y.long() - folded                                     # the two last 12-mers of a read + 12-mers stopped by an N
    assert bool((extra >= 0).all()) and 2 * N <= int(extra.sum()) == total - t14 <= 2 * N + total // 1000
    knob("y_atomics")
    assert torch.equal(ds.count_y(12), y)
    # the sampled isoform ids and starts are the reference's stream: the histogram of the reads' first 12-mers follows
    assert info["theta_sampled"].shape[0] == len(info["lengths"]) and abs(float(info["theta_sampled"].sum()) - 1) < 1e-9


def test_config4_full_size_build_count_fit(full):
    """X (K=14) + y (K=14) of the full-size inputs -> the model in HBM -> logp_grad -> a short fit."""
    import torch
    from kmers_amd.model import DeviceModel
    K, M = 14, 4 ** 14
    xs = Store(full["iso"], MODE_X)
    dx = xs.to_device()
    buf = dx.x_buffers(K, xs.n_windows(K))
    nnz = dx.build_x(buf)
    buf["workspace"] = None
    counts = Store(full["reads"], MODE_Y_FASTA).to_device().count_y(K)
    T = xs.n_rec_total
    dm = DeviceModel(buf["data"], buf["col_ind"], buf["row_ind"], T, counts)
    assert dm.n_masked == int((counts > 0).sum()) and 0 < dm.nnz <= nnz
    dev = counts.device
    g = torch.Generator(device="cpu").manual_seed(3)
    theta = torch.rand(T, generator=g, dtype=torch.float64).to(dev)
    theta = (theta / theta.sum()).float()
    # (1) sum over ALL rows of X . theta = sum_t theta_t * (column sum = 1 where the isoform has a k-mer)
    data, col = buf["data"][:nnz], buf["col_ind"][:nnz]
    live = torch.zeros(T, dtype=torch.bool, device=dev)
    live[col] = True
    assert abs(float((data.double() * theta.double()[col]).sum()) - float(theta.double()[live].sum())) < 1e-6
    # (2) logp_grad against float64 on the masked rows, from the resident arrays (torch as the checker)
    val, grad = dm.logp_grad(theta)
    row = buf["row_ind"]
    rows_of = torch.repeat_interleave(torch.arange(M, device=dev), row[1:] - row[:-1])
    mask = counts > 0
    keep = mask[rows_of]
    r, c, d = rows_of[keep], col[keep], data[keep].double()
    del rows_of, keep
    xt = torch.zeros(M, dtype=torch.float64, device=dev).index_add_(0, r, d * theta.double()[c])
    ynorm = counts.double() / float(counts.sum(dtype=torch.int64))
    want_val = float((ynorm[mask] * torch.log(xt[mask])).sum())
    w = torch.zeros(M, dtype=torch.float64, device=dev)
    w[mask] = ynorm[mask] / xt[mask]
    want_grad = torch.zeros(T, dtype=torch.float64, device=dev).index_add_(0, c, d * w[r])
    assert abs(val - want_val) < 1e-5 * abs(want_val)
    g_dev = torch.as_tensor(grad, device=dev).double()
    assert float((g_dev - want_grad).abs().max()) < 2e-4 * float(want_grad.abs().max())
    # (2b) the float64 checker itself against oracle/logp_grad.py on a sample of 10^5 masked rows (the oracle's scipy
    # statement of kmerexpr/utils.py:46-73 finishes in seconds at that size): same val and gradient contributions
    import scipy.sparse as sp
    from oracle.logp_grad import logp_grad_ref
    rows_m = mask.nonzero().squeeze(1)
    pick = rows_m[torch.randperm(rows_m.numel(), generator=g)[:100_000].to(dev)].sort().values
    sel = torch.zeros(M, dtype=torch.bool, device=dev)
    sel[pick] = True
    new_id = torch.cumsum(sel, 0) - 1
    k2 = sel[r]
    xs_ = sp.csr_matrix((d[k2].cpu().numpy(), (new_id[r[k2]].cpu().numpy(), c[k2].cpu().numpy())), shape=(pick.numel(), T))
    o_val, o_grad, o_xt = logp_grad_ref(theta.cpu().numpy(), 1.0, xs_, ynorm[pick].cpu().numpy(), np.ones(T))
    assert np.allclose(o_xt, xt[pick].cpu().numpy(), rtol=2e-7, atol=0)          # the oracle rounds xtheta to float32
    assert abs(o_val - float((ynorm[pick] * torch.log(xt[pick])).sum())) < 1e-6 * abs(o_val)
    t_grad = torch.zeros(T, dtype=torch.float64, device=dev).index_add_(0, c[k2], d[k2] * w[r[k2]]).cpu().numpy()
    assert np.abs(o_grad - t_grad).max() < 1e-6 * np.abs(t_grad).max()
    del r, c, d, xt, w, k2, sel, new_id
    # (3) exponentiated-gradient steps (kmerexpr/exp_grad_solver.py:60-64, default step) raise the likelihood
    theta = torch.full((T,), 1.0 / T, dtype=torch.float32, device=dev)
    v0, g0 = dm.logp_grad(theta)
    lr = 2 ** -0.5 / float(torch.as_tensor(g0).abs().max())
    vals = [v0]
    for _ in range(10):
        v, gr = dm.logp_grad(theta)
        gr = torch.as_tensor(gr, device=dev)
        theta = theta * torch.exp(lr * (gr - gr.max()))
        theta = theta / theta.sum()
        vals.append(dm.logp_grad(theta, nograd=True))
    assert all(b > a for a, b in zip(vals, vals[1:])), vals          # strictly increasing


def test_x15_structure(full):
    """K=15: 2^30 rows.  Structural checks, and against K=14: a 15-mer's first 14 bases are a 14-mer of the same isoform,
    so the rows of X15 folded by 4 have at most the entries of X14's rows, and every isoform with a 15-mer has a 14-mer."""
    import torch
    K, M = 15, 4 ** 15
    st = Store(full["iso"], MODE_X)
    ds = st.to_device()
    buf = ds.x_buffers(K, st.n_windows(K))
    nnz = ds.build_x(buf)
    live15 = _csr_checks(torch, buf, nnz, M, st.n_rec)
    y = Store(full["iso"], MODE_Y_FASTA).to_device().count_y(K)
    assert torch.equal(y > 0, buf["row_ind"][1:] > buf["row_ind"][:-1])
    per14 = (buf["row_ind"][4::4] - buf["row_ind"][:-4:4]).clone()       # entries of the 4 extensions of each 14-mer
    del buf, y
    torch.cuda.empty_cache()
    b14 = ds.x_buffers(14, st.n_windows(14))
    n14 = ds.build_x(b14)
    rows14 = b14["row_ind"][1:] - b14["row_ind"][:-1]
    assert nnz <= 4 * n14
    assert bool(((per14 > 0) <= (rows14 > 0)).all())                     # a 15-mer row implies its 14-mer row
    assert bool((per14 <= 4 * rows14).all())
    live14 = torch.zeros(st.n_rec, dtype=torch.bool, device=ds.device)
    live14[b14["col_ind"][:n14]] = True
    assert bool((live15 <= live14).all())
