"""BASELINE.json's full sizes (config 2: ~170 k isoforms / ~400 Mbases; 10 M reads of 100 bp), where no oracle run
finishes in seconds: the HIP path is checked through properties that do not depend on the size, and through
independent code paths that must agree bit for bit:

  * X at K=8: 4-byte-entry path == 8-byte-entry path (KMERS_FORCE_WIDE); rows / columns ordered as the reference's CSR;
    every non-empty column sums to 1; row k is non-empty exactly when the k-mer occurs (y of the same file);
  * X at K=14 (four passes, 2^28 rows): same structural checks, and the whole == the interleave of two column shards;
  * y at K=8 (LDS tables), 12 (one partition pass, 2-byte payloads), 14 (two passes) == the plain global-atomics
    kernel; totals == the number of valid windows.
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


def test_x8_full_size(full, monkeypatch):
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
    monkeypatch.setenv("KMERS_FORCE_WIDE", "1")                 # the 8-byte-entry path on the same input
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
def test_y_full_size(full, K, monkeypatch):
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
        monkeypatch.setenv("KMERS_Y_ATOMICS", "1")
        assert torch.equal(ds.count_y(K), y)
