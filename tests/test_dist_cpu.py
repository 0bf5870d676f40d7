"""The N > 1 path on CPU: world_size-2 and -3 gloo jobs (one process per rank, rendezvous on 127.0.0.1).
Shards come from the real ingest, the per-shard compute from the oracle (no GPU here), and the combined result
must equal the oracle on the whole input: y through one all-reduce, X through gather + kmers_csr_merge."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import kmtest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,K,names", [(2, 8, ("syn_small.fsa", "ref_test4.fsa")), (3, 5, ("syn_small.fsa", "ref_test4.fsa")),
                                           (4, 2, ("ref_test2.fsa", "ref_test4.fsa"))])   # six records on four ranks: empty shards
def test_sharded_y_and_x_equal_whole(oracle, tmp_path, world, K, names):
    files = [os.path.join(kmtest.INP, n) for n in names]
    out = str(tmp_path / "res.npz")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), KMER_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_worker.py"), ",".join(files), str(K), out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=300)
        assert p.returncode == 0, o.decode()[-2000:]
    res = np.load(out)
    rc, y = oracle.y(files, K)
    assert rc == 0 and np.array_equal(res["y"], y)
    for r in range(1, world):       # every rank holds the full histogram after the all-reduce
        assert np.array_equal(np.load(out + f".rank{r}.npy"), y)
    rc, data, row, col, n_cols = oracle.x(files, K)
    assert int(res["n_cols"]) == n_cols
    assert np.array_equal(res["row"], row) and np.array_equal(res["col"], col)
    assert np.array_equal(res["data"].view(np.uint32), data.view(np.uint32))
    rc, data, row, col, n_cols = oracle.x(files, K, L=30, cat=True)
    assert int(res["cn"]) == n_cols == 2
    assert np.array_equal(res["crow"], row) and np.array_equal(res["ccol"], col)
    assert np.array_equal(res["cdata"].view(np.uint32), data.view(np.uint32))


def test_csr_merge_unit():
    from kmers_amd.distributed import merge_csr_parts
    rng = np.random.default_rng(3)
    n_rows, parts, dense = 50, [], []
    for g in range(4):
        d = (rng.random((n_rows, 7)) < 0.2) * rng.random((n_rows, 7))
        dense.append(d)
        r, c = np.nonzero(d)
        row = np.zeros(n_rows + 1, np.uint64)
        np.add.at(row, r + 1, 1)
        parts.append((d[r, c].astype(np.float32), np.cumsum(row).astype(np.uint64), (c + 7 * g).astype(np.uint64)))
    data, row, col = merge_csr_parts(parts, n_rows)
    whole = np.concatenate(dense, axis=1)
    r, c = np.nonzero(whole)
    assert np.array_equal(col, c.astype(np.uint64)) and np.array_equal(data, whole[r, c].astype(np.float32))
    assert np.array_equal(np.diff(row.astype(np.int64)), np.bincount(r, minlength=n_rows))


def test_csr_place_unit():
    """kmers_csr_place, one call per "rank" into one shared output, == kmers_csr_merge of the same parts (empty parts and
    empty rows included)."""
    from kmers_amd import _lib
    from kmers_amd.distributed import merge_csr_parts
    lib = _lib.handle()
    rng = np.random.default_rng(8)
    n_rows, parts = 300, []
    for g in range(5):
        dens = 0.0 if g == 2 else 0.15
        d = (rng.random((n_rows, 9)) < dens) * rng.random((n_rows, 9))
        d[100:180] = 0
        r, c = np.nonzero(d)
        row = np.zeros(n_rows + 1, np.uint64)
        np.add.at(row, r + 1, 1)
        parts.append((d[r, c].astype(np.float32), np.cumsum(row).astype(np.uint64), (c + 9 * g).astype(np.uint64)))
    want = merge_csr_parts(parts, n_rows)
    glob = np.sum([p[1] for p in parts], axis=0).astype(np.uint64)
    total = int(glob[-1])
    out_d, out_c, out_r = np.zeros(total, np.float32), np.zeros(total, np.uint64), np.zeros(n_rows + 1, np.uint64)
    before = np.zeros(n_rows + 1, np.uint64)
    for g, (d, r, c) in enumerate(parts):
        rc = lib.kmers_csr_place(n_rows, r.ctypes.data, before.ctypes.data, glob.ctypes.data, d.ctypes.data if d.size else None,
                                 c.ctypes.data if c.size else None, out_d.ctypes.data, out_c.ctypes.data,
                                 out_r.ctypes.data if g == 0 else None)
        assert rc == 0
        before = before + r
    assert np.array_equal(out_r, want[1]) and np.array_equal(out_c, want[2]) and np.array_equal(out_d, want[0])


def _run_model_workers(mode, world, iso, reads, K, out, timeout=300):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), KMER_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_model_worker.py"), mode, iso, reads, str(K),
                                       out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=timeout)
        assert p.returncode == 0, o.decode()[-2000:]
    return [np.load(out + f".rank{r}.npz") for r in range(world)]


def check_sharded_model(oracle, res, iso, reads, K):
    """Every rank's (val, grad) against the single-process float64 oracle on the whole X and y."""
    from scipy.sparse import csr_matrix
    from oracle import logp_grad as orc
    rc, data, row, col, n_cols = oracle.x([iso], K)
    rc2, y = oracle.y([reads], K)
    assert rc == 0 and rc2 == 0
    x = csr_matrix((data, col.astype(np.int64), row.astype(np.int64)), shape=(4 ** K, n_cols))
    rows, xnnz, ynnz = orc.model_setup(x, y)
    theta = res[0]["theta"]
    want_val, want_grad, _ = orc.logp_grad_ref(theta, 1.0, xnnz, ynnz, np.ones(n_cols))
    covered = 0
    for r in res:
        assert int(r["N"]) == int(y.sum()) and int(r["first"]) == covered
        covered += int(r["n"])
        # xtheta is a sum of float32 partial products here (one rounding per rank), hence the wider tolerance
        np.testing.assert_allclose(float(r["val"]), want_val, rtol=0, atol=2e-6)
        np.testing.assert_allclose(r["grad"], want_grad, rtol=5e-6, atol=1e-30)
        assert np.array_equal(r["grad"], res[0]["grad"]) and float(r["val"]) == float(res[0]["val"])
    assert covered == n_cols


@pytest.mark.parametrize("world,K", [(2, 6), (3, 9)])
def test_sharded_logp_grad(oracle, tmp_path, world, K):
    """ShardedModel: X sharded by columns, y all-reduced, one all-reduce of xtheta per evaluation."""
    iso = os.path.join(kmtest.INP, "syn_small.fsa")
    reads = os.path.join(kmtest.INP, "syn_small.fsa")            # the transcriptome's own k-mers: every row of y has an X row
    res = _run_model_workers("cpu", world, iso, reads, K, str(tmp_path / "m"))
    check_sharded_model(oracle, res, iso, reads, K)
