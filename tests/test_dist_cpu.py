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


@pytest.mark.parametrize("world,K", [(2, 8), (3, 5)])
def test_sharded_y_and_x_equal_whole(oracle, tmp_path, world, K):
    files = [os.path.join(kmtest.INP, "syn_small.fsa"), os.path.join(kmtest.INP, "ref_test4.fsa")]
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
