"""The N > 1 flows with the REAL kernels: two ranks on the one GPU of the box (collectives over gloo), through
kmers_amd.distributed.build_x_sharded and count_y_sharded, at K = 8 (4-byte entries, fused emit) and K = 14 (8-byte
entries, 2^28 rows: the count all-reduce is 1 GB per rank), bit-exact against the oracle on the whole input."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from kmers_amd import synth
from tests import kmtest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_two_ranks_on_one_gpu_equal_the_whole(oracle, tmp_path, world):
    iso = str(tmp_path / "iso.fsa")
    synth.synth_transcriptome(iso, n_records=700, total_bases=900_000, seed=31)
    files = [iso, os.path.join(kmtest.INP, "ref_test4.fsa")]
    Ks = [8, 14] if world == 2 else [5, 9]
    out = str(tmp_path / "res.npz")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), KMER_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "dist_gpu_worker.py"), ",".join(files),
                                       ",".join(map(str, Ks)), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        o, _ = p.communicate(timeout=280)
        assert p.returncode == 0, o.decode()[-3000:]
    res = np.load(out)
    for K in Ks:
        rc, y = oracle.y(files, K)
        idx = np.flatnonzero(y)
        assert rc == 0 and np.array_equal(res[f"y{K}_idx"], idx) and np.array_equal(res[f"y{K}_cnt"], y[idx])
        rc, data, row, col, n_cols = oracle.x(files, K)
        assert rc == 0 and int(res[f"x{K}_n"]) == n_cols
        cnt = np.diff(row)
        nz = np.flatnonzero(cnt)
        assert np.array_equal(res[f"x{K}_rows"], nz) and np.array_equal(res[f"x{K}_cnt"], cnt[nz])
        assert np.array_equal(res[f"x{K}_col"], col)
        assert np.array_equal(res[f"x{K}_data"].view(np.uint32), data.view(np.uint32))
