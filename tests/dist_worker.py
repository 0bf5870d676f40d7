"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job on CPU.  The HIP calls of
kmers_amd.distributed are replaced by the oracle (count_fn / build_fn hooks, test infrastructure only), so what is
exercised is the real sharding (kmers_store_ingest shard arguments), the all-reduce / gather, and the CSR merge."""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KMERS_NO_TORCH"] = "1"

from kmers_amd import distributed as kd  # noqa: E402
from tests import kmtest  # noqa: E402


def _rewrite(store, path):
    with open(path, "wb") as f:
        for i, s in enumerate(store.sequences()):
            f.write(b">r%d\n" % i + s + b"\n")


def main():
    files, K, out = sys.argv[1].split(","), int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    oracle = kmtest.load_oracle()
    tmp = tempfile.mkdtemp()

    def count_fn(store, K):
        p = os.path.join(tmp, f"y{rank}.fsa")
        _rewrite(store, p)
        rc, y = oracle.y([p], K)
        assert rc == 0
        return torch.from_numpy(y.copy())

    def build_fn(store, K):
        p = os.path.join(tmp, f"x{rank}.fsa")
        _rewrite(store, p)
        rc, data, row, col, n_cols = oracle.x([p], K, max_nz=store.n_pos + 16)
        assert rc == 0 and n_cols == store.n_rec
        return data, row, col + np.uint64(store.first_rec)

    y = kd.count_y_sharded(files, K, count_fn=count_fn)
    x = kd.build_x_sharded(files, K, build_fn=build_fn)
    xc = kd.build_x_sharded(files, K, L=30, concatenate_subseq=True, build_fn=build_fn)
    if rank == 0:
        np.savez(out, y=y, data=x[0], row=x[1], col=x[2], n_cols=x[3], cdata=xc[0], crow=xc[1], ccol=xc[2], cn=xc[3])
    else:
        assert x is None and xc is None
        np.save(out + f".rank{rank}.npy", y)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
