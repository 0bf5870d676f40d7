"""Worker of tests/test_dist_cpu.py: one rank of a world_size-N gloo job on CPU.  The per-rank engine is the oracle,
patched in HERE (tests.kmtest.patch_oracle_engine replaces Store.to_device; the product's functions have no hooks), so what
is exercised is the product's real sharding (kmers_store_ingest_bytes), its collectives and the CSR placement."""
import os
import sys
import tempfile

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["KMERS_NO_TORCH"] = "1"

from kmers_amd import distributed as kd  # noqa: E402
from tests import kmtest  # noqa: E402


def main():
    files, K, out = sys.argv[1].split(","), int(sys.argv[2]), sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    kmtest.patch_oracle_engine(tempfile.mkdtemp(), f"r{rank}")
    y = kd.count_y_sharded(files, K, device="cpu")
    lo, part = kd.count_y_sharded(files, K, device="cpu", reduce_scatter=True)
    per = -(-(4 ** K) // world)
    assert lo == min(rank * per, 4 ** K) and np.array_equal(part, y[lo:lo + per])
    x = kd.build_x_sharded(files, K, device="cpu")
    xc = kd.build_x_sharded(files, K, L=30, concatenate_subseq=True, device="cpu")
    if rank == 0:
        np.savez(out, y=y, data=x[0], row=x[1], col=x[2], n_cols=x[3], cdata=xc[0], crow=xc[1], ccol=xc[2], cn=xc[3])
    else:
        assert x is None and xc is None
        np.save(out + f".rank{rank}.npy", y)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
