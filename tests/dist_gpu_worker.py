"""Worker of tests/test_dist_gpu.py: one rank of a world_size-N job whose ranks all sit on GPU 0 of a one-GPU box.
The engine is the PRODUCT's (HIP kernels through the C ABI); only the collectives travel over gloo, because RCCL refuses
two ranks on one device (kmers_amd.distributed._all_reduce_sum hops through the host for them, the arithmetic around them
stays on the GPU exactly as with RCCL).  What is exercised: byte-range ingest per rank, kmers_hip_build_x / _count_y on a
shard with its global column numbers, the count all-reduce + rank prefix + device-side placement lists of concat_x_shards,
the shared-memory CSR, and the y all-reduce / reduce-scatter."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from kmers_amd import distributed as kd  # noqa: E402


def main():
    files, Ks, out = sys.argv[1].split(","), [int(k) for k in sys.argv[2].split(",")], sys.argv[3]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    res = {}
    for K in Ks:
        y = kd.count_y_sharded(files, K, device="cuda:0")
        lo, part = kd.count_y_sharded(files, K, device="cuda:0", reduce_scatter=True)
        per = -(-(4 ** K) // world)
        assert lo == min(rank * per, 4 ** K) and np.array_equal(part, y[lo:lo + per])
        x = kd.build_x_sharded(files, K, device="cuda:0")
        if rank == 0:
            idx = np.flatnonzero(y)
            res.update({f"y{K}_idx": idx, f"y{K}_cnt": y[idx], f"x{K}_data": x[0], f"x{K}_col": x[2], f"x{K}_n": x[3],
                        f"x{K}_rows": np.flatnonzero(np.diff(x[1])), f"x{K}_cnt": np.diff(x[1])[np.flatnonzero(np.diff(x[1]))]})
        else:
            assert x is None
    if rank == 0:
        np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
