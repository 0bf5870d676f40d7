"""Worker of the sharded-fit tests: one rank of a world_size-N gloo job.
  cpu  (tests/test_dist_cpu.py): the per-shard engine is the numpy oracle (patched in by this worker, test infrastructure only);
       what is exercised is the column sharding, the all-reduce of xtheta and the assembly of the gradient.
  gpu  (tests/test_model_gpu.py): the real HIP model on every rank, all ranks on GPU 0 (a one-GPU box; RCCL refuses two
       ranks on one device, so the collective runs over gloo): a rehearsal of the multi-GPU control flow."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode == "cpu":
    os.environ["KMERS_NO_TORCH"] = "1"
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from scipy.sparse import csr_matrix  # noqa: E402

from kmers_amd import distributed as kd  # noqa: E402
from oracle import logp_grad as orc  # noqa: E402
from tests import kmtest  # noqa: E402


class OracleShard:
    """Stand-in for kmers_amd.model.DeviceModel (same constructor and methods) on the numpy oracle: CPU workers only."""

    def __init__(self, data, col_ind, row_ind, n_cols, counts, first_col=0):
        M = row_ind.numel() - 1
        nnz = int(row_ind[-1])
        x = csr_matrix((data.numpy()[:nnz], col_ind.numpy()[:nnz] - first_col, row_ind.numpy()), shape=(M, n_cols))
        self.rows, self.xnnz, self.ynnz = orc.model_setup(x, counts.numpy())
        self.n_cols, self.n_masked = n_cols, self.rows.shape[0]

    def xtheta_into(self, theta, out):
        out.copy_(torch.from_numpy(self.xnnz.dot(theta.numpy().astype(np.float64)).astype(np.float32)))

    def grad_from(self, xtheta, nograd=False):
        xt = xtheta.numpy().astype(np.float64)
        with np.errstate(divide="ignore"):
            val = float(self.ynnz.dot(np.log(xt)))
            w = (self.ynnz / xt).astype(np.float32).astype(np.float64)
        return val if nograd else (val, torch.from_numpy((w @ self.xnnz).astype(np.float32)))


def main():
    iso, reads, K, out = sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    if mode == "cpu":       # the engine is swapped HERE, in the test worker: the product's classes have no hooks
        from kmers_amd import model as _model
        kmtest.patch_oracle_engine(tempfile.mkdtemp(), f"m{rank}")
        _model.DeviceModel = OracleShard
        model = kd.ShardedModel(iso, reads, K, device="cpu")
    else:
        torch.cuda.set_device(0)
        model = kd.ShardedModel(iso, reads, K, device="cuda:0")
    theta = np.random.default_rng(5).dirichlet(np.full(model.T, 0.7))
    val, grad = model.logp_grad(theta)
    val2 = model.logp_grad(theta, nograd=True)
    assert val2 == val
    np.savez(out + f".rank{rank}.npz", val=val, grad=grad, theta=theta, N=model.N, first=model.first_col, n=model.n_local)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
