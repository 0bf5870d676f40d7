"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  SURVEY.md section 8e:

  y  reads are partitioned into contiguous record ranges of equal size (kmers_store_ingest's shard arguments);
     every rank counts its range into a private int32[4^K] in HBM; ONE all-reduce(sum) combines them.  Integer
     sums are order-independent, so the result is bit-identical to the single-GPU one.
  X  records (columns) are partitioned the same way; columns are independent, so there is no reduction: every
     rank builds the CSR of its column range (global column numbers via first_col) and rank 0 interleaves the
     per-rank CSRs row by row (kmers_csr_merge), which keeps columns ascending inside each row.
"""
import ctypes as C

import numpy as np

from . import _lib
from .store import MODE_X, MODE_X_CAT, MODE_Y_FASTA, MODE_Y_FASTQ, Store


def merge_csr_parts(parts, n_rows):
    """parts: list of (data f32[nnz_g], row_ind u64[n_rows+1], col_ind u64[nnz_g]) with ascending disjoint column
    ranges.  Returns (data, row_ind, col_ind) of the interleaved matrix."""
    lib = _lib.handle()
    n = len(parts)
    data = [np.ascontiguousarray(p[0], np.float32) for p in parts]
    rows = [np.ascontiguousarray(p[1], np.uint64) for p in parts]
    cols = [np.ascontiguousarray(p[2], np.uint64) for p in parts]
    for r in rows:
        assert r.shape[0] == n_rows + 1
    nnz = int(sum(int(r[-1]) for r in rows))
    out_d = np.empty(nnz, np.float32)
    out_r = np.empty(n_rows + 1, np.uint64)
    out_c = np.empty(nnz, np.uint64)
    vp = C.c_void_p
    pd = (vp * n)(*[a.ctypes.data for a in data])
    pr = (vp * n)(*[a.ctypes.data for a in rows])
    pc = (vp * n)(*[a.ctypes.data for a in cols])
    _lib.check(lib.kmers_csr_merge(n, n_rows, pd, pr, pc, out_d.ctypes.data, out_r.ctypes.data, out_c.ctypes.data),
               "kmers_csr_merge")
    return out_d, out_r, out_c


def _dist():
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    return dist


def count_y_sharded(files, K, fastq=False, device=None, count_fn=None):
    """Every rank returns the full histogram (int32 numpy[4^K]) of `files`, each having counted 1/world of the
    records.  count_fn(store, K) -> int32 tensor replaces the HIP call in CPU tests only."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    st = Store(files, MODE_Y_FASTQ if fastq else MODE_Y_FASTA, shard=rank, n_shards=world)
    if count_fn is None:
        counts = st.to_device(device or "cuda").count_y(K)
    else:
        counts = count_fn(st, K)
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)      # RCCL ring over xGMI: 4^K * 4 bytes
    return counts.cpu().numpy()


def build_x_sharded(files, K, L=0, concatenate_subseq=False, device=None, build_fn=None):
    """Rank 0 returns (data, row_ind, col_ind, n_cols) of the whole matrix; other ranks return None.
    build_fn(store, K) -> (data, row_ind, col_ind) numpy replaces the HIP call in CPU tests only."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    st = Store(files, MODE_X_CAT if concatenate_subseq else MODE_X, L=L, shard=rank, n_shards=world)
    if build_fn is None:
        ds = st.to_device(device or "cuda")
        buf = ds.x_buffers(K, st.n_windows(K))
        nnz = ds.build_x(buf)
        part = (buf["data"][:nnz].cpu().numpy(), buf["row_ind"].cpu().numpy().view(np.uint64),
                buf["col_ind"][:nnz].cpu().numpy().view(np.uint64))
    else:
        part = build_fn(st, K)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object(part, gathered, dst=0)          # host-side concatenation of column shards
    if rank != 0:
        return None
    d, r, c = merge_csr_parts(gathered, 4 ** K)
    return d, r, c, st.n_rec_total


class ShardedModel:
    """logp_grad with X sharded by columns over the ranks (BASELINE.json configs[4]): every
    rank builds the CSR of its own isoform range and keeps it in HBM (no host-side merge), counts its share of the
    reads, and the histogram is all-reduced so that every rank masks the same rows.  One evaluation is then

        xtheta_g = xnnz_g . theta_g            per rank (its columns of theta)
        xtheta   = all_reduce(sum, xtheta_g)   float32[n_masked]: the path's one exchange step per evaluation
        val, grad_g = terms(xtheta), (ynnz / xtheta) . xnnz_g          per rank, no further exchange for grad_g

    (reference kmerexpr/utils.py:76-108 on kmerexpr/multinomial_simplex_model.py:56-62's xnnz / ynnz).  The float32
    partial products are summed by the collective, so the result depends on the number of ranks in the last bits.
    make_model(store, K, y) -> an object with n_cols, n_masked, xtheta_into(theta, out), grad_from(xtheta, nograd)
    replaces the HIP model in CPU tests only."""

    def __init__(self, iso_files, reads_files, K, L=0, fastq=False, device=None, count_fn=None, make_model=None):
        import torch
        dist = _dist()
        rank, world = dist.get_rank(), dist.get_world_size()
        self.K = K
        xs = Store(iso_files, MODE_X, L=L, shard=rank, n_shards=world)
        ys = Store(reads_files, MODE_Y_FASTQ if fastq else MODE_Y_FASTA, shard=rank, n_shards=world)
        self.T, self.first_col, self.n_local = xs.n_rec_total, xs.first_rec, xs.n_rec
        if make_model is None:
            dev = torch.device(device or "cuda")
            counts = ys.to_device(dev).count_y(K)
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
            dx = xs.to_device(dev)
            buf = dx.x_buffers(K, xs.n_windows(K))
            dx.build_x(buf)
            buf["workspace"] = None
            from .model import DeviceModel
            self.model = DeviceModel(buf["data"], buf["col_ind"], buf["row_ind"], max(self.n_local, 1), counts,
                                     first_col=self.first_col)
            self._x = buf                      # the model holds its own copy of xnnz; X itself stays for the caller
            self.device = dev
        else:
            counts = count_fn(ys, K)
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
            self.model = make_model(xs, K, counts)
            self.device = torch.device("cpu")
        self.N = int(counts.sum(dtype=torch.int64).item())
        self._xtheta = torch.empty(self.model.n_masked, dtype=torch.float32, device=self.device)
        self._theta = torch.zeros(max(self.n_local, 1), dtype=torch.float32, device=self.device)
        self._full = torch.zeros(self.T, dtype=torch.float32, device=self.device)

    def logp_grad(self, theta, nograd=False):
        """theta: numpy [T] (the same on every rank).  Returns val, or (val, grad numpy float32 [T]), on every rank:
        the data terms only, like kmers_hip_model_logp_grad."""
        import torch
        dist = _dist()
        theta = np.asarray(theta).astype(np.float32)
        assert theta.shape == (self.T,)
        lo, n = self.first_col, self.n_local
        self._theta[:n].copy_(torch.from_numpy(theta[lo:lo + n]))
        self.model.xtheta_into(self._theta, self._xtheta)
        dist.all_reduce(self._xtheta, op=dist.ReduceOp.SUM)
        out = self.model.grad_from(self._xtheta, nograd=nograd)
        if nograd:
            return out
        val, grad_local = out
        self._full.zero_()
        self._full[lo:lo + n].copy_(grad_local[:n])
        dist.all_reduce(self._full, op=dist.ReduceOp.SUM)     # T floats: assembling the shards, not a reduction of data
        return val, self._full.cpu().numpy()
