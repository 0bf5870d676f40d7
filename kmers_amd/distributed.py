"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  SURVEY.md section 8e:

  y  reads are partitioned into contiguous record ranges of equal size (kmers_store_ingest's shard arguments);
     every rank counts its range into a private int32[4^K] in HBM; ONE all-reduce(sum) combines them.  Integer
     sums are order-independent, so the result is bit-identical to the single-GPU one.
  X  records (columns) are partitioned the same way; columns are independent, so there is no reduction: every
     rank builds the CSR of its column range (global column numbers via first_col).  For the host-side whole the
     per-row counts are combined with one all-reduce and an exclusive prefix over the ranks, and every rank writes its
     rows in place in one shared-memory CSR (concat_x_shards); merge_csr_parts (kmers_csr_merge) interleaves parts
     that are already on one host.
"""
import ctypes as C
import itertools
import os
import tempfile

import numpy as np

from . import _lib
from .store import MODE_X, MODE_X_CAT, MODE_Y_FASTA, MODE_Y_FASTQ, Store


def ingest_sharded(files, mode, L=0):
    """This rank's shard of the records of `files`, with no rank parsing the whole input: rank r scans only its byte range
    for headers (kmers_store_ingest_bytes), and one all-gather of the per-range record counts gives every shard its
    first global record number.  The two one-record-per-file modes are sharded by records as before."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    if mode in (MODE_X_CAT, MODE_Y_FASTQ):
        return Store(files, mode, L=L, shard=rank, n_shards=world)
    st, err = None, 0
    try:
        st = Store(files, mode, L=L, shard=rank, n_shards=world, by_bytes=True)
    except Exception as e:                       # every rank must reach the collective
        err = e
    mine = torch.tensor([st.n_rec if st is not None else -1], dtype=torch.int64)
    if dist.get_backend() != "gloo":
        mine = mine.cuda()
    counts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(counts, mine)
    counts = [int(c.item()) for c in counts]
    if err or min(counts) < 0:
        raise err or RuntimeError("kmers_store_ingest_bytes failed on rank %d" % counts.index(min(counts)))
    st.set_numbering(sum(counts[:rank]), sum(counts))
    return st


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


def _all_reduce_sum(t):
    """Sum a tensor over the ranks, in place.  The math stays where the tensor is; only the collective goes through the host
    when the backend cannot take device tensors (gloo: CPU tests, and rehearsals of the multi-rank flow on a one-GPU
    box -- RCCL refuses two ranks on one device)."""
    dist = _dist()
    if t.is_cuda and dist.get_backend() == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def count_y_sharded(files, K, fastq=False, device=None, reduce_scatter=False):
    """Every rank returns the full histogram (int32 numpy[4^K]) of `files`, each having counted 1/world of the records:
    ONE all-reduce(sum) of int32[4^K] over RCCL (the reference's shared-array increment, src/libkmers.cpp:213-216, across
    GPUs).  reduce_scatter=True (the consumer is the host): every rank reduces and returns only its 1/world slice of the
    bins, (first_bin, int32 numpy) -- 1/world of the bytes per GPU over xGMI and over PCIe."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    st = ingest_sharded(files, MODE_Y_FASTQ if fastq else MODE_Y_FASTA)
    counts = st.to_device(device or "cuda").count_y(K)
    if not reduce_scatter:
        _all_reduce_sum(counts)      # RCCL ring over xGMI: 4^K * 4 bytes
        return counts.cpu().numpy()
    M = counts.numel()
    per = -(-M // world)
    if counts.is_cuda and dist.get_backend() != "gloo":
        padded = counts if per * world == M else torch.cat([counts, counts.new_zeros(per * world - M)])
        mine = torch.empty(per, dtype=counts.dtype, device=counts.device)
        dist.reduce_scatter_tensor(mine, padded, op=dist.ReduceOp.SUM)
    else:                            # gloo has no reduce-scatter: the same result from the all-reduce
        _all_reduce_sum(counts)
        mine = counts[rank * per:(rank + 1) * per]
    lo = min(rank * per, M)
    return lo, mine[:max(0, min(per, M - lo))].cpu().numpy()


_SHM_DIR = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
_shm_serial = itertools.count()


def _same_host_or_raise(dist, rank, world):
    """concat_x_shards maps one set of /dev/shm files on every rank: all ranks must share a host (one node, as
    BASELINE.json's configurations are).  Every rank learns it and raises alike otherwise."""
    import socket
    names = [None] * world
    dist.all_gather_object(names, socket.gethostname())
    if len(set(names)) != 1:
        raise RuntimeError("concat_x_shards needs all ranks on one host (shared-memory output); got hosts %s" % sorted(set(names)))


def _rank_prefix(row, dist, rank, world):
    """Exclusive prefix over the ranks of a tensor (Hillis-Steele: ceil(log2 world) rounds of one send and one receive
    of the whole tensor per rank; an all-gather would hold `world` copies -- 16 GB of row pointers at K=14 on 8 GPUs)."""
    import torch
    if row.is_cuda and dist.get_backend() == "gloo":      # (the collective's hop through the host, see _all_reduce_sum)
        return _rank_prefix(row.cpu(), dist, rank, world).to(row.device)
    inc = row.clone()
    d = 1
    while d < world:
        ops, recv = [], None
        if rank + d < world:
            ops.append(dist.P2POp(dist.isend, inc.clone(), rank + d))
        if rank - d >= 0:
            recv = torch.empty_like(inc)
            ops.append(dist.P2POp(dist.irecv, recv, rank - d))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        if recv is not None:
            inc += recv
        d *= 2
    return inc - row


def concat_x_shards(data, col_ind, row_ind, nnz, n_rows):
    """The host-side concatenation of north_star's multi-GPU X: every rank holds the CSR of its column range (torch
    tensors on its GPU, or on the CPU; global column numbers; row_ind int64[n_rows + 1]); rank 0 returns
    (data f32, row_ind u64, col_ind u64) of the whole matrix as numpy arrays, the others None.

    SURVEY.md section 8e's scheme, no pickling and no merge on one rank:
      * ONE all-reduce (sum) of the per-row COUNTS, int32[n_rows] (a row holds at most one entry per column: 4 bytes
        per row, 1 GB at K = 14 -- half of what the int64 row pointers were), whose exclusive scan is the global row
        pointer;
      * rank g's entries of row r go to  global[r] + (entries of row r on ranks < g), an exclusive prefix over the
        ranks of the same count vector (_rank_prefix);
      * every rank copies its shard to the host over its own PCIe link and writes its rows in place in one POSIX
        shared-memory CSR (kmers_csr_place_rows, all host threads), so the copies and the placement run on all ranks at
        once.
    All of the arithmetic runs where the tensors are (on the GPU in production and in the one-GPU rehearsal); the backend
    only decides how the two collectives travel (_all_reduce_sum).  One node: every rank maps the same /dev/shm files."""
    import torch
    dist = _dist()
    rank, world = dist.get_rank(), dist.get_world_size()
    lib = _lib.handle()
    assert row_ind.dtype == torch.int64 and row_ind.numel() == n_rows + 1
    _same_host_or_raise(dist, rank, world)
    on_gpu = row_ind.is_cuda
    row_c = row_ind.contiguous()
    cnt = (row_c[1:] - row_c[:-1]).to(torch.int32)
    glob_cnt = _all_reduce_sum(cnt.clone())
    before = _rank_prefix(cnt, dist, rank, world)
    glob = torch.zeros(n_rows + 1, dtype=torch.int64, device=row_c.device)
    torch.cumsum(glob_cnt, 0, dtype=torch.int64, out=glob[1:])
    del glob_cnt
    total = int(glob[-1].item())
    def all_ok(err):
        """Every rank learns whether any rank failed its local step, so that all of them leave together (a rank that
        raised on its own would leave the others waiting in the next collective)."""
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int32)
        if on_gpu and dist.get_backend() != "gloo":
            flag = flag.cuda()
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            raise RuntimeError("concat_x_shards: %s" % (err if err is not None else "another rank failed"))

    # this rank's shard on the host (pinned when it comes from the GPU: the copy then runs at the link's rate; pageable if
    # the pinned allocation is refused)
    def to_host(t, n):
        if not t.is_cuda:
            return t[:n].contiguous().numpy()
        try:
            h = torch.empty(n, dtype=t.dtype, pin_memory=True)
        except RuntimeError:
            h = torch.empty(n, dtype=t.dtype)
        h.copy_(t[:n], non_blocking=True)
        return h.numpy()                     # valid after the synchronize below

    err = None
    try:
        # destination of this rank's row r = global start of the row + the row's entries on the ranks in front.  Formed
        # where the vectors are (on the GPU with RCCL) and only for the rows this rank has entries in -- a column shard at
        # K = 14 fills a sixth of its 2^28 rows --, so the list of (source start, length, destination) crosses PCIe, not
        # the row vectors (rank 0 also takes the global pointer)
        rows = torch.nonzero(cnt).squeeze(1)
        n_list = int(rows.numel())
        src = row_c[:-1][rows]
        dst = glob[:-1][rows] + before[rows]
        ln = cnt[rows]
        del before, cnt
        h_data, h_col = to_host(data, nnz), to_host(col_ind, nnz)
        h_src, h_dst, h_len = to_host(src, n_list), to_host(dst, n_list), to_host(ln, n_list)
        h_glob = to_host(glob, n_rows + 1) if rank == 0 else None
        if row_ind.is_cuda:
            torch.cuda.synchronize()
    except (RuntimeError, MemoryError) as e:
        err = e
    all_ok(err)
    # the shared output: three files in /dev/shm mapped by every rank, unlinked as soon as all ranks have written (the
    # mapping rank 0 returns keeps the pages until its arrays go)
    shapes = ((max(total, 1), np.float32), (max(total, 1), np.uint64), (n_rows + 1, np.uint64))
    base = [None]
    if rank == 0:       # a directory with room for the whole matrix; the space is reserved now (posix_fallocate), so a full
        need = sum(n * np.dtype(dt).itemsize for n, dt in shapes)      # tmpfs shows as an error here, not as SIGBUS later
        for d in (_SHM_DIR, tempfile.gettempdir()):
            stem = os.path.join(d, "kmers_x_%d_%d" % (os.getpid(), next(_shm_serial)))
            try:
                fs = os.statvfs(d)
                if fs.f_bavail * fs.f_frsize < need + (64 << 20):
                    continue
                for i, (n, dt) in enumerate(shapes):
                    fd = os.open(stem + ".%d" % i, os.O_CREAT | os.O_RDWR | os.O_EXCL, 0o600)
                    try:
                        os.posix_fallocate(fd, 0, n * np.dtype(dt).itemsize)
                    finally:
                        os.close(fd)
                base[0] = stem
                break
            except OSError:
                for i in range(3):
                    try:
                        os.unlink(stem + ".%d" % i)
                    except OSError:
                        pass
    dist.broadcast_object_list(base, src=0)           # one short string
    if base[0] is None:                               # (every rank raises: no collective is left half done)
        raise RuntimeError("concat_x_shards: no room for the whole matrix (%d non-zeros) in %s or %s"
                           % (total, _SHM_DIR, tempfile.gettempdir()))
    err, out, maps = None, None, None
    try:
        maps = [np.memmap(base[0] + ".%d" % i, dtype=dt, mode="r+", shape=(n,)) for i, (n, dt) in enumerate(shapes)]
    except (OSError, ValueError) as e:
        err = e
    # every rank has the files mapped (or has failed): rank 0 unlinks them NOW, so that nothing stays in /dev/shm if a
    # rank dies from here on; the mappings keep the pages until they go
    try:
        dist.barrier()
    finally:
        if rank == 0:
            for i in range(3):
                try:
                    os.unlink(base[0] + ".%d" % i)
                except OSError:
                    pass
    all_ok(err)
    try:
        o_data, o_col, o_row = maps
        rc = lib.kmers_csr_place_rows(n_list, h_src.ctypes.data, h_len.ctypes.data, h_dst.ctypes.data, h_data.ctypes.data,
                                      h_col.ctypes.data, o_data.ctypes.data, o_col.ctypes.data)
        _lib.check(rc, "kmers_csr_place_rows")
        if rank == 0:
            o_row[:] = h_glob.view(np.uint64)
            out = (o_data[:total], o_row, o_col[:total])
    except (OSError, ValueError, RuntimeError, MemoryError) as e:
        err = e
    all_ok(err)                                       # (also the barrier: every rank has written its rows)
    return out


def build_x_sharded(files, K, L=0, concatenate_subseq=False, device=None):
    """Rank 0 returns (data, row_ind, col_ind, n_cols) of the whole matrix; other ranks return None."""
    dist = _dist()
    rank = dist.get_rank()
    st = ingest_sharded(files, MODE_X_CAT if concatenate_subseq else MODE_X, L=L)
    M = 4 ** K
    ds = st.to_device(device or "cuda")
    buf = ds.x_buffers(K, st.n_windows(K))
    nnz = ds.build_x(buf)
    res = concat_x_shards(buf["data"], buf["col_ind"], buf["row_ind"], nnz, M)
    if rank != 0:
        return None
    return np.asarray(res[0]), np.asarray(res[1]), np.asarray(res[2]), st.n_rec_total


class ShardedModel:
    """logp_grad with X sharded by columns over the ranks (BASELINE.json configs[4]): every
    rank builds the CSR of its own isoform range and keeps it in HBM (no host-side merge), counts its share of the
    reads, and the histogram is all-reduced so that every rank masks the same rows.  One evaluation is then

        xtheta_g = xnnz_g . theta_g            per rank (its columns of theta)
        xtheta   = all_reduce(sum, xtheta_g)   float32[n_masked]: the path's one exchange step per evaluation
        val, grad_g = terms(xtheta), (ynnz / xtheta) . xnnz_g          per rank, no further exchange for grad_g

    (reference kmerexpr/utils.py:76-108 on kmerexpr/multinomial_simplex_model.py:56-62's xnnz / ynnz).  The float32
    partial products are summed by the collective, so the result depends on the number of ranks in the last bits."""

    def __init__(self, iso_files, reads_files, K, L=0, fastq=False, device=None):
        import torch
        dist = _dist()
        rank, world = dist.get_rank(), dist.get_world_size()
        self.K = K
        xs = ingest_sharded(iso_files, MODE_X, L=L)
        ys = ingest_sharded(reads_files, MODE_Y_FASTQ if fastq else MODE_Y_FASTA)
        self.T, self.first_col, self.n_local = xs.n_rec_total, xs.first_rec, xs.n_rec
        dev = torch.device(device or "cuda")
        counts = ys.to_device(dev).count_y(K)
        _all_reduce_sum(counts)
        dx = xs.to_device(dev)
        buf = dx.x_buffers(K, xs.n_windows(K))
        dx.build_x(buf)
        buf["workspace"] = None
        from . import model as _model
        self.model = _model.DeviceModel(buf["data"], buf["col_ind"], buf["row_ind"], max(self.n_local, 1), counts,
                                        first_col=self.first_col)
        self._x = buf                      # the model holds its own copy of xnnz; X itself stays for the caller
        self.device = counts.device
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
        _all_reduce_sum(self._xtheta)
        out = self.model.grad_from(self._xtheta, nograd=nograd)
        if nograd:
            return out
        val, grad_local = out
        self._full.zero_()
        self._full[lo:lo + n].copy_(grad_local[:n])
        _all_reduce_sum(self._full)     # T floats: assembling the shards, not a reduction of data
        return val, self._full.cpu().numpy()
