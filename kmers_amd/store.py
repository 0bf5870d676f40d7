"""Packed sequence store (host side) and its device-resident counterpart.

`Store` wraps the C-ABI ingest of include/kmers_hip.h (kmers_store_*): FASTA / FASTQ(.gz) files parsed with the
reference's record rules and packed to 2 bits + 1 validity bit per base.  `DeviceStore` is the same three arrays
in HBM (torch tensors are used purely as device allocations) plus thin callers of the device entry points
kmers_hip_count_y / kmers_hip_build_x."""
import ctypes as C

import numpy as np

from . import _lib

MODE_X, MODE_X_CAT, MODE_Y_FASTA, MODE_Y_FASTQ, MODE_FASTA_PY = 0, 1, 2, 3, 4


class Store:
    def __init__(self, files, mode, L=0, shard=0, n_shards=1, by_bytes=False):
        """by_bytes: the shard is a byte range of the files and only that range is parsed (kmers_store_ingest_bytes);
        record numbers are local until set_numbering() -- kmers_amd.distributed.ingest_sharded does both."""
        files = [files] if isinstance(files, str) else list(files)
        self._lib = _lib.handle()
        h = C.c_void_p()
        fn = self._lib.kmers_store_ingest_bytes if by_bytes else self._lib.kmers_store_ingest
        rc = fn(len(files), _lib.cfiles(files), mode, L, shard, n_shards, C.byref(h))
        _lib.check(rc, "kmers_store_ingest_bytes" if by_bytes else "kmers_store_ingest")
        self._h = h
        self.mode = mode

    def set_numbering(self, first_rec, n_rec_total):
        self._lib.kmers_store_set_numbering(self._h, first_rec, n_rec_total)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kmers_store_free(self._h)
            self._h = None

    __del__ = close

    n_pos = property(lambda s: int(s._lib.kmers_store_n_pos(s._h)))
    n_rec = property(lambda s: int(s._lib.kmers_store_n_rec(s._h)))
    n_rec_total = property(lambda s: int(s._lib.kmers_store_n_rec_total(s._h)))
    first_rec = property(lambda s: int(s._lib.kmers_store_first_rec(s._h)))
    n_bases = property(lambda s: int(s._lib.kmers_store_n_bases(s._h)))

    def n_windows(self, K):
        return int(self._lib.kmers_store_n_windows(self._h, K))

    def _view(self, ptr, n, dtype):
        if n == 0:
            return np.zeros(0, dtype)
        ct = {np.uint32: C.c_uint32, np.uint64: C.c_uint64}[dtype]
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(n,))

    @property
    def bases(self):   # uint32 words, valid while the store lives
        return self._view(self._lib.kmers_store_bases(self._h), self._lib.kmers_store_bases_words(self._h), np.uint32)

    @property
    def valid(self):
        return self._view(self._lib.kmers_store_valid(self._h), self._lib.kmers_store_valid_words(self._h), np.uint32)

    @property
    def rec_start(self):
        return self._view(self._lib.kmers_store_rec_start(self._h), self.n_rec + 1, np.uint64)

    def sequences(self):
        """Decode back to one bytes object per record ('N' for every invalid position).  Test helper."""
        n = self.n_pos
        pos = np.arange(n, dtype=np.int64)
        code = (self.bases[pos >> 4] >> (30 - 2 * (pos & 15)).astype(np.uint32)) & 3
        ok = (self.valid[pos >> 5] >> (pos & 31).astype(np.uint32)) & 1
        chars = np.where(ok == 1, np.frombuffer(b"ACGT", np.uint8)[code], ord("N")).astype(np.uint8)
        rs = self.rec_start.astype(np.int64)
        return [chars[rs[r]:rs[r + 1] - 1].tobytes() for r in range(self.n_rec)]

    def to_device(self, device="cuda"):
        return DeviceStore(self, device)


class DeviceStore:
    """The store's arrays resident in HBM."""

    def __init__(self, store, device="cuda"):
        import torch
        self._lib = _lib.handle()
        self.n_pos, self.n_rec, self.first_rec = store.n_pos, store.n_rec, store.first_rec
        self.n_rec_total = store.n_rec_total
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        # int32 / int64 tensors: torch has no unsigned 32/64-bit arithmetic types, and none is needed -- these
        # tensors are only device allocations whose addresses go to the C ABI.  The copy is the library's own
        # hipMemcpyAsync out of the store's pinned arrays (no pageable staging copy on the way).
        lib = self._lib
        nb, nv = int(lib.kmers_store_bases_words(store._h)), int(lib.kmers_store_valid_words(store._h))
        self.bases = torch.empty(nb, dtype=torch.int32, device=self.device)
        self.valid = torch.empty(nv, dtype=torch.int32, device=self.device)
        self.rec_start = torch.empty(self.n_rec + 1, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(lib.kmers_hip_store_upload(store._h, self.bases.data_ptr(), self.valid.data_ptr(),
                                                  self.rec_start.data_ptr(), None), "kmers_hip_store_upload")

    @classmethod
    def from_tensors(cls, bases, valid, n_pos, n_rec, rec_start=None):
        """A device store whose arrays were produced on the GPU (kmers_hip_simulate_reads): no host Store behind it.
        Enough for count_y; build_x needs rec_start."""
        self = cls.__new__(cls)
        self._lib = _lib.handle()
        self.n_pos, self.n_rec, self.first_rec, self.n_rec_total = int(n_pos), int(n_rec), 0, int(n_rec)
        self.device = bases.device
        self.bases, self.valid, self.rec_start = bases, valid, rec_start
        return self

    def y_workspace(self, K, nbytes=None):
        """Scratch for count_y at K >= 9 (None for K <= 8); keep it across repeated counts."""
        import torch
        if nbytes is None:
            nbytes = int(self._lib.kmers_hip_y_workspace_bytes(self.n_pos, K))
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device) if nbytes else None

    def count_y(self, K, counts=None, stream=None, workspace=None):
        """counts (int32[4**K] on the device) += histogram of this store.  Asynchronous; K >= 9 works in `workspace`
        (from y_workspace; allocated here when not given)."""
        import torch
        if counts is None:
            counts = torch.zeros(4 ** K, dtype=torch.int32, device=self.device)
        assert counts.dtype == torch.int32 and counts.numel() == 4 ** K and counts.is_contiguous()
        own_ws = workspace is None
        if own_ws:
            workspace = self.y_workspace(K)
        with torch.cuda.device(self.device):          # the C ABI launches on the CURRENT device
            rc = self._lib.kmers_hip_count_y_ws(self.bases.data_ptr(), self.bases.numel(), self.valid.data_ptr(),
                                                self.valid.numel(), self.n_pos, K, counts.data_ptr(),
                                                workspace.data_ptr() if workspace is not None else None,
                                                workspace.numel() if workspace is not None else 0, _stream_ptr(stream))
        _lib.check(rc, "kmers_hip_count_y_ws")
        if own_ws and workspace is not None and stream is not None:
            # the kernels on `stream` still use the scratch after this returns: keep the caching allocator from
            # handing the block out on another stream before they are done (callers who pass a workspace own it)
            workspace.record_stream(stream)
        return counts

    def x_buffers(self, K, max_nz):
        """Allocate outputs + workspace for build_x (kept by the caller across repeated builds)."""
        import torch
        M = 4 ** K
        ws = int(self._lib.kmers_hip_x_workspace_bytes(self.n_pos, self.n_rec, K))
        return dict(
            data=torch.empty(max(max_nz, 1), dtype=torch.float32, device=self.device),
            col_ind=torch.empty(max(max_nz, 1), dtype=torch.int64, device=self.device),
            row_ind=torch.empty(M + 1, dtype=torch.int64, device=self.device),
            workspace=torch.empty(ws, dtype=torch.uint8, device=self.device),
            max_nz=max_nz, K=K)

    def build_x(self, buf, stream=None):
        """CSR of X for this store into `buf` (from x_buffers).  Returns nnz.  Synchronises the stream."""
        import torch
        nnz = C.c_uint64(0)
        with torch.cuda.device(self.device):
            rc = self._call_build_x(buf, nnz, stream)
        _lib.check(rc, "kmers_hip_build_x")
        return int(nnz.value)

    def _call_build_x(self, buf, nnz, stream):
        return self._lib.kmers_hip_build_x(self.bases.data_ptr(), self.bases.numel(), self.valid.data_ptr(),
                                         self.valid.numel(), self.rec_start.data_ptr(), self.n_pos, self.n_rec,
                                         self.first_rec, buf["K"], buf["data"].data_ptr(), buf["col_ind"].data_ptr(),
                                         buf["row_ind"].data_ptr(), buf["max_nz"], buf["workspace"].data_ptr(),
                                         buf["workspace"].numel(), C.byref(nnz), _stream_ptr(stream))


def _stream_ptr(stream):
    if stream is None:
        return None
    return C.c_void_p(stream.cuda_stream)
