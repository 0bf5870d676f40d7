"""Device-resident state of the reference's multinomial_simplex_model and the data terms of its logp_grad
(kmers_hip_model_* of include/kmers_hip.h; reference kmerexpr/multinomial_simplex_model.py:56-62 and
kmerexpr/utils.py:76-108).  torch tensors are device allocations only; the products run in the library's kernels."""
import ctypes as C

import numpy as np

from . import _lib
from .store import _stream_ptr


class DeviceModel:
    """xnnz = x[y != 0] (and its column-sorted copy), ynnz = y[y != 0] / sum(y), in HBM."""

    def __init__(self, data, col_ind, row_ind, n_cols, y, stream=None, first_col=0):
        """data f32[>=nnz], col_ind i64/u64[>=nnz], row_ind i64/u64[n_rows + 1], y int32[n_rows]: torch tensors on the
        GPU, e.g. the buffers DeviceStore.build_x filled and the counts of DeviceStore.count_y.  For a column shard
        (multi-GPU fit) n_cols is the shard's width, first_col its first global column and y the histogram of all
        reads."""
        import torch
        self._lib = _lib.handle()
        assert data.dtype == torch.float32 and col_ind.dtype == torch.int64 and row_ind.dtype == torch.int64
        assert y.dtype == torch.int32 and y.numel() == row_ind.numel() - 1
        assert all(t.is_cuda and t.is_contiguous() for t in (data, col_ind, row_ind, y))
        self.device = y.device
        self.n_rows, self.n_cols = y.numel(), int(n_cols)
        h = C.c_void_p()
        with torch.cuda.device(self.device):     # the C ABI allocates and launches on the CURRENT device
            rc = self._lib.kmers_hip_model_create_shard(data.data_ptr(), col_ind.data_ptr(), row_ind.data_ptr(),
                                                        self.n_rows, int(first_col), self.n_cols, y.data_ptr(),
                                                        _stream_ptr(stream), C.byref(h))
        _lib.check(rc, "kmers_hip_model_create_shard")
        self._h = h
        self.n_masked = int(self._lib.kmers_hip_model_n_masked(h))
        self.nnz = int(self._lib.kmers_hip_model_nnz(h))
        self.sum_y = int(self._lib.kmers_hip_model_sum_y(h))
        self._grad = torch.empty(self.n_cols, dtype=torch.float32, device=self.device)

    @classmethod
    def from_host(cls, x, y, device="cuda"):
        """x: scipy CSR (float32), y: integer ndarray of x.shape[0] counts."""
        import torch
        x = x.tocsr()
        dev = torch.device(device)
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        nz = max(x.nnz, 1)
        data = torch.zeros(nz, dtype=torch.float32, device=dev)
        col = torch.zeros(nz, dtype=torch.int64, device=dev)
        data[:x.nnz] = up(x.data, np.float32)
        col[:x.nnz] = up(x.indices, np.int64)
        return cls(data, col, up(x.indptr, np.int64), x.shape[1], up(y, np.int32))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kmers_hip_model_free(self._h)
            self._h = None

    __del__ = close

    def logp_grad(self, theta, nograd=False, stream=None):
        """theta: float32 torch tensor [n_cols] on the device.  Returns val (float) or (val, grad): grad is a float32
        device tensor owned by the model and overwritten by the next call."""
        import torch
        assert theta.dtype == torch.float32 and theta.numel() == self.n_cols and theta.is_cuda and theta.is_contiguous()
        val = C.c_double(0.0)
        with torch.cuda.device(self.device):
            rc = self._lib.kmers_hip_model_logp_grad(self._h, theta.data_ptr(), 1 if nograd else 0, C.byref(val),
                                                     None if nograd else self._grad.data_ptr(), _stream_ptr(stream))
        _lib.check(rc, "kmers_hip_model_logp_grad")
        return val.value if nograd else (val.value, self._grad)

    def xtheta_into(self, theta, out, stream=None):
        """out (float32 device tensor [n_masked]) = xnnz . theta of this (column shard of the) model; asynchronous."""
        import torch
        assert theta.dtype == torch.float32 and theta.numel() == self.n_cols and theta.is_cuda and theta.is_contiguous()
        assert out.dtype == torch.float32 and out.numel() == self.n_masked and out.is_cuda and out.is_contiguous()
        with torch.cuda.device(self.device):
            rc = self._lib.kmers_hip_model_xtheta_into(self._h, theta.data_ptr(), out.data_ptr(), _stream_ptr(stream))
        _lib.check(rc, "kmers_hip_model_xtheta_into")
        return out

    def grad_from(self, xtheta, nograd=False, stream=None):
        """val (and this model's gradient entries) from a given xtheta, e.g. the sum over column shards."""
        import torch
        val = C.c_double(0.0)
        with torch.cuda.device(self.device):
            rc = self._lib.kmers_hip_model_grad_from(self._h, xtheta.data_ptr(), 1 if nograd else 0, C.byref(val),
                                                     None if nograd else self._grad.data_ptr(), _stream_ptr(stream))
        _lib.check(rc, "kmers_hip_model_grad_from")
        return val.value if nograd else (val.value, self._grad)

    def _read(self, what, dtype):
        import torch
        out = np.empty(self.n_masked, dtype)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.kmers_hip_model_read(self._h, what, out.ctypes.data), "kmers_hip_model_read")
        return out

    def rows(self):
        """ymask: ascending row numbers of the non-zeros of y (int64 numpy)."""
        return self._read(0, np.uint32).astype(np.int64)

    def xtheta(self):
        """xnnz . theta of the last logp_grad call (float32 numpy)."""
        return self._read(1, np.float32)
