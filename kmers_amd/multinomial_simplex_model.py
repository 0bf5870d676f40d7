"""`multinomial_simplex_model` with the reference's constructor and `logp_grad` (reference
kmerexpr/multinomial_simplex_model.py:10-95, kmerexpr/utils.py:76-108), X and y resident in HBM.

What is here: the constructor, `logp_grad(theta, nograd)`, the three `initialize_iterates_*` starts and a `fit` that
hands `self.logp_grad` to the reference's own solvers (exp_grad_solver / mg, imported from an installed `kmerexpr`: they
are host-side Python outside this library's scope and only ever call `logp_grad`).  X and y live in HBM
(`self.device_model`), so the attributes `ymask / xnnz / ynnz / scratch` do not exist: `initialize_iterates_Xy` is one
`w . xnnz` product on the device, `initialize_iterates_lsq` (scipy's lsqr, host-side as in the reference) builds its
host copy of xnnz when it is called.  The log term is accumulated in float64 where the reference's numba loop works on its float32 scratch:
agreement to float32 rounding, not bit parity (tests/test_model_gpu.py states the tolerance)."""
import numpy as np

from .model import DeviceModel
from .transcriptome_reader import load_xy


class multinomial_simplex_model:
    def __init__(self, x_file=None, y_file=None, beta=1.0, lengths=None, solver_name="exp_grad", device="cuda"):
        """Same arguments as the reference (multinomial_simplex_model.py:42-44); x_file / y_file may also be a scipy
        CSR matrix / an ndarray, as its load_xy allows (transcriptome_reader.py:6-16)."""
        import torch
        x, y = load_xy(x_file, y_file)
        assert len(x.shape) == 2 and len(y.shape) == 1 and y.shape[0] == x.shape[0]
        self.M, self.T = x.shape
        self.N = np.sum(y)
        self.beta = beta
        self.name = "mirror"
        self.solver_name = solver_name
        if lengths is not None:
            assert len(lengths) == self.T
            self.lengths = lengths
        else:
            self.lengths = np.ones(self.T)
        self._torch = torch
        self._x, self._y = x, y                      # (references, for initialize_iterates_lsq only)
        self.device_model = DeviceModel.from_host(x, y, device)
        self._theta = torch.empty(self.T, dtype=torch.float32, device=self.device_model.device)

    def logp_grad(self, theta, nograd=False):
        """reference kmerexpr/utils.py:76-108: the data terms on the GPU, the (beta - 1) terms here as there."""
        theta = np.asarray(theta).astype(np.float32)
        self._theta.copy_(self._torch.from_numpy(theta))
        out = self.device_model.logp_grad(self._theta, nograd=nograd)
        val = out if nograd else out[0]
        beta, lengths = self.beta, self.lengths
        mask = theta > 0
        thetamask = theta[mask]
        if beta != 1.0:
            val += (beta - 1.0) * np.sum(np.log(thetamask / lengths[mask]))
            val -= (beta - 1.0) * np.log(np.sum(thetamask / lengths[mask]))
        if nograd:
            return val
        grad = out[1].cpu().numpy()
        if beta != 1.0:
            grad[mask] += (beta - 1.0) / thetamask
            grad[mask] -= (beta - 1.0) / (np.sum(thetamask / lengths[mask]) * lengths[mask])
        return val, grad

    def initialize_iterates_uniform(self, lengths=None):
        """reference multinomial_simplex_model.py:97-101"""
        alpha = np.ones(self.T)
        return alpha / alpha.sum()

    def initialize_iterates_Xy(self):
        """reference multinomial_simplex_model.py:103-106: theta0 = ynnz @ xnnz, normalised.  On the device: the gradient
        product with x.theta = 1, i.e. w = ynnz."""
        torch = self._torch
        ones = torch.ones(self.device_model.n_masked, dtype=torch.float32, device=self.device_model.device)
        theta0 = self.device_model.grad_from(ones)[1].double().cpu().numpy()
        return theta0 / theta0.sum()

    def initialize_iterates_lsq(self, iterations=200, mult_fact_neg=10):
        """reference multinomial_simplex_model.py:108-115 (scipy lsqr on the host, then negatives lifted and normalised)."""
        from scipy.sparse.linalg import lsqr
        ymask = self._y.nonzero()
        xnnz = self._x.tocsr()[ymask]
        ynnz = self._y[ymask] / np.sum(self._y)
        theta0 = lsqr(xnnz, ynnz / self.N, iter_lim=iterations)[0]
        theta0[theta0 <= 0] = np.min(theta0[theta0 > 0]) / mult_fact_neg
        return theta0 / theta0.sum()

    def fit(self, model_parameters, theta0=None, tol=1e-20, gtol=1e-20, n_iters=100, opt_method="mg"):
        """The reference's dispatch (multinomial_simplex_model.py:117-146) onto its own solvers, which must be importable
        (`kmerexpr.exp_grad_solver`, `kmerexpr.mg`)."""
        if theta0 is None:
            init = getattr(model_parameters, "init_iterates", "uniform")
            theta0 = (self.initialize_iterates_lsq() if init == "lsq" else
                      self.initialize_iterates_Xy() if init == "Xy" else self.initialize_iterates_uniform())
        try:  # the reference's imports (multinomial_simplex_model.py:3-4)
            from kmerexpr.exp_grad_solver import exp_grad_solver
            from kmerexpr.mg import mg
        except ImportError as e:
            raise ImportError("fit() runs the reference's solvers: install kmerexpr, or call a solver of your own with "
                              "model.logp_grad") from e
        if opt_method == "mg":
            return mg(self.logp_grad, theta0, tol=tol, max_iter=n_iters)
        return exp_grad_solver(self.logp_grad, theta0, lrs=model_parameters.lrs, tol=tol, gtol=gtol, n_iters=n_iters)
