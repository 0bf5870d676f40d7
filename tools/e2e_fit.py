"""BASELINE.json configs[4] on one GPU, by stage: transcriptome -> X, reads -> y, the model, and a short fit, with X and
y never leaving HBM between the build and the fit.  `python tools/e2e_fit.py --workload g14`

The fit is the reference's exponentiated-gradient step restated in five lines (kmerexpr/exp_grad_solver.py's update
theta <- theta * exp(lr * grad) / sum with its default step lr = 2^-1/2 / |grad_0|_inf, exp_grad_solver.py:60-64,117):
a consumer of logp_grad, here only to time the loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kmers_amd.model import DeviceModel
from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store

args = bench.parse()
iso, reads = bench.make_inputs(args, 0, want_reads=True)
K = int(args.workload[1:])
import torch
torch.zeros(1, device="cuda")                      # runtime up before timing
dev = torch.device("cuda")
T0 = time.perf_counter()

def t(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label:58s} {dt:8.3f} s", flush=True)
    return r

xs = t("transcriptome: host ingest (FASTA -> packed store)", lambda: Store(iso, MODE_X))
dx = t("  H2D", lambda: xs.to_device(dev))
buf = t("  output + workspace allocation", lambda: dx.x_buffers(K, xs.n_windows(K)))
nnz = t("  kmers_hip_build_x", lambda: dx.build_x(buf))
buf["workspace"] = None
ys = t("reads: host ingest", lambda: Store(reads, MODE_Y_FASTA))
dy = t("  H2D", lambda: ys.to_device(dev))
counts = t("  kmers_hip_count_y", lambda: dy.count_y(K))
dm = t("kmers_hip_model_create (xnnz, transposed copy)", lambda: DeviceModel(buf["data"], buf["col_ind"], buf["row_ind"],
                                                                            xs.n_rec_total, counts))
T = xs.n_rec_total
theta = torch.full((T,), 1.0 / T, dtype=torch.float32, device=dev)
val0 = dm.logp_grad(theta, nograd=True)
iters = 50
lr = 2 ** -0.5 / float(dm.logp_grad(theta)[1].abs().max())

def fit():
    global theta
    for _ in range(iters):
        val, grad = dm.logp_grad(theta)
        theta = theta * torch.exp(lr * (grad - grad.max()))      # torch here is the solver's arithmetic, O(T) per step
        theta = theta / theta.sum()
    return val

_ = torch.exp(theta - theta.max()) / theta.sum() * theta     # torch loads each op's code object on first use: not the loop's cost
from kmers_amd import _lib
_lib.profile(True)
val = t(f"{iters} exponentiated-gradient steps (logp_grad each)", fit)
print("  kernels, ms per step:", {k: round(ms / iters, 4) for k, ms, n in _lib.profile_table()})
_lib.profile(False)
print(f"total {time.perf_counter() - T0:.3f} s; nnz {nnz}, masked rows {dm.n_masked}, nnz(xnnz) {dm.nnz}; "
      f"log-likelihood {val0:.6f} -> {val:.6f}")
assert val > val0
