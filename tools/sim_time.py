"""Time of simulating N reads into HBM from the bench's config-2 transcriptome (kmers_amd.simulate_reads), by stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kmers_amd import simulate_reads as sr
from kmers_amd.store import MODE_X, Store

args = bench.parse()
iso, _ = bench.make_inputs(args, 0, want_reads=False)
N, L, K = args.reads, 100, int(args.workload[1:])
import torch
torch.zeros(1, device="cuda")

def t(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{label:58s} {dt:8.3f} s", flush=True)
    return r

st = t("transcriptome ingest (MODE_X, L=100)", lambda: Store(iso, MODE_X, L=L))
lens = np.diff(st.rec_start.astype(np.int64)) - 1
info = t(f"sample_reads: numpy dirichlet + choice(T, {N}) + starts replay", lambda: sr.sample_reads(lens, N, L, 0.1))
ds, _ = t("simulate_reads_device (all of the above + H2D + kernel)", lambda: sr.simulate_reads_device(N, L, 0.1, iso))
y = t(f"count_y K={K}", lambda: ds.count_y(K))
print("reads", N, "positions", ds.n_pos, "k-mers counted", int(y.sum(dtype=torch.int64).item()))
