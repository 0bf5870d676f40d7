"""Skew of the bench's y workload: heaviest bins and heaviest partition buckets (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import bench
from kmers_amd.store import MODE_Y_FASTA, Store

sys.argv = [sys.argv[0]] + sys.argv[1:]
ap_args = bench.parse()
K = int(ap_args.workload[1:])
iso, reads = bench.make_inputs(ap_args, 0, want_reads=True)
st = Store(reads, MODE_Y_FASTA)
ds = st.to_device()
c = ds.count_y(K).to(torch.int64)
tot = int(c.sum())
top = torch.topk(c, 10).values.tolist()
print("windows", tot, "bins", c.numel(), "nonzero", int((c > 0).sum()), "top bins", top)
for bits in (8, 16):
    if 2 * K - bits < 0:
        continue
    b = c.view(1 << bits, -1).sum(1)
    print(f"top-{bits}-bit buckets: max {int(b.max())} mean {tot / b.numel():.0f} ratio {float(b.max()) * b.numel() / tot:.1f}",
          "top5", torch.topk(b, 5).values.tolist())
