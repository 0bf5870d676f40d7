"""One-off check above 2^31 positions (entry indices beyond int32, two y chunks): X at K=8 with 4-byte entries ==
8-byte entries, structure checks, and y at K=12 partition path == atomics kernel.  ~2.6e9 bases, ~1.1 M records.
Needs ~100 GB of HBM and a few minutes (mostly generating the FASTA)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kmers_amd import synth
from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store

t0 = time.time()
path = "/tmp/big_tx.fsa"
synth.synth_transcriptome(path, n_records=1_100_000, total_bases=2_600_000_000, seed=5)
print("generated", round(time.time() - t0), "s", os.path.getsize(path) / 1e9, "GB", flush=True)
st = Store(path, MODE_X)
print("ingested: positions", st.n_pos, "records", st.n_rec, round(time.time() - t0), "s", flush=True)
assert st.n_pos > 2 ** 31
ds = st.to_device()
K = 8
buf = ds.x_buffers(K, st.n_windows(K))
from kmers_amd import _lib as _l
_knob = _l.handle().kmers_hip_debug_knob
nnz = ds.build_x(buf)
row = buf["row_ind"]
assert int(row[-1]) == nnz and bool((row[1:] >= row[:-1]).all())
col = buf["col_ind"][:nnz]
assert int(col.max()) < st.n_rec
sums = torch.zeros(st.n_rec, dtype=torch.float64, device=ds.device).index_add_(0, col, buf["data"][:nnz].double())
live = sums > 0
assert float((sums[live] - 1).abs().max()) < 1e-4
print("x8 narrow ok: nnz", nnz, "> 2^31:", nnz > 2 ** 31, flush=True)
keep_row = row.clone()
chk = (int(col.sum()), float(buf["data"][:nnz].double().sum()))
_knob(b"force_wide", 1)
assert ds.build_x(buf) == nnz
assert torch.equal(buf["row_ind"], keep_row)
assert (int(buf["col_ind"][:nnz].sum()), float(buf["data"][:nnz].double().sum())) == chk
_knob(b"force_wide", 0)
print("x8 wide == narrow (row pointer equal, column and value checksums equal)", flush=True)
del buf
torch.cuda.empty_cache()
sy = Store(path, MODE_Y_FASTA).to_device()
y = sy.count_y(12)
_knob(b"y_atomics", 1)
assert torch.equal(sy.count_y(12), y)
_knob(b"y_atomics", 0)
print("y12 partition (2 chunks) == atomics; total", int(y.sum(dtype=torch.int64)), flush=True)
os.remove(path)
print("big check ok", round(time.time() - t0), "s")
