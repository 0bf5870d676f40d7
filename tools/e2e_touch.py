"""fasta_to_kmers_csr through the C ABI into (a) fresh np.zeros arrays (what the reference wrapper allocates) and
(b) arrays whose pages were touched before: how much of the D2H stage is first-touch page faults."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kmers_amd import _lib

args = bench.parse()
iso = bench.transcriptome_file(args)[0]
K = int(args.workload[1:])
import torch
torch.zeros(1, device="cuda")
lib = _lib.handle()
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), flush=True)
max_nz = os.path.getsize(iso)
M = 4 ** K

def call(data, row, col):
    nnz, ncols = C.c_uint64(0), C.c_int(0)
    t0 = time.perf_counter()
    rc = lib.fasta_to_kmers_csr(1, _lib.cfiles([iso]), K, 0, data.ctypes.data_as(C.POINTER(C.c_float)),
                                row.ctypes.data_as(C.POINTER(C.c_uint64)), col.ctypes.data_as(C.POINTER(C.c_uint64)),
                                max_nz, C.byref(nnz), C.byref(ncols))
    dt = time.perf_counter() - t0
    assert rc == 0
    return dt, _lib.last_call_stages()

fresh = lambda: (np.zeros(max_nz, np.float32), np.zeros(M + 1, np.uint64), np.zeros(max_nz, np.uint64))
call(*fresh())
for i in range(3):
    dt, st = call(*fresh())
    print(f"fresh   {dt:.4f} s  {st}", flush=True)
arrs = fresh()
for a in arrs:
    a[::512] = 1        # touch every page
for i in range(3):
    dt, st = call(*arrs)
    print(f"touched {dt:.4f} s  {st}", flush=True)
