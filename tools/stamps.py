#!/usr/bin/env python3
"""Phase breakdown of the partition kernels from the -DKK_STAMPS diagnostic build (never the product build).
    KMERS_LIB=kmers_amd/lib/libkmers_stamps.so python tools/stamps.py [--workload x8]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KMERS_LIB", os.path.join(ROOT, "kmers_amd", "lib", "libkmers_stamps.so"))

import bench  # noqa: E402
from kmers_amd import _lib  # noqa: E402
from kmers_amd.store import MODE_X, Store  # noqa: E402

sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
K = int(args.workload[1:])
iso, _ = bench.make_inputs(args, 0, False)
st = Store(iso, MODE_X)
ds = st.to_device("cuda")
buf = ds.x_buffers(K, st.n_windows(K))
lib = _lib.handle()
lib.kmers_hip_debug_stamps.restype = C.c_int
lib.kmers_hip_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
lib.kmers_hip_debug_stamps(out, 1)      # also applies KK_ABLATE
ds.build_x(buf)
assert lib.kmers_hip_debug_stamps(out, 1) == 0
n = 3
_lib.profile(True)
for _ in range(n):
    ds.build_x(buf)
print("ablate =", os.environ.get("KK_ABLATE", "0"), {r[0]: round(r[1] / n, 3) for r in _lib.profile_table()})
assert lib.kmers_hip_debug_stamps(out, 1) == 0
names = ["zero+commit", "ranking", "barrier", "digit scan", "lds scatter+prefetch", "barrier", "write-out", "end barrier"]
for which, off in (("k_scatter<bases> (old pass 0)", 0), ("k_scatter<entries> / k_emit (fused)", 8)):
    tot = sum(out[off:off + 8]) or 1
    print(which, "total cycles (sum over workgroups, per build):", tot // n)
    for i, nm in enumerate(names):
        print(f"   {nm:24s} {100.0 * out[off + i] / tot:5.1f} %")
