#!/usr/bin/env python3
"""Phase breakdown of the small-K table kernels (k_dd) from the -DKK_STAMPS diagnostic build (never the product build).
    python -m kmers_amd.build --stamps && python tools/stamps_dd.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("KMERS_LIB", os.path.join(ROOT, "kmers_amd", "lib", "libkmers_stamps.so"))

import bench  # noqa: E402
from kmers_amd import _lib  # noqa: E402
from kmers_amd.store import MODE_X, Store  # noqa: E402

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
lib.kmers_hip_debug_stamps(out, 1)
ds.build_x(buf)
assert lib.kmers_hip_debug_stamps(out, 1) == 0
n = 3
_lib.profile(True)
for _ in range(n):
    ds.build_x(buf)
print({r[0]: round(r[1] / n, 3) for r in _lib.profile_table()})
assert lib.kmers_hip_debug_stamps(out, 1) == 0
names = ["tile start", "isoform bounds / tail", "count atomics", "barrier", "claim + stage", "isoform end", "flush", "loop top"]
for which, off in (("k_dd<hist>", 0), ("k_dd<scatter>", 8)):
    tot = sum(out[off:off + 8]) or 1
    print(which, "cycles per workgroup and build:", tot // n // 256)
    for i, nm in enumerate(names):
        print(f"   {nm:24s} {100.0 * out[off + i] / tot:5.1f} %   {out[off + i] // n // 256:>10d}")
