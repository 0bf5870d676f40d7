#!/usr/bin/env python3
"""tools/kres.py [regex] [extra hipcc flags...] -- VGPRs / spills / LDS of the kernels of kernels.hip (compile only, no GPU)."""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "."
src = next((a for a in sys.argv[2:] if a.endswith(".hip")), "kmers_amd/csrc/kernels.hip")
extra = [a for a in sys.argv[2:] if not a.endswith(".hip")]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Iinclude", "-Ikmers_amd/csrc", *extra, src,
       "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, cwd=root, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(": ", 1)[1]
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
names = list(rows)
if not names:
    print(out[-3000:])
    sys.exit(1)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, stdin=subprocess.DEVNULL).stdout.split("\n")
for n, d in zip(names, dem):
    r = rows[n]
    d = re.sub(r"\(.*", "", d).replace("void kk::", "")
    if re.search(pat, d):
        print("%-56s vgpr %4s spill %3s sgpr %4s lds %7s occ %s" % (
            d[:56], r.get("VGPRs"), r.get("VGPRs Spill"), r.get("TotalSGPRs"), r.get("LDS Size [bytes/block]"),
            r.get("Occupancy [waves/SIMD]")))
