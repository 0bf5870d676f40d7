#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output (tools/pmc.sh): per kernel, per counter: mean per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "kk::" not in k and "k_spmv" not in k and "k_logp_terms" not in k:
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
