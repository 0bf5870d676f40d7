#!/usr/bin/env python3
"""HBM traffic per kernel launch from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc.sh, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 128-byte requests as 64: doubled;
both counters are in KiB).  Writes profiles/pmc_traffic.json, stamped with the hash of the kernel sources the passes ran
(bench.kernels_sha); bench.py quotes it as roofline.traffic only while that hash is the tree's.

    tools/pmc_traffic.py gpurun_out/pmc3 x8
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

root, workload = sys.argv[1], sys.argv[2]
LABEL = [  # kernel symbol (regex) -> the label of bench.py's per-kernel table
    (r"k_spmv<0", "k_spmv<x.theta>"),
    (r"k_spmv<1", "k_spmv<w.x>"),
    (r"k_logp_terms", "k_logp_terms"),
    (r"k_rle<", "k_rle"),
    (r"k_dd<false", "k_dd<hist>"),
    (r"k_dd<true", "k_dd<scatter>"),
    (r"k_final_scatter", "k_final_scatter"),
    (r"k_scatter<true, unsigned int, true|k_yscatter", "k_scatter<bases,y>"),
    (r"k_scatter<true|k_xscatter0", "k_scatter<bases>"),
    (r"k_scatter<false|k_scatter_wide", "k_scatter<entries>"),
    (r"k_heads_hist", "k_heads_hist"),
    (r"k_emit", "k_emit"),
    (r"k_hist_words_y", "k_hist_words_y"),
    (r"k_hist_bases", "k_hist_bases"),
    (r"k_hist_entries", "k_hist_entries"),
    (r"k_count_y_sorted", "k_count_y_sorted"),
    (r"k_count_y_lds", "k_count_y_lds"),
]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            for pat, lab in LABEL:
                if re.search(pat, r["Kernel_Name"]):
                    acc[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    break
out = {}
for lab, c in acc.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch = 2.0 * 1024.0 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        write = 1024.0 * sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        out[lab] = {"bytes": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
                    "launches_sampled": len(c["FETCH_SIZE"])}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256()
for f in ("kernels.hip", "model.hip"):                 # the same hash as bench.kernels_sha()
    with open(os.path.join(ROOT, "kmers_amd", "csrc", f), "rb") as fh:
        sha.update(fh.read())
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
allw = json.load(open(path)) if os.path.exists(path) else {}
allw[workload] = {"kernels_sha": sha.hexdigest()[:16],
                  "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH doubled (gfx950), KiB -> bytes; "
                            "mean per launch", "kernels": out}
json.dump(allw, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps(allw[workload], indent=1))
