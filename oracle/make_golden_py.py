#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- build container only (/root/reference does not exist on the GPU box).

Pins the isoform reader of the read simulator to the reference's OWN python parser: imports
/root/reference/kmerexpr/fasta.py by path (standard library only; the package __init__, which needs MKL, is never run),
feeds it the awkward FASTA files of tests/test_simreads_cpu.py, a dozen seeded fuzz files and every edge_*.fsa / ref_*.fsa
input that python can decode, and writes what it yields -- (header, sequence) per record, or the exception's type -- to
tests/golden/fasta_py.json.  The generated input files are committed beside the other golden inputs
(tests/golden/inputs/py_*.fsa); tests/test_fasta_py_golden.py checks oracle/reads_oracle.read_fasta and
Store(..., MODE_FASTA_PY) against the fixture.  Only data travels; the reference's source never does.

    python oracle/make_golden_py.py            (reference: kmerexpr/fasta.py:19-46, simulate_reads.py:78-92)
"""
import glob
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/kmerexpr/fasta.py"
INP = os.path.join(ROOT, "tests", "golden", "inputs")


def digest(records):
    import hashlib
    return {"n": len(records), "sha256": hashlib.sha256("\n".join(h + "\t" + q for h, q in records).encode()).hexdigest()}


def main():
    from tests import kmtest
    spec = importlib.util.spec_from_file_location("ref_fasta", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    made = kmtest.write_fasta_py_inputs(INP)
    files = sorted(made + glob.glob(os.path.join(INP, "edge_*.fsa")) + glob.glob(os.path.join(INP, "ref_*.fsa")) +
                   [os.path.join(INP, "syn_small.fsa")])
    out = {}
    for path in files:
        name = os.path.basename(path)
        try:
            with open(path, "r") as f:            # text mode, universal newlines: exactly how simulate_reads.py:78 opens it
                out[name] = {"records": [[r.header, r.sequence] for r in ref.read_fasta(f)]}
        except Exception as e:                    # bytes python cannot decode
            out[name] = {"error": type(e).__name__}
        if "records" in out[name] and len(out[name]["records"]) > 200:   # large: the count and a checksum instead of the text
            out[name] = digest(out[name]["records"])
    with open(os.path.join(ROOT, "tests", "golden", "fasta_py.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"{len(out)} files -> tests/golden/fasta_py.json")


if __name__ == "__main__":
    main()
