"""The read simulator's isoform reader against the reference's OWN python parser (kmerexpr/fasta.py:19-46).

tests/golden/fasta_py.json holds what the reference's read_fasta yields on the committed inputs (made in the build
container by oracle/make_golden_py.py, which imports the reference file by path).  Checked here, on the CPU:
  * oracle/reads_oracle.read_fasta (the restatement the other simulator tests use) yields the same records;
  * Store(..., MODE_FASTA_PY) holds the sequences of the records simulate_reads.py:78-92 keeps (not PREDICTED, len >= L)."""
import hashlib
import json
import os

import pytest

from oracle import reads_oracle as orc
from tests import kmtest

with open(os.path.join(kmtest.GOLD, "fasta_py.json")) as _f:
    GOLDEN = json.load(_f)


def _digest(records):
    return {"n": len(records), "sha256": hashlib.sha256("\n".join(h + "\t" + q for h, q in records).encode()).hexdigest()}


def _norm(text):
    return bytes(c if c in b"ACGT" else ord("N") for c in text.encode("latin-1"))


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_restated_reader_equals_reference_reader(name):
    want = GOLDEN[name]
    path = os.path.join(kmtest.INP, name)
    if "error" in want:
        with pytest.raises(Exception):
            with open(path, "r") as f:
                list(orc.read_fasta(f))
        return
    with open(path, "r") as f:
        got = [[h, q] for h, q in orc.read_fasta(f)]
    if "records" in want:
        assert got == want["records"]
    else:
        assert _digest(got) == want


@pytest.mark.parametrize("L", [0, 40])
@pytest.mark.parametrize("name", sorted(n for n, v in GOLDEN.items() if "records" in v))
def test_store_under_fasta_py_rules_equals_reference_reader(name, L):
    from kmers_amd.store import MODE_FASTA_PY, Store
    path = os.path.join(kmtest.INP, name)
    with open(path, "rb") as f:
        raw = f.read()
    if any(b >= 0x80 for b in raw):
        with pytest.raises(Exception):     # python would decode these bytes; the store refuses them
            Store(path, MODE_FASTA_PY, L=L)
        return
    want = [_norm(q) for h, q in GOLDEN[name]["records"] if "PREDICTED" not in h and len(q) >= L]
    assert Store(path, MODE_FASTA_PY, L=L).sequences() == want
    for n in (2, 5):
        got = sum((Store(path, MODE_FASTA_PY, L=L, shard=r, n_shards=n, by_bytes=True).sequences() for r in range(n)), [])
        assert got == want, n
