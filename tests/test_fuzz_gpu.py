"""Seeded fuzzing of the HIP path against the oracle (itself pinned to the compiled reference by tests/golden):
small FASTA files full of the things the parser rules are about -- invalid bytes, lower case, CR, blanks (inner and
trailing), NUL, '>' inside lines, empty lines and records, PREDICTED headers, records around K-1 / K / L bases,
several files -- through all four reference entry points at random K and L.  Bit-exact."""
import numpy as np
import pytest

from tests import kmtest

pytestmark = pytest.mark.gpu

ALPHABETS = [b"ACGT", b"ACGTN", b"ACGTacgtN", b"ACGT \r", b"ACGT\x00N", b"AC", b"ACGT>;\t"]


def _record(rng, i, min_len=0):
    hdr = b">r%d" % i
    if rng.random() < 0.1:
        hdr += b" PREDICTED: something"
    if rng.random() < 0.05:
        hdr += b" predicted"                                   # lower case: not filtered
    lines = []
    for _ in range(int(rng.integers(0, 5))):
        alpha = np.frombuffer(ALPHABETS[int(rng.integers(0, len(ALPHABETS)))], np.uint8)
        n = int(rng.choice([0, 1, 2, 3, 5, 8, 13, 40, 70, 200]))
        line = bytes(rng.choice(alpha, n)) if n else b""
        if line.startswith(b">"):
            line = b"A" + line[1:]                              # a leading '>' would start a record: covered by _file
        if rng.random() < 0.2:
            line += b" " * int(rng.integers(1, 4))
        lines.append(line)
    if sum(len(l.rstrip(b" ")) for l in lines) < min_len:       # (the reference crashes on records shorter than K-1)
        lines.append(b"ACGTTGCAACGTAC"[:max(min_len, 1)])
    return hdr + b"\n" + b"".join(l + b"\n" for l in lines)


def _file(rng, path, min_len=0):
    n = int(rng.integers(0, 12))
    body = b"".join(_record(rng, i, min_len) for i in range(n))
    r = rng.random()
    if r < 0.1:
        body = b"no header marker\nACGTACGTACGT\n" + body           # the first line is a header whatever it is
    elif r < 0.15:
        body = b"\n" + body                                     # an empty first line ends the file
    if rng.random() < 0.3 and body.endswith(b"\n"):
        body = body[:-1]                                        # no newline at the end
    with open(path, "wb") as f:
        f.write(body)


def _fastq(rng, path):
    out = []
    for i in range(int(rng.integers(0, 30))):
        n = int(rng.choice([0, 1, 4, 30, 100]))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), n)) if n else b""
        out.append(b"@q%d\n%s\n+\n%s\n" % (i, seq, b"I" * n))
    body = b"".join(out)
    if rng.random() < 0.3:
        body += b"@tail\nACGT"                                  # unterminated record
    with open(path, "wb") as f:
        f.write(body)


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_fasta(product, oracle, tmp_path, seed):
    rng = np.random.default_rng(1000 + seed)
    files = []
    for i in range(int(rng.integers(1, 4))):
        p = str(tmp_path / f"f{i}.fsa")
        _file(rng, p)
        files.append(p)
    K = int(rng.integers(1, 11))
    L = int(rng.choice([0, 0, 1, K - 1, K, K + 3, 60]))
    cat = bool(rng.random() < 0.3)
    kmtest.assert_x_equal(product.x(files, K, L, cat), oracle.x(files, K, L, cat))
    rc, y = product.y(files, K)
    rc2, y2 = oracle.y(files, K)
    assert rc == rc2 == 0 and np.array_equal(y, y2)


@pytest.mark.parametrize("seed", range(15))
def test_fuzz_fastq(product, oracle, tmp_path, seed):
    rng = np.random.default_rng(5000 + seed)
    files = []
    for i in range(int(rng.integers(1, 3))):
        p = str(tmp_path / f"q{i}.fastq")
        _fastq(rng, p)
        files.append(p)
    K = int(rng.integers(1, 10))
    rc, y = product.y(files, K, fastq=True)
    rc2, y2 = oracle.y(files, K, fastq=True)
    assert rc == rc2 == 0 and np.array_equal(y, y2)
