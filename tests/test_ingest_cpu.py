"""Host ingest (kmers_amd/csrc/ingest.cpp) on CPU: the packed store, decoded back to sequences and run through
the oracle, must reproduce every golden fixture -- so record rules, filters, column numbering, NUL cut, FASTQ
joining and the 2-bit/validity packing are all pinned without a GPU.  Also: the C-ABI library loads and exports
every symbol that include/*.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from kmers_amd import _lib
from kmers_amd.store import MODE_X, MODE_X_CAT, MODE_Y_FASTA, MODE_Y_FASTQ, Store
from tests import kmtest

CASES = kmtest.manifest()


def _mode(case):
    if case["kind"] == "x":
        return MODE_X_CAT if case["cat"] else MODE_X
    return MODE_Y_FASTQ if case["kind"] == "y_fastq" else MODE_Y_FASTA


def _rewrite(seqs, path):
    with open(path, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">r%d\n" % i + s + b"\n")


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_store_roundtrip_matches_golden(oracle, case, tmp_path):
    st = Store(kmtest.case_paths(case), _mode(case), L=case.get("L", 0))
    seqs = st.sequences()
    assert st.n_rec == st.n_rec_total == len(seqs)
    assert st.n_pos == sum(len(s) + 1 for s in seqs)
    K = case["K"]
    assert st.n_windows(K) == sum(max(len(s) - K + 1, 0) for s in seqs)
    p = str(tmp_path / "rt.fsa")
    _rewrite(seqs, p)
    if case["kind"] == "x":
        out = oracle.x([p], K, 0, False, max_nz=st.n_pos + 16)
        kmtest.assert_x_equal_golden(case, *out)
    else:
        rc, y = oracle.y([p], K)
        kmtest.assert_y_equal_golden(case, rc, y)
    st.close()


def test_store_padding_is_zero():
    st = Store([os.path.join(kmtest.INP, "syn_small.fsa")], MODE_X)
    n = st.n_pos
    assert st.bases.shape[0] >= (n + 15) // 16 + 4 and st.valid.shape[0] >= (n + 31) // 32 + 4
    assert not st.bases[(n + 15) // 16:].any() and not st.valid[(n + 31) // 32:].any()
    # separators are invalid positions
    rs = st.rec_start.astype(np.int64)
    sep = rs[1:] - 1
    assert not ((st.valid[sep >> 5] >> (sep & 31).astype(np.uint32)) & 1).any()


@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_store_shards_partition_records(n_shards):
    files = [os.path.join(kmtest.INP, "syn_small.fsa"), os.path.join(kmtest.INP, "ref_test4.fsa")]
    whole = Store(files, MODE_X).sequences()
    got, first = [], 0
    sizes = []
    for s in range(n_shards):
        st = Store(files, MODE_X, shard=s, n_shards=n_shards)
        assert st.first_rec == first and st.n_rec_total == len(whole)
        got += st.sequences()
        first += st.n_rec
        sizes.append(st.n_pos)
    assert got == whole
    assert max(sizes) - min(sizes) <= max(len(s) for s in whole) + 1   # balanced by positions


def test_x_range_count_leaves_room_for_the_longest_record(tmp_path):
    """kmers_x_range_count (the chunked route of the file-level X calls, libkmers.cpp build_files): with the count it
    returns, no range of the real ingest is above the cap -- on record lengths chosen to overshoot a plain
    ceil(n_pos / cap): the long records sit right where the even split would cut."""
    from kmers_amd import _lib
    lib = _lib.handle()
    rng = np.random.default_rng(5)
    for trial in range(6):
        lens = rng.integers(20, 400, size=300)
        lens[rng.integers(0, 300, size=4)] = rng.integers(3000, 6000, size=4)
        p = str(tmp_path / f"t{trial}.fsa")
        with open(p, "w") as f:
            for i, n in enumerate(lens):
                f.write(f">r{i}\n" + "".join(rng.choice(list("ACGT"), size=int(n))) + "\n")
        whole = Store([p], MODE_X)
        n_pos, longest = whole.n_pos, int(np.diff(whole.rec_start.astype(np.int64)).max())
        for cap in (longest, longest + 1, longest + 50, 2 * longest, n_pos // 3, n_pos - 1):
            n = lib.kmers_x_range_count(n_pos, whole.n_rec, longest, cap)
            if n == 0:
                assert (n_pos + (cap - longest)) // (cap - longest + 1) > 4096
                continue
            sizes = [Store([p], MODE_X, shard=s, n_shards=n).n_pos for s in range(n)]
            assert sum(sizes) == n_pos and max(sizes) <= cap, (trial, cap, n, max(sizes))
        assert lib.kmers_x_range_count(n_pos, whole.n_rec, longest, n_pos) == 1
        assert lib.kmers_x_range_count(n_pos, whole.n_rec, longest, longest - 1) == 0       # one record above the cap
    assert lib.kmers_x_range_count(10, 2 ** 32, 5, 0) == 3                                  # by records: 2^31 - 1 a call
    assert lib.kmers_x_range_count(2 ** 34, 1000, 2 ** 20, 0) == 5                          # the built-in cap


@pytest.mark.parametrize("n_shards", [2, 3, 7, 64])
def test_byte_range_shards_are_the_whole_in_order(tmp_path, n_shards):
    """kmers_store_ingest_bytes: every rank parses only its byte range; the shards in order are the whole record list
    under all three record rules, with headers on and next to the range boundaries, several files, an empty file, a
    file whose empty first line ends it (C++ rules) and one that starts with text (fasta.py rules)."""
    from kmers_amd.store import MODE_FASTA_PY
    from tests.test_simreads_cpu import awkward_fasta
    e1 = tmp_path / "empty.fsa"
    e1.write_bytes(b"")
    e2 = tmp_path / "blankfirst.fsa"
    e2.write_bytes(b"\n>a\nACGTACGT\n>b\nACGT\n")
    tiny = tmp_path / "tiny.fsa"
    tiny.write_bytes(b">t1\nAC\n>t2\n\n>t3 PREDICTED\nACGT\n>t4\nACGTACGTAC")
    files = [os.path.join(kmtest.INP, "syn_small.fsa"), str(e1), awkward_fasta(str(tmp_path / "awk.fsa")), str(e2),
             os.path.join(kmtest.INP, "ref_test4.fsa"), str(tiny)]
    for mode, L in ((MODE_X, 0), (MODE_X, 9), (MODE_Y_FASTA, 0), (MODE_FASTA_PY, 0), (MODE_FASTA_PY, 100)):
        whole = Store(files, mode, L=L).sequences()
        got = []
        for s in range(n_shards):
            st = Store(files, mode, L=L, shard=s, n_shards=n_shards, by_bytes=True)
            assert st.first_rec == 0 and st.n_rec_total == st.n_rec          # local until set_numbering
            st.set_numbering(len(got), len(whole))
            assert st.first_rec == len(got) and st.n_rec_total == len(whole)
            got += st.sequences()
        assert got == whole, (mode, L)
    # every split point of a small file, including the ones on a header and on the byte before it
    size = os.path.getsize(str(tiny))
    whole = Store([str(tiny)], MODE_Y_FASTA).sequences()
    for n in range(2, size + 2):
        assert sum((Store([str(tiny)], MODE_Y_FASTA, shard=s, n_shards=n, by_bytes=True).sequences() for s in range(n)), []) == whole
    with pytest.raises(RuntimeError):
        Store([str(tiny)], MODE_X_CAT, shard=0, n_shards=2, by_bytes=True)    # one record per file: sharded by records
    with pytest.raises(RuntimeError, match="Failed to open FASTA"):
        Store([str(tmp_path / "nope.fsa")], MODE_X, shard=0, n_shards=2, by_bytes=True)


def test_threads_do_not_change_the_store(monkeypatch):
    files = [os.path.join(kmtest.INP, "syn_small.fsa")]
    monkeypatch.setenv("KMER_NUM_THREADS", "1")
    a = Store(files, MODE_X)
    monkeypatch.setenv("KMER_NUM_THREADS", "7")
    b = Store(files, MODE_X)
    assert np.array_equal(a.bases, b.bases) and np.array_equal(a.valid, b.valid)
    assert np.array_equal(a.rec_start, b.rec_start)


def test_ingest_errors(tmp_path):
    missing = str(tmp_path / "nope.fsa")
    with pytest.raises(RuntimeError, match="Failed to open FASTA"):
        Store([missing], MODE_X)
    with pytest.raises(RuntimeError, match="Failed to open FASTA"):
        Store([missing], MODE_Y_FASTA)
    st = Store([missing, os.path.join(kmtest.INP, "edge_two_a.fsa")], MODE_X_CAT)   # empty column, no error
    assert st.n_rec == 2 and st.sequences()[0] == b""
    st = Store([missing], MODE_Y_FASTQ)                                             # reference: stderr + empty
    assert st.n_rec == 1 and st.n_bases == 0


def test_negative_L_skips_everything():
    st = Store([os.path.join(kmtest.INP, "ref_test4.fsa")], MODE_X, L=-1)   # unsigned compare in the reference
    assert st.n_rec == 0


def test_scalars_and_exports(product):
    import json
    with open(os.path.join(kmtest.GOLD, "scalars.json")) as f:
        scal = json.load(f)
    for w, (kid, ok) in scal.items():
        assert product.fn("valid_kmer")(w.encode(), len(w)) == ok
        assert product.fn("kmer_to_id")(w.encode(), len(w)) == kid
    # every function declared in include/*.h is exported
    lib = C.CDLL(kmtest.PRODUCT_SO)
    names = set()
    for h in ("libkmers.h", "kmers_hip.h"):
        src = open(os.path.join(kmtest.ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(k?mers?_\w+|kmer_to_id|valid_kmer|fast[aq]\w+)\s*\(", src))
    assert {"kmer_to_id", "valid_kmer", "fasta_count_kmers", "fastq_gz_count_kmers", "fasta_to_kmers_csr",
            "fasta_to_kmers_csr_cat_subseq", "kmers_hip_build_x", "kmers_hip_count_y", "kmers_hip_count_y_ws",
            "kmers_hip_y_workspace_bytes"} <= names
    for n in names:
        assert hasattr(lib, n), n


def test_no_gpu_fails_loudly(product):
    """Without a HIP device the product has no fallback: the file-level calls return KMERS_ERR_NO_DEVICE."""
    if _lib.handle().kmers_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    p = [os.path.join(kmtest.INP, "ref_test1.fsa")]
    assert product.x(p, 1)[0] == -10
    assert product.y(p, 1)[0] == -10
    from kmers_amd import transcriptome_reader as tr
    with pytest.raises(RuntimeError, match="no usable HIP device"):
        tr.transcriptome_to_x(1, p[0])


@pytest.mark.parametrize("mode", [MODE_X, MODE_Y_FASTA])
def test_many_small_records(tmp_path, mode):
    """3e5 tiny records (several of the ingest's 65,536-record numbering chunks, two files): numbering, filters,
    lengths and the packed bases against a plain Python parse, whole and in shards."""
    rng = np.random.default_rng(77)
    want = []                                             # kept records, in file order
    L = 6
    paths = []
    for fi, n in enumerate((200_000, 100_000)):
        p = str(tmp_path / f"m{fi}.fsa")
        paths.append(p)
        lens = rng.integers(0, 40, n)
        pred = rng.random(n) < 0.02
        bases = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, int(lens.sum()), p=[.24, .24, .24, .24, .04])]
        off = np.concatenate([[0], np.cumsum(lens)])
        with open(p, "wb") as f:
            chunks = []
            for i in range(n):
                s = bases[off[i]:off[i + 1]].tobytes()
                chunks.append(b">r%d%s\n%s\n" % (i, b" PREDICTED" if pred[i] else b"", s))
                if pred[i] or (mode == MODE_X and len(s) < L):
                    continue
                want.append(s)
            f.write(b"".join(chunks))
    st = Store(paths, mode, L=L)
    assert st.n_rec == len(want) and st.n_rec_total == len(want)
    got = st.sequences()
    assert got == want
    # three shards: contiguous, complete, same records
    merged, first = [], 0
    for s in range(3):
        sh = Store(paths, mode, L=L, shard=s, n_shards=3)
        assert sh.first_rec == first and sh.n_rec_total == len(want)
        merged += sh.sequences()
        first += sh.n_rec
    assert merged == want


REF_WRAPPER = "/root/reference/kmerexpr/libkmers.py"


@pytest.mark.skipif(not os.path.exists(REF_WRAPPER), reason="the reference tree exists in the build container only")
def test_unmodified_reference_wrapper_binds_and_failures_are_loud(tmp_path):
    """The zero-change route: the reference's own kmerexpr/libkmers.py (loaded by path, unmodified) on the product .so.
    All six symbols bind with the reference's argtypes (kmerexpr/libkmers.py:37-74).  Without a GPU the file-level calls
    cannot compute: the failure is one line on stderr, and with KMERS_REFERENCE_CODES=1 the code is folded into -1 so
    that the unchanged wrapper raises (it only looks for -1 / -2, :102-103, :173-176).  Runs in a child process: the
    wrapper finds the library relative to the working directory."""
    import subprocess
    import sys
    lib_dir = tmp_path / "lib"
    lib_dir.mkdir()
    os.symlink(kmtest.PRODUCT_SO, lib_dir / "libkmers.so")
    fsa = os.path.join(kmtest.INP, "ref_test1.fsa")
    script = f"""
import importlib.util, sys
import numpy as np
spec = importlib.util.spec_from_file_location("ref_libkmers", {REF_WRAPPER!r})
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
for name in ("fasta_count_kmers", "fastq_gz_count_kmers", "fasta_to_kmers_csr", "fasta_to_kmers_csr_cat_subseq"):
    assert getattr(m, "_" + name).argtypes is not None, name       # bound by the wrapper with its own argtypes
assert m._libkmers.kmer_to_id(b"ACGT", 4) == 27 and m._libkmers.valid_kmer(b"ACGT", 4)   # the two it leaves unbound
try:
    y = m.fasta_count_kmers({fsa!r}, 1)
    print("Y", int(y.sum()))
except RuntimeError as e:
    print("RAISED", e)
"""
    env = dict(os.environ, KMERS_NO_TORCH="1")
    env.pop("KMERS_REFERENCE_CODES", None)
    have_gpu = _lib.handle().kmers_hip_device_count() > 0
    r = subprocess.run([sys.executable, "-c", script], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    if have_gpu:
        assert "Y 10" in r.stdout              # test1.fsa: ACGT + ACGTAA... (10 bases)
        return
    assert "Y 0" in r.stdout                   # the unchanged wrapper does not look at -10 ...
    assert "libkmers: fasta_count_kmers failed with code -10" in r.stderr and "no usable HIP device" in r.stderr
    env["KMERS_REFERENCE_CODES"] = "1"         # ... unless the codes are folded into the one it raises on
    r = subprocess.run([sys.executable, "-c", script], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "RAISED" in r.stdout and "no usable HIP device" in r.stderr


# ---- multi-member FASTQ.gz on all host threads (reference src/libkmers.cpp:240-300 is one serial
# gzread loop): members are found by speculation and only the chain from offset 0 counts.

def _fastq_text(rng, n_reads, read_len=100, weird=False):
    out = bytearray()
    bases = np.frombuffer(b"ACGT", np.uint8)
    for i in range(n_reads):
        seq = bytes(rng.choice(bases, read_len))
        qual = b"I" * read_len
        if weird and i % 97 == 0:
            qual = b"\x1f\x8b\x08\x00" + qual[4:]       # looks like a gzip member header inside the text
        out += b"@r%d\n%s\n+\n%s\n" % (i, seq, qual)
    return bytes(out)


def _members(text, n, level=6, cut_lines=False):
    import gzip
    step = len(text) // n
    cuts = [0] + [step * i + (37 if cut_lines else 0) for i in range(1, n)] + [len(text)]
    return b"".join(gzip.compress(text[a:b], compresslevel=level) for a, b in zip(cuts, cuts[1:]))


def _reads_of(text):
    lines = text.split(b"\n")
    terminated = lines[:-1]                                  # the last piece has no newline behind it
    return b"".join(terminated[1::4])


@pytest.mark.parametrize("level,weird", [(6, False), (0, True), (9, True)])
def test_multi_member_fastq_gz_equals_serial(tmp_path, monkeypatch, level, weird):
    """16 members cut in the middle of lines; level 0 keeps the text verbatim in the file, so the planted header
    look-alikes are candidates that the chain must skip.  One thread (serial gzread) == eight threads (members side by
    side) == the lines read in python."""
    rng = np.random.default_rng(11 + level)
    text = _fastq_text(rng, 4000, weird=weird)
    p = tmp_path / "m.fastq.gz"
    p.write_bytes(_members(text, 16, level, cut_lines=True))
    want = [bytes(c if c in b"ACGT" else ord("N") for c in _reads_of(text))]
    for threads in ("1", "8"):
        monkeypatch.setenv("KMER_NUM_THREADS", threads)
        assert Store(str(p), MODE_Y_FASTQ).sequences() == want, threads
    # a truncated last member, trailing garbage, an unterminated tail: whatever the serial reader makes of it
    blob = p.read_bytes()
    for i, damaged in enumerate((blob[:-9], blob + b"garbage", _members(text[:-40], 5, level))):
        q = tmp_path / f"d{i}.fastq.gz"
        q.write_bytes(damaged)
        monkeypatch.setenv("KMER_NUM_THREADS", "1")
        serial = Store(str(q), MODE_Y_FASTQ).sequences()
        monkeypatch.setenv("KMER_NUM_THREADS", "8")
        assert Store(str(q), MODE_Y_FASTQ).sequences() == serial, i


def test_multi_member_fastq_gz_is_faster_on_more_threads(tmp_path, monkeypatch):
    import time
    rng = np.random.default_rng(5)
    text = _fastq_text(rng, 150_000)                        # ~32 MB of text
    p = tmp_path / "big.fastq.gz"
    p.write_bytes(_members(text * 4, 16, 6))
    times = {}
    for threads in ("1", "8"):
        monkeypatch.setenv("KMER_NUM_THREADS", threads)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            st = Store(str(p), MODE_Y_FASTQ)
            best = min(best, time.perf_counter() - t0)
            n = st.n_bases
            st.close()
        times[threads] = best
        assert n == 4 * 150_000 * 100
    assert times["8"] < 0.85 * times["1"], times           # (measured here: 0.74 s -> 0.34 s; the line scan stays serial)


def test_simd_packer_equals_scalar_packer(tmp_path):
    """The SSSE3 packer (16 bytes per step, any bit alignment of the destination) against the byte-at-a-time one
    (KMERS_NO_SIMD=1, read when the library is loaded: hence two child processes) on ragged files that hold every byte
    value: the packed bases, the validity mask and the record table must be the same words.  Also with the arrays
    poisoned before packing (KMERS_INGEST_POISON): only the words between pieces are zeroed, the rest must be written."""
    import hashlib
    import subprocess
    import sys
    rng = np.random.default_rng(99)
    paths = []
    for i in range(6):
        p = tmp_path / f"f{i}.fsa"
        p.write_bytes(kmtest.fuzz_fasta_bytes(100 + i))
        paths.append(str(p))
    p = tmp_path / "allbytes.fsa"      # every byte value in sequence lines of every length 1..70
    with open(p, "wb") as f:
        for n in range(1, 71):
            f.write(b">r%d\n" % n)
            line = bytes(rng.integers(0, 256, size=n, dtype=np.uint8)).replace(b"\n", b"A").replace(b">", b"C")
            f.write(line + b"\n" + bytes(rng.choice(np.frombuffer(b"ACGTacgtN", np.uint8), size=3 * n)) + b"\n")
    paths.append(str(p))
    code = ("import sys, hashlib, os\n"
            "os.environ['KMERS_NO_TORCH'] = '1'\n"
            f"sys.path.insert(0, {kmtest.ROOT!r})\n"
            "from kmers_amd.store import Store, MODE_X, MODE_Y_FASTA, MODE_FASTA_PY\n"
            "h = hashlib.sha256()\n"
            "for mode in (MODE_X, MODE_Y_FASTA):\n"
            "    for p in sys.argv[1:]:\n"
            "        st = Store([p], mode)\n"
            "        for a in (st.bases, st.valid, st.rec_start):\n"
            "            h.update(a.tobytes())\n"
            "st = Store(sys.argv[1:], MODE_X)\n"
            "h.update(st.bases.tobytes()); h.update(st.valid.tobytes())\n"
            "print(h.hexdigest())\n")
    out = {}
    for simd in ("1", "0", "poison", "poison0"):    # poison: the arrays start as 0xA5 instead of whatever malloc gave
        env = dict(os.environ)
        env.pop("KMERS_NO_SIMD", None)
        env.pop("KMERS_INGEST_POISON", None)
        if simd in ("0", "poison0"):
            env["KMERS_NO_SIMD"] = "1"
        if simd.startswith("poison"):
            env["KMERS_INGEST_POISON"] = "1"
        out[simd] = subprocess.check_output([sys.executable, "-c", code] + paths, env=env, text=True).strip()
    assert out["1"] == out["0"] == out["poison"] == out["poison0"] and len(out["1"]) == 64


def test_debug_knob_and_stage_record_without_a_gpu(tmp_path):
    """kmers_hip_debug_knob: known names return 0, an unknown one KMERS_ERR_BAD_ARG (-12), "reset" re-reads the
    environment.  kmers_last_call_stages: a file-level call that fails for want of a device has still parsed the file, and
    says so; a buffer too small for the record gets what fits, NUL-terminated."""
    lib = _lib.handle()
    for name in (b"y_atomics", b"y_cap", b"force_wide", b"x_path", b"old_pass0", b"x_chunk_pos", b"verbose", b"wide14"):
        assert lib.kmers_hip_debug_knob(name, 1) == 0
    assert lib.kmers_hip_debug_knob(b"no_such_knob", 1) == -12
    assert lib.kmers_hip_debug_knob(b"reset", 0) == 0
    p = tmp_path / "a.fsa"
    p.write_text(">a\nACGTACGTAC\n")
    y = np.zeros(4 ** 3, np.int32)
    rc = lib.fasta_count_kmers(1, _lib.cfiles([str(p)]), 3, y.ctypes.data_as(C.POINTER(C.c_int)))
    st = _lib.last_call_stages()
    assert "host ingest" in st and st["host ingest"] >= 0.0
    if rc != 0:                         # no GPU here: nothing after the parse ran
        assert rc == -10 and list(st) == ["host ingest"] and not y.any()
    small = C.create_string_buffer(8)
    lib.kmers_last_call_stages(small, 8)
    assert len(small.value) <= 7 and b"host ingest".startswith(small.value[:7])
