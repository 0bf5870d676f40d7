#!/usr/bin/env python3
"""Generate tests/golden/ -- TEST INFRASTRUCTURE ONLY, run in the build container.

Runs the UNMODIFIED reference library (oracle/_ref/libkmers_ref.so, compiled by `make -C oracle ref` from
/root/reference/src/libkmers.cpp) through ctypes on

  * the data files the reference's own tests use (test/data/test1..5.fsa, data/test5.fsa: copied as data),
  * small edge-case FASTA / FASTQ files synthesised below,
  * seeded random transcriptome / read files,

and stores inputs + expected outputs as small fixtures:

  tests/golden/inputs/<file>             the input files
  tests/golden/manifest.json             one entry per case: name, kind, files, K, L, cat
  tests/golden/<case>.npz                X cases: rows,cols (uint32), data (float32), n_cols, nnz
                                         y cases: idx (uint32), cnt (int32) of the non-zero bins

X is stored as (row, col, value) triples in CSR order so that K=12/14 cases stay small; the test expands
the CSR arrays it gets back and compares them element for element.

Every input keeps each record at least K-1 bases long: shorter records make the reference read out of
bounds (libkmers.cpp:221-222, :321), so no reference answer exists for them.
"""
import ctypes as C
import gzip
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
INP = os.path.join(GOLD, "inputs")
REFERENCE = os.environ.get("KMERS_REFERENCE", "/root/reference")

os.environ.setdefault("KMER_NUM_THREADS", "4")


def load_ref():
    lib = C.CDLL(os.path.join(HERE, "_ref", "libkmers_ref.so"))
    xargs = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint64),
             C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    for name in ("fasta_to_kmers_csr", "fasta_to_kmers_csr_cat_subseq"):
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = xargs
    for name in ("fasta_count_kmers", "fastq_gz_count_kmers"):
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_int)]
    lib.kmer_to_id.restype = C.c_uint32
    lib.kmer_to_id.argtypes = [C.c_char_p, C.c_int]
    lib.valid_kmer.restype = C.c_bool
    lib.valid_kmer.argtypes = [C.c_char_p, C.c_int]
    return lib


def cfiles(paths):
    arr = (C.c_char_p * len(paths))()
    arr[:] = [p.encode() for p in paths]
    return arr


def ref_x(lib, paths, K, L, cat):
    M = 4 ** K
    max_nz = sum(os.stat(p).st_size for p in paths if os.path.isfile(p)) + 16
    data = np.zeros(max_nz, np.float32)
    col = np.zeros(max_nz, np.uint64)
    row = np.zeros(M + 1, np.uint64)
    nnz = C.c_uint64(0)
    ncols = C.c_int(0)
    fn = lib.fasta_to_kmers_csr_cat_subseq if cat else lib.fasta_to_kmers_csr
    rc = fn(len(paths), cfiles(paths), K, L, data.ctypes.data_as(C.POINTER(C.c_float)),
            row.ctypes.data_as(C.POINTER(C.c_uint64)), col.ctypes.data_as(C.POINTER(C.c_uint64)), max_nz,
            C.byref(nnz), C.byref(ncols))
    assert rc == 0, rc
    n = nnz.value
    cnt = np.diff(row)
    nzr = np.flatnonzero(cnt)
    rows = np.repeat(nzr.astype(np.uint32), cnt[nzr].astype(np.int64))
    assert rows.shape[0] == n
    return dict(rows=rows, cols=col[:n].astype(np.uint32), data=data[:n].copy(), n_cols=np.int64(ncols.value),
                nnz=np.int64(n))


def ref_y(lib, paths, K, fastq):
    M = 4 ** K
    y = np.zeros(M, np.int32)
    fn = lib.fastq_gz_count_kmers if fastq else lib.fasta_count_kmers
    rc = fn(len(paths), cfiles(paths), K, y.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0, rc
    idx = np.flatnonzero(y).astype(np.uint32)
    return dict(idx=idx, cnt=y[idx].copy())


def write(name, content):
    mode = "wb" if isinstance(content, bytes) else "w"
    with open(os.path.join(INP, name), mode) as f:
        f.write(content)
    return name


def wrap(seq, width):
    return "\n".join(seq[i:i + width] for i in range(0, len(seq), width))


def random_transcriptome(rng, n_rec, mean_len, min_len, width=70, p_n=0.01, p_predicted=0.02, polya=0.3):
    """Seeded, RefSeq-flavoured: shared blocks between neighbouring records, poly-A tails, N runs."""
    out = []
    alphabet = np.array(list("ACGT"))
    prev = None
    for i in range(n_rec):
        ln = int(max(min_len, rng.lognormal(np.log(mean_len), 0.8)))
        s = "".join(rng.choice(alphabet, size=ln, p=[0.26, 0.24, 0.24, 0.26]))
        if prev is not None and rng.random() < 0.5:  # share a block with the previous record
            a = rng.integers(0, max(1, len(prev) - 50))
            blk = prev[a:a + rng.integers(50, 400)]
            b = rng.integers(0, len(s))
            s = s[:b] + blk + s[b:]
        if rng.random() < polya:
            s += "A" * int(rng.integers(1, 200))
        if rng.random() < p_n:
            a = rng.integers(0, len(s))
            s = s[:a] + "N" * int(rng.integers(1, 50)) + s[a:]
        if rng.random() < 0.01:
            a = rng.integers(0, len(s) - 10)
            s = s[:a] + s[a:a + 10].lower() + s[a + 10:]
        hdr = f">NM_{i:06d}.1 " + ("PREDICTED: " if rng.random() < p_predicted else "") + f"synthetic gene {i}"
        out.append(hdr + "\n" + wrap(s, width) + "\n")
        prev = s
    return "".join(out)


def random_reads(rng, n, length, source):
    out = []
    for i in range(n):
        a = int(rng.integers(0, len(source) - length))
        out.append(f">sim-{i}/x\n{source[a:a + length]}\n")
    return "".join(out)


def main():
    if not os.path.isdir(REFERENCE):
        sys.exit("reference tree not present: golden fixtures can only be generated in the build container")
    shutil.rmtree(GOLD, ignore_errors=True)
    os.makedirs(INP)
    lib = load_ref()
    cases = []

    def x_case(name, files, K, L=0, cat=False):
        paths = [os.path.join(INP, f) for f in files]
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **ref_x(lib, paths, K, L, cat))
        cases.append(dict(name=name, kind="x", files=files, K=K, L=L, cat=cat))

    def y_case(name, files, K, fastq=False):
        paths = [os.path.join(INP, f) for f in files]
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **ref_y(lib, paths, K, fastq))
        cases.append(dict(name=name, kind="y_fastq" if fastq else "y_fasta", files=files, K=K))

    # ---- the reference's own data files (data, not source) -------------------------------------------
    for i in range(1, 6):
        shutil.copy(os.path.join(REFERENCE, "test", "data", f"test{i}.fsa"), os.path.join(INP, f"ref_test{i}.fsa"))
    shutil.copy(os.path.join(REFERENCE, "data", "test5.fsa"), os.path.join(INP, "ref_data_test5.fsa"))
    # shortest records: test1 4 bases, test2 4, test3 3 ("TTT"), test4/5 16  -> K-1 <= shortest
    for i, ks in ((1, (1, 2, 3, 4, 5)), (2, (1, 2, 3, 4, 5)), (3, (1, 2, 3, 4)), (4, (1, 2, 3, 8, 12)),
                  (5, (1, 2, 3, 8, 12))):
        for K in ks:
            x_case(f"ref_test{i}_K{K}_x", [f"ref_test{i}.fsa"], K)
            y_case(f"ref_test{i}_K{K}_y", [f"ref_test{i}.fsa"], K)
    x_case("cfg1_data_test5_K8_L14_x", ["ref_data_test5.fsa"], 8, L=14)       # BASELINE.json configs[0]
    x_case("cfg1_data_test5_K8_L0_x", ["ref_data_test5.fsa"], 8, L=0)
    x_case("cfg1_data_test5_K8_L15_x", ["ref_data_test5.fsa"], 8, L=15)       # every record filtered out
    y_case("cfg1_data_test5_K8_y", ["ref_data_test5.fsa"], 8)
    x_case("cfg1_data_test5_K14_x", ["ref_data_test5.fsa"], 14)               # one window per record
    write("edge_len13.fsa", ">a\nACGTACGTACGTA\n>b\nACGTACGTACGTAC\n>c\nTTTTTTTTTTTTTTT\n")
    x_case("edge_len13_K14_x", ["edge_len13.fsa"], 14)                        # a: len == K-1, no windows, column kept

    # ---- edge cases ---------------------------------------------------------------------------------
    write("edge_mixed.fsa", ">a\nACGTNACGT\n>b\nacgtACGTA\n>c\nNNNN\n>d\nACG\n")
    x_case("edge_mixed_K3_x", ["edge_mixed.fsa"], 3)
    y_case("edge_mixed_K3_y", ["edge_mixed.fsa"], 3)
    x_case("edge_mixed_K4_x", ["edge_mixed.fsa"], 4)          # record d has K-1 bases: zero windows, column kept
    write("edge_predicted.fsa", ">x1 PREDICTED: foo\nAAAAAA\n>x2\nACGTAC\n>x3\nACG\n>x4 xPREDICTEDx\nGGGGGG\n>x5\nTTTTTT\n")
    x_case("edge_predicted_K2_L4_x", ["edge_predicted.fsa"], 2, L=4)
    x_case("edge_predicted_K2_L0_x", ["edge_predicted.fsa"], 2, L=0)
    y_case("edge_predicted_K2_y", ["edge_predicted.fsa"], 2)
    write("edge_two_a.fsa", ">r1\nAACC\n>r2\nCCGG\n")
    write("edge_two_b.fsa", ">r3\nTTTT")
    x_case("edge_twofiles_K2_x", ["edge_two_a.fsa", "edge_two_b.fsa"], 2)
    x_case("edge_twofiles_K2_cat_x", ["edge_two_a.fsa", "edge_two_b.fsa"], 2, cat=True)
    x_case("edge_twofiles_K2_cat_L5_x", ["edge_two_a.fsa", "edge_two_b.fsa"], 2, L=5, cat=True)
    y_case("edge_twofiles_K2_y", ["edge_two_a.fsa", "edge_two_b.fsa"], 2)
    write("edge_crlf.fsa", b">a desc\r\nACGT\r\nACGT\r\n>b\r\nGGCA\r\nTTAC\r\n")
    x_case("edge_crlf_K2_x", ["edge_crlf.fsa"], 2)
    y_case("edge_crlf_K2_y", ["edge_crlf.fsa"], 2)
    write("edge_spaces.fsa", ">a\nACGT  \nACGT\n>b\nAC GT\n   \nTTGA \t\nCC\t \nGG\n>c\n \nACGT\n")
    x_case("edge_spaces_K2_x", ["edge_spaces.fsa"], 2)
    y_case("edge_spaces_K2_y", ["edge_spaces.fsa"], 2)
    write("edge_nohdr.fsa", "ACGTACGT\nGGGG\n>b\nTTTTACGT\nAC>GT\n >c\nAAAA")   # first line is a header whatever it is
    x_case("edge_nohdr_K2_x", ["edge_nohdr.fsa"], 2)
    y_case("edge_nohdr_K3_y", ["edge_nohdr.fsa"], 3)
    write("edge_blankfirst.fsa", "\n>a\nACGT\n")                                # empty first line ends the file
    x_case("edge_blankfirst_K1_x", ["edge_blankfirst.fsa"], 1)
    y_case("edge_blankfirst_K1_y", ["edge_blankfirst.fsa"], 1)
    write("edge_empty.fsa", "")
    x_case("edge_empty_K2_x", ["edge_empty.fsa"], 2)
    y_case("edge_empty_K2_y", ["edge_empty.fsa"], 2)
    write("edge_hdronly.fsa", ">a\nACGT\n>b\n>c\nGG\n>d")                      # empty records: K=1 only (len >= K-1)
    x_case("edge_hdronly_K1_x", ["edge_hdronly.fsa"], 1)
    y_case("edge_hdronly_K1_y", ["edge_hdronly.fsa"], 1)
    rng = np.random.default_rng(7)
    longline = "".join(rng.choice(list("ACGT"), size=70000))
    write("edge_longline.fsa", ">titin-like one line\n" + longline + "\n>polyA\n" + "A" * 70000 + "\n>short\nACGTACGTACGTACGT\n")
    x_case("edge_longline_K8_x", ["edge_longline.fsa"], 8)                      # count 69993 > 65535 in one column
    y_case("edge_longline_K8_y", ["edge_longline.fsa"], 8)
    write("edge_highbytes.fsa", b">a\nACGT\x00ACGT\x7fAC\x01GT\n>b\nAC*GT-ACGU\n")
    x_case("edge_highbytes_K2_x", ["edge_highbytes.fsa"], 2)
    y_case("edge_highbytes_K2_y", ["edge_highbytes.fsa"], 2)

    # ---- seeded synthetic transcriptomes ---------------------------------------------------------------
    rng = np.random.default_rng(20240521)
    write("syn_small.fsa", random_transcriptome(rng, 300, 700, 40))
    for K in (8, 12, 14):
        x_case(f"syn_small_K{K}_x", ["syn_small.fsa"], K)
    x_case("syn_small_K8_L500_x", ["syn_small.fsa"], 8, L=500)
    x_case("syn_small_K12_cat_x", ["syn_small.fsa", "ref_test4.fsa"], 12, cat=True)
    for K in (8, 12):
        y_case(f"syn_small_K{K}_y", ["syn_small.fsa"], K)
    src = "".join(rng.choice(list("ACGT"), size=20000))
    write("syn_reads.fsa", random_reads(rng, 4000, 100, src))
    for K in (8, 12):
        y_case(f"syn_reads_K{K}_y", ["syn_reads.fsa"], K)

    # ---- FASTQ ----------------------------------------------------------------------------------------
    fq = "".join(f"@r{i}\n{src[37 * i:37 * i + 50]}\n+\n{'I' * 50}\n" for i in range(200))
    write("reads_plain.fastq", fq)
    with gzip.open(os.path.join(INP, "reads.fastq.gz"), "wb") as f:
        f.write(fq.encode())
    write("reads_tail.fastq", fq + "@tail\nACGTACGT")       # unterminated sequence line is dropped
    write("reads_crlf.fastq", b"@r\r\nACGTAC\r\n+\r\nIIIIII\r\n@s\r\nGGTTAA\r\n+\r\nIIIIII\r\n")
    for K in (3, 8):
        y_case(f"fastq_plain_K{K}_y", ["reads_plain.fastq"], K, fastq=True)
        y_case(f"fastq_gz_K{K}_y", ["reads.fastq.gz"], K, fastq=True)
        y_case(f"fastq_two_K{K}_y", ["reads.fastq.gz", "reads_plain.fastq"], K, fastq=True)
    y_case("fastq_tail_K3_y", ["reads_tail.fastq"], 3, fastq=True)
    y_case("fastq_crlf_K3_y", ["reads_crlf.fastq"], 3, fastq=True)
    write("reads_longline.fastq", "@r\n" + "ACGT" * 2048 + "\n+\n" + "I" * 8192 + "\n")  # 8192-byte line: file dropped
    y_case("fastq_longline_K1_y", ["reads_longline.fastq"], 1, fastq=True)
    write("reads_8191.fastq", "@r\n" + ("ACGT" * 2048)[:8191] + "\n+\n" + "I" * 8191 + "\n")
    y_case("fastq_8191_K3_y", ["reads_8191.fastq"], 3, fastq=True)

    # ---- scalar helpers ---------------------------------------------------------------------------------
    words = ["A", "C", "G", "T", "AA", "AC", "TT", "ACGT", "TTTTTTTTTTTTTTTT", "GATTACA", "ACGTN", "acgt", "AC GT",
             "NNNN", "ANA", ""]
    scal = {w: [int(lib.kmer_to_id(w.encode(), len(w))), bool(lib.valid_kmer(w.encode(), len(w)))] for w in words}
    with open(os.path.join(GOLD, "scalars.json"), "w") as f:
        json.dump(scal, f, indent=1)

    with open(os.path.join(GOLD, "manifest.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print(f"wrote {len(cases)} cases to {GOLD}")


if __name__ == "__main__":
    main()
