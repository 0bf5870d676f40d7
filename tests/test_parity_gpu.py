"""Parity of the HIP path (through the C ABI of kmers_amd/lib/libkmers.so) with the reference.

  * every golden fixture produced by the unmodified reference library (tests/golden/) -- bit-exact;
  * seeded random transcriptomes / reads against the oracle -- bit-exact (int32 counts, uint64 indices,
    float32 values compared as bit patterns);
  * the edge cases the reference has no answer for (records shorter than K-1) against the oracle's definition;
  * error codes.
Needs a real MI355X: run with -m gpu."""
import os

import numpy as np
import pytest

from kmers_amd import synth
from tests import kmtest

pytestmark = pytest.mark.gpu
CASES = kmtest.manifest()


@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] == "x"], ids=lambda c: c["name"])
def test_x_matches_reference(product, case):
    out = product.x(kmtest.case_paths(case), case["K"], case["L"], case["cat"])
    kmtest.assert_x_equal_golden(case, *out)


@pytest.mark.parametrize("case", [c for c in CASES if c["kind"] != "x"], ids=lambda c: c["name"])
def test_y_matches_reference(product, case):
    rc, y = product.y(kmtest.case_paths(case), case["K"], fastq=(case["kind"] == "y_fastq"))
    kmtest.assert_y_equal_golden(case, rc, y)


@pytest.fixture(scope="module")
def syn(tmp_path_factory):
    d = tmp_path_factory.mktemp("syn")
    iso = str(d / "iso.fsa")
    info = synth.synth_transcriptome(iso, n_records=3000, total_bases=3_000_000, seed=11, return_sequences=True)
    reads = str(d / "reads.fsa")
    synth.synth_reads(reads, info["concat"], info["offsets"], 30000, read_len=100, seed=5)
    uni = str(d / "uni.fsa")
    synth.synth_transcriptome(uni, n_records=500, total_bases=1_000_000, seed=3, uniform=True)
    return dict(iso=iso, reads=reads, uni=uni)


@pytest.mark.parametrize("K", [1, 2, 3, 4, 5, 7, 8, 9, 11, 12, 13])
def test_x_random_vs_oracle(product, oracle, syn, K):
    a = product.x([syn["iso"]], K)
    b = oracle.x([syn["iso"]], K)
    assert a[1].shape[0] > 0
    kmtest.assert_x_equal(a, b)


@pytest.mark.parametrize("K", [2, 4, 5, 8])
def test_x_wide_entries_for_small_k(product, oracle, syn, K, knob):
    """K <= 8 normally runs on 4-byte entries; the force_wide knob selects the 8-byte form (used for K >= 9 or more
    than 2^24 records), which must give the same matrix."""
    knob("force_wide")
    kmtest.assert_x_equal(product.x([syn["iso"], syn["uni"]], K), oracle.x([syn["iso"], syn["uni"]], K))


@pytest.mark.parametrize("K", [9, 12, 16])
def test_x_first_and_last_row_in_one_round(product, oracle, tmp_path, K):
    """A handful of distinct k-mers that include the smallest and the largest id (poly-A, poly-T): the row heads of ONE
    64-entry round of k_rle span all 4^K rows (2^32 at K=16: the row count of the round does not fit 32 bits), and
    long gaps lie between them."""
    p = tmp_path / "span.fsa"
    rng = np.random.default_rng(K)
    rnd = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)).decode()
    p.write_text(">a\n" + "A" * (K + 1) + "C\n>b\n" + "T" * (K + 2) + "\n>c\n" + rnd(K + 9) + "\n>d\n" + "G" * K + "\n>e\n"
                 + "A" * K + "\n")
    kmtest.assert_x_equal(product.x([str(p)], K), oracle.x([str(p)], K))


@pytest.mark.parametrize("K,L,cat", [(8, 0, False), (13, 900, False), (6, 0, True), (11, 2000, True)])
def test_x_in_record_ranges_above_the_per_call_limit(product, oracle, syn, K, L, cat, knob):
    """An input of more positions than one device call indexes (2^32 - 65536) is built in contiguous record ranges that
    are interleaved row by row on the host.  The x_chunk_pos knob lowers the limit so that the route runs here: 4 M
    positions in ranges of at most 300,000 (the count leaves room for the longest record: kmers_x_range_count); in the
    cat mode one file is one column, 1 + 3 + 1 M positions under a limit of 3.3 M (17 ranges, 14 of them empty)."""
    files = [syn["uni"], syn["iso"], syn["uni"]] if cat else [syn["iso"], syn["uni"]]
    want = oracle.x(files, K, L, cat)
    knob("x_chunk_pos", 3300000 if cat else 300000)
    kmtest.assert_x_equal(product.x(files, K, L, cat), want)
    nnz = want[1].shape[0]
    assert product.x(files, K, L, cat, max_nz=nnz)[0] == 0
    assert product.x(files, K, L, cat, max_nz=nnz - 1)[0] == -2


@pytest.mark.parametrize("K,L,cat", [(8, 1500, False), (12, 0, True), (6, 2000, True), (14, 0, False)])
def test_x_variants_vs_oracle(product, oracle, syn, K, L, cat):
    files = [syn["iso"], syn["uni"]]
    kmtest.assert_x_equal(product.x(files, K, L, cat), oracle.x(files, K, L, cat))


@pytest.mark.parametrize("K", [1, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14])
def test_y_random_vs_oracle(product, oracle, syn, K):
    rc, y = product.y([syn["reads"], syn["uni"]], K)
    rc2, y2 = oracle.y([syn["reads"], syn["uni"]], K)
    assert rc == rc2 == 0
    assert int(y.sum()) > 0
    assert np.array_equal(y, y2)


@pytest.mark.parametrize("K", [9, 12, 13, 14])
def test_y_small_workspace_chunks(oracle, syn, K):
    """kmers_hip_count_y_ws with a scratch buffer far smaller than recommended: the store is counted in several
    chunks of tiles (windows that span a chunk boundary included) and the histogram is the same."""
    import torch
    from kmers_amd.store import MODE_Y_FASTA, Store
    rc, want = oracle.y([syn["reads"]], K)
    assert rc == 0
    st = Store(syn["reads"], MODE_Y_FASTA)
    ds = st.to_device()
    full = ds._lib.kmers_hip_y_workspace_bytes(st.n_pos, K)
    assert st.n_pos > 20 * 8192
    for nbytes in (full, full // 7, 800_000):                # one chunk, ~7 chunks, chunks of a few tiles
        ws = ds.y_workspace(K, nbytes)
        got = ds.count_y(K, workspace=ws)
        torch.cuda.synchronize()
        assert np.array_equal(got.cpu().numpy(), want), nbytes
    with pytest.raises(RuntimeError):
        ds.count_y(K, workspace=ds.y_workspace(K, 4096))    # not even one tile


@pytest.mark.parametrize("K", [9, 12, 14, 16])
def test_y_hot_bucket_overflow_queue(oracle, tmp_path, K, knob):
    """A slice with more entries than the per-workgroup cap hands the rest to the overflow kernel (several chunks,
    atomics into bins the slice's own workgroup also writes).  Poly-A reads put ~4e5 windows into one bin; the cap is
    forced down so that ordinary slices overflow too."""
    import torch
    from kmers_amd.store import MODE_Y_FASTA, Store
    rng = np.random.default_rng(K)
    p = tmp_path / "hot.fsa"
    with open(p, "wb") as f:
        for i in range(6000):
            seq = b"A" * 100 if i % 4 else bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 100))
            f.write(b">r%d\n%s\n" % (i, seq))
    st = Store(str(p), MODE_Y_FASTA)
    ds = st.to_device()
    if K <= 14:
        rc, want = oracle.y([str(p)], K)
        assert rc == 0
        want = torch.from_numpy(want)
    else:
        knob("y_atomics", 1)
        want = ds.count_y(K).cpu()
        knob("y_atomics", 0)
    for cap in (1000, 7, 0):           # 0: the library's own cap
        knob("y_cap", cap)
        got = ds.count_y(K)
        torch.cuda.synchronize()
        assert torch.equal(got.cpu(), want), cap
    assert int(want[0]) >= 4500 * (100 - K + 1)


@pytest.mark.parametrize("K", [15, 16])
def test_y_largest_k_partition_equals_atomics(syn, K, knob):
    """K = 15, 16 (4 / 16 GiB of bins: no oracle run at that size): the partition path against the plain
    global-atomics kernel on the GPU, plus the total against the host-side window count."""
    import torch
    from kmers_amd.store import MODE_Y_FASTA, Store
    st = Store(syn["reads"], MODE_Y_FASTA)
    ds = st.to_device()
    a = ds.count_y(K)
    knob("y_atomics")
    b = ds.count_y(K)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    total = int(a.sum(dtype=torch.int64).item())
    assert 0 < total <= st.n_windows(K)
    if K == 16:
        return
    # every bin index decodes to a k-mer that occurs: spot-check the heaviest bin against a host count at K=15
    top = int(torch.argmax(a).item())
    kmer = "".join("ACGT"[(top >> (2 * (K - 1 - i))) & 3] for i in range(K)).encode()
    n = sum(sum(1 for i in range(len(seq) - K + 1) if seq[i:i + K] == kmer) for seq in st.sequences())
    assert n == int(a[top].item())


def test_y_accumulates_into_callers_buffer(product, syn):
    """The reference only increments the caller's array (libkmers.cpp:213-216)."""
    rc, y = product.y([syn["reads"]], 6)
    pre = np.full(4 ** 6, 7, np.int32)
    rc2, y2 = product.y([syn["reads"]], 6, into=pre)
    assert rc == rc2 == 0 and np.array_equal(y2, y + 7)


def test_short_and_empty_records(product, oracle, tmp_path):
    """Records shorter than K-1: the reference reads out of bounds; defined here (and in the oracle) as zero
    k-mers, column kept."""
    p = tmp_path / "short.fsa"
    p.write_text(">a\nAC\n>b\n\n>c\nACGTACGTNNACGTACGTAC\n>d\nA\n>e\n>f\nACGTA\n")
    for K in (1, 2, 5, 6, 9):
        kmtest.assert_x_equal(product.x([str(p)], K), oracle.x([str(p)], K))
        assert np.array_equal(product.y([str(p)], K)[1], oracle.y([str(p)], K)[1])
    rc, data, row, col, n_cols = product.x([str(p)], 5)
    assert rc == 0 and n_cols == 6 and set(col.tolist()) == {2, 5}


def test_no_files_and_empty_inputs(product, tmp_path):
    rc, data, row, col, n_cols = product.x([], 3)
    assert rc == 0 and n_cols == 0 and data.shape[0] == 0 and not row.any()
    e = tmp_path / "e.fsa"
    e.write_text("")
    rc, data, row, col, n_cols = product.x([str(e)], 3)
    assert rc == 0 and n_cols == 0 and not row.any()
    n = tmp_path / "n.fsa"
    n.write_text(">a\nNNNNNNNN\n>b\nNNNN\n")
    rc, data, row, col, n_cols = product.x([str(n)], 3)
    assert rc == 0 and n_cols == 2 and data.shape[0] == 0 and not row.any()
    rc, y = product.y([str(n)], 3)
    assert rc == 0 and not y.any()


def test_error_codes(product, tmp_path):
    missing = str(tmp_path / "nope.fsa")
    assert product.x([missing], 2)[0] == -1
    assert product.y([missing], 2)[0] == -1
    four = os.path.join(kmtest.INP, "ref_test4.fsa")
    assert product.x([four], 2, max_nz=3)[0] == -2           # 31 non-zeros do not fit
    assert product.x([four], 2, max_nz=31)[0] == 0           # exactly enough
    assert product.x([four], 0)[0] == -12 and product.x([four], 17)[0] == -12
    rc, data, row, col, n_cols = product.x([missing, four], 2, cat=True)
    assert rc == 0 and n_cols == 2 and set(col.tolist()) == {1}


def test_row_pointer_gap_fill_paths(product, oracle, tmp_path):
    """Very sparse rows at large K exercise the queued (long-gap) row-pointer fill and its chunk splitting."""
    p = tmp_path / "sparse.fsa"
    p.write_text(">a\nAAAAAAAAAAAAAC\n>b\nTTTTTTTTTTTTTT\n>c\nCATGCATGCATGCATGCA\n")
    for K in (11, 12, 14):
        kmtest.assert_x_equal(product.x([str(p)], K), oracle.x([str(p)], K))


def test_one_giant_run(product, oracle, tmp_path):
    """A single (k-mer, column) run that spans hundreds of tiles: the run-length reduce must find its end by search,
    not by walking (a poly-A record is the worst case for the CSR emit)."""
    p = tmp_path / "polya.fsa"
    p.write_text(">a\n" + "A" * 3_000_000 + "\n>b\nACGTACGTACGT\n" + ">c\n" + "A" * 100_000 + "C\n")
    for K in (8, 12):
        kmtest.assert_x_equal(product.x([str(p)], K), oracle.x([str(p)], K))
        assert np.array_equal(product.y([str(p)], K)[1], oracle.y([str(p)], K)[1])


@pytest.mark.parametrize("K", [2, 5, 8])
def test_x_fused_emit_long_column_groups(product, oracle, tmp_path, K):
    """K <= 8: the last partition pass writes the CSR arrays itself and closes a (k-mer, column) group that straddles a
    tile by looking back / ahead over the column's entries (kernels.hip, "fused emit").  Records whose groups are
    hundreds to tens of thousands of entries long -- repeats, poly-A, one long random record -- between short ones, so
    that groups straddle tiles and wave chunks at every offset; all still below the length at which the library falls
    back to the general path (that one is test_one_giant_run)."""
    rng = np.random.default_rng(77 + K)
    rnd = lambda n: "".join(rng.choice(list("ACGT"), n))
    recs = []
    for i in range(40):
        recs.append(rnd(int(rng.integers(1, 400))))
    recs += ["A" * 30_000, "ACGT" * 8_000, rnd(17), "AC" * 20_000 + rnd(3000), rnd(60_000), "T" * 9000 + "G" * 9000,
             rnd(5), ("ACGTTGCA" * 3 + "N") * 2000, rnd(45_000) + "A" * 5000, rnd(300)]
    for i in range(300):
        recs.append(rnd(int(rng.integers(1, 900))))
    p = tmp_path / "groups.fsa"
    p.write_text("".join(f">r{i}\n{q}\n" for i, q in enumerate(recs)))
    kmtest.assert_x_equal(product.x([str(p)], K), oracle.x([str(p)], K))


def test_x_fused_emit_equals_the_run_length_path(product, oracle, syn, knob):
    """K <= 8 has two ways from the partitioned instances to the CSR arrays: the fused emit of the last pass (default) and
    k_scatter<entries> + k_rle (knob x_path = 2, also the fallback for records too long for the fused emit); pass 0 has
    two kernels as well (old_pass0).  All of them against the oracle on the same input."""
    files = [syn["iso"], syn["uni"]]
    for K in (3, 6, 8):
        want = oracle.x(files, K)
        for x_path, old0 in ((0, 0), (2, 0), (2, 1), (0, 1)):
            knob("x_path", x_path)
            knob("old_pass0", old0)
            kmtest.assert_x_equal(product.x(files, K), want)


def test_x14_in_three_passes_equals_four(product, oracle, syn, knob):
    """K = 14 has a second pass plan (10 + 9 + 9 bits: 1024-bin pass 0 with the compact staging, two 512-bin passes; knob
    wide14), measured no faster than four 8-bit passes and therefore not the default: same matrix either way."""
    files = [syn["iso"], syn["uni"]]
    want = oracle.x(files, 14)
    kmtest.assert_x_equal(product.x(files, 14), want)
    knob("wide14")
    kmtest.assert_x_equal(product.x(files, 14), want)
    knob("x_path", 2)          # rounds 1-2: 8-bit digits, position-ordered kernels
    kmtest.assert_x_equal(product.x(files, 14), want)
    kmtest.assert_x_equal(product.x(files, 13), oracle.x(files, 13))
    kmtest.assert_x_equal(product.x(files, 9), oracle.x(files, 9))


@pytest.mark.parametrize("K", [9, 10, 12, 13, 14])
def test_y_several_hot_kmers_in_one_slice(oracle, tmp_path, K):
    """k_count_y_sorted keeps ONE hot entry value per lane in a register (the poly-A case) and changes it when a load
    with an equal pair no longer holds it: reads that mix poly-A, AAAA...C, dinucleotide repeats (two alternating
    k-mers), poly-T and random sequence put several hot k-mers into the same and into different slices, in runs of every
    length, so the hand-over between hot values and the final flush are exercised."""
    rng = np.random.default_rng(1000 + K)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    p = tmp_path / "hots.fsa"
    kinds = [lambda n: b"A" * n, lambda n: (b"A" * (n - 1)) + b"C", lambda n: (b"AC" * n)[:n], lambda n: b"T" * n,
             lambda n: (b"AAAC" * n)[:n], lambda n: bytes(rng.choice(acgt, n)), lambda n: (b"A" * (n // 2)) + bytes(rng.choice(acgt, n - n // 2))]
    with open(p, "wb") as f:
        for i in range(30000):
            n = int(rng.integers(K, 160))
            f.write(b">r%d\n%s\n" % (i, kinds[int(rng.integers(0, len(kinds)))](n)))
    rc, want = oracle.y([str(p)], K)
    from kmers_amd.store import MODE_Y_FASTA, Store
    got = Store(str(p), MODE_Y_FASTA).to_device().count_y(K).cpu().numpy()
    assert rc == 0 and int(want[0]) > 100000 and np.array_equal(got, want)
