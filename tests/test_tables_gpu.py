"""The small-K X path (5 <= K <= 8: per-isoform LDS tables, kmers_amd/csrc/kernels.hip "X for 5 <= K <= 8") against the
reference's golden fixtures and the oracle, bit for bit, through the drop-in C ABI.

The x_path knob (KMERS_X_PATH=tables in the environment) selects the path (every test here sets it); without it the library takes the general partition
path, and the two must agree with the oracle and hence with each other."""
import numpy as np
import pytest

from kmers_amd import synth
from tests import kmtest
from tests.test_fuzz_gpu import _file

pytestmark = pytest.mark.gpu
CASES = [c for c in kmtest.manifest() if c["kind"] == "x" and 5 <= c["K"] <= 8]


@pytest.fixture(autouse=True)
def _tables(knob):
    knob("x_path", 1)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["name"])
def test_golden_fixtures(product, case):
    out = product.x(kmtest.case_paths(case), case["K"], case["L"], case["cat"])
    kmtest.assert_x_equal_golden(case, *out)


@pytest.fixture(scope="module")
def syn(tmp_path_factory):
    d = tmp_path_factory.mktemp("syn_tables")
    iso = str(d / "iso.fsa")
    info = synth.synth_transcriptome(iso, n_records=3000, total_bases=3_000_000, seed=11, return_sequences=True)
    reads = str(d / "reads.fsa")
    synth.synth_reads(reads, info["concat"], info["offsets"], 30000, read_len=100, seed=5)
    uni = str(d / "uni.fsa")
    synth.synth_transcriptome(uni, n_records=500, total_bases=1_000_000, seed=3, uniform=True)
    big = str(d / "big.fsa")                     # ~3100 tiles: every workgroup gets a range of its own
    synth.synth_transcriptome(big, n_records=8000, total_bases=19_000_000, seed=77)
    return dict(iso=iso, reads=reads, uni=uni, big=big)


@pytest.mark.parametrize("K", [5, 6, 7, 8])
def test_random_vs_oracle(product, oracle, syn, K):
    a = product.x([syn["iso"], syn["uni"]], K)
    assert a[1].shape[0] > 0
    kmtest.assert_x_equal(a, oracle.x([syn["iso"], syn["uni"]], K))


@pytest.mark.parametrize("K,L,cat", [(8, 1500, False), (6, 2000, True), (8, 0, True), (7, 300, False)])
def test_variants_vs_oracle(product, oracle, syn, K, L, cat):
    files = [syn["iso"], syn["uni"]]
    kmtest.assert_x_equal(product.x(files, K, L, cat), oracle.x(files, K, L, cat))


def test_every_workgroup_busy(product, oracle, syn):
    kmtest.assert_x_equal(product.x([syn["big"]], 8), oracle.x([syn["big"]], 8))


def test_reads_as_records(product, oracle, syn):
    """30,000 records of 100 bases: ~60 records per tile, handled one after the other."""
    for K in (5, 8):
        kmtest.assert_x_equal(product.x([syn["reads"]], K), oracle.x([syn["reads"]], K))


def test_both_paths_on_the_same_input(product, oracle, syn, knob):
    want = oracle.x([syn["big"]], 7)
    kmtest.assert_x_equal(product.x([syn["big"]], 7), want)       # the tables (set by the fixture)
    knob("x_path", 0)
    kmtest.assert_x_equal(product.x([syn["big"]], 7), want)       # the default: the partition path


def test_short_empty_and_invalid_records(product, oracle, tmp_path):
    p = tmp_path / "short.fsa"
    p.write_text(">a\nAC\n>b\n\n>c\nACGTACGTNNACGTACGTAC\n>d\nA\n>e\n>f\nACGTA\n>g\nNNNNNNNNNNNN\n>h\nACGTACG\n>i\nACGTACGT\n")
    for K in (5, 6, 7, 8):
        kmtest.assert_x_equal(product.x([str(p)], K), oracle.x([str(p)], K))
    e = tmp_path / "n.fsa"
    e.write_text(">a\nNNNNNNNN\n>b\nNNNN\n")
    rc, data, row, col, n_cols = product.x([str(e)], 6)
    assert rc == 0 and n_cols == 2 and data.shape[0] == 0 and not row.any()


def test_counts_beyond_the_entry_field_and_long_isoforms(product, oracle, tmp_path):
    """Poly-A records: counts far above the entry's count field (patched by k_fix_escapes), isoforms of many tiles whose
    k-mers are claimed in the tile of their first occurrence, a repeat with period 5, and neighbours of every kind."""
    p = tmp_path / "polya.fsa"
    rng = np.random.default_rng(5)
    rnd = lambda n: bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), n)).decode()
    p.write_text(">a\n" + "A" * 30_000 + "\n>b\nACGTACGTACGT\n>c\n" + "A" * 9_000 + "C" + rnd(20_000) + "A" * 300 + "\n>d\n"
                 + "ACGTT" * 7_000 + "\n>e\n" + rnd(50_000) + "\n>f\n" + rnd(700) + "\n")
    for K in (5, 8):
        kmtest.assert_x_equal(product.x([str(p)], K), oracle.x([str(p)], K))


def test_a_count_that_wraps_the_table_falls_back(product, oracle, tmp_path, capfd, knob):
    """70,000 windows of one k-mer in one isoform do not fit a 16-bit field: the library notices and rebuilds X on the
    general path; the result is the reference's either way."""
    p = tmp_path / "wrap.fsa"
    p.write_text(">a\nACGTACGTAC\n>b\n" + "A" * 70_007 + "CGT\n>c\nTTTTTTTTTTGA\n")
    knob("verbose", 1)
    kmtest.assert_x_equal(product.x([str(p)], 8), oracle.x([str(p)], 8))
    assert "rebuilt on the general path" in capfd.readouterr().err
    p.write_text(">a\nACGTACGTAC\n>b\n" + "A" * 65_542 + "CGT\n>c\nTTTTTTTTTTGA\n")     # 65,535 windows: the largest that fits
    kmtest.assert_x_equal(product.x([str(p)], 8), oracle.x([str(p)], 8))
    assert "rebuilt on the general path" not in capfd.readouterr().err


def test_max_nz(product, syn):
    rc, data, row, col, n_cols = product.x([syn["uni"]], 8)
    nnz = data.shape[0]
    assert rc == 0 and product.x([syn["uni"]], 8, max_nz=nnz)[0] == 0
    assert product.x([syn["uni"]], 8, max_nz=nnz - 1)[0] == -2


@pytest.mark.parametrize("seed", range(40))
def test_fuzz(product, oracle, tmp_path, seed):
    rng = np.random.default_rng(7000 + seed)
    files = []
    for i in range(int(rng.integers(1, 4))):
        p = str(tmp_path / f"f{i}.fsa")
        _file(rng, p)
        files.append(p)
    K = int(rng.integers(5, 9))
    L = int(rng.choice([0, 0, 1, K - 1, K, K + 3, 60]))
    cat = bool(rng.random() < 0.3)
    kmtest.assert_x_equal(product.x(files, K, L, cat), oracle.x(files, K, L, cat))
