"""reads simulated straight into HBM (kmers_hip_simulate_reads) against the reads FILE the
restated reference simulator writes (oracle/reads_oracle.py, reference kmerexpr/simulate_reads.py:54-122): the same
read sequences, hence the same y -- bit-exact.  Needs a real MI355X: run with -m gpu."""
import numpy as np
import pytest

from kmers_amd import synth
from oracle import reads_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def iso(tmp_path_factory):
    d = tmp_path_factory.mktemp("simg")
    p = str(d / "iso.fsa")
    synth.synth_transcriptome(p, n_records=400, total_bases=500_000, seed=8)
    with open(p, "a") as f:      # filtered records, a record with exactly one start, and non-ACGT bases inside reads
        f.write(">x PREDICTED: dropped\n" + "ACGT" * 100 + "\n>short\nACGTACGT\n>exact\n" + "ACGTTGCA" * 10 + "\n")
        f.write(">withN\n" + ("ACGTNACGTTTGACCANNNACGT" * 40) + "\n")
    return p


def decode(ds, N, L):
    pos = np.arange(ds.n_pos, dtype=np.int64)
    bases = ds.bases.cpu().numpy().view(np.uint32)
    valid = ds.valid.cpu().numpy().view(np.uint32)
    code = (bases[pos >> 4] >> (30 - 2 * (pos & 15)).astype(np.uint32)) & 3
    ok = (valid[pos >> 5] >> (pos & 31).astype(np.uint32)) & 1
    chars = np.where(ok == 1, np.frombuffer(b"ACGT", np.uint8)[code], ord("N")).astype(np.uint8).reshape(N, L + 1)
    assert (ok.reshape(N, L + 1)[:, L] == 0).all()           # separators
    assert not bases[(ds.n_pos + 15) // 16:].any() and not valid[(ds.n_pos + 31) // 32:].any()   # padding
    return [r[:L].tobytes() for r in chars]


@pytest.mark.parametrize("N,L,alpha", [(4000, 80, 0.1), (1500, 100, 1.0), (5000, 14, 0.5)])
def test_simulated_store_equals_reference_reads(oracle, iso, tmp_path, N, L, alpha):
    from kmers_amd.simulate_reads import simulate_reads_device
    reads = str(tmp_path / "reads.fsa")
    want = orc.simulate_reads(N, L, alpha, reads, iso)
    ds, info = simulate_reads_device(N, L, alpha, iso)
    assert np.array_equal(info["iso_sampled"], want["iso_sampled"])
    assert np.array_equal(info["starts"].astype(np.int64), want["starts"])
    assert np.array_equal(info["theta_true"], want["theta_true"])
    with open(reads) as f:
        texts = [line.strip().encode() for line in f if not line.startswith(">")]
    got = decode(ds, N, L)
    norm = lambda t: bytes(c if c in b"ACGT" else ord("N") for c in t)     # the store keeps validity, not the letter
    assert got == [norm(t) for t in texts]
    for K in (6, 12, 14):
        if K > L:
            continue
        rc, y = oracle.y([reads], K)
        assert rc == 0 and np.array_equal(ds.count_y(K).cpu().numpy(), y), K


def test_reads_from_a_file_the_two_reference_readers_see_differently(oracle, tmp_path):
    """CRLF / lone-CR line ends, blanks python strips, text before the first header, a NUL: the isoforms are those of
    kmerexpr/fasta.py (the simulator's reader), the reads file is then counted by the C++ reader's rules -- as in the
    reference, where simulate_reads writes the file and fasta_count_kmers reads it."""
    from kmers_amd.simulate_reads import simulate_reads_device
    from tests.test_simreads_cpu import awkward_fasta
    p = awkward_fasta(str(tmp_path / "awk.fsa"), seed=9)
    for N, L in ((3000, 100), (2000, 31)):
        reads = str(tmp_path / f"reads{L}.fsa")
        want = orc.simulate_reads(N, L, 0.3, reads, p)
        ds, info = simulate_reads_device(N, L, 0.3, p)
        assert np.array_equal(info["iso_sampled"], want["iso_sampled"])
        assert np.array_equal(info["starts"].astype(np.int64), want["starts"])
        assert np.array_equal(info["lengths"], want["lengths"])
        for K in (5, 12):
            rc, y = oracle.y([reads], K)
            assert rc == 0 and np.array_equal(ds.count_y(K).cpu().numpy(), y), (L, K)


def test_bad_isoform_or_start_is_rejected(iso):
    import torch
    from kmers_amd import _lib
    from kmers_amd.store import MODE_X, Store
    lib = _lib.handle()
    st = Store(iso, MODE_X, L=50)
    ds = st.to_device()
    lens = np.diff(st.rec_start.astype(np.int64)) - 1
    N, L = 100, 50
    n_pos = N * (L + 1)
    ob = torch.empty((n_pos + 15) // 16 + 4, dtype=torch.int32, device="cuda")
    ov = torch.empty((n_pos + 31) // 32 + 4, dtype=torch.int32, device="cuda")
    def run(iso_idx, starts, nb=None):
        a = torch.from_numpy(np.asarray(iso_idx, np.uint32).view(np.int32)).cuda()
        b = torch.from_numpy(np.asarray(starts, np.uint32).view(np.int32)).cuda()
        return lib.kmers_hip_simulate_reads(ds.bases.data_ptr(), ds.valid.data_ptr(), ds.rec_start.data_ptr(), st.n_rec,
                                            a.data_ptr(), b.data_ptr(), N, L, ob.data_ptr(), nb or ob.numel(), ov.data_ptr(),
                                            ov.numel(), None)
    good_iso = np.zeros(N, np.uint32)
    assert run(good_iso, np.zeros(N)) == 0
    bad = good_iso.copy()
    bad[7] = st.n_rec
    assert run(bad, np.zeros(N)) == -12
    s = np.zeros(N)
    s[3] = lens[0] - L + 1
    assert run(good_iso, s) == -12
    assert run(good_iso, np.zeros(N), nb=ob.numel() - 5) == -14


def test_shards_add_up(iso):
    """Reads split over ranks: the shard stores' histograms sum to the whole one."""
    from kmers_amd.simulate_reads import simulate_reads_device
    N, L, K = 3001, 60, 9
    whole, _ = simulate_reads_device(N, L, 0.3, iso)
    want = whole.count_y(K)
    got = None
    for r in range(3):
        part, info = simulate_reads_device(N, L, 0.3, iso, shard=r, n_shards=3)
        assert info["iso_sampled"].shape[0] == N
        got = part.count_y(K, counts=got)
    assert bool((got == want).all())
