"""`simulate_reads` of the reference (kmerexpr/simulate_reads.py:54-122) with the reads written straight into a packed
store in HBM instead of a FASTA file.

Every random choice is the reference's, bit for bit: the same legacy numpy calls in the same order for psi
(np.random.seed(42), np.random.dirichlet, :55,:96) and the isoform of every read (np.random.choice(T, N, p), :105), and
the same MT19937 stream continued for the per-read starts (np.random.choice(len - L + 1), :116, replayed in C by
kmers_sim_starts instead of N Python-level calls).  Only the copying of the L bases of each read runs on the GPU
(kmers_hip_simulate_reads).  The isoform list is the reference's: the records kmerexpr/fasta.py:19-46 yields (python
text-mode lines, strip() on both sides, nothing before the first '>' line), filtered as simulate_reads.py:86-90 does
(no "PREDICTED" in the header, at least L characters) -- a Store in MODE_FASTA_PY with the same L.  On a well-formed
FASTA ('\n' line ends, no blanks, first byte '>') that is also the column list of fasta_to_kmers_csr (MODE_X); on
anything else the two readers of the reference differ, and each of them is followed where the reference uses it
(tests/test_simreads_cpu.py, tests/test_simreads_gpu.py: CRLF, inner blanks, text before the first header).
Not restated: a read whose first character is '>' (possible only when a sequence line holds one) starts a new record
when the reference's reads file is parsed back; here it stays part of the read."""
import ctypes as C

import numpy as np

from . import _lib
from .store import MODE_FASTA_PY, DeviceStore, Store


def length_adjustment(psi, lengths):
    """reference simulate_reads.py:42-45"""
    theta = psi * lengths
    return theta / theta.sum()


def length_adjustment_inverse(theta, lengths):
    """reference simulate_reads.py:48-51"""
    psi = theta / lengths
    return psi / psi.sum()


def _choice_with_p(T, N, p):
    """np.random.choice(T, size=N, replace=True, p=p) of numpy's LEGACY generator, whose algorithm is frozen
    (numpy/random/mtrand.pyx, RandomState.choice): cdf = p.cumsum(); cdf /= cdf[-1]; u = random_sample(N);
    idx = cdf.searchsorted(u, side='right').  The same calls on the same stream -- only the N binary searches, nine
    tenths of its time, run on all host threads (searchsorted releases the GIL).  The first few thousand results are
    checked against numpy's own choice() on a copy of the generator state, so a numpy that ever did it differently would
    be noticed here and not in y."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    p = np.ascontiguousarray(p, dtype=np.float64)
    n_check = min(N, 4096)
    state = np.random.get_state()
    check = np.random.choice(T, size=n_check, replace=True, p=p)
    np.random.set_state(state)
    cdf = p.cumsum()
    cdf /= cdf[-1]
    u = np.random.random_sample(N)
    idx = np.empty(N, np.int64)
    n_thr = max(1, min(int(os.environ.get("KMER_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1), 64))
    step = max(1 << 16, -(-N // n_thr))
    with ThreadPoolExecutor(n_thr) as ex:
        list(ex.map(lambda lo: idx.__setitem__(slice(lo, lo + step), cdf.searchsorted(u[lo:lo + step], side="right")),
                    range(0, N, step)))
    if not np.array_equal(idx[:n_check], check):
        raise RuntimeError("numpy's legacy choice() no longer matches its restatement here: refusing to guess")
    return idx


def sample_reads(rec_lengths, N, L, alpha):
    """The reference's random choices for isoforms of the given lengths (all >= L): returns
    dict(psi, theta_true, theta_sampled, lengths, iso_sampled int64[N], starts uint32[N])."""
    lib = _lib.handle()
    np.random.seed(42)                                        # :55
    T = len(rec_lengths)
    lengths = np.asarray(rec_lengths, dtype=np.int64) - L + 1    # :91,:98
    assert T > 0 and lengths.min() >= 1
    psi = np.random.dirichlet(alpha * np.ones(T))             # :95-96
    theta_true = length_adjustment(psi, lengths)              # :101
    iso_sampled = _choice_with_p(T, N, theta_true)            # :105 np.random.choice(T, size=N, replace=True, p=theta_true)
    theta_sampled = np.bincount(iso_sampled, minlength=T) / N                # :107-108
    # :116, N times: start = np.random.choice(len(seq) - L + 1), replayed over raw 32-bit outputs of the same stream
    n_choices = np.ascontiguousarray(lengths[iso_sampled], dtype=np.uint32)
    starts = np.empty(N, np.uint32)
    done, left = 0, np.empty(0, np.uint32)
    while done < N:
        fresh = np.random.randint(0, 2 ** 32, size=2 * (N - done) + 1024, dtype=np.uint32)
        raw = np.concatenate([left, fresh]) if left.size else fresh
        rd, ru = C.c_uint64(0), C.c_uint64(0)
        rc = lib.kmers_sim_starts(raw.ctypes.data, raw.size, n_choices[done:].ctypes.data, N - done,
                                  starts[done:].ctypes.data, C.byref(rd), C.byref(ru))
        _lib.check(rc, "kmers_sim_starts")
        done += int(rd.value)
        left = raw[int(ru.value):]
    return dict(psi=psi, theta_true=theta_true, theta_sampled=theta_sampled, lengths=lengths, iso_sampled=iso_sampled,
                starts=starts)


def simulate_reads_device(N, L, alpha, ISO_FILE, device="cuda", shard=0, n_shards=1):
    """The reads of simulate_reads(problem(N, L, alpha), ..., ISO_FILE) as a DeviceStore (count_y-ready), plus the
    dict of sample_reads (what the reference saves with save_simulation_parameters / save_lengths, :109-110).
    Multi-GPU: every rank makes all N choices (the random stream is serial by definition) and materialises only reads
    [shard N / n_shards, (shard + 1) N / n_shards); the histograms are then summed with one all-reduce."""
    import torch
    lib = _lib.handle()
    st = Store(ISO_FILE, MODE_FASTA_PY, L=L)
    rs = st.rec_start.astype(np.int64)
    info = sample_reads(np.diff(rs) - 1, N, L, alpha)
    ds = st.to_device(device)
    dev = ds.device
    lo, hi = shard * N // n_shards, (shard + 1) * N // n_shards
    iso = torch.from_numpy(info["iso_sampled"][lo:hi].astype(np.uint32).view(np.int32)).to(dev)
    starts = torch.from_numpy(info["starts"][lo:hi].copy().view(np.int32)).to(dev)
    N = hi - lo
    n_pos = N * (L + 1)
    out_bases = torch.empty((n_pos + 15) // 16 + 4, dtype=torch.int32, device=dev)
    out_valid = torch.empty((n_pos + 31) // 32 + 4, dtype=torch.int32, device=dev)
    rc = lib.kmers_hip_simulate_reads(ds.bases.data_ptr(), ds.valid.data_ptr(), ds.rec_start.data_ptr(), st.n_rec,
                                      iso.data_ptr(), starts.data_ptr(), N, L, out_bases.data_ptr(), out_bases.numel(),
                                      out_valid.data_ptr(), out_valid.numel(), None)
    _lib.check(rc, "kmers_hip_simulate_reads")
    return DeviceStore.from_tensors(out_bases, out_valid, n_pos, N), info
