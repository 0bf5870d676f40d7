"""Seeded synthetic inputs of the shapes BASELINE.json names.  numpy only.

  synth_transcriptome  GRCh38-RefSeq-shaped FASTA: log-normal lengths (median 1800, sigma 0.8, clipped to
                       [200, 110000]) rescaled to a target number of bases; "genes" whose isoforms are cuts of one
                       gene sequence (so isoforms of a gene share most of their k-mers); 26/24/24/26 % A/C/G/T;
                       poly-A tails on 30 % of the records; N runs in 0.05 %; "PREDICTED" in 0.5 % of the headers;
                       70-column lines.  Default seed 20240521.
  synth_reads          reads drawn the way the reference's simulate_reads does (kmerexpr/simulate_reads.py:95-116:
                       psi ~ Dirichlet(alpha), theta ~ psi * (len - L + 1), uniform start), written as two-line
                       FASTA records.  Default seed 42.

The generator is not part of the parity claim: parity is on what the library computes from a file, and every
test compares against the oracle run on the same file.
"""
import os

import numpy as np

_ACGT = np.frombuffer(b"ACGT", np.uint8)


def _lengths(rng, n_records, total_bases, min_len, max_len):
    ln = np.clip(np.rint(rng.lognormal(np.log(1800.0), 0.8, n_records)), 200, 110000)
    ln = np.clip(np.rint(ln * (total_bases / ln.sum())), min_len, max_len).astype(np.int64)
    return ln


def _wrap(seq, width):
    n = seq.size
    if n == 0:
        return b"\n"
    nl = -(-n // width)
    m = np.full((nl, width + 1), 10, np.uint8)
    pad = nl * width - n
    if pad:
        m[:-1, :width] = seq[:(nl - 1) * width].reshape(nl - 1, width)
        m[-1, :width - pad] = seq[(nl - 1) * width:]
        return m.ravel()[:(nl - 1) * (width + 1) + (width - pad)].tobytes() + b"\n"
    m[:, :width] = seq.reshape(nl, width)
    return m.tobytes()


def synth_transcriptome(path, n_records=170_000, total_bases=400_000_000, seed=20240521, wrap=70, uniform=False,
                        min_len=None, return_sequences=False):
    """Write the FASTA to `path`.  Returns dict(n_records, n_bases, kept (records without PREDICTED), lengths)
    and, if return_sequences, the kept records' ASCII bases concatenated + offsets (for synth_reads).

    uniform=True: i.i.d. uniform bases, no shared genes, no tails (the worst case for nnz: nnz ~ #k-mers)."""
    rng = np.random.default_rng(seed)
    T = int(n_records)
    scale = total_bases / (T * 2350.0)
    lo = int(min_len) if min_len is not None else max(20, int(200 * min(1.0, scale)))
    hi = max(lo, int(110000 * max(1.0, scale)))
    lengths = _lengths(rng, T, total_bases, lo, hi)

    # genes: isoform counts ~ 1 + Geometric (mean 4), contiguous blocks of records
    n_genes = max(1, int(round(T * 42.0 / 170.0)))
    per_gene = rng.geometric(0.25, n_genes)
    gene_of = np.repeat(np.arange(n_genes), per_gene)[:T]
    if gene_of.size < T:
        gene_of = np.concatenate([gene_of, np.full(T - gene_of.size, n_genes - 1)])
    if uniform:
        gene_of = np.arange(T)
    # gene sequence: 1.25 x its longest isoform
    gmax = np.zeros(gene_of.max() + 1, np.int64)
    np.maximum.at(gmax, gene_of, lengths)
    glen = (gmax * (1.0 if uniform else 1.25)).astype(np.int64) + 1
    goff = np.concatenate([[0], np.cumsum(glen)])
    p = [0.25, 0.25, 0.25, 0.25] if uniform else [0.26, 0.24, 0.24, 0.26]
    pool = _ACGT[np.searchsorted(np.cumsum(p)[:3], rng.random(int(goff[-1]), dtype=np.float32)).astype(np.uint8)]

    tail = np.where(rng.random(T) < (0.0 if uniform else 0.30), rng.integers(0, 201, T), 0)
    nrun = rng.random(T) < (0.0 if uniform else 0.0005)
    predicted = rng.random(T) < (0.0 if uniform else 0.005)
    cut_frac = rng.random(T)
    start_frac = rng.random(T)
    skip_frac = rng.random(T)

    kept_seqs = [] if return_sequences else None
    n_bases = 0
    out_len = np.zeros(T, np.int64)
    with open(path, "wb") as f:
        chunk = []
        for r in range(T):
            g = gene_of[r]
            G = pool[goff[g]:goff[g + 1]]
            ell = int(lengths[r])
            slack = G.size - ell
            d = int(skip_frac[r] * slack)            # one internal deletion ("skipped exon")
            a = int(start_frac[r] * (slack - d + 1))
            c = int(cut_frac[r] * ell)
            if d:
                seq = np.concatenate([G[a:a + c], G[a + c + d:a + ell + d]])
            else:
                seq = G[a:a + ell].copy() if (tail[r] or nrun[r]) else G[a:a + ell]
            if tail[r]:
                seq = np.concatenate([seq, np.full(int(tail[r]), 65, np.uint8)])
            if nrun[r]:
                k = int(rng.integers(1, 51))
                at = int(rng.integers(0, max(1, seq.size - k)))
                seq[at:at + k] = 78
            hdr = b">NM_%07d.1 %ssynthetic gene %d\n" % (r, b"PREDICTED: " if predicted[r] else b"", g)
            chunk.append(hdr)
            chunk.append(_wrap(seq, wrap))
            n_bases += seq.size
            out_len[r] = seq.size
            if return_sequences and not predicted[r]:
                kept_seqs.append(seq)
            if len(chunk) >= 4096:
                f.write(b"".join(chunk))
                chunk = []
        f.write(b"".join(chunk))
    info = dict(n_records=T, n_bases=int(n_bases), kept=int(T - predicted.sum()), lengths=out_len[~predicted])
    if return_sequences:
        info["offsets"] = np.concatenate([[0], np.cumsum([s.size for s in kept_seqs])]).astype(np.int64)
        info["concat"] = np.concatenate(kept_seqs) if kept_seqs else np.zeros(0, np.uint8)
    return info


def synth_reads(path, concat, offsets, n_reads, read_len=100, alpha=0.1, seed=42, chunk=200_000):
    """Two-line FASTA reads sampled as simulate_reads does; isoforms shorter than read_len are not sampled
    (the reference filters them with L, kmerexpr/simulate_reads.py:84-85)."""
    rng = np.random.default_rng(seed)
    lens = np.diff(offsets)
    ok = lens >= read_len
    T = int(ok.sum())
    if T == 0:
        raise ValueError("no isoform is as long as a read")
    starts0 = offsets[:-1][ok]
    span = (lens[ok] - read_len + 1).astype(np.float64)
    psi = rng.dirichlet(np.full(T, alpha))
    theta = psi * span
    theta /= theta.sum()
    cdf = np.cumsum(theta)
    cdf[-1] = 1.0
    hw = len(str(max(n_reads - 1, 1)))
    ar = np.arange(read_len, dtype=np.int64)
    with open(path, "wb") as f:
        for lo in range(0, n_reads, chunk):
            n = min(chunk, n_reads - lo)
            iso = np.searchsorted(cdf, rng.random(n), side="right").clip(0, T - 1)
            st = starts0[iso] + np.floor(rng.random(n) * span[iso]).astype(np.int64)
            rec = np.empty((n, 5 + hw + 1 + read_len + 1), np.uint8)
            rec[:, :5] = np.frombuffer(b">sim-", np.uint8)
            ids = np.arange(lo, lo + n, dtype=np.int64)
            for k in range(hw):
                rec[:, 5 + hw - 1 - k] = 48 + (ids // 10 ** k) % 10
            rec[:, 5 + hw] = 10
            rec[:, 5 + hw + 1:5 + hw + 1 + read_len] = concat[st[:, None] + ar[None, :]]
            rec[:, -1] = 10
            f.write(rec.tobytes())
    return dict(n_reads=int(n_reads), read_len=int(read_len))


def cached(path, make):
    """Generate `path` with make(tmp_path) unless it already exists (atomic rename)."""
    if not os.path.exists(path):
        tmp = f"{path}.tmp{os.getpid()}"
        info = make(tmp)
        os.replace(tmp, path)
        return info
    return None
