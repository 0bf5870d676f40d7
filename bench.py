#!/usr/bin/env python3
"""bench.py -- throughput of the k-mer hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload x8|x14|y12|y14|y8|g8|g14 ...] [--only]

A step = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  x8   BASELINE.json configs[1], the headline line: synthetic GRCh38-RefSeq-shaped transcriptome (seed 20240521), ~170k
       isoforms / ~400 Mbases, K=8 -> CSR X (kmers_hip_build_x)
  x14  configs[2]: the same transcriptome at K=14
  y12  configs[3]: 10^8 simulated 100 bp reads (kmers_amd.simulate_reads: the reference's sampling, reads written straight
       into HBM) -> dense int32 histogram at K=12 (kmers_hip_count_y); y14: the same reads at K=14
  g8, g14 (not a BASELINE.json config): one logp_grad evaluation on the resident X and y
With no --workload the line is x8 and carries an `also` list with x14, y12 and y14 timed in the same process, each with
its own ms_per_step, roofline and cpu_baseline (--only: the headline alone).

N > 1 (one process per GPU, torchrun): STRONG scaling of the same job.  The ONE transcriptome is split into N contiguous
record ranges (kmers_store_ingest_bytes: rank r parses only its N-th of the file's bytes) and rank r builds the CSR of its columns; the ONE set of reads is
split the same way and the histograms are summed with one RCCL all-reduce inside the timed step.  X needs no data-path
collective; what north_star calls "concatenated on the host" is timed once, apart, after the steps (`concat`: every rank
copies its shard over its own PCIe link and places it in a shared host buffer, kmers_amd.distributed.concat_x_shards)
and never enters `value`, which is HBM-resident throughput as the bench contract asks.
Rank 0 prints ONE JSON line.  `value` is k-mers/s over all ranks, a k-mer being one window position
(sum over records of max(len-K+1, 0)).
"""
import argparse
import hashlib
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "k-mers/sec (transcriptome_to_x + reads->y) at K=8/14; achieved HBM GB/s"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALSO = ["x14", "y12", "y14"]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, help="x<K>, y<K> or g<K>, K in 1..16; default x8 with x14, y12, y14 in `also`")
    ap.add_argument("--only", action="store_true", help="no `also` list")
    ap.add_argument("--records", type=int, default=170_000)
    ap.add_argument("--bases", type=int, default=400_000_000)
    ap.add_argument("--reads", type=int, default=100_000_000, help="simulated reads of the y workloads, over ALL ranks")
    ap.add_argument("--model-reads", type=int, default=10_000_000, help="reads of the g workloads (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the wall time of the file-level drop-in call")
    ap.add_argument("--no-concat", action="store_true", help="N > 1: skip the host concatenation of the X shards")
    ap.add_argument("--cache", default=os.environ.get("KMERS_BENCH_CACHE", "/tmp/kmers_bench"))
    args = ap.parse_args(argv)
    args.explicit = args.workload is not None
    if args.workload is None:
        args.workload = "x8"
    if not re.fullmatch(r"[xyg]([1-9]|1[0-6])", args.workload):
        ap.error("--workload must be x<K>, y<K> or g<K> with K in 1..16")
    return args


# ------------------------------------------------------------------------------------------------ inputs (not timed)

def transcriptome_file(args, seed=20240521, with_reads=0, reads_seed=42):
    """The synthetic transcriptome FASTA (cached on local disk) and, if asked, a reads FASTA drawn from it."""
    from kmers_amd import synth
    os.makedirs(args.cache, exist_ok=True)
    iso = os.path.join(args.cache, f"tx_s{seed}_r{args.records}_b{args.bases}.fsa")
    reads = os.path.join(args.cache, f"reads_s{reads_seed}_n{with_reads}_tx{seed}_r{args.records}_b{args.bases}.fsa")
    need_reads = with_reads and not os.path.exists(reads)
    if not os.path.exists(iso) or need_reads:
        tmp = f"{iso}.tmp{os.getpid()}"
        info = synth.synth_transcriptome(tmp, n_records=args.records, total_bases=args.bases, seed=seed,
                                         return_sequences=bool(need_reads))
        os.replace(tmp, iso)
        if need_reads:
            tmp = f"{reads}.tmp{os.getpid()}"
            synth.synth_reads(tmp, info["concat"], info["offsets"], with_reads, read_len=100, seed=reads_seed)
            os.replace(tmp, reads)
    return iso, (reads if with_reads else None)


def make_inputs(args, rank, want_reads):
    """(transcriptome FASTA, reads FASTA) of the g workloads and the tools: a transcriptome of this rank's own."""
    return transcriptome_file(args, seed=20240521 + rank, with_reads=args.model_reads if want_reads else 0,
                              reads_seed=42 + rank)


class Ctx:
    """Process-wide state of one bench run."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        # KMERS_BENCH_REHEARSAL=1: several ranks on ONE GPU over gloo (RCCL refuses two ranks on a device) -- to rehearse
        # the multi-rank control flow on a one-GPU box; never a measurement
        self.rehearsal = os.environ.get("KMERS_BENCH_REHEARSAL") == "1"
        if self.rehearsal:
            local = 0
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        if self.world > 1:
            if self.rehearsal:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)
        self._iso = None
        self._xstore = None
        self._reads = None

    def fence(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def all_max_sum(self, t_local, *sums):
        """max over ranks of t_local, sums over ranks of the rest."""
        torch, dist = self.torch, self.dist
        if self.world == 1:
            return (t_local,) + tuple(float(s) for s in sums)
        dev = torch.device("cpu") if self.rehearsal else self.dev
        tm = torch.tensor([t_local], dtype=torch.float64, device=dev)
        sm = torch.tensor([float(s) for s in sums] or [0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        return (float(tm[0].item()),) + tuple(float(v) for v in sm[:len(sums)].tolist())

    def all_reduce_dev(self, t):
        """sum of a device tensor over the ranks (RCCL over xGMI; through the host in a gloo rehearsal)."""
        if self.world == 1:
            return
        if self.rehearsal:
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
        else:
            self.dist.all_reduce(t)

    def iso_file(self):
        if self._iso is None:
            if self.rank == 0:
                self._iso = transcriptome_file(self.args)[0]
            if self.world > 1:
                self.dist.barrier()          # one node: the other ranks read rank 0's file
            self._iso = transcriptome_file(self.args)[0]
        return self._iso

    def x_store(self):
        """This rank's shard (contiguous record range) of the ONE transcriptome, host and device side."""
        if self._xstore is None:
            from kmers_amd.store import MODE_X, Store
            if self.world > 1:       # rank r parses only its byte range of the file (kmers_store_ingest_bytes)
                from kmers_amd.distributed import ingest_sharded
                st = ingest_sharded(self.iso_file(), MODE_X)
            else:
                st = Store(self.iso_file(), MODE_X)
            self._xstore = (st, st.to_device(self.dev))
        return self._xstore

    def reads_store(self):
        """This rank's share of the ONE set of simulated reads, in HBM (kmers_hip_simulate_reads)."""
        if self._reads is None:
            from kmers_amd.simulate_reads import simulate_reads_device
            ds, _ = simulate_reads_device(self.args.reads, 100, 0.1, self.iso_file(), device=self.dev, shard=self.rank,
                                          n_shards=self.world)
            self._reads = ds
        return self._reads


# ------------------------------------------------------------------------------------------------ measurement helpers

def kernels_sha():
    h = hashlib.sha256()
    for f in ("kernels.hip", "model.hip"):
        with open(os.path.join(ROOT, "kmers_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(workload, kernel, default_size):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of this same command (tools/pmc.sh +
    tools/pmc_traffic.py write profiles/pmc_traffic.json; counters cannot be read from inside the process).  Only quoted
    when the file was made from the kernel sources that are in the tree now and for the default sizes."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not default_size or not os.path.exists(path):
        return None
    with open(path) as f:
        allw = json.load(f)
    w = allw.get(workload, {})
    if w.get("kernels_sha") != kernels_sha():
        return None
    return w.get("kernels", {}).get(kernel, {}).get("bytes")


def timed(ctx, step):
    """W warm-up steps, then exactly K steps between two fences; per-kernel HIP-event timings of the K steps."""
    from kmers_amd import _lib
    a = ctx.args
    for _ in range(a.warmup):
        step()
    ctx.fence()
    _lib.profile(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    ctx.fence()
    dt = time.perf_counter() - t0
    table = _lib.profile_table()
    _lib.profile(False)
    return dt, table


def roofline(table, kernel_bytes, path_bytes, ms_per_step, steps, workload, default_size):
    dom = max((r for r in table if r[0] in kernel_bytes), key=lambda r: r[1], default=None)
    if dom is None:
        ach = path_bytes / (ms_per_step * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "whole step", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_ms": round(ms_per_step, 4),
                "algorithmic_bytes": int(path_bytes)}
    avg_ms = dom[1] / dom[2]
    ach = kernel_bytes[dom[0]] / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": dom[0], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(workload, dom[0], default_size),
            "avg_ms": round(avg_ms, 4), "launches_per_step": dom[2] // max(steps, 1),
            "algorithmic_bytes": int(kernel_bytes[dom[0]])}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_threads():
    """Threads this process may use: the box's CPU share (OMP_NUM_THREADS is set to it there), else all CPUs."""
    for var in ("KMERS_BENCH_THREADS", "OMP_NUM_THREADS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            return int(v)
    return os.cpu_count() or 1


def sample_file(args, path, max_bytes):
    """A prefix of a FASTA file that ends at a record boundary (the bounded CPU sample), cached beside it."""
    if os.path.getsize(path) <= max_bytes:
        return path
    out = f"{path}.first{max_bytes}"
    if not os.path.exists(out):
        with open(path, "rb") as f:
            buf = f.read(max_bytes)
        cut = buf.rfind(b"\n>")
        tmp = f"{out}.tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(buf[:cut + 1])
        os.replace(tmp, out)
    return out


def cpu_baseline(args, K, kind_x):
    """The reference library itself (oracle/_ref, built in the build container from /root/reference/src) -- or, when that
    build is absent, the oracle port -- timed on the host cores as SURVEY.md section 8d asks: with all the threads this
    process may use AND with KMER_NUM_THREADS=1 (the reference's knob, src/libkmers.cpp:50-52), median over up to 5 calls
    each, on a bounded sample of the GPU run's own input (a prefix of the transcriptome file / reads drawn from it the same
    way) sized so that both legs together take some tens of seconds."""
    from tests import kmtest
    from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libkmers_ref.so")
    threads = host_threads()
    if os.path.exists(ref_so):
        lib, kind = kmtest.Lib(ref_so, "", hip=False), "reference"
    else:
        lib, kind, threads = kmtest.load_oracle(), "port", 1
    if kind_x:
        full = transcriptome_file(args)[0]
        path = sample_file(args, full, 60_000_000)
        path1 = sample_file(args, full, 15_000_000)        # the single-thread leg: a quarter of it, so that 5 calls fit
        what = (f"the first {os.path.getsize(path) / 1e6:.0f} MB of the GPU run's transcriptome file"
                if path != full else "the transcriptome file of the GPU run")
    else:
        n_reads = max(1000, min(500_000, args.reads // 200))
        path = transcriptome_file(args, with_reads=n_reads)[1]
        path1 = transcriptome_file(args, with_reads=max(1000, n_reads // 4))[1]
        what = f"{n_reads} reads of 100 bp drawn from the same transcriptome"

    def n_windows(p):
        st = Store(p, MODE_X if kind_x else MODE_Y_FASTA)
        try:
            return st.n_windows(K)
        finally:
            st.close()

    windows, windows1 = n_windows(path), n_windows(path1)

    def one_call(path):
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(1)
        sys.stdout.flush()
        os.dup2(devnull, 1)                  # the reference prints phase timings to stdout
        try:
            t0 = time.perf_counter()
            rc = (lib.x([path], K) if kind_x else lib.y([path], K))[0]
            dt = time.perf_counter() - t0
        finally:
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)
        assert rc == 0, rc
        return dt

    def leg(path, n_threads, budget_s):
        os.environ["KMER_NUM_THREADS"] = str(n_threads)
        times = [one_call(path)]
        while len(times) < 5 and sum(times) + times[0] < budget_s:
            times.append(one_call(path))
        return float(np.median(times)), len(times)

    saved_threads = os.environ.get("KMER_NUM_THREADS")
    try:
        t_all, n_all = leg(path, threads, 8.0)
        t_one, n_one = leg(path1, 1, 12.0)
    finally:
        if saved_threads is None:
            os.environ.pop("KMER_NUM_THREADS", None)
        else:
            os.environ["KMER_NUM_THREADS"] = saved_threads
    return {"value": windows / t_all, "unit": "k-mers/s", "cores": threads, "kind": kind,
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
            "single_thread": {"value": windows1 / t_one, "calls": n_one, "median_s": round(t_one, 3), "kmers": windows1},
            "calls": n_all, "median_s": round(t_all, 3),
            "sample": f"{what} ({windows} k-mers), {'fasta_to_kmers_csr' if kind_x else 'fasta_count_kmers'} at K={K}, "
                      f"file in page cache; value = median of {n_all} calls with KMER_NUM_THREADS={threads}, "
                      f"single_thread = median of {n_one} calls with KMER_NUM_THREADS=1 on a quarter of that sample"}


def e2e_record(args, K, kind_x):
    """Wall time of the drop-in call users make (kmerexpr/libkmers.py:130-184 / :86-105): file in the page cache -> the
    caller's numpy arrays, with the library's own stage split.  Never `value`: it includes host parsing and PCIe."""
    from kmers_amd import _lib, libkmers
    if kind_x:
        path = transcriptome_file(args)[0]
        call, what = "fasta_to_kmers_csr", f"the transcriptome file of the GPU run ({os.path.getsize(path) / 1e6:.0f} MB)"
        fn = lambda: libkmers.fasta_to_kmers_csr(path, K)
    else:
        n_reads = max(1000, min(2_000_000, args.reads // 50))
        path = transcriptome_file(args, with_reads=n_reads)[1]
        call, what = "fasta_count_kmers", f"{n_reads} reads of 100 bp as a FASTA file ({os.path.getsize(path) / 1e6:.0f} MB)"
        fn = lambda: libkmers.fasta_count_kmers(path, K)
    fn()                                      # page cache, pinned pools, code objects
    times, stages = [], None
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
        if times[-1] == min(times):
            stages = _lib.last_call_stages()
    rec = {"call": call, "input": what, "wall_s": round(min(times), 4), "calls": len(times),
           "stages_ms": {k: round(v, 2) for k, v in (stages or {}).items()},
           "host_threads": host_threads()}
    if kind_x:       # the call the reference's users make (kmerexpr/transcriptome_reader.py:19-41): the arrays + scipy's csr_matrix
        from kmers_amd import transcriptome_reader as tr
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            x = tr.transcriptome_to_x(K, path)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            del x
        rec["transcriptome_to_x_s"] = round(best, 4)
    return rec


# ------------------------------------------------------------------------------------------------ the workloads

def run_x(ctx, K, tag):
    """transcriptome_to_x's device stage on this rank's columns of the one transcriptome."""
    args = ctx.args
    st, ds = ctx.x_store()
    windows, n_bases, n_rec, n_pos = st.n_windows(K), st.n_bases, st.n_rec, st.n_pos
    M = 4 ** K
    instances = int(ds.count_y(K).sum().item())      # valid windows (entries the X path sorts); not timed
    buf = ds.x_buffers(K, windows)
    nnz = 0

    def step():
        nonlocal nnz
        nnz = ds.build_x(buf)

    dt, table = timed(ctx, step)
    concat = concat_err = None
    if ctx.world > 1 and not args.no_concat:        # what north_star calls "X columns are concatenated on the host"
        from kmers_amd.distributed import concat_x_shards
        ctx.fence()
        t0 = time.perf_counter()
        try:
            whole = concat_x_shards(buf["data"], buf["col_ind"], buf["row_ind"], nnz, M)
            ctx.fence()
            concat = time.perf_counter() - t0
            if ctx.rank == 0:
                assert int(whole[1][-1]) == whole[0].shape[0]
            del whole
        except RuntimeError as e:       # no room on the host for the whole matrix: raised on every rank alike
            concat_err = str(e)
    dt_max, windows_all, nnz_all, bases_all, rec_all = ctx.all_max_sum(dt, windows, nnz, n_bases, n_rec)
    concat_max = ctx.all_max_sum(concat or 0.0)[0]
    del buf
    ctx.torch.cuda.empty_cache()
    if ctx.rank != 0:
        return None
    ms = dt_max / args.steps * 1e3
    E = instances
    path_bytes = n_bases / 4 + n_bases / 8 + 8 * (n_rec + 1) + nnz * 12 + 8 * (M + 1)       # this rank
    es = 4 if (K <= 8 and n_rec <= 1 << 24) else 8      # bytes per partition entry
    kernel_bytes = {    # what each kernel has to move at minimum for its own job (per launch)
        "k_hist_bases": n_pos * 3 / 8 + 8 * (n_rec + 1),
        "k_scatter<bases>": n_pos * 3 / 8 + 8 * (n_rec + 1) + es * E,
        "k_hist_entries": es * E,
        "k_scatter<entries>": 2 * es * E,
        "k_rle": es * E + 12 * nnz + 8 * (M + 1),
        # K <= 8, round 3: the last pass writes the CSR arrays itself (k_heads_hist counts heads, k_emit partitions + emits)
        "k_heads_hist": es * E,
        "k_emit": es * E + 12 * nnz + 4 * n_rec,
        # the small-K table path (KMERS_X_PATH=tables): entries are the nnz distinct (k-mer, isoform) pairs
        "k_dd<hist>": n_pos * 3 / 8 + 8 * (n_rec + 1),
        "k_dd<scatter>": n_pos * 3 / 8 + 8 * (n_rec + 1) + 4 * nnz,
        "k_final_scatter": 4 * nnz + 12 * nnz + 4 * n_rec,
    }
    if "k_dd<hist>" in {r[0] for r in table}:
        kernel_bytes["k_hist_entries"] = 4 * nnz
    full_size = (args.records, args.bases) == (170_000, 400_000_000)
    default_size = full_size and ctx.world == 1
    gbs = path_bytes / (ms * 1e-3) / 1e9
    out = {
        "metric": METRIC, "value": windows_all * args.steps / dt_max, "unit": "k-mers/s", "n_gpus": ctx.world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong" if ctx.world > 1 else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"transcriptome_to_x K={K}: {int(rec_all)} records / {bases_all / 1e6:.0f} Mbases ("
                               + (("BASELINE.json configs[1]" if tag == "x8" else "BASELINE.json configs[2]" if tag == "x14"
                                   else tag) if full_size else f"{tag}, NOT a BASELINE.json size")
                               + (f"; the one transcriptome in {ctx.world} column shards, one per GPU)" if ctx.world > 1 else ")"),
                   "K": K, "records": int(rec_all), "bases": int(bases_all), "kmers": int(windows_all),
                   "nnz": int(nnz_all), "rank0": {"records": n_rec, "bases": n_bases, "valid_instances": E, "nnz": nnz},
                   "parallelism": f"column-shards{ctx.world}"},
        "roofline": roofline(table, kernel_bytes, path_bytes, ms, args.steps, tag, default_size),
        "path": {"algorithmic_bytes": int(path_bytes), "bytes_per_kmer": round(path_bytes / max(windows, 1), 3),
                 "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)},
        "kernels_ms": {r[0]: round(r[1] / args.steps, 4) for r in table},
    }
    if concat_err is not None:
        out["concat"] = {"error": concat_err}
    if concat is not None:
        out["concat"] = {"ms": round(concat_max * 1e3, 2),
                         "what": "D2H of every rank's CSR shard over its own link + placement in one shared host CSR "
                                 "(kmers_amd.distributed.concat_x_shards), once, outside `value`",
                         "kmers_per_s_with_concat": windows_all / (dt_max / args.steps + concat_max)}
    return out


def run_y(ctx, K, tag):
    """reads -> y on this rank's share of the one set of simulated reads; N > 1: + one all-reduce of the histogram."""
    args, torch = ctx.args, ctx.torch
    ds = ctx.reads_store()
    n_reads = ds.n_rec
    n_pos = ds.n_pos
    windows = n_reads * max(100 - K + 1, 0)
    M = 4 ** K
    counts = torch.zeros(M, dtype=torch.int32, device=ctx.dev)
    y_ws = ds.y_workspace(K)             # K >= 9: partition scratch, kept across steps

    def step():
        counts.zero_()
        ds.count_y(K, counts, workspace=y_ws)
        ctx.all_reduce_dev(counts)       # RCCL ring over xGMI: the reference's shared-array "+=" across GPUs

    dt, table = timed(ctx, step)
    E = int(counts.sum(dtype=torch.int64).item())      # valid windows of ALL reads (the histogram is all-reduced)
    allreduce_ms = None
    if ctx.world > 1:                    # the collective alone, timed apart (after the steps) so the curve says where time goes
        ctx.fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.all_reduce_dev(counts)
        ctx.fence()
        allreduce_ms = ctx.all_max_sum((time.perf_counter() - t0) / args.steps * 1e3)[0]
    dt_max, windows_all, reads_all = ctx.all_max_sum(dt, windows, n_reads)
    del y_ws
    torch.cuda.empty_cache()
    if ctx.rank != 0:
        return None
    E = E / ctx.world                                  # per rank, for the per-launch byte counts
    ms = dt_max / args.steps * 1e3
    n_bases = n_reads * 100
    path_bytes = n_bases / 4 + n_bases / 8 + 4 * M
    kernel_bytes = {
        "k_count_y_lds": n_pos * 3 / 8 * (2 if K == 8 else 1) + 4 * M,
        "k_count_y_global": n_pos * 3 / 8 + 8 * E,
        "k_hist_words_y": n_pos * 3 / 8,
        "k_scatter<bases,y>": n_pos * 3 / 8 + (2 if K <= 12 else 4) * E,   # 2-byte payloads with one pass
        "k_hist_entries": 4 * E,
        "k_scatter<entries>": 8 * E,
        "k_count_y_sorted": (2 if K <= 12 else 4) * E + 8 * M,
    }
    n_chunks = max(1, -(-n_pos // (1 << 31)))
    # chunked stores launch every kernel once per chunk: bytes per launch
    kernel_bytes = {k: v / n_chunks for k, v in kernel_bytes.items()}
    full_size = (args.records, args.bases, args.reads) == (170_000, 400_000_000, 100_000_000)
    default_size = full_size and ctx.world == 1
    gbs = path_bytes / (ms * 1e-3) / 1e9
    return {
        "metric": METRIC, "value": windows_all * args.steps / dt_max, "unit": "k-mers/s", "n_gpus": ctx.world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong" if ctx.world > 1 else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": f"reads->y K={K}: {int(reads_all)} simulated 100 bp reads ("
                               + (("BASELINE.json configs[3]" if tag == "y12" else tag) if full_size
                                  else f"{tag}, NOT a BASELINE.json size")
                               + (f"; {ctx.world} shards, histogram all-reduced over RCCL)" if ctx.world > 1 else ")"),
                   "K": K, "reads": int(reads_all), "kmers": int(windows_all), "reads_rank0": n_reads,
                   "chunks_per_step": n_chunks, "parallelism": f"read-shards{ctx.world}"},
        "roofline": roofline(table, kernel_bytes, path_bytes, ms, args.steps, tag, default_size),
        "path": {"algorithmic_bytes": int(path_bytes), "bytes_per_kmer": round(path_bytes / max(windows, 1), 3),
                 "achieved_GBs": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4)},
        "kernels_ms": {r[0]: round(r[1] / args.steps, 4) for r in table},
        **({"allreduce_ms": round(allreduce_ms, 4), "allreduce_bytes": 4 * M} if allreduce_ms is not None else {}),
    }


def run_model(ctx, K):
    """g<K>: logp_grad on the resident X and y (kmers_hip_model_*).  N > 1: weak scaling over column shards -- every
    rank's transcriptome is its shard of the columns, the histogram and X.theta are all-reduced."""
    args, torch, dist = ctx.args, ctx.torch, ctx.dist
    rank, world, dev = ctx.rank, ctx.world, ctx.dev
    from kmers_amd.model import DeviceModel
    from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store
    iso, reads = make_inputs(args, rank, want_reads=True)
    xs = Store(iso, MODE_X)
    dx = xs.to_device(dev)
    buf = dx.x_buffers(K, xs.n_windows(K))
    nnz_x = dx.build_x(buf)
    buf["workspace"] = None
    ys = Store(reads, MODE_Y_FASTA)
    counts = ys.to_device(dev).count_y(K)
    ctx.all_reduce_dev(counts)               # every rank masks the same rows: y of all ranks' reads
    torch.cuda.synchronize()
    T = xs.n_rec_total
    t0 = time.perf_counter()
    dm = DeviceModel(buf["data"], buf["col_ind"], buf["row_ind"], T, counts)
    setup_s = time.perf_counter() - t0
    theta = torch.full((T,), 1.0 / T, dtype=torch.float32, device=dev)
    val = 0.0
    xtheta = torch.empty(dm.n_masked, dtype=torch.float32, device=dev)

    def step():
        nonlocal val
        if world == 1:
            val, _ = dm.logp_grad(theta)
        else:                                # column shards: this rank's isoforms are its columns of X
            dm.xtheta_into(theta, xtheta)
            ctx.all_reduce_dev(xtheta)       # RCCL: float32[masked rows], the evaluation's one exchange step
            val, _ = dm.grad_from(xtheta)

    dt, table = timed(ctx, step)
    dt, nnz_all = ctx.all_max_sum(dt, dm.nnz)
    if rank != 0:
        return None
    ms_per_step = dt / args.steps * 1e3
    nnz, n_m = dm.nnz, dm.n_masked
    kernel_bytes = {      # per launch: 8 bytes per non-zero, row pointers in, one float per row out, the gathered vector once
        "k_spmv<x.theta>": 8 * nnz + 4 * (n_m + 1) + 4 * n_m + 4 * T,
        "k_spmv<w.x>": 8 * nnz + 4 * (T + 1) + 4 * T + 4 * n_m,
        "k_logp_terms": (4 + 4 + 4) * n_m,
    }
    path_bytes = sum(kernel_bytes.values())
    default_size = (args.records, args.bases, args.model_reads) == (170_000, 400_000_000, 10_000_000) and world == 1
    out = {
        "metric": "logp_grad non-zeros/s (X.theta + X^T.(y/X.theta) on the resident X; SURVEY.md 8f row 4)",
        "value": 2 * nnz_all * args.steps / dt, "unit": "non-zeros/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 in, f64 accumulate", "data": "synthetic",
        "config": {"workload": f"logp_grad K={K}: X of {T} isoforms / {xs.n_bases / 1e6:.0f} Mbases, y of {args.model_reads} "
                               f"simulated 100 bp reads per GPU (g{K}"
                               + ("; column shards, xtheta all-reduced)" if world > 1 else ")"),
                   "K": K, "nnz_x": nnz_x, "masked_rows": n_m, "nnz_xnnz": nnz, "val": val,
                   "model_setup_s": round(setup_s, 3), "parallelism": f"column-shards{world}"},
        "roofline": roofline(table, kernel_bytes, path_bytes, ms_per_step, args.steps, f"g{K}", default_size),
        "path": {"algorithmic_bytes": int(path_bytes),
                 "achieved_GBs": round(path_bytes / (ms_per_step * 1e-3) / 1e9, 1)},
        "kernels_ms": {r[0]: round(r[1] / args.steps, 4) for r in table},
    }
    if world == 1 and not args.no_cpu_baseline:
        # the reference's products are MKL's (sparse_dot_mkl, absent here): scipy's CSR kernels on one core stand in,
        # on the same xnnz downloaded from the GPU run's inputs
        try:
            from scipy.sparse import csr_matrix
            n = nnz_x
            x = csr_matrix((buf["data"][:n].cpu().numpy(), buf["col_ind"][:n].cpu().numpy(),
                            buf["row_ind"].cpu().numpy()), shape=(4 ** K, T))
            y = counts.cpu().numpy()
            ymask = y.nonzero()
            xnnz, ynnz = x[ymask], y[ymask] / y.sum()
            th = np.full(T, 1.0 / T, np.float32)
            t0 = time.perf_counter()
            xt = xnnz.dot(th)
            v = float(ynnz.dot(np.log(xt)))
            g = (ynnz / xt).astype(np.float32) @ xnnz
            cdt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": 2 * xnnz.nnz / cdt, "unit": "non-zeros/s", "cores": 1, "kind": "port",
                                   "sample": f"the same logp_grad (utils.py:46-73 in scipy float32, single thread; the "
                                             f"reference's MKL is absent), full size, {cdt:.2f} s, val {v:.6f}, "
                                             f"|grad|_1 {float(np.abs(g).sum()):.4f}"}
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    return out


def run_workload(ctx, tag):
    kind, K = tag[0], int(tag[1:])
    if kind == "g":
        return run_model(ctx, K)
    out = run_x(ctx, K, tag) if kind == "x" else run_y(ctx, K, tag)
    if out is not None and ctx.world == 1 and not ctx.args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(ctx.args, K, kind == "x")
        except Exception as e:  # the baseline must never hide the GPU number
            out["cpu_baseline"] = {"value": None, "error": repr(e)}
    if out is not None and ctx.world == 1 and not ctx.args.no_e2e:
        try:
            out["e2e"] = e2e_record(ctx.args, K, kind == "x")
        except Exception as e:
            out["e2e"] = {"wall_s": None, "error": repr(e)}
    return out


def main():
    args = parse()
    if os.environ.get("KMERS_BENCH_WATCHDOG"):     # seconds: dump every thread's stack and exit instead of hanging
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["KMERS_BENCH_WATCHDOG"]), exit=True)
    ctx = Ctx(args)
    out = run_workload(ctx, args.workload)
    if not args.explicit and not args.only:
        also = []
        for tag in ALSO:
            try:
                line = run_workload(ctx, tag)
            except Exception as e:                 # a side workload must never hide the headline
                line = {"config": {"workload": tag}, "error": repr(e)}
            if line is not None:
                line["workload"] = tag
                also.append(line)
        if out is not None:
            out["also"] = also
    if ctx.rank == 0:
        if ctx.rehearsal:
            out["rehearsal"] = "several ranks on one GPU over gloo: control flow only, not a measurement"
        print(json.dumps(out))
    if ctx.world > 1:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
