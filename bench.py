#!/usr/bin/env python3
"""bench.py -- throughput of the k-mer hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload x8|x14|x12|y12|y8|y14|g8|g14]

A step = one pass of the hot path over one batch of synthetic input that is already resident in HBM:
  x8  (default) BASELINE.json configs[1]: synthetic GRCh38-RefSeq-shaped transcriptome, ~170k isoforms / ~400 Mbases,
      K=8 -> CSR X (kmers_hip_build_x: shred, stable radix partition, run-length reduce, emit)
  x14 configs[2]: the same transcriptome at K=14
  y12 configs[3] per-GPU share: simulated 100 bp reads -> dense int32 histogram at K=12 (kmers_hip_count_y)
  g8, g14 (not a BASELINE.json config): one logp_grad evaluation (X.theta, the log-likelihood
      term and X^T.(y / X.theta)) on the X and y of the configs above, resident in HBM; unit non-zeros/s; for N > 1 every
      rank holds its own isoforms as a column shard of X, y and X.theta are all-reduced (kmers_amd.distributed.ShardedModel)
For N > 1 (one process per GPU, torchrun) every rank works on its own shard of the same shape (weak scaling: X column
shards need no collective; y shards are summed with one RCCL all-reduce inside the timed step).
Rank 0 prints ONE JSON line.  `value` is k-mers/s over all ranks, a k-mer being one window position
(sum over records of max(len-K+1, 0)).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "k-mers/sec (transcriptome_to_x + reads->y) at K=8/14; achieved HBM GB/s"
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="x8", help="x<K>, y<K> or g<K>, K in 1..16 (x8, x14, y12 are BASELINE.json's)")
    ap.add_argument("--records", type=int, default=170_000)
    ap.add_argument("--bases", type=int, default=400_000_000)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU for the y workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cache", default=os.environ.get("KMERS_BENCH_CACHE", "/tmp/kmers_bench"))
    args = ap.parse_args()
    import re
    if not re.fullmatch(r"[xyg]([1-9]|1[0-6])", args.workload):
        ap.error("--workload must be x<K>, y<K> or g<K> with K in 1..16")
    return args


def make_inputs(args, rank, want_reads):
    """Synthetic FASTA files for this rank (cached on local disk; generation is not timed)."""
    from kmers_amd import synth
    os.makedirs(args.cache, exist_ok=True)
    seed = 20240521 + rank
    iso = os.path.join(args.cache, f"tx_s{seed}_r{args.records}_b{args.bases}.fsa")
    reads = os.path.join(args.cache, f"reads_s{42 + rank}_n{args.reads}_tx{seed}_r{args.records}_b{args.bases}.fsa")
    need_reads = want_reads and not os.path.exists(reads)
    if not os.path.exists(iso) or need_reads:
        tmp = f"{iso}.tmp{os.getpid()}"
        info = synth.synth_transcriptome(tmp, n_records=args.records, total_bases=args.bases, seed=seed,
                                         return_sequences=need_reads)
        os.replace(tmp, iso)
        if need_reads:
            tmp = f"{reads}.tmp{os.getpid()}"
            synth.synth_reads(tmp, info["concat"], info["offsets"], args.reads, read_len=100, seed=42 + rank)
            os.replace(tmp, reads)
    return iso, reads


def cpu_baseline(args, K, kind_x):
    """The reference library itself (oracle/_ref, built in the build container from /root/reference/src) -- or, when
    that build is absent, the oracle port -- timed on the host cores on a bounded sample of the same workload."""
    from kmers_amd import synth
    from tests import kmtest
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libkmers_ref.so")
    cores = min(16, os.cpu_count() or 1)
    os.environ["KMER_NUM_THREADS"] = str(cores)
    if os.path.exists(ref_so):
        lib, kind = kmtest.Lib(ref_so, "", hip=False), "reference"
    else:
        lib, kind, cores = kmtest.load_oracle(), "port", 1
    frac = 4 if kind_x else 10          # ~20-40 core-seconds of the reference either way
    n_rec, n_bases = max(100, args.records // frac), max(100_000, args.bases // frac)
    sample = os.path.join(args.cache, f"cpu_sample_r{n_rec}_b{n_bases}.fsa")
    info = None
    if not os.path.exists(sample):
        info = synth.synth_transcriptome(sample + ".tmp", n_records=n_rec, total_bases=n_bases, seed=20240521,
                                         return_sequences=not kind_x)
        os.replace(sample + ".tmp", sample)
    path = sample
    what = f"{n_rec} isoforms / {n_bases / 1e6:.0f} Mbases of the same synthetic transcriptome"
    if not kind_x:
        path = sample + ".reads"
        n_reads = max(1000, args.reads // 10)
        if not os.path.exists(path):
            if info is None or "concat" not in info:
                info = synth.synth_transcriptome(sample + ".tmp", n_records=n_rec, total_bases=n_bases, seed=20240521,
                                                 return_sequences=True)
                os.replace(sample + ".tmp", sample)
            synth.synth_reads(path + ".tmp", info["concat"], info["offsets"], n_reads, read_len=100, seed=42)
            os.replace(path + ".tmp", path)
        what = f"{n_reads} simulated 100 bp reads"
    from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store
    st = Store(path, MODE_X if kind_x else MODE_Y_FASTA)
    windows = st.n_windows(K)
    st.close()
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    sys.stdout.flush()
    os.dup2(devnull, 1)                  # the reference prints phase timings to stdout
    try:
        t0 = time.perf_counter()
        if kind_x:
            rc = lib.x([path], K)[0]
        else:
            rc = lib.y([path], K)[0]
        dt = time.perf_counter() - t0
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
        os.close(saved)
    assert rc == 0, rc
    return {"value": windows / dt, "unit": "k-mers/s", "cores": cores, "kind": kind,
            "sample": f"{what}, one call of {'fasta_to_kmers_csr' if kind_x else 'fasta_count_kmers'} at K={K}, "
                      f"{dt:.1f} s wall, file in page cache"}


def bench_model(args, K, dev, rank, world, fence):
    """g<K>: logp_grad on the resident X and y (kmers_hip_model_*).  N > 1: weak scaling over column shards -- every
    rank's transcriptome is its shard of the columns, the histogram and X.theta are all-reduced."""
    import torch
    from kmers_amd import _lib
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
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(counts)              # every rank masks the same rows: y of all ranks' reads
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
            dist.all_reduce(xtheta)          # RCCL: float32[masked rows], the evaluation's one exchange step
            val, _ = dm.grad_from(xtheta)

    for _ in range(args.warmup):
        step()
    fence()
    _lib.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    table = _lib.profile_table()
    _lib.profile(False)
    if world > 1:
        t = torch.tensor([dt, float(dm.nnz)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, nnz_all = float(tmax[0].item()), float(t[1].item())
    else:
        nnz_all = float(dm.nnz)
    if rank != 0:
        return
    ms_per_step = dt / args.steps * 1e3
    nnz, n_m = dm.nnz, dm.n_masked
    kernel_bytes = {      # per launch: 8 bytes per non-zero, row pointers in, one float per row out, the gathered vector once
        "k_spmv<x.theta>": 8 * nnz + 4 * (n_m + 1) + 4 * n_m + 4 * T,
        "k_spmv<w.x>": 8 * nnz + 4 * (T + 1) + 4 * T + 4 * n_m,
        "k_logp_terms": (4 + 4 + 4) * n_m,
    }
    dom = max((r for r in table if r[0] in kernel_bytes), key=lambda r: r[1])
    avg_ms = dom[1] / dom[2]
    ach = kernel_bytes[dom[0]] / (avg_ms * 1e-3) / 1e9
    path_bytes = sum(kernel_bytes.values())
    traffic = None            # measured by rocprofv3 PMC passes of this same command (tools/pmc.sh + tools/pmc_traffic.py)
    tpath = os.path.join(ROOT, "kmers_amd", "data", "pmc_traffic.json")
    if (args.records, args.bases, args.reads) == (170_000, 400_000_000, 10_000_000) and os.path.exists(tpath):
        with open(tpath) as f:
            traffic = json.load(f).get(args.workload, {}).get("kernels", {}).get(dom[0], {}).get("bytes")
    out = {
        "metric": "logp_grad non-zeros/s (X.theta + X^T.(y/X.theta) on the resident X; SURVEY.md 8f row 4)",
        "value": 2 * nnz_all * args.steps / dt, "unit": "non-zeros/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 in, f64 accumulate", "data": "synthetic",
        "config": {"workload": f"logp_grad K={K}: X of {T} isoforms / {xs.n_bases / 1e6:.0f} Mbases, y of {args.reads} "
                               f"simulated 100 bp reads per GPU ({args.workload}"
                               + ("; column shards, xtheta all-reduced)" if world > 1 else ")"),
                   "K": K, "nnz_x": nnz_x, "masked_rows": n_m, "nnz_xnnz": nnz, "val": val,
                   "model_setup_s": round(setup_s, 3), "parallelism": f"column-shards{world}"},
        "roofline": {"bound": "hbm", "kernel": dom[0], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_ms": round(avg_ms, 4),
                     "launches_per_step": dom[2] // max(args.steps, 1), "algorithmic_bytes": int(kernel_bytes[dom[0]])},
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
    print(json.dumps(out))


def main():
    args = parse()
    if os.environ.get("KMERS_BENCH_WATCHDOG"):     # seconds: dump every thread's stack and exit instead of hanging
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["KMERS_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    kind_x = args.workload[0] == "x"
    K = int(args.workload[1:])

    import torch
    import torch.distributed as dist
    from kmers_amd import _lib
    from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store

    # KMERS_BENCH_REHEARSAL=1: several ranks on ONE GPU over gloo (RCCL refuses two ranks on a device) -- to rehearse the
    # multi-rank control flow on a one-GPU box; never a measurement
    rehearsal = os.environ.get("KMERS_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if args.workload[0] == "g":
        def fence0():
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
        bench_model(args, K, dev, rank, world, fence0)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    iso, reads = make_inputs(args, rank, want_reads=not kind_x)
    st = Store(iso if kind_x else reads, MODE_X if kind_x else MODE_Y_FASTA)
    ds = st.to_device(dev)
    windows = st.n_windows(K)
    n_bases, n_rec, n_pos = st.n_bases, st.n_rec, st.n_pos
    M = 4 ** K
    instances = int(ds.count_y(K).sum().item())      # valid windows (entries the X path sorts); not timed

    if kind_x:
        buf = ds.x_buffers(K, windows)
        nnz = 0

        def step():
            nonlocal nnz
            nnz = ds.build_x(buf)
    else:
        counts = torch.zeros(M, dtype=torch.int32, device=dev)
        y_ws = ds.y_workspace(K)             # K >= 9: partition scratch, kept across steps

        def step():
            counts.zero_()
            ds.count_y(K, counts, workspace=y_ws)
            if world > 1:
                dist.all_reduce(counts)      # RCCL ring over xGMI: the reference's shared-array "+=" across GPUs

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    _lib.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    table = _lib.profile_table()
    _lib.profile(False)

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    w = torch.tensor([float(windows)], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
    dt_max, windows_all = float(t.item()), float(w.item())

    if rank == 0:
        ms_per_step = dt_max / args.steps * 1e3
        value = windows_all * args.steps / dt_max
        # ---- algorithmic bytes per rank
        B = n_bases
        E = instances
        if kind_x:
            path_bytes = B / 4 + B / 8 + 8 * (n_rec + 1) + nnz * 12 + 8 * (M + 1)
            es = 4 if (K <= 8 and n_rec <= 1 << 24) else 8      # bytes per partition entry
            # what each kernel has to move at minimum for its own job (per launch)
            kernel_bytes = {
                "k_hist_bases": n_pos * 3 / 8 + 8 * (n_rec + 1),
                "k_scatter<bases>": n_pos * 3 / 8 + 8 * (n_rec + 1) + es * E,
                "k_hist_entries": es * E,
                "k_scatter<entries>": 2 * es * E,
                "k_rle": es * E + 12 * nnz + 8 * (M + 1),
            }
        else:
            path_bytes = B / 4 + B / 8 + 4 * M
            kernel_bytes = {
                "k_count_y_lds": n_pos * 3 / 8 * (2 if K == 8 else 1) + 4 * M,
                "k_count_y_global": n_pos * 3 / 8 + 8 * E,
                "k_hist_words_y": n_pos * 3 / 8,
                "k_scatter<bases,y>": n_pos * 3 / 8 + (2 if K <= 12 else 4) * E,   # 2-byte payloads with one pass
                "k_hist_entries": 4 * E,
                "k_scatter<entries>": 8 * E,
                "k_count_y_sorted": (2 if K <= 12 else 4) * E + 8 * M,
            }
        dom = max((r for r in table if r[0] in kernel_bytes), key=lambda r: r[1], default=None)
        # HBM traffic per launch measured with rocprofv3 PMC passes of this same command (tools/pmc.sh +
        # tools/pmc_traffic.py; counters cannot be read from inside the process), default sizes only
        traffic = None
        default_size = (args.records, args.bases, args.reads) == (170_000, 400_000_000, 10_000_000)
        tpath = os.path.join(ROOT, "kmers_amd", "data", "pmc_traffic.json")
        if dom is not None and default_size and os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(args.workload, {}).get("kernels", {}).get(dom[0], {}).get("bytes")
        roof = None
        if dom is not None:
            avg_ms = dom[1] / dom[2]
            ach = kernel_bytes[dom[0]] / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom[0], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "avg_ms": round(avg_ms, 4),
                    "launches_per_step": dom[2] // max(args.steps, 1), "algorithmic_bytes": int(kernel_bytes[dom[0]])}
        else:
            ach = path_bytes / (ms_per_step * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_count_y", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_ms": round(ms_per_step, 4),
                    "algorithmic_bytes": int(path_bytes)}
        path_gbs = path_bytes / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": METRIC, "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{'transcriptome_to_x' if kind_x else 'reads->y'} K={K}: "
                                   f"{n_rec} records / {n_bases / 1e6:.0f} Mbases per GPU "
                                   f"({'BASELINE.json configs[1]' if args.workload == 'x8' else args.workload})",
                       "K": K, "records_per_gpu": n_rec, "bases_per_gpu": n_bases, "kmers_per_gpu": windows,
                       "valid_instances_per_gpu": instances, "nnz_per_gpu": (nnz if kind_x else None),
                       "parallelism": f"shards{world}"},
            "roofline": roof,
            "path": {"algorithmic_bytes": int(path_bytes), "bytes_per_kmer": round(path_bytes / max(windows, 1), 3),
                     "achieved_GBs": round(path_gbs, 1), "frac_of_hbm_peak": round(path_gbs / HBM_PEAK_GBS, 4)},
            "kernels_ms": {r[0]: round(r[1] / args.steps, 4) for r in table},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, K, kind_x)
            except Exception as e:  # the baseline must never hide the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
