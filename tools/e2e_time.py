"""Wall time of the drop-in calls on the bench's config-2 file (host ingest + PCIe + GPU + scipy + .npz), by stage."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from kmers_amd import libkmers, transcriptome_reader as tr
from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store

args = bench.parse()
iso, reads = bench.make_inputs(args, 0, want_reads=True)
K = int(args.workload[1:])
import torch
torch.zeros(1, device="cuda")                      # runtime up before timing

def t(label, fn):
    t0 = time.perf_counter(); r = fn(); dt = time.perf_counter() - t0
    print(f"{label:50s} {dt:8.3f} s", flush=True)
    return r

st = t("Store(iso) host ingest", lambda: Store(iso, MODE_X)); st.close()
t("fasta_to_kmers_csr (ABI: ingest+H2D+GPU+D2H)", lambda: libkmers.fasta_to_kmers_csr(iso, K))
t("fasta_to_kmers_csr again", lambda: libkmers.fasta_to_kmers_csr(iso, K))
x = t("transcriptome_to_x (no file)", lambda: tr.transcriptome_to_x(K, iso))
t("transcriptome_to_x (x_file on /tmp)", lambda: tr.transcriptome_to_x(K, iso, "/tmp/x_e2e.npz"))
os.remove("/tmp/x_e2e.npz")
t("fasta_count_kmers(reads) K=12", lambda: libkmers.fasta_count_kmers(reads, 12))
t("fasta_count_kmers(reads) K=12 again", lambda: libkmers.fasta_count_kmers(reads, 12))
print("nnz", x.nnz, "threads", os.environ.get("KMER_NUM_THREADS"), "cpus", os.cpu_count())
