"""Host ingest alone (kmers_store_ingest) on the bench's transcriptome and reads files, with the stage laps (KMERS_VERBOSE=2)."""
import os, sys, time
os.environ.setdefault("KMERS_VERBOSE", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kmers_amd.store import MODE_X, MODE_Y_FASTA, Store

args = bench.parse()
iso, reads = bench.transcriptome_file(args, with_reads=2_000_000)
for path, mode in ((iso, MODE_X), (reads, MODE_Y_FASTA)):
    for i in range(3):
        t0 = time.perf_counter()
        st = Store(path, mode)
        dt = time.perf_counter() - t0
        print(f"ingest {os.path.basename(path)} {dt * 1e3:.1f} ms", flush=True)
        st.close()
