#!/bin/bash
# tools/refresh_profiles.sh TAG ["workloads"] -- bench lines (with cpu_baseline) and rocprofv3 kernel statistics of every workload
# quoted in DESIGN.md, into gpurun_out/prof_TAG/.  Run on the GPU box; copy what is judged into profiles/.
set -u
TAG=${1:-v8}
WLS=${2:-"x8 y12 x14 y14 y8 g8 g14"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for wl in $WLS; do
  timeout -k 10 400 python3 "$ROOT/bench.py" --workload $wl > "$OUT/${wl}_bench.json" 2> "$OUT/${wl}_bench.err" || { echo "bench $wl failed"; tail -5 "$OUT/${wl}_bench.err"; exit 1; }
  echo "bench $wl ok"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${wl}_stats" -- python3 "$ROOT/bench.py" --workload $wl --no-cpu-baseline > "$OUT/${wl}_stats.log" 2>&1 || { echo "stats $wl failed"; tail -5 "$OUT/${wl}_stats.log"; exit 1; }
  f=$(find "$OUT/${wl}_stats" -name '*kernel_stats.csv' | head -1)
  cp "$f" "$OUT/${wl}_kernel_stats.csv" && rm -rf "$OUT/${wl}_stats"
  echo "stats $wl ok"
done
