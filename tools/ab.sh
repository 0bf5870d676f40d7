#!/bin/bash
# tools/ab.sh TAG "workloads" [ENV=VAL ...] -- short bench lines (no cpu baseline) into gpurun_out/ab_TAG_<wl>.json; run on the GPU box
TAG=$1; WLS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for wl in $WLS; do
  env "$@" timeout -k 10 300 python3 "$ROOT/bench.py" --workload $wl --only --no-cpu-baseline --steps 5 --warmup 2 > "$ROOT/gpurun_out/ab_${TAG}_$wl.json" 2> "$ROOT/gpurun_out/ab_${TAG}_$wl.err" || { echo "$TAG $wl FAILED"; tail -5 "$ROOT/gpurun_out/ab_${TAG}_$wl.err"; exit 1; }
  python3 - "$ROOT/gpurun_out/ab_${TAG}_$wl.json" "$TAG" $wl <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], "ms/step %.3f" % d["ms_per_step"], {k: v for k, v in d["kernels_ms"].items()})
PY
done
