#!/bin/bash
# tools/shares.sh -- what ONE GPU does with the N-th of configs[1..3] (N = 2, 4, 8): short bench lines into gpurun_out/share_<wl>_<N>.json.
# Compute only: nothing inter-GPU is in these numbers.  Run on the GPU box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for N in 2 4 8; do
  for wl in x8 x14 y12 y14; do
    timeout -k 10 200 python3 "$ROOT/bench.py" --workload $wl --only --no-cpu-baseline --no-e2e --steps 5 --warmup 2 \
      --records $((170000 / N)) --bases $((400000000 / N)) --reads $((100000000 / N)) > "$ROOT/gpurun_out/share_${wl}_$N.json" 2> "$ROOT/gpurun_out/share_${wl}_$N.err" \
      || { echo "share $wl $N FAILED"; tail -3 "$ROOT/gpurun_out/share_${wl}_$N.err"; exit 1; }
    python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], 'ms/step %.3f' % d['ms_per_step'])" "$ROOT/gpurun_out/share_${wl}_$N.json" $wl $N
  done
done
