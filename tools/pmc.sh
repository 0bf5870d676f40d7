#!/bin/bash
# tools/pmc.sh OUTDIR WORKLOAD [traffic] -- rocprofv3 counter passes over a short bench run (one pass per run: the SQ / TCC
# blocks have few slots and rocprofv3 does not split a request over passes).  Run on the GPU box.
set -u
OUT=${1:-gpurun_out/pmc}; WL=${2:-x8}; ONLY=${3:-all}   # traffic: the FETCH_SIZE / WRITE_SIZE passes alone
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/bench.py" --workload $WL --steps 2 --warmup 1 --no-cpu-baseline > "$ROOT/$OUT/$name.log" 2>&1
  echo "$name rc=$?"
}
run fetch FETCH_SIZE || exit 1
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
[ "$ONLY" = traffic ] && exit 0
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU || exit 1
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS || exit 1
run grbm GRBM_GUI_ACTIVE || exit 1
