#!/bin/bash
# PMC counters of the heavy early frames (bench.py --steps N): two SQ passes. usage: tools/pmc_search.sh <tag> [steps]
set -u
TAG=${1:-pmc}; STEPS=${2:-12}
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; SUM=$ROOT/gpurun_out/summary_$TAG
mkdir -p "$OUT" "$SUM"; cd /tmp
ARGS="--steps $STEPS --warmup 1 --cpu-seconds 0"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq" -- python "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_sq.err"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_wait" -- python "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_wait.err"
cd "$ROOT"; python tools/summarize_rocprof.py "$OUT" "$SUM" "$TAG" | grep -A22 "seed_search"
