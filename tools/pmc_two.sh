#!/bin/bash
# two short PMC passes (instruction counts, WRITE_SIZE) + fault check. usage: tools/pmc_two.sh <tag> [steps]
set -u
TAG=${1:-pmc2}; STEPS=${2:-20}
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; SUM=$ROOT/gpurun_out/summary_$TAG
mkdir -p "$OUT" "$SUM"; cd /tmp
ARGS="--steps $STEPS --warmup 2 --cpu-seconds 0"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_insts" -- python "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_insts.err"
echo "insts pass faults: $(grep -c 'Memory access fault' $OUT/pmc_insts.err)"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_write.err"
echo "write pass faults: $(grep -c 'Memory access fault' $OUT/pmc_write.err)"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$ROOT/bench.py" $ARGS > /dev/null 2> "$OUT/pmc_fetch.err"
echo "fetch pass faults: $(grep -c 'Memory access fault' $OUT/pmc_fetch.err)"
cd "$ROOT"; python tools/summarize_rocprof.py "$OUT" "$SUM" "$TAG" > /dev/null 2>&1; ls "$SUM"
