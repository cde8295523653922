#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of the bench command, raw output under
# gpurun_out/prof_<tag>/, compact summaries (what gets committed under profiles/) under gpurun_out/summary_<tag>/.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
BENCH_ARGS=${*:-"--steps 199 --warmup 10 --cpu-seconds 0"}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/summary_$TAG
mkdir -p "$OUT" "$SUM"
cd /tmp
echo "== kernel trace + stats: bench.py $BENCH_ARGS"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" $BENCH_ARGS > "$SUM/bench_under_trace.json" 2> "$OUT/trace.err"
PMC_ARGS="--steps 30 --warmup 2 --cpu-seconds 0"
if [ "${PMC:-1}" = "1" ]; then
echo "== PMC pass 1 (instruction counts)"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_insts" -- python "$ROOT/bench.py" $PMC_ARGS > /dev/null 2> "$OUT/pmc_insts.err"
echo "== PMC pass 2 (FETCH_SIZE)"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$ROOT/bench.py" $PMC_ARGS > /dev/null 2> "$OUT/pmc_fetch.err"
echo "== PMC pass 3 (WRITE_SIZE)"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$ROOT/bench.py" $PMC_ARGS > /dev/null 2> "$OUT/pmc_write.err"
fi
cd "$ROOT"
python tools/summarize_rocprof.py "$OUT" "$SUM" "$TAG"
ls -la "$SUM"
