#!/bin/bash
# one PMC pass with an arbitrary counter list. usage: tools/pmc_one.sh <tag> <steps> <counter...>
set -u
TAG=$1; STEPS=$2; shift 2
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; SUM=$ROOT/gpurun_out/summary_$TAG
mkdir -p "$OUT" "$SUM"; cd /tmp
timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_x" -- python "$ROOT/bench.py" --steps $STEPS --warmup 2 --cpu-seconds 0 > /dev/null 2> "$OUT/pmc_x.err"
echo "faults: $(grep -c 'Memory access fault' $OUT/pmc_x.err)"; tail -2 "$OUT/pmc_x.err"
cd "$ROOT"; python tools/summarize_rocprof.py "$OUT" "$SUM" "$TAG" > /dev/null 2>&1
python - <<PY
import json
c=json.load(open("$SUM/${TAG}_counters.json"))
for k,v in c.items():
    if 'seed_' in k: print(k[:56], {a:round(b,1) for a,b in v.items()})
PY
