#!/bin/bash
# PMC pass over tools/frame_ab.py (one variant, no parity), per-dispatch table of one kernel.
# usage: tools/pmc_ab.sh <tag> <variant> <frames> <kernel-substring> <counter...>
set -u
TAG=$1; VAR=$2; FRAMES=$3; KSUB=$4; shift 4
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"; cd /tmp
timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- python "$ROOT/tools/frame_ab.py" --no-parity --variants $VAR --frames $FRAMES --passes 1 ${AB_ARGS:-} > "$OUT/run.log" 2> "$OUT/run.err"
echo "faults: $(grep -c 'Memory access fault' $OUT/run.err)"; tail -1 "$OUT/run.log"
cd "$ROOT"; python tools/pmc_dispatch.py "$OUT" "$KSUB" ${MAXROWS:-14}
