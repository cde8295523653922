#!/bin/bash
# A/B on the GPU box: rebuild librmd_hip.so with extra hipcc flags and run the bench (JSON value + per-kernel event time).
# usage: tools/ab_build_bench.sh "<label>" "<extra flags>" [bench args]
LABEL=$1; FLAGS=$2; shift 2
ARGS=${*:-"--steps 199 --warmup 10 --cpu-seconds 0"}
python - <<PY
from rpg_open_remode_amd import build
build.build_hip(force=True, extra_flags="$FLAGS".split())
PY
python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$LABEL', '| Mpix/s', d['value'], '| ms/step', d['ms_per_step'], '| update pipeline us', d['roofline']['avg_launch_us'])
"
