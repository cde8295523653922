#!/bin/bash
# On the GPU box: bench every variant built by tools/ab_make.sh.  usage: tools/ab_run.sh "<bench args>" <label> [<label> ...]
ARGS=$1; shift
cp rpg_open_remode_amd/librmd_hip.so /tmp/librmd_hip_orig.so
for L in "$@"; do
  cp build_ab/librmd_hip_$L.so rpg_open_remode_amd/librmd_hip.so
  for rep in 1 2; do
  python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', '| Mpix/s', d['value'], '| ms/step', d['ms_per_step'], '| update pipeline us', d['roofline']['avg_launch_us'])
"
  done
done
cp /tmp/librmd_hip_orig.so rpg_open_remode_amd/librmd_hip.so
