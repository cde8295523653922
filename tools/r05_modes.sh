#!/bin/bash
# GPU box: how a host frame gets from the pinned ring to the current image (RMD_HIP_HOST_FRAMES), per frame size.   usage: tools/r05_modes.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-modes}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_$TAG; mkdir -p $OUT
{ for S in 1920x1080 1280x960 960x720; do
    for M in staged_ahead staged inplace inplace_ahead; do
      echo "== $S x 300 frames, RMD_HIP_HOST_FRAMES=$M"
      RMD_HIP_HOST_FRAMES=$M RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $S --frames 300 --steps 2 --warmup 1 --modes u8 2>&1 | python3 -c "
import sys,re
for l in sys.stdin:
    l=l.strip()
    if l.startswith('[rmd_hip'): print('   ',l)
    elif l.startswith('{'): print('   ', ', '.join(re.findall(r'\"value\": [\d.]+|\"us_per_update_wall\": [\d.]+|\"host_cores_busy\": [\d.]+',l)))"
    done
    echo "== $S x 300 frames, resident"; apps/bench_main --size $S --frames 300 --steps 2 --warmup 1 --modes resident 2>&1 | python3 -c "
import sys,re
for l in sys.stdin:
    if l.startswith('{'): print('   ', ', '.join(re.findall(r'\"value\": [\d.]+|\"us_per_update_wall\": [\d.]+',l)))"
  done
} > $OUT/modes.txt 2>&1
cat $OUT/modes.txt
