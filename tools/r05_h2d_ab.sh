#!/bin/bash
# GPU box: 1920x1080 host frames, the round-4 tree (build_ab/r04_tree, built from commit ca7d0d1 with tools' recipe in LAB.md) against this
# tree ON THE SAME BOX: is the lower configs[4] host-frame figure of round 5 the code or the box's host link?   usage: tools/r05_h2d_ab.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-h2dab}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_$TAG; mkdir -p $OUT
{ for rep in 1 2; do
    for T in r04 r05; do
      if [ $T = r04 ]; then D=$ROOT/build_ab/r04_tree; else D=$ROOT; fi
      echo "== $T tree, 1920x1080 x 400 frames, 8-bit host frames (bench.py --no-extras)"
      ( cd $D && python bench.py --size 1920x1080 --frames 400 --steps 1 --warmup 1 --cpu-seconds 0 --batch "" --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'us/update', d['roofline']['avg_launch_us'], 'host', d.get('host_cores_busy'))" )
    done
  done
  echo "== copy engine: pinned -> device, 2 MB (tools/ubench/copy_rate)"; ( cd $ROOT/tools/ubench && ls; test -x ./copy_rate && ./copy_rate 2>&1 | tail -12 )
} > $OUT/h2d_ab.txt 2>&1
cat $OUT/h2d_ab.txt
