#!/bin/bash
# GPU box: with the frames on copy engines addressed directly (route 2) -- conversion one step ahead (default) against conversion in the frame's own setup kernel (staged).
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_v; mkdir -p $OUT
for S in 640x480:200 1280x960:500 1920x1080:1000; do
  SZ=${S%:*}; F=${S#*:}
  echo "== $SZ x $F resident"; apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes resident 2>&1 | grep -oE '"value": [0-9.]+|"us_per_update_wall": [0-9.]+' | paste - -
  for M in staged_ahead staged staged_ahead staged; do
    echo "== $SZ x $F u8, RMD_HIP_HOST_FRAMES=$M"
    RMD_HIP_HOST_FRAMES=$M RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes u8 2>&1 | grep -E "wait for slot|converted by|value" | sed -E 's/.*("value": [0-9.]+).*("us_per_update_wall": [0-9.]+).*("host_cores_busy": [0-9.]+).*/    \1 \2 \3/' | sed -E 's/.*(converted by their own.*)/    \1/' | cut -c1-200
  done
done > $OUT/rates.txt 2>&1
cat $OUT/rates.txt
