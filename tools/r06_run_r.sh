#!/bin/bash
# GPU box: staged host frames on copy engines addressed directly (RMD_HIP_COPY_ENGINES = 0 / 1 / 2): host-frame tests, then rates per size and route.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06_r; mkdir -p $OUT
timeout 900 python -m pytest tests/test_host_frame_modes.py tests/test_full_speed.py tests/test_abi.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
build_ab/link_probe 100 2>&1 | grep -E "A  hipMemcpyAsync|F  frames" | grep -v "64 MB" > $OUT/probe.txt
for S in 1920x1080:300 1920x1080:1000 1280x960:500 640x480:200; do
  SZ=${S%:*}; F=${S#*:}
  echo "== $SZ x $F resident"; apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes resident 2>&1 | grep -oE '"value": [0-9.]+|"us_per_update_wall": [0-9.]+' | paste - -
  for E in 0 1 2 0 2; do
    echo "== $SZ x $F u8, RMD_HIP_COPY_ENGINES=$E"
    RMD_HIP_COPY_ENGINES=$E RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes u8 2>&1 | grep -E "staged frames|wait for slot|converted by|value" | sed -E 's/.*("value": [0-9.]+).*("us_per_update_wall": [0-9.]+).*("host_cores_busy": [0-9.]+).*/    \1 \2 \3/' | cut -c1-230
  done
done > $OUT/rates.txt 2>&1
cat $OUT/probe.txt; cat $OUT/rates.txt
