#!/bin/bash
# GPU box: frames the caller keeps in pinned host memory (rmd_hip_seeds_update_u8_pinned) against the copying update_u8 and resident frames: rate and host cores.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_p2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_pinned_frames.py tests/test_cpp_facade.py tests/test_abi.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for S in 640x480:200 1280x960:500 1920x1080:1000; do
  SZ=${S%:*}; F=${S#*:}
  for rep in 1 2; do
    echo "== $SZ x $F"
    RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes resident,u8,pinned 2>&1 | grep -E "wait for slot|value" | sed -E 's/.*("mode": "[a-z0-9]+").*("value": [0-9.]+).*("us_per_update_wall": [0-9.]+).*("host_cores_busy": [0-9.]+).*("host_submit_us_per_update": [0-9.]+).*/    \1 \2 \3 \4 \5/' | cut -c1-220
  done
done > $OUT/rates.txt 2>&1
cat $OUT/rates.txt
