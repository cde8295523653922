#!/bin/bash
# GPU box: batches with the new default (staged on one engine with eight staging buffers while a step is at most 3 MB) -- tests, then batch sizes against in place.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_b3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_batch.py tests/test_host_frame_modes.py tests/test_full_speed.py tests/test_concurrency.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for B in 2 4 8 12 16; do
  echo "== batch of $B resident"; python tools/batch_bench.py --b $B --passes 3 2>&1 | grep -E "Mpix/s" | cut -c1-120
  for M in default inplace default inplace; do
    echo "== batch of $B u8 $M"
    if [ $M = default ]; then RMD_HIP_INGEST_PROFILE=1 python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | grep -E "Mpix/s|group 0" | cut -c1-200
    else RMD_HIP_HOST_FRAMES=$M RMD_HIP_INGEST_PROFILE=1 python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | grep -E "Mpix/s|group 0" | cut -c1-200; fi
  done
done > $OUT/batch.txt 2>&1
cat $OUT/batch.txt
