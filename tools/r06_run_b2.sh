#!/bin/bash
# GPU box: a batch's host frames staged on ONE copy engine addressed directly with the staging ring in HBM eight steps deep, against in place (the default).
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_b2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_batch.py tests/test_host_frame_modes.py tests/test_full_speed.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for B in 8 16 4 2; do
  echo "== batch of $B resident"; python tools/batch_bench.py --b $B --passes 3 2>&1 | grep -E "Mpix/s" | cut -c1-140
  for M in inplace staged inplace staged; do
    echo "== batch of $B u8 $M"; RMD_HIP_HOST_FRAMES=$M RMD_HIP_INGEST_PROFILE=1 python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | grep -E "Mpix/s|wait for slot|group 0" | cut -c1-330
  done
done > $OUT/batch.txt 2>&1
cat $OUT/batch.txt
