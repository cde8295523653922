#!/bin/bash
# GPU box: the ring wait that keeps waiting through long kernels: host-frame tests, the stall account at 1920x1080 / VGA (waits that gave up must read 0), default bench.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r06_q; mkdir -p $OUT
timeout 900 python -m pytest tests/test_host_frame_modes.py tests/test_full_speed.py tests/test_batch.py tests/test_concurrency.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for S in 1920x1080:300 1280x960:300 640x480:200; do
  SZ=${S%:*}; F=${S#*:}
  for M in default staged; do
    echo "== $SZ x $F u8 $M"
    if [ $M = default ]; then RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes u8 2>&1 | grep -E "rmd_hip|value" | cut -c1-260
    else RMD_HIP_HOST_FRAMES=staged RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes u8 2>&1 | grep -E "rmd_hip|value" | cut -c1-260; fi
  done
done > $OUT/stall.txt 2>&1
cat $OUT/stall.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
