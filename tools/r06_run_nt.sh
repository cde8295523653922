#!/bin/bash
# GPU box: a frame's copy into the pinned ring with streaming (non-temporal) stores (product) against memcpy (build_ab/plainmemcpy): rates, host copy time, cores.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_nt; mkdir -p $OUT
timeout 600 python -m pytest tests/test_host_frame_modes.py tests/test_full_speed.py tests/test_batch.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -1 $OUT/pytest.log
LD0=${LD_LIBRARY_PATH:-}
for S in 1920x1080:1000 1280x960:500 640x480:200; do
  SZ=${S%:*}; F=${S#*:}
  for V in product plain product plain; do
    if [ $V = product ]; then export LD_LIBRARY_PATH=$LD0; else export LD_LIBRARY_PATH=$ROOT/build_ab/plainmemcpy:$LD0; fi
    echo "== $SZ x $F u8, copy: $V"
    RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes u8 2>&1 | grep -E "wait for slot|value" | sed -E 's/.*(host copy [0-9.]+ us).*/    \1/; s/.*("value": [0-9.]+).*("us_per_update_wall": [0-9.]+).*("host_cores_busy": [0-9.]+).*/    \1 \2 \3/' | paste - - | cut -c1-200
  done
done > $OUT/rates.txt 2>&1
export LD_LIBRARY_PATH=$LD0
for V in product plain product plain; do
  if [ $V = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_plainmemcpy.so; fi
  for B in 8 16; do echo "== batch of $B u8, copy: $V"; RMD_HIP_INGEST_PROFILE=1 python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | grep -E "Mpix/s|wait for slot" | sed -E 's/.*(host copy [0-9.]+ us).*/    \1/' | cut -c1-120; done
done >> $OUT/rates.txt 2>&1
cat $OUT/rates.txt
