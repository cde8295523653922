#!/bin/bash
# GPU box: copy engines in rotation (RMD_HIP_COPY_ENGINES = 2 / 3 / 4; writes gpurun_out/r06_x) -- host-frame tests, probe, rates per size and route, the 8-rank rehearsal per route (incl. 0).
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_x; mkdir -p $OUT/nranks
timeout 900 python -m pytest tests/test_host_frame_modes.py tests/test_full_speed.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
build_ab/link_probe 100 2>&1 | grep -E "A  hipMemcpyAsync \+|F  frames alt|G  frames" | grep -v "64 MB" > $OUT/probe.txt; cat $OUT/probe.txt
for S in 1920x1080:1000 1280x960:500 640x480:200; do
  SZ=${S%:*}; F=${S#*:}
  echo "== $SZ x $F resident"; apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes resident 2>&1 | grep -oE '"value": [0-9.]+|"us_per_update_wall": [0-9.]+' | paste - -
  for E in 2 3 4 2 3 4; do
    echo "== $SZ x $F u8, RMD_HIP_COPY_ENGINES=$E"
    RMD_HIP_COPY_ENGINES=$E RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes u8 2>&1 | grep -E "staged frames|wait for slot|converted by|value" | sed -E 's/.*("value": [0-9.]+).*("us_per_update_wall": [0-9.]+).*("host_cores_busy": [0-9.]+).*/    \1 \2 \3/' | cut -c1-230
  done
done > $OUT/rates.txt 2>&1
cat $OUT/rates.txt | grep -E "^==|value"
for E in 2 3 4 0; do
  ( time RMD_HIP_COPY_ENGINES=$E timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2958$E bench.py --gpus 8 --steps 3 --warmup 1 --batch-per-gpu 1 \
      > $OUT/nranks/bench_gpus8_b1_e$E.json 2> $OUT/nranks/bench_gpus8_b1_e$E.err ) 2> $OUT/nranks/bench_gpus8_b1_e$E.time
  python3 - $OUT/nranks/bench_gpus8_b1_e$E.json $E <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('route', sys.argv[2], 'ranks', d['n_gpus'], 'value', d['value'], [round(r['mpix']/r['elapsed_s']) for r in d['per_rank']])
except Exception as e: print('no line', e)
PY
done
