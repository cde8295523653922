#!/bin/bash
# GPU box: the closing check of the copy-engine work -- the whole GPU suite, smoke(), rates per size with the defaults (route 2 on engines 0x1 / 0x4, conversion in the
# frame's own setup kernel) against route 0 + staged_ahead (the defaults until now), the default bench line, the 8-rank rehearsal.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_w; mkdir -p $OUT/nranks
timeout 1800 python -m pytest tests -m gpu -x -q -rs --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -14 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
for S in 640x480:200 1280x960:500 1920x1080:1000; do
  SZ=${S%:*}; F=${S#*:}
  echo "== $SZ x $F resident"; apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes resident 2>&1 | grep -oE '"value": [0-9.]+|"us_per_update_wall": [0-9.]+' | paste - -
  for V in new old new old; do
    echo "== $SZ x $F u8, $V defaults"
    if [ $V = new ]; then E="A=1"; else E="RMD_HIP_COPY_ENGINES=0"; fi
    env $E RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes u8 2>&1 | grep -E "staged frames|converted by|value" | sed -E 's/.*("value": [0-9.]+).*("us_per_update_wall": [0-9.]+).*("host_cores_busy": [0-9.]+).*/    \1 \2 \3/' | sed -E 's/.*(converted by their own.*)/    \1/' | cut -c1-200
  done
done > $OUT/rates.txt 2>&1
cat $OUT/rates.txt
( time timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
python3 - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('headline',d['value'],d['pass_ms'],'resident',d['resident']['value'],'route',d.get('host_frame_route'))
print({k:(v['resident']['value'],v['u8_host_frames']['value']) for k,v in d['batched_per_gpu'].items() if k.startswith('B=')})
print({k:(v['value'],v['resident']['value'],v['u8_over_resident'],v['u8_host_frames'].get('frames_on_copy_engines_addressed_directly')) for k,v in d['configs'].items()}, 'live', d['live']['publication_in_the_callback']['ms'], d['live']['publication_off_the_update_stream']['ms'])
PY
for B in 1 2; do
  ( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2959$B bench.py --gpus 8 --steps 3 --warmup 1 --batch-per-gpu $B \
      > $OUT/nranks/bench_gpus8_b$B.json 2> $OUT/nranks/bench_gpus8_b$B.err ) 2> $OUT/nranks/bench_gpus8_b$B.time
  python3 - $OUT/nranks/bench_gpus8_b$B.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('ranks', d['n_gpus'], 'B', d['config']['batch_per_gpu'], 'value', d['value'], [round(r['mpix']/r['elapsed_s']) for r in d['per_rank']])
except Exception as e: print('no line', e)
PY
done
