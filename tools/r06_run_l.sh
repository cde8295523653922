# GPU box: after the ring depth went back to 4 / 3 -- the stall hunt with the defaults, the host-frame tests, the bench lines.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_l; mkdir -p $OUT
timeout 900 python -m pytest tests/test_host_frame_modes.py tests/test_full_speed.py tests/test_batch.py tests/test_publish_async.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
{ for rep in 1 2 3 4; do echo "== defaults (ring 4, one copy stream)"; RMD_HIP_INGEST_PROFILE=1 python tools/r06_stall.py 80 2>&1 | cut -c1-700; done; } > $OUT/stall_defaults.txt 2>&1; cat $OUT/stall_defaults.txt
timeout 1200 python bench.py > $OUT/r06_bench_default.json 2> $OUT/bench_default.err
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extras > $OUT/r06_bench_steps20.json 2> $OUT/bench_steps20.err
python3 - $OUT <<'PY'
import json,sys
for f in ('r06_bench_default.json','r06_bench_steps20.json'):
    d=json.loads(open(sys.argv[1]+'/'+f).read().strip().splitlines()[-1])
    print(f, d['value'], d['pass_ms'], d['host_cores_busy'], (d.get('resident') or {}).get('value'))
    if d.get('batched_per_gpu'): print({k:(v['resident']['value'],v['u8_host_frames']['value']) for k,v in d['batched_per_gpu'].items() if k.startswith('B=')})
    if d.get('configs'): print({k:(v['value'],v['resident']['value'],v['u8_over_resident']) for k,v in d['configs'].items()})
    if d.get('live'): print('live', d['live']['publication_in_the_callback']['ms'], d['live']['publication_off_the_update_stream']['ms'])
PY
