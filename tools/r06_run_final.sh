#!/bin/bash
# GPU box: the round's last check -- the whole GPU suite, smoke(), the default bench line and the driver's --steps 20 line.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_final; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q -rs --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
( time timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
timeout 900 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extras > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python3 - $OUT <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+'/bench_default.json').read().strip().splitlines()[-1])
print('headline',d['value'],d['pass_ms'],'resident',d['resident']['value'],'pinned',(d.get('caller_pinned_frames') or {}).get('value'),(d.get('caller_pinned_frames') or {}).get('host_cores_busy_incl_warmup_pass'),'route',d.get('host_frame_route'))
print({k:(v['resident']['value'],v['u8_host_frames']['value']) for k,v in d['batched_per_gpu'].items() if k.startswith('B=')})
print({k:(v['value'],v['resident']['value'],v['u8_over_resident'],(v.get('caller_pinned_frames') or {}).get('value'),(v.get('caller_pinned_frames') or {}).get('host_cores_busy')) for k,v in d['configs'].items()}, 'live', d['live']['publication_in_the_callback']['ms'], d['live']['publication_off_the_update_stream']['ms'])
e=json.loads(open(sys.argv[1]+'/bench_steps20.json').read().strip().splitlines()[-1])
print('steps20', e['value'], e['pass_ms'])
PY
