# GPU box: the last check of the round -- the whole GPU suite, smoke(), the default bench line.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_p; mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -x -q -rs --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -16 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( time timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
python3 - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('headline',d['value'],d['pass_ms'],'resident',d['resident']['value'],'valu',d['roofline_valu'].get('frac'),d['roofline_valu'].get('stale'))
print({k:(v['resident']['value'],v['u8_host_frames']['value']) for k,v in d['batched_per_gpu'].items() if k.startswith('B=')})
print({k:(v['value'],v['resident']['value'],v['u8_over_resident']) for k,v in d['configs'].items()}, 'live', d['live']['publication_in_the_callback']['ms'], d['live']['publication_off_the_update_stream']['ms'])
PY
