# GPU box: the whole GPU suite, the default bench line (headline + configs[2] / [4] + live), and the 8-rank rehearsal on the one-GPU lease exactly
# as the driver launches N > 1 (ranks share the device, control plane on gloo).   usage: tools/r06_run_g.sh <tag> [parts: tests,bench,ranks]
set -u
export TMPDIR=/tmp
TAG=${1:-g}; PARTS=${2:-tests,bench,ranks}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_$TAG; mkdir -p $OUT/nranks
if [[ $PARTS == *tests* ]]; then
  timeout 1800 python -m pytest tests -m gpu -x -q -rs --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -25 $OUT/pytest.log
fi
if [[ $PARTS == *bench* ]]; then
  ( time timeout 1500 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err ) 2> $OUT/bench_default.time; tail -3 $OUT/bench_default.time
  python3 - $OUT/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('headline',d['value'],'resident',d['resident']['value'],'float',d['float_frames']['value'],'float_other',(d.get('float_frames_not_8bit_levels') or {}).get('value'),'host',d['host_cores_busy'])
print('batched',{k:(v['resident']['value'],v['u8_host_frames']['value']) for k,v in d['batched_per_gpu'].items() if k.startswith('B=')})
for k,v in (d.get('configs') or {}).items(): print(k, {a:v.get(a) for a in ('value','u8_over_resident','host_render_and_upload_s')}, v.get('resident',{}).get('value'), v.get('u8_host_frames',{}).get('host_cores_busy'), v.get('roofline',{}).get('frac'), v.get('roofline_denoiser',{}).get('frac'), v.get('unavailable'))
l=d.get('live') or {}
print('live', l.get('value'), (l.get('publication_in_the_callback') or {}).get('ms'), (l.get('publication_off_the_update_stream') or {}).get('ms'), l.get('unavailable'))
PY
fi
if [[ $PARTS == *ranks* ]]; then
  for B in 1 2; do
    ( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2957$B bench.py --gpus 8 --steps 3 --warmup 1 --batch-per-gpu $B \
        > $OUT/nranks/bench_gpus8_b$B.json 2> $OUT/nranks/bench_gpus8_b$B.err ) 2> $OUT/nranks/bench_gpus8_b$B.time
    python3 - $OUT/nranks/bench_gpus8_b$B.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print('ranks', d['n_gpus'], 'B', d['config']['batch_per_gpu'], 'value', d['value'], 'control', d['control_plane'], 'per_rank', [(r['device'], round(r['mpix']/r['elapsed_s']), r['host_cores_busy'], r['host_submit_us_per_update']) for r in d['per_rank']])
except Exception as e:
    print('rehearsal failed', e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
  done
fi
