#!/bin/bash
# GPU box: where the 8-bit-host-frame path spends its extra 1.3 us per update against resident frames -- per-kernel averages (rocprofv3 --kernel-trace
# --stats) of apps/bench_main with frames resident, with the default host-frame mode (staged + converted one step ahead inside the search kernel) and
# with the conversion left to the frame's own setup kernel (RMD_HIP_HOST_FRAMES=staged: no extra workgroups in the search kernel).   usage: tools/r05_u8gap.sh <tag>
set -u
export TMPDIR=/tmp; ROOT=$(pwd); TAG=${1:-u8gap}; OUT=$ROOT/gpurun_out/r05_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp
run() {  # label, env, modes
  local L=$1 E=$2 M=$3
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$L -- $ROOT/apps/bench_main --modes $M --steps 3 --warmup 1 > $OUT/$L.log 2>&1
  { echo "== $L ($E, --modes $M)"; grep '"value"' $OUT/$L.log | python3 -c "
import sys,re
for l in sys.stdin: print('   ', ', '.join(re.findall(r'\"value\": [\d.]+|\"us_per_update_wall\": [\d.]+|\"us_per_update_device\": [\d.]+',l)))"
    python3 - $OUT/$L <<'PY'
import sys,glob,csv
for f in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r.get('Name','')
        if 'seed_search' in n or 'seed_setup' in n:
            print('   ', n[:40].ljust(40), 'calls', r.get('Calls'), 'avg_us', round(float(r.get('AverageNs',0))/1e3,2), 'min', round(float(r.get('MinNs',0))/1e3,2), 'max', round(float(r.get('MaxNs',0))/1e3,2))
PY
  } >> $OUT/summary.txt 2>&1
  rm -rf $OUT/$L
}
for rep in 1 2; do
  run resident_$rep "A=1" resident
  run staged_ahead_$rep "RMD_HIP_HOST_FRAMES=staged_ahead" u8
  run staged_$rep "RMD_HIP_HOST_FRAMES=staged" u8
  run ahead32_$rep "RMD_HIP_AHEAD_WGS=32" u8
done
cat $OUT/summary.txt
