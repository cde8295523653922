# GPU box: the bringers of a frame one step ahead at the head / at the tail of the search grid (RMD_HIP_AHEAD_LAST), with and without split frames.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_j; mkdir -p $OUT
brief() { python3 -c "
import sys,re
for l in sys.stdin:
    l=l.strip()
    if l.startswith('[rmd_hip ingest] frames'): print('   ',l[150:330])
    elif l.startswith('{'): print('   ', ', '.join(re.findall(r'\"mode\": \"\w+\"|\"value\": [\d.]+|\"us_per_update_wall\": [\d.]+|\"host_cores_busy\": [\d.]+',l)))"; }
{ for rep in 1 2; do
  for S in 1920x1080:600 1280x960:500 640x480:200; do
    SZ=${S%:*}; F=${S#*:}
    echo "== $SZ x $F resident"; apps/bench_main --size $SZ --frames $F --steps 2 --warmup 1 --modes resident 2>&1 | brief
    for L in 0 1; do for P in 0 40 60; do
      [ $SZ = 640x480 ] && [ $P != 0 ] && continue
      echo "== $SZ x $F u8, AHEAD_LAST=$L INPLACE_PERCENT=$P"; RMD_HIP_AHEAD_LAST=$L RMD_HIP_INPLACE_PERCENT=$P RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 2 --warmup 1 --modes u8 2>&1 | brief
    done; done
  done
done } > $OUT/rates.txt 2>&1
cat $OUT/rates.txt
