# GPU box: large staged frames over the host link from both sides (RMD_HIP_INPLACE_PERCENT): parity in every host-frame mode, then the rate per share.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_i; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_host_frame_modes.py tests/test_full_speed.py tests/test_batch.py tests/test_full_size.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
brief() { python3 -c "
import sys,re
for l in sys.stdin:
    l=l.strip()
    if l.startswith('[rmd_hip'): print('   ',l[:330])
    elif l.startswith('{'): print('   ', ', '.join(re.findall(r'\"mode\": \"\w+\"|\"value\": [\d.]+|\"us_per_update_wall\": [\d.]+|\"host_cores_busy\": [\d.]+',l)))"; }
{ for S in 1920x1080:600 1280x960:500; do
    SZ=${S%:*}; F=${S#*:}
    echo "== $SZ x $F resident"; apps/bench_main --size $SZ --frames $F --steps 2 --warmup 1 --modes resident 2>&1 | brief
    for P in 0 40 60 80 100; do
      echo "== $SZ x $F u8, RMD_HIP_INPLACE_PERCENT=$P"; RMD_HIP_INPLACE_PERCENT=$P RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 2 --warmup 1 --modes u8 2>&1 | brief
    done
    echo "== $SZ x $F float (not packed: RMD_HIP_FLOAT_AS_BYTES=0), RMD_HIP_INPLACE_PERCENT=0 / 60"
    RMD_HIP_FLOAT_AS_BYTES=0 RMD_HIP_INPLACE_PERCENT=0 apps/bench_main --size $SZ --frames 200 --steps 2 --warmup 1 --modes float 2>&1 | brief
    RMD_HIP_FLOAT_AS_BYTES=0 RMD_HIP_INPLACE_PERCENT=60 apps/bench_main --size $SZ --frames 200 --steps 2 --warmup 1 --modes float 2>&1 | brief
  done
  echo "== 640x480 float not packed (1.2 MB frames), RMD_HIP_INPLACE_PERCENT=0 / 60"
  RMD_HIP_FLOAT_AS_BYTES=0 RMD_HIP_INPLACE_PERCENT=0 apps/bench_main --modes float --steps 3 --warmup 1 2>&1 | brief
  RMD_HIP_FLOAT_AS_BYTES=0 RMD_HIP_INPLACE_PERCENT=60 apps/bench_main --modes float --steps 3 --warmup 1 2>&1 | brief
} > $OUT/rates.txt 2>&1
cat $OUT/rates.txt
