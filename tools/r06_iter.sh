#!/bin/bash
# GPU box: one iteration of the round-6 host-frame work.   usage: tools/r06_iter.sh <tag> <tests: full|quick|host|none> [parts: big,vga,batch] [variant labels... ("product" = in-tree)]
# Per variant, untraced, with RMD_HIP_INGEST_PROFILE=1 (host time per frame, lead of the caller over the device, frames the setup kernel converted /
# waited for): apps/bench_main at 1920x1080, 1280x960 (300 frames) and 640x480 (200) with frames resident, as 8-bit host frames in the default mode
# and with RMD_HIP_HOST_FRAMES=staged; tools/batch_bench.py for batches of 8 and 16 (resident, in place, staged).
set -u
export TMPDIR=/tmp
TAG=${1:-x}; TESTS=${2:-quick}; PARTS=${3:-big,vga,batch}; shift; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_$TAG; mkdir -p $OUT; : > $OUT/rates.txt
if [ $TESTS = full ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -8 $OUT/pytest.log
elif [ $TESTS = quick ]; then
  timeout 900 python -m pytest tests/test_hip_parity.py tests/test_batch.py tests/test_golden_vga.py tests/test_host_frame_modes.py tests/test_full_speed.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -8 $OUT/pytest.log
elif [ $TESTS = host ]; then
  timeout 900 python -m pytest tests/test_host_frame_modes.py tests/test_full_speed.py tests/test_batch.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -8 $OUT/pytest.log
fi
brief() { python3 -c "
import sys,re
for l in sys.stdin:
    l=l.strip()
    if l.startswith('[rmd_hip'): print('   ',l)
    elif l.startswith('{'): print('   ', ', '.join(re.findall(r'\"mode\": \"\w+\"|\"value\": [\d.]+|\"us_per_update_wall\": [\d.]+|\"host_cores_busy\": [\d.]+|\"host_submit_us_per_update\": [\d.]+',l)))
    elif 'Mpix/s' in l: print('   ', l[:150])"; }
[ $# -eq 0 ] && set -- product
LD0=${LD_LIBRARY_PATH:-}
for L in "$@"; do
  # (python: RMD_HIP_LIB; apps/bench_main: a directory build_ab/<label>/ with that variant as librmd_hip.so, found first through LD_LIBRARY_PATH)
  if [ $L = product ]; then unset RMD_HIP_LIB; export LD_LIBRARY_PATH=${LD0:-}; else export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$L.so; export LD_LIBRARY_PATH=$ROOT/build_ab/$L:${LD0:-}; fi
  { echo "==== $L"
    for S in 1920x1080:300 1280x960:300 640x480:200; do
      SZ=${S%:*}; F=${S#*:}
      case $SZ in 640x480) [[ $PARTS == *vga* ]] || continue;; *) [[ $PARTS == *big* ]] || continue;; esac
      echo "== $SZ x $F resident"; apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes resident 2>&1 | brief
      for M in default staged; do
        echo "== $SZ x $F u8, host frames: $M"
        if [ $M = default ]; then RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes u8 2>&1 | brief
        else RMD_HIP_HOST_FRAMES=$M RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size $SZ --frames $F --steps 3 --warmup 1 --modes u8 2>&1 | brief; fi
      done
    done
    if [[ $PARTS == *batch* ]]; then
      for B in 8 16; do
        echo "== batch of $B resident"; python tools/batch_bench.py --b $B --passes 3 2>&1 | brief
        echo "== batch of $B u8 (default: in place)"; RMD_HIP_INGEST_PROFILE=1 python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | brief
        echo "== batch of $B u8 staged"; RMD_HIP_HOST_FRAMES=staged RMD_HIP_INGEST_PROFILE=1 python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | brief
      done
    fi
  } >> $OUT/rates.txt 2>&1
done
unset RMD_HIP_LIB; export LD_LIBRARY_PATH=$LD0
cat $OUT/rates.txt
