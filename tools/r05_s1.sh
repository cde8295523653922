#!/bin/bash
# GPU box: the round's last kernel iteration (first units taken where they lie, argument-segment lines requested together, reservation first)
# against the library of the closing measurement set before it (build_ab/librmd_hip_r5final.so), and the host's ring-slot wait per mode at
# 1920x1080 and 640x480.   usage: tools/r05_s1.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-s1}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_batch.py tests/test_golden_vga.py tests/test_host_frame_modes.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
: > $OUT/rates.txt
for L in product r5final product r5final; do
  if [ $L = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$L.so; fi
  echo "== $L" >> $OUT/rates.txt
  python tools/batch_bench.py --b 1,8 --passes 3 >> $OUT/rates.txt 2>&1
  python tools/batch_bench.py --b 1 --passes 3 --u8 >> $OUT/rates.txt 2>&1
done
unset RMD_HIP_LIB
python tools/first_update_bench.py --b 1,8 --label product >> $OUT/rates.txt 2>&1
python tools/search_timeline.py --brief > $OUT/timeline_product.txt 2>&1
cat $OUT/rates.txt
{ for W in 1 0; do  # (the run recorded in profiles/r05_host_wait_modes.txt also had an adaptive variant, since dropped)
    echo "== RMD_HIP_HOST_WAIT=$W, 1920x1080 x 300 frames, 8-bit host frames (apps/bench_main)"; RMD_HIP_HOST_WAIT=$W RMD_HIP_INGEST_PROFILE=1 apps/bench_main --size 1920x1080 --frames 300 --steps 2 --warmup 1 --modes u8 2>&1 | cut -c1-700
    echo "== RMD_HIP_HOST_WAIT=$W, 640x480 x 200 frames"; RMD_HIP_HOST_WAIT=$W apps/bench_main --steps 5 --warmup 1 --modes u8 2>&1 | cut -c1-700
  done
  echo "== RMD_HIP_HOST_WAIT=1, 1280x960 x 300 frames"; RMD_HIP_HOST_WAIT=1 apps/bench_main --size 1280x960 --frames 300 --steps 2 --warmup 1 --modes u8 2>&1 | cut -c1-700
  echo "== RMD_HIP_HOST_WAIT=0, 1280x960 x 300 frames"; RMD_HIP_HOST_WAIT=0 apps/bench_main --size 1280x960 --frames 300 --steps 2 --warmup 1 --modes u8 2>&1 | cut -c1-700
} > $OUT/host_wait.txt 2>&1
cat $OUT/host_wait.txt
