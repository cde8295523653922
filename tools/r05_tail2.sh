#!/bin/bash
# GPU box: graduated unit sizes, second sweep (one sequence only: a batch loses with every variant); libraries as in tools/r05_tail.sh.
set -u
export TMPDIR=/tmp
TAG=${1:-tail2}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_$TAG; mkdir -p $OUT; : > $OUT/rates.txt
for rep in 1 2 3; do
  for L in tail0 tail2 tail4 tail5; do
    export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$L.so
    echo "== $L" >> $OUT/rates.txt
    python tools/batch_bench.py --b 1 --passes 3 >> $OUT/rates.txt 2>&1
    [ $rep = 1 ] && python tools/batch_bench.py --b 1 --passes 3 --u8 >> $OUT/rates.txt 2>&1
    [ $rep = 1 ] && [ $L != tail0 ] && { echo "-- unit target 1 / 3" >> $OUT/rates.txt; python tools/batch_bench.py --b 1 --passes 3 --unit-target 1 >> $OUT/rates.txt 2>&1; python tools/batch_bench.py --b 1 --passes 3 --unit-target 3 >> $OUT/rates.txt 2>&1; }
  done
done
unset RMD_HIP_LIB
cut -c1-140 $OUT/rates.txt
