#!/bin/bash
# GPU box: graduated unit sizes -- libraries build_ab/librmd_hip_tailN.so, each built (tools/ab_make.sh) from a variant of unit_tail_shift()
# in csrc/rmd_frame.hpp (N = 0: no graduation; the variants are listed in profiles/r05_ab_unit_tail.txt), alternating on one box.
# usage: tools/r05_tail.sh <tag>
set -u
export TMPDIR=/tmp
TAG=${1:-tail}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_$TAG; mkdir -p $OUT; : > $OUT/rates.txt
for rep in 1 2; do
  for L in tail0 tail1 tail2 tail3; do
    export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$L.so
    echo "== $L" >> $OUT/rates.txt
    python tools/batch_bench.py --b 1,8 --passes 3 >> $OUT/rates.txt 2>&1
    [ $rep = 1 ] && python tools/first_update_bench.py --b 1,8 --label $L >> $OUT/rates.txt 2>&1
    [ $rep = 1 ] && python tools/search_timeline.py --brief > $OUT/timeline_$L.txt 2>&1
  done
done
unset RMD_HIP_LIB
grep -v "^\[flags 6\] batch of 8 (8-bit" $OUT/rates.txt | cut -c1-150
