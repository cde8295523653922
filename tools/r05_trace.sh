#!/bin/bash
# GPU box: kernel timeline (rocprofv3 --kernel-trace) and instruction counts (--pmc, separate runs) of a batch of 8 and of one sequence, per library variant.
# usage: tools/r05_trace.sh <tag> <variant labels...>   ("product" = the in-tree library)
set -u
export TMPDIR=/tmp; ROOT=$(pwd); TAG=$1; shift
OUT=$ROOT/gpurun_out/r05_trace_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for L in "$@"; do
  if [ $L = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$L.so; fi
  for B in 8 1; do
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/t_${L}_$B -- python $ROOT/tools/batch_bench.py --b $B --passes 1 > $OUT/t_${L}_$B.log 2>&1
    { echo "== $L, batch of $B: kernel trace"; grep "batch of" $OUT/t_${L}_$B.log | tail -1; python $ROOT/tools/batch_trace.py $OUT/t_${L}_$B; } >> $OUT/summary.txt 2>&1
    rm -rf $OUT/t_${L}_$B
    timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/p_${L}_$B -- python $ROOT/tools/batch_bench.py --b $B --passes 1 > $OUT/p_${L}_$B.log 2>&1
    { echo "== $L, batch of $B: instruction counts"; python $ROOT/tools/pmc_batch.py $OUT/p_${L}_$B $B 398; } >> $OUT/summary.txt 2>&1
    rm -rf $OUT/p_${L}_$B
  done
done
unset RMD_HIP_LIB
cat $OUT/summary.txt
