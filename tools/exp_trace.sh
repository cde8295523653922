# kernel timeline of a batch of 8 (and of one sequence) under rocprofv3 --kernel-trace; usage: tools/exp_trace.sh [groups ...]
export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04_trace; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for G in ${@:-3}; do
export RMD_HIP_BATCH_GROUPS=$G
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/g$G -- python $ROOT/tools/batch_bench.py --b 8 --passes 1 > $OUT/g$G.log 2>&1
python $ROOT/tools/batch_trace.py $OUT/g$G > $OUT/g${G}_summary.txt; echo "== batch of 8, $G stream groups"; grep "batch of" $OUT/g$G.log | tail -1; cat $OUT/g${G}_summary.txt
rm -rf $OUT/g$G
done
unset RMD_HIP_BATCH_GROUPS
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/b1 -- python $ROOT/tools/batch_bench.py --b 1 --passes 1 > $OUT/b1.log 2>&1
python $ROOT/tools/batch_trace.py $OUT/b1 > $OUT/b1_summary.txt; echo "== one sequence"; grep "batch of" $OUT/b1.log | tail -1; cat $OUT/b1_summary.txt; rm -rf $OUT/b1
