export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04_trace; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for G in 1 2 3; do
export RMD_HIP_BATCH_GROUPS=$G
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/g$G -- python $ROOT/tools/batch_bench.py --b 8 --passes 1 > $OUT/g$G.log 2>&1
python $ROOT/tools/batch_trace.py $OUT/g$G > $OUT/g${G}_summary.txt; echo "== groups $G"; cat $OUT/g$G.log | tail -1; cat $OUT/g${G}_summary.txt
rm -rf $OUT/g$G
done
