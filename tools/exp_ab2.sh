# single-sequence rates + update 1 + batch of 8 for build variants; usage: tools/exp_ab2.sh <out> <label> ...
OUT=$1; shift; : > $OUT
for V in "$@"; do
  if [ $V = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$(pwd)/build_ab/librmd_hip_$V.so; fi
  echo "== $V" >> $OUT
  python tools/first_update_bench.py --b 1,8 --label $V >> $OUT 2>&1
  bash tools/exp_single.sh $V 2>&1 | head -4 >> $OUT
  python tools/batch_bench.py --b 8 --passes 3 2>&1 | grep flags >> $OUT
done
cat $OUT
