# usage: tools/exp_ab.sh <out> <label> [<label> ...]   (labels: product or a build_ab variant)
OUT=$1; shift; : > $OUT
for V in "$@"; do
  if [ $V = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$(pwd)/build_ab/librmd_hip_$V.so; fi
  echo "== $V" >> $OUT
  python tools/first_update_bench.py --b 1,8 --label $V >> $OUT 2>&1
  python tools/batch_bench.py --b 1,4,8 --passes 3 >> $OUT 2>&1
  python tools/batch_bench.py --b 8 --passes 3 --u8 >> $OUT 2>&1
done
cat $OUT
