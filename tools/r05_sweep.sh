export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_h; mkdir -p $OUT
{
for L in product unitwin; do
  if [ $L = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$L.so; fi
  echo "== $L"; python tools/batch_bench.py --b 1,8 --passes 3 2>&1 | grep flags
  python tools/batch_bench.py --b 1 --passes 3 --unit-target 1 2>&1 | grep flags | sed 's/^/unit target 1: /'
  python tools/batch_bench.py --b 1 --passes 3 --unit-target 3 2>&1 | grep flags | sed 's/^/unit target 3: /'
  python tools/batch_bench.py --b 8 --passes 3 --unit-target 2 2>&1 | grep flags | sed 's/^/unit target 2: /'
done
unset RMD_HIP_LIB
for N in 32 64 128 256; do echo "== RMD_HIP_AHEAD_WGS=$N"; RMD_HIP_AHEAD_WGS=$N apps/bench_main --modes u8 --steps 5 --warmup 1 | cut -c1-330; done
for M in staged staged_ahead inplace inplace_ahead; do echo "== RMD_HIP_HOST_FRAMES=$M"; RMD_HIP_HOST_FRAMES=$M apps/bench_main --modes u8 --steps 5 --warmup 1 | cut -c1-330; done
} > $OUT/out.txt 2>&1
cat $OUT/out.txt
