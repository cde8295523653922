set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_o; mkdir -p $OUT
for rep in 1 2; do
  for B in 8 16; do
    echo "== batch of $B resident"; python tools/batch_bench.py --b $B --passes 3 2>&1 | grep Mpix | cut -c1-150
    echo "== batch of $B u8 in place (default)"; python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | grep "Mpix" | cut -c1-200
    echo "== batch of $B u8, conversion kernel on a high-priority stream"; RMD_HIP_HOST_FRAMES=inplace_ahead python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | grep "Mpix" | cut -c1-200
  done
done > $OUT/rates.txt 2>&1
cat $OUT/rates.txt
