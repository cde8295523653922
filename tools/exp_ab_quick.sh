#!/bin/bash
# GPU box: quick A/B of librmd_hip variants (tools/ab_make.sh): bit-exact parity subset once per variant, then rates (update 1, one sequence and a batch
# of 8, resident / 8-bit host frames).  usage: tools/exp_ab_quick.sh <out name> <label> [<label> ...]   ("product" = the in-tree library)
set -u
export TMPDIR=/tmp
ROOT=$(pwd); NAME=$1; shift
OUT=$ROOT/gpurun_out/$NAME; mkdir -p $OUT; : > $OUT/rates.txt
for L in "$@"; do
  if [ $L = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$L.so; fi
  echo "== $L" >> $OUT/rates.txt
  if [ $L != product ] && [ -z "${NO_PARITY:-}" ]; then
    timeout 600 python -m pytest tests/test_hip_parity.py tests/test_batch.py tests/test_golden_vga.py tests/test_host_frame_modes.py -m gpu -x -q 2>&1 | tail -2 >> $OUT/rates.txt
  fi
  python tools/first_update_bench.py --b 1,8 --label $L --unit-target 2 >> $OUT/rates.txt 2>&1
  python tools/batch_bench.py --b 1,8 --passes 3 --unit-target 2 >> $OUT/rates.txt 2>&1
  python tools/batch_bench.py --b 1,8 --passes 3 --unit-target 2 --u8 >> $OUT/rates.txt 2>&1
done
cat $OUT/rates.txt
