#!/bin/bash
# GPU box: one iteration of the round-5 kernel work.  usage: tools/r05_iter.sh <tag> <tests: full|quick|none> <variant labels...>   ("product" = the in-tree library)
# Per variant: (product only) the GPU suite; update 1, one sequence and a batch of 8 with resident / 8-bit host frames; per-update search timeline (brief).
set -u
export TMPDIR=/tmp
TAG=${1:-x}; TESTS=${2:-quick}; shift; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_$TAG; mkdir -p $OUT; : > $OUT/rates.txt
if [ $TESTS = full ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -8 $OUT/pytest.log
elif [ $TESTS = quick ]; then
  timeout 900 python -m pytest tests/test_hip_parity.py tests/test_batch.py tests/test_golden_vga.py tests/test_parity_glibc.py tests/test_host_frame_modes.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -8 $OUT/pytest.log
fi
for L in "$@"; do
  if [ $L = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$L.so; fi
  echo "== $L" >> $OUT/rates.txt
  python tools/first_update_bench.py --b 1,8 --label $L --unit-target 2 >> $OUT/rates.txt 2>&1
  python tools/batch_bench.py --b 1,8 --passes 3 >> $OUT/rates.txt 2>&1
  python tools/batch_bench.py --b 1,8 --passes 3 --u8 >> $OUT/rates.txt 2>&1
  python tools/search_timeline.py --brief > $OUT/timeline_$L.txt 2>&1
done
unset RMD_HIP_LIB
cat $OUT/rates.txt
