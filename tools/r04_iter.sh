#!/bin/bash
# GPU box: one iteration of the round-4 kernel work -- the GPU suite (bit-exact parity), the per-update instruction counts of the two update
# kernels, and the rates that the instruction counts are supposed to move.  usage: tools/r04_iter.sh <tag> [quick]
set -u
export TMPDIR=/tmp
TAG=${1:-x}; MODE=${2:-full}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04_$TAG; mkdir -p $OUT
if [ $MODE = quick ]; then
  timeout 900 python -m pytest tests/test_hip_parity.py tests/test_batch.py tests/test_golden_vga.py tests/test_parity_glibc.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1
else
  timeout 1200 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest.log 2>&1
fi
echo "pytest rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
python tools/frame_stats.py $OUT/frame_stats.json > $OUT/frame_stats.log 2>&1
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras --resident > /dev/null 2> $OUT/pmc.err
cd $ROOT
python tools/pmc_frames.py $OUT/pmc $OUT/frame_stats.json > $OUT/pmc_frames.txt 2>&1; tail -1 $OUT/pmc_frames.txt
rm -rf $OUT/pmc
python tools/first_update_bench.py --b 1,8 --label $TAG > $OUT/rates.txt 2>&1
python tools/batch_bench.py --b 1,4,8 --passes 3 >> $OUT/rates.txt 2>&1
python tools/batch_bench.py --b 1,8 --passes 3 --u8 >> $OUT/rates.txt 2>&1
cat $OUT/rates.txt
