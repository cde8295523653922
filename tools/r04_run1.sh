#!/bin/bash
# GPU box, round 4, first call: the GPU suite on the new parity tests, the per-update instruction counts of the round-3 kernels (the
# baseline of the VALU work), the FMA-contraction diagnostic, the live bench with the device colouring.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04a; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
python tools/frame_stats.py $OUT/frame_stats.json > $OUT/frame_stats.log 2>&1
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras --resident > /dev/null 2> $OUT/pmc.err
cd $ROOT
python tools/pmc_frames.py $OUT/pmc $OUT/frame_stats.json > $OUT/pmc_frames.txt 2>&1; tail -3 $OUT/pmc_frames.txt
rm -rf $OUT/pmc/*/*agent_info.csv
for V in product fma; do
  if [ $V = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$V.so; fi
  python tools/first_update_bench.py --b 1,8 --label $V >> $OUT/fma_ab.txt 2>&1
  python tools/batch_bench.py --b 1,8 --passes 3 >> $OUT/fma_ab.txt 2>&1
done
unset RMD_HIP_LIB
cat $OUT/fma_ab.txt
python tools/live_bench.py --breakdown > $OUT/live.txt 2>&1; tail -12 $OUT/live.txt
