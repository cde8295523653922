#!/bin/bash
# GPU box, a round's closing measurement set: the GPU suite, the per-update instruction counts of the two update kernels
# (tools/pmc_frames.py), then tools/profile_round.sh (trace, counters, 1080p denoiser, the bench lines with this run's counters installed).
set -u
export TMPDIR=/tmp
TAG=${1:-r05}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${TAG}_final; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
python tools/frame_stats.py $OUT/frame_stats.json > $OUT/frame_stats.log 2>&1
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras --resident > /dev/null 2> $OUT/pmc.err
cd $ROOT
python tools/pmc_frames.py $OUT/pmc $OUT/frame_stats.json > $OUT/pmc_frames.txt 2>&1; tail -3 $OUT/pmc_frames.txt
rm -rf $OUT/pmc
bash tools/profile_round.sh $TAG
