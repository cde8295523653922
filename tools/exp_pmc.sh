# per-update instruction counts of the update kernels (tools/pmc_frames.py); usage: tools/exp_pmc.sh <tag>
export TMPDIR=/tmp; TAG=${1:-x}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04_$TAG; mkdir -p $OUT
python tools/frame_stats.py $OUT/frame_stats.json > $OUT/frame_stats.log 2>&1
cd /tmp
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc -- python $ROOT/bench.py --steps 1 --warmup 0 --cpu-seconds 0 --no-extras --resident > /dev/null 2> $OUT/pmc.err
cd $ROOT
python tools/pmc_frames.py $OUT/pmc $OUT/frame_stats.json > $OUT/pmc_frames.txt 2>&1; tail -1 $OUT/pmc_frames.txt; rm -rf $OUT/pmc
