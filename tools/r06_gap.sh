#!/bin/bash
# GPU box: timelines (rocprofv3 --kernel-trace --memory-copy-trace, no counters) of the host-frame paths that lose against resident frames:
# one 1920x1080 / 1280x960 sequence (apps/bench_main) and a batch of 8 / 16 640x480 sequences (tools/batch_bench.py), each with frames resident
# and as 8-bit host frames, condensed by tools/r06_timeline.py.   usage: tools/r06_gap.sh <tag> [parts: big,batch]
set -u
export TMPDIR=/tmp; ROOT=$(pwd); TAG=${1:-gap}; PARTS=${2:-big,batch}; OUT=$ROOT/gpurun_out/r06_$TAG; rm -rf $OUT; mkdir -p $OUT; cd /tmp
one() {  # label, env, frame bytes, command...
  local L=$1 E=$2 FB=$3; shift 3
  env $E RMD_HIP_INGEST_PROFILE=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$L -- "$@" > $OUT/$L.log 2>&1
  { echo "== $L ($E) $*"; grep -h '"value"\|rmd_hip ingest\|Mpix' $OUT/$L.log | cut -c1-400; python3 $ROOT/tools/r06_timeline.py $OUT/$L 150 $FB; } >> $OUT/summary.txt 2>&1
  rm -rf $OUT/$L
}
if [[ $PARTS == *big* ]]; then
  for S in 1920x1080 1280x960; do
    W=${S%x*}; H=${S#*x}; FB=$((W*H))
    one res_$S "A=1" 0 $ROOT/apps/bench_main --size $S --frames 200 --steps 2 --warmup 1 --modes resident
    one u8_$S "A=1" $FB $ROOT/apps/bench_main --size $S --frames 200 --steps 2 --warmup 1 --modes u8
    one u8staged_$S "RMD_HIP_HOST_FRAMES=staged" $FB $ROOT/apps/bench_main --size $S --frames 200 --steps 2 --warmup 1 --modes u8
  done
fi
if [[ $PARTS == *batch* ]]; then
  for B in 8 16; do
    one res_b$B "A=1" 0 python3 $ROOT/tools/batch_bench.py --b $B --passes 2
    one u8_b$B "A=1" $((640*480*B)) python3 $ROOT/tools/batch_bench.py --b $B --passes 2 --u8
    one u8staged_b$B "RMD_HIP_HOST_FRAMES=staged" $((640*480*B)) python3 $ROOT/tools/batch_bench.py --b $B --passes 2 --u8
  done
fi
cat $OUT/summary.txt
