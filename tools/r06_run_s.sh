#!/bin/bash
# GPU box: copy engines addressed directly -- tests again, then timelines of 1920x1080 with host frames by route (rocprofv3 kernel + memory-copy trace).
set -u
export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_s; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests/test_host_frame_modes.py tests/test_full_speed.py tests/test_abi.py tests/test_concurrency.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
cd /tmp
one() {  # label, env, frame bytes, command...
  local L=$1 E=$2 FB=$3; shift 3
  env $E RMD_HIP_INGEST_PROFILE=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/$L -- "$@" > $OUT/$L.log 2>&1
  { echo "== $L ($E) $*"; grep -h '"value"\|rmd_hip ingest\|Mpix' $OUT/$L.log | cut -c1-300; python3 $ROOT/tools/r06_timeline.py $OUT/$L 400 $FB; } >> $OUT/summary.txt 2>&1
  ls $OUT/$L/*/* | head -5 >> $OUT/files.txt; head -3 $OUT/$L/*/*memory_copy_trace.csv >> $OUT/files.txt 2>&1
  rm -rf $OUT/$L
}
for E in 2 0 1; do
  one u8_1080_e$E "RMD_HIP_COPY_ENGINES=$E" $((1920*1080)) $ROOT/apps/bench_main --size 1920x1080 --frames 600 --steps 2 --warmup 1 --modes u8
done
one u8_960_e2 "RMD_HIP_COPY_ENGINES=2" $((1280*960)) $ROOT/apps/bench_main --size 1280x960 --frames 500 --steps 2 --warmup 1 --modes u8
one u8_vga_e2 "RMD_HIP_COPY_ENGINES=2" $((640*480)) $ROOT/apps/bench_main --size 640x480 --frames 200 --steps 3 --warmup 1 --modes u8
cat $OUT/summary.txt; cat $OUT/files.txt
