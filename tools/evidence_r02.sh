#!/bin/bash
# Runs on the GPU box (via gpurun): the round's supporting evidence next to tools/profile_r02.sh (A/B tables, timelines, other configs,
# the reference's own programs).  Everything lands in gpurun_out/summary_<tag>/ and is copied into profiles/ by hand.
set -u
TAG=${1:-r02}; PARTS=${2:-bench,timeline,h2d,ab,configs,refprogs}
export TMPDIR=/tmp
ROOT=$(pwd)
SUM=$ROOT/gpurun_out/summary_$TAG
mkdir -p "$SUM"
if [[ $PARTS == *bench* ]]; then
  timeout 600 python bench.py > "$SUM/${TAG}_bench_default.json" 2> "$SUM/bench_default.err"; tail -c 300 "$SUM/${TAG}_bench_default.json"; echo
fi
if [[ $PARTS == *timeline* ]]; then
  { echo "# python tools/search_timeline.py (frames resident in HBM)"; timeout 250 python tools/search_timeline.py --show 1,4,15,25,35,60,90,105,125,150,190 2>&1
    echo; echo "# python tools/search_timeline.py --u8 (8-bit frames handed over in host memory)"; timeout 250 python tools/search_timeline.py --u8 --show 4,25,60,125,190 2>&1 | grep -v "^        \|slowest"; } > "$SUM/${TAG}_timeline.txt"
  tail -3 "$SUM/${TAG}_timeline.txt"
fi
if [[ $PARTS == *h2d* ]]; then
  bash tools/h2d_ab.sh > "$SUM/${TAG}_h2d_ab.txt" 2>&1; tail -4 "$SUM/${TAG}_h2d_ab.txt"
fi
if [[ $PARTS == *ab* ]]; then
  timeout 400 python tools/frame_ab.py --variants 0,3,1,2,21,31,32 > "$SUM/${TAG}_matcher_ab.txt" 2>&1; tail -8 "$SUM/${TAG}_matcher_ab.txt"
fi
if [[ $PARTS == *configs* ]]; then
  timeout 900 python bench.py --size 1280x960 --steps 2 --warmup 1 --cpu-seconds 0 > "$SUM/${TAG}_bench_config2_1280x960x500.json" 2> "$SUM/bench_config2.err"; tail -c 200 "$SUM/${TAG}_bench_config2_1280x960x500.json"; echo
  timeout 900 python bench.py --size 1920x1080 --frames 200 --tv-iters 500 --steps 2 --warmup 1 --cpu-seconds 0 > "$SUM/${TAG}_bench_1920x1080x200.json" 2> "$SUM/bench_1080.err"; tail -c 200 "$SUM/${TAG}_bench_1920x1080x200.json"; echo
fi
if [[ $PARTS == *refprogs* ]]; then
  D=/tmp/over_table_$$
  python -c "from rpg_open_remode_amd import dataset as D; D.export_synthetic('$D', 640, 480, 200, image_ext='pgm', depth_every=1)" 
  { echo "# oracle/_ref/rmd_gtests_ref: the reference's test/{seed_matrix,epipolar,reduction,main}_test.cpp, unmodified, on librmd_hip.so"; RMD_TEST_DATA_PATH=$D timeout 300 oracle/_ref/rmd_gtests_ref 2>&1 | grep -v "^DEBUG"
    echo; echo "# oracle/_ref/dataset_main_ref: the reference's test/dataset_main.cpp, unmodified (200 frames, 8-bit PGM + .depth per frame read inside the loop)"
    RMD_TEST_DATA_PATH=$D timeout 600 oracle/_ref/dataset_main_ref 2>&1 | grep -v "^T_world_curr\|^\[\|^RUN EXPERIMENT\|^$\|^  *[-0-9]" | tail -12; } > "$SUM/${TAG}_reference_programs.txt"
  tail -8 "$SUM/${TAG}_reference_programs.txt"
  rm -rf "$D"
fi
ls -la "$SUM"
