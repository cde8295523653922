#!/bin/bash
# Runs on the GPU box (via gpurun): supporting evidence of round 3 next to tools/profile_r03.sh -- the other BASELINE sizes, host-frame
# paths, batched mode (one launch pair for B sequences, two stream groups), stream-level alternatives, live use, the reference's own
# programs on the library.  Output: gpurun_out/summary_<tag>/.   usage: tools/evidence_r03.sh <tag> [parts: sizes,host,batch,live,ref]
set -u
TAG=${1:-r03}; PARTS=${2:-sizes,host,batch,live,ref,nranks,abmatch}
export TMPDIR=/tmp
ROOT=$(pwd)
SUM=$ROOT/gpurun_out/summary_$TAG
mkdir -p "$SUM"
if [[ $PARTS == *sizes* ]]; then
  echo "== configs[2]: 1280x960 x 500"
  timeout 900 python bench.py --size 1280x960 --steps 2 --warmup 1 --cpu-seconds 0 --batch "" > "$SUM/${TAG}_bench_config2_1280x960x500.json" 2> "$SUM/config2.err"; cut -c1-200 "$SUM/${TAG}_bench_config2_1280x960x500.json"; echo
  echo "== configs[4]: 1920x1080 x 1000 + TV-L1 500"
  timeout 1500 python bench.py --size 1920x1080 --steps 1 --warmup 1 --cpu-seconds 0 --batch "" > "$SUM/${TAG}_bench_config4_1920x1080x1000.json" 2> "$SUM/config4.err"; cut -c1-200 "$SUM/${TAG}_bench_config4_1920x1080x1000.json"; echo
fi
if [[ $PARTS == *host* ]]; then
  echo "== frames handed over in host memory (tools/h2d_rate.py)"
  { echo "# default: a single sequence stages its frames with the copy engine and converts them one step ahead; with the host-side time per frame"
    RMD_HIP_INGEST_PROFILE=1 python tools/h2d_rate.py 2>&1
    for m in staged inplace inplace_ahead; do
      echo "# RMD_HIP_HOST_FRAMES=$m"; RMD_HIP_HOST_FRAMES=$m python tools/h2d_rate.py 2>&1 | tail -6 | head -3
    done
    echo "# batches of 4 / 8, 8-bit host frames: default (read in place) and RMD_HIP_HOST_FRAMES=staged"
    python tools/batch_bench.py --b 4,8 --u8 --passes 2 2>&1 | grep flags
    RMD_HIP_HOST_FRAMES=staged python tools/batch_bench.py --b 4,8 --u8 --passes 2 2>&1 | grep flags
    echo "# tools/ubench/copy_rate: one 640x480 float / 8-bit frame from pinned memory, copy engine vs a kernel reading the host link"
    tools/ubench/copy_rate; tools/ubench/copy_rate 307200
  } > "$SUM/${TAG}_h2d.txt" 2>&1; cat "$SUM/${TAG}_h2d.txt"
fi
if [[ $PARTS == *batch* ]]; then
  echo "== batched mode: A/B of the stream groups and of the search loop's switches (same scene for every member)"
  { echo "# python tools/batch_bench.py --same-scene (default: up to three stream groups)"; python tools/batch_bench.py --b 1,2,3,4,6,8 --same-scene --passes 2 2>&1 | grep flags
    echo "# RMD_HIP_BATCH_GROUPS=1 (one launch pair for all members)"; RMD_HIP_BATCH_GROUPS=1 python tools/batch_bench.py --b 2,4,8 --same-scene --passes 2 2>&1 | grep flags
    echo "# RMD_HIP_BATCH_GROUPS=2"; RMD_HIP_BATCH_GROUPS=2 python tools/batch_bench.py --b 3,4,6,8 --same-scene --passes 2 2>&1 | grep flags
    echo "# RMD_HIP_BATCH_GROUPS=4 (the fourth group shares a hardware-queue pool with the first)"; RMD_HIP_BATCH_GROUPS=4 python tools/batch_bench.py --b 4,8 --same-scene --passes 2 2>&1 | grep flags
    echo "# search-loop switches (RMD_HIP_OPT_SEARCH_FLAGS: 1 prefetch the next unit, 2 sixteen hand-out counters, 4 tile box with the unit)"; python tools/batch_bench.py --b 1,4 --same-scene --passes 2 --flags 6,0,4,2,7 2>&1 | grep flags
    echo "# stream-level alternative: S independent batches on S streams / host threads (tools/multi_batch.py)"; python tools/multi_batch.py --configs 1x4,2x2,4x1,1x8,2x4 2>&1 | tail -5
  } > "$SUM/${TAG}_batch_ab.txt" 2>&1; cat "$SUM/${TAG}_batch_ab.txt"
fi
if [[ $PARTS == *live* ]]; then
  echo "== live use (node state machine)"
  python tools/live_bench.py --breakdown > "$SUM/${TAG}_live.txt" 2>&1; tail -12 "$SUM/${TAG}_live.txt"
fi
if [[ $PARTS == *ref* ]]; then
  echo "== the reference's own programs on the library"
  D=/tmp/over_table_$$
  python -c "from rpg_open_remode_amd import dataset as D; D.export_synthetic('$D', 640, 480, 200, image_ext='pgm', depth_every=1)"
  { echo "# oracle/_ref/rmd_gtests_ref: the reference's test/{seed_matrix,epipolar,reduction,main}_test.cpp, unmodified, on librmd_hip.so"; RMD_TEST_DATA_PATH=$D timeout 300 oracle/_ref/rmd_gtests_ref 2>&1 | grep -v "^DEBUG"
    echo; echo "# oracle/_ref/dataset_main_ref: the reference's test/dataset_main.cpp, unmodified (200 frames, 8-bit PGM + .depth per frame read inside the loop)"
    RMD_TEST_DATA_PATH=$D timeout 600 oracle/_ref/dataset_main_ref 2>&1 | grep -v "^T_world_curr\|^\[\|^RUN EXPERIMENT\|^$\|^  *[-0-9]" | tail -12; } > "$SUM/${TAG}_reference_programs.txt"
  tail -8 "$SUM/${TAG}_reference_programs.txt"
  rm -rf "$D"
fi
ls -la "$SUM"
if [[ $PARTS == *nranks* ]]; then
  echo "== two ranks (one GPU on this box: both ranks compute on device 0; the launch path the driver uses for --gpus N)"
  mkdir -p "$SUM/${TAG}_nranks"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --dist --steps 3 --warmup 1 --no-extras > "$SUM/${TAG}_nranks/one_rank_rccl.json" 2> "$SUM/${TAG}_nranks/one_rank.err"; cut -c1-160 "$SUM/${TAG}_nranks/one_rank_rccl.json"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --no-extras > "$SUM/${TAG}_nranks/two_ranks_one_gpu.json" 2> "$SUM/${TAG}_nranks/two_ranks.err"; cut -c1-160 "$SUM/${TAG}_nranks/two_ranks_one_gpu.json"
fi
if [[ $PARTS == *abmatch* ]]; then
  echo "== retired matchers (A/B build of the library, build_ab/librmd_hip_ab.so): parity test of variants 1, 2, 21"
  if [ -f build_ab/librmd_hip_ab.so ]; then
    cp rpg_open_remode_amd/librmd_hip.so /tmp/librmd_hip_product.so
    cp build_ab/librmd_hip_ab.so rpg_open_remode_amd/librmd_hip.so
    python -m pytest tests/test_hip_parity.py -q -k "other_matchers" 2>&1 | tail -2 | tee "$SUM/${TAG}_ab_matchers_parity.txt"
    cp /tmp/librmd_hip_product.so rpg_open_remode_amd/librmd_hip.so
  fi
fi
