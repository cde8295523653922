#!/bin/bash
# Runs on the GPU box (via gpurun): supporting evidence of a round next to tools/profile_round.sh -- the other BASELINE sizes, batched mode
# (stream groups, TV-L1 for all members), N ranks x B sequences on the one-GPU lease, live use, the reference's own programs on the
# library, the retired matchers.  Output: gpurun_out/summary_<tag>/.   usage: tools/evidence_round.sh <tag> [parts]
set -u
TAG=${1:-r05}; PARTS=${2:-sizes,batch,live,ref,nranks,first}
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
if [[ $PARTS == *batch* ]]; then
  echo "== batched mode: stream groups (same scene for every member), host frames"
  { echo "# python tools/batch_bench.py --same-scene (default: up to three stream groups)"; python tools/batch_bench.py --b 1,2,3,4,6,8,12,16,24 --same-scene --passes 2 2>&1 | grep flags
    echo "# RMD_HIP_BATCH_GROUPS=1 (one launch pair for all members)"; RMD_HIP_BATCH_GROUPS=1 python tools/batch_bench.py --b 2,4,8 --same-scene --passes 2 2>&1 | grep flags
    echo "# RMD_HIP_BATCH_GROUPS=2"; RMD_HIP_BATCH_GROUPS=2 python tools/batch_bench.py --b 4,8 --same-scene --passes 2 2>&1 | grep flags
    echo "# scenes 0..B-1, frames resident / 8-bit host frames"; python tools/batch_bench.py --b 1,2,4,8,16 --passes 3 2>&1 | grep flags; python tools/batch_bench.py --b 4,8,16 --passes 3 --u8 2>&1 | grep flags
  } > "$SUM/${TAG}_batch_ab.txt" 2>&1; cat "$SUM/${TAG}_batch_ab.txt"
fi
if [[ $PARTS == *first* ]]; then
  python tools/first_update_bench.py --b 1,8 --label "update 1 (every seed live)" > "$SUM/${TAG}_first_update.txt" 2>&1; cat "$SUM/${TAG}_first_update.txt"
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
if [[ $PARTS == *nranks* ]]; then
  echo "== ranks on the one-GPU lease (all ranks compute on device 0; the launch path the driver uses for --gpus N)"
  mkdir -p "$SUM/${TAG}_nranks"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --dist --steps 3 --warmup 1 --no-extras | grep '^{"metric"' > "$SUM/${TAG}_nranks/one_rank_rccl.json" 2> "$SUM/${TAG}_nranks/one_rank.err"; cut -c1-160 "$SUM/${TAG}_nranks/one_rank_rccl.json"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --no-extras | grep '^{"metric"' > "$SUM/${TAG}_nranks/two_ranks_one_gpu.json" 2> "$SUM/${TAG}_nranks/two_ranks.err"; cut -c1-160 "$SUM/${TAG}_nranks/two_ranks_one_gpu.json"
  echo "-- two ranks x four sequences each (N GPUs x B sequences, configs[3] composed with the batched mode)"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --batch-per-gpu 4 --steps 3 --warmup 1 --no-extras | grep '^{"metric"' > "$SUM/${TAG}_nranks/two_ranks_x_four_sequences_one_gpu.json" 2> "$SUM/${TAG}_nranks/two_ranks_x4.err"; cut -c1-160 "$SUM/${TAG}_nranks/two_ranks_x_four_sequences_one_gpu.json"
  echo "-- one rank x eight sequences (--batch-per-gpu 8)"
  timeout 600 python bench.py --batch-per-gpu 8 --steps 3 --warmup 1 --no-extras | grep '^{"metric"' > "$SUM/${TAG}_nranks/one_rank_x_eight_sequences.json" 2> "$SUM/${TAG}_nranks/one_rank_x8.err"; cut -c1-160 "$SUM/${TAG}_nranks/one_rank_x_eight_sequences.json"
fi
ls -la "$SUM"
