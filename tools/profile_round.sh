#!/bin/bash
# Runs on the GPU box (via gpurun): a round's measurement set (rounds 4 and 5).  Raw rocprofv3 output under gpurun_out/prof_<tag>/,
# compact summaries (what gets committed under profiles/) under gpurun_out/summary_<tag>/.
# usage: tools/profile_round.sh <tag> [parts: bench,trace,restrace,pmc,tv1080]
set -u
TAG=${1:-r05}; PARTS=${2:-bench,trace,restrace,pmc,tv1080}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
SUM=$ROOT/gpurun_out/summary_$TAG
mkdir -p "$OUT" "$SUM"
: > "$OUT/commands.txt"
cd /tmp
ONE="--steps 1 --warmup 0 --cpu-seconds 0 --no-extras"
if [[ $PARTS == *trace* ]]; then
  echo "== kernel trace + stats (2 timed passes after 1 warm-up pass)"
  printf 'trace\tpython bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extras\n' >> "$OUT/commands.txt"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 2 --warmup 1 --cpu-seconds 0 --no-extras > "$SUM/${TAG}_bench_under_trace.json" 2> "$OUT/trace.err"
  (cd "$ROOT" && python tools/trace_breakdown.py "$OUT" 199 398 > "$SUM/${TAG}_timed_pass_breakdown.txt"; cat "$SUM/${TAG}_timed_pass_breakdown.txt")
fi
if [[ $PARTS == *restrace* ]]; then
  # the same passes with the frames resident in HBM: under the tracer the host thread (a frame copied into the pinned ring and two intercepted
  # launches per update) is slower than the device, so the default command's setup kernels spend part of their time waiting for their frame
  echo "== kernel trace, frames resident (2 timed passes after 1 warm-up pass)"
  printf 'trace_resident\tpython bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-extras --resident\n' >> "$OUT/commands.txt"
  mkdir -p "$OUT/res"
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/res/trace" -- python "$ROOT/bench.py" --steps 2 --warmup 1 --cpu-seconds 0 --no-extras --resident > "$SUM/${TAG}_bench_under_trace_resident.json" 2> "$OUT/res_trace.err"
  (cd "$ROOT" && python tools/trace_breakdown.py "$OUT/res" 199 398 > "$SUM/${TAG}_timed_pass_breakdown_resident.txt"; cat "$SUM/${TAG}_timed_pass_breakdown_resident.txt")
fi
if [[ $PARTS == *pmc* ]]; then
  echo "== PMC pass 1 (instruction counts, one complete pass of the sequence + the denoise)"
  printf 'pmc_insts\tpython bench.py %s  (640x480, one pass + TV-L1)\n' "$ONE" >> "$OUT/commands.txt"
  timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d "$OUT/pmc_insts" -- python "$ROOT/bench.py" $ONE > /dev/null 2> "$OUT/pmc_insts.err"
  echo "faults: $(grep -c 'Memory access fault' $OUT/pmc_insts.err)"
  echo "== PMC pass 2 (FETCH_SIZE)"
  printf 'pmc_fetch\tpython bench.py %s  (640x480, one pass + TV-L1)\n' "$ONE" >> "$OUT/commands.txt"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python "$ROOT/bench.py" $ONE > /dev/null 2> "$OUT/pmc_fetch.err"
  echo "== PMC pass 3 (WRITE_SIZE)"
  printf 'pmc_write\tpython bench.py %s  (640x480, one pass + TV-L1)\n' "$ONE" >> "$OUT/commands.txt"
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python "$ROOT/bench.py" $ONE > /dev/null 2> "$OUT/pmc_write.err"
fi
if [[ $PARTS == *tv1080* ]]; then
  echo "== denoiser at 1920x1080 (TV-L1 500 iterations after a 3-frame sequence): trace, FETCH_SIZE, WRITE_SIZE"
  TV="--size 1920x1080 --frames 3 --tv-iters 500 --steps 1 --warmup 0 --cpu-seconds 0 --no-extras"
  for p in tv1080_trace tv1080_fetch tv1080_write; do printf '%s\tpython bench.py %s  (1920x1080 denoiser run)\n' "$p" "$TV" >> "$OUT/commands.txt"; done
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tv1080_trace" -- python "$ROOT/bench.py" $TV > "$SUM/${TAG}_bench_1080p_tv.json" 2> "$OUT/tv1080_trace.err"
  timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/tv1080_fetch" -- python "$ROOT/bench.py" $TV > /dev/null 2> "$OUT/tv1080_fetch.err"
  timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/tv1080_write" -- python "$ROOT/bench.py" $TV > /dev/null 2> "$OUT/tv1080_write.err"
fi
cd "$ROOT"
python tools/summarize_rocprof.py "$OUT" "$SUM" "$TAG" > /dev/null
python tools/make_traffic.py "$OUT" "$SUM/${TAG}_traffic.json" || true
# the bench lines come LAST: bench.py takes its instruction counts from profiles/traffic.json and refuses counts of other kernel sources, so the
# counts of this very run are installed (in the box's copy of the repository; copy them into profiles/ at home as well) before it is started
if [[ $PARTS == *pmc* && -s "$SUM/${TAG}_traffic.json" ]]; then cp "$SUM/${TAG}_traffic.json" "$ROOT/profiles/traffic.json"; fi
if [[ $PARTS == *bench* ]]; then
  echo "== plain default run"
  timeout 900 python bench.py > "$SUM/${TAG}_bench_default.json" 2> "$OUT/bench_default.err"; tail -c 400 "$SUM/${TAG}_bench_default.json"; echo
  echo "== as the driver runs it (--steps 20 --warmup 5)"
  timeout 900 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extras > "$SUM/${TAG}_bench_steps20.json" 2> "$OUT/bench_steps20.err"
fi
ls -la "$SUM"
