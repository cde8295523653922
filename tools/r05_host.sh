#!/bin/bash
# GPU box: the host side of a rank: the C++ timed driver next to the Python one, the ring-slot wait spin vs sleep, eight ranks
# on the lease.  usage: tools/r05_host.sh <tag>
set -u
export TMPDIR=/tmp; ROOT=$(pwd); TAG=${1:-x}; OUT=$ROOT/gpurun_out/r05_host_$TAG; mkdir -p $OUT
nproc > $OUT/cpus.txt; cat /sys/fs/cgroup/cpu.max >> $OUT/cpus.txt 2>/dev/null
{ echo "# apps/bench_main (C++), default (RMD_HIP_HOST_WAIT=1: spin 5 us, then sleep)"; apps/bench_main --steps 5 --warmup 1 --modes u8,resident,float
  echo "# apps/bench_main, RMD_HIP_HOST_WAIT=0 (spin only)"; RMD_HIP_HOST_WAIT=0 apps/bench_main --steps 5 --warmup 1 --modes u8
  echo "# bench.py (Python), same timed region"; python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras 2>/dev/null
  echo "# bench.py, RMD_HIP_HOST_WAIT=0"; RMD_HIP_HOST_WAIT=0 python bench.py --steps 5 --warmup 1 --cpu-seconds 0 --no-extras 2>/dev/null
} > $OUT/cpp_vs_python.txt 2>&1
{ echo "# eight ranks of apps/bench_main (u8 host frames, scenes 0..7) on this lease's device(s) under taskset -c 0-15: host CPU per rank"; taskset -c 0-15 apps/bench_main --ranks-probe 8 --steps 3 --warmup 1
  echo "# the same with RMD_HIP_HOST_WAIT=0"; RMD_HIP_HOST_WAIT=0 taskset -c 0-15 apps/bench_main --ranks-probe 8 --steps 3 --warmup 1
  echo "# one rank alone under the same taskset"; taskset -c 0-15 apps/bench_main --steps 3 --warmup 1 --modes u8
} > $OUT/eight_ranks.txt 2>&1
cat $OUT/cpus.txt $OUT/cpp_vs_python.txt $OUT/eight_ranks.txt | cut -c1-400
