set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_k; mkdir -p $OUT
{ for E in "A=1" "RMD_HIP_COPY_STREAMS=1" "RMD_HIP_RING_DEPTH=4" "RMD_HIP_COPY_STREAMS=1 RMD_HIP_RING_DEPTH=4" "RMD_HIP_HOST_WAIT=0" "RMD_HIP_HOST_FRAMES=staged"; do
    for rep in 1 2; do echo "== $E"; env $E RMD_HIP_INGEST_PROFILE=1 python tools/r06_stall.py 80 2>&1 | cut -c1-900; done
  done
  echo "== r05 library (build_ab/librmd_hip_r05.so cannot be loaded by this api.py: new symbols) -- bench_main against build_ab/r05"
  for rep in 1 2 3; do LD_LIBRARY_PATH=$ROOT/build_ab/r05 apps/bench_main --modes u8 --steps 40 --warmup 3 2>&1 | python3 -c "
import sys,re
for l in sys.stdin:
    if l.startswith('{'): print('   r05', ', '.join(re.findall(r'\"value\": [\d.]+|\"us_per_update_wall\": [\d.]+',l)))"; done
  for rep in 1 2 3; do apps/bench_main --modes u8 --steps 40 --warmup 3 2>&1 | python3 -c "
import sys,re
for l in sys.stdin:
    if l.startswith('{'): print('   r06', ', '.join(re.findall(r'\"value\": [\d.]+|\"us_per_update_wall\": [\d.]+',l)))"; done
} > $OUT/stall.txt 2>&1
cat $OUT/stall.txt
