set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_m; mkdir -p $OUT
{ for rep in 1 2 3 4 5 6; do echo "== defaults, run $rep"; RMD_HIP_INGEST_PROFILE=1 python tools/r06_stall.py 80 2>&1 | grep -v "frames handed over" | cut -c1-500; done; } > $OUT/stall_phases.txt 2>&1; cat $OUT/stall_phases.txt
