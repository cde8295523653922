#!/bin/bash
# GPU box: the round's last measurement set after the copy-engine work (device code unchanged: the counter files of the r06 closing set stay valid) -- the stall
# hunt with the new defaults, then tools/profile_round.sh r06e (kernel traces with host frames and resident, the default bench line, the driver's --steps 20 line).
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_z; mkdir -p $OUT
{ for rep in 1 2 3; do echo "== defaults (copy engines in rotation, plain staged)"; RMD_HIP_INGEST_PROFILE=1 python tools/r06_stall.py 80 2>&1 | cut -c1-700; done; } > $OUT/stall_defaults.txt 2>&1; cat $OUT/stall_defaults.txt | cut -c1-300
bash tools/profile_round.sh r06e bench,trace,restrace 2>&1 | tail -40 | cut -c1-300
