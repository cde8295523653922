#!/bin/bash
# On the GPU box: SQ issue/stall counters of the search kernel on the heaviest kind of update (update 1 of a batch of 8: every seed of
# eight sequences searches its whole range).  usage: tools/pmc_heavy.sh [batch size]
cd /tmp; export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; B=${1:-8}; OUT=$ROOT/gpurun_out/pmc_heavy; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/a -- python $ROOT/tools/first_update_bench.py --b $B --reps 3 > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAVES SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $OUT/b -- python $ROOT/tools/first_update_bench.py --b $B --reps 3 > $OUT/b.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("a", "b"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "seed_search" in r["Kernel_Name"] or "seed_setup" in r["Kernel_Name"]:
                acc[("search" if "seed_search" in r["Kernel_Name"] else "setup", r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{k:7s} {c:24s} mean {sum(v)/len(v):14.0f}  n={len(v)}")
PY
