# GPU box: A/B of the LDS window's row stride (window_stride(), csrc/rmd_frame.hpp): product (ww | 1) against build_ab/librmd_hip_stride3.so (ww | 3) and
# _stride5.so ((ww + 2) | 1): parity subset, batches of 8 / 16 and one sequence with resident frames (two repetitions), and the LDS counters of the search
# kernel for a batch of 8 (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS, one --pmc pass each way the guide prescribes: own runs, kernel trace only).
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_stride; rm -rf $OUT; mkdir -p $OUT
for L in product stride3 stride5; do
  if [ $L = product ]; then unset RMD_HIP_LIB; else export RMD_HIP_LIB=$ROOT/build_ab/librmd_hip_$L.so; fi
  { echo "==== $L"
    timeout 600 python -m pytest tests/test_hip_parity.py tests/test_batch.py -m gpu -x -q 2>&1 | tail -1
    for rep in 1 2; do python tools/batch_bench.py --b 1,8,16 --passes 3 2>&1 | grep Mpix | cut -c1-120; done
    cd /tmp
    timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_$L -- python $ROOT/tools/batch_bench.py --b 8 --passes 1 > $OUT/pmc_$L.log 2>&1
    python3 - $OUT/pmc_$L <<'PY'
import csv,glob,sys
from collections import defaultdict
tot=defaultdict(float); n=0
for f in glob.glob(sys.argv[1]+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'seed_search_compact' in r.get('Kernel_Name',''):
            tot[r['Counter_Name']]+=float(r['Counter_Value'])
print('   search kernel, batch of 8, all launches of the run:', {k: f'{v:.4g}' for k,v in sorted(tot.items())})
if tot.get('SQ_INSTS_LDS'): print('   conflict cycles per LDS instruction: %.3f' % (tot.get('SQ_LDS_BANK_CONFLICT',0)/tot['SQ_INSTS_LDS']))
PY
    rm -rf $OUT/pmc_$L; cd $ROOT
  } >> $OUT/summary.txt 2>&1
done
unset RMD_HIP_LIB
cat $OUT/summary.txt
