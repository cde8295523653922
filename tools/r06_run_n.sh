# GPU box: a batch's host frames converted by a kernel of their own on the copy stream (RMD_HIP_HOST_FRAMES=inplace_ahead) -- parity of the batch tests in
# that mode, then rates against the default (the step's setup kernels read the pinned block).
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_n; mkdir -p $OUT
RMD_HIP_HOST_FRAMES=inplace_ahead timeout 900 python -m pytest tests/test_batch.py tests/test_full_speed.py tests/test_host_frame_modes.py tests/test_concurrency.py -m gpu -x -q -rs > $OUT/pytest_mode.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_mode.log; tail -4 $OUT/pytest_mode.log
timeout 600 python -m pytest tests/test_batch.py tests/test_hip_parity.py -m gpu -x -q > $OUT/pytest_default.log 2>&1; tail -2 $OUT/pytest_default.log
for rep in 1 2; do
  for B in 4 8 16; do
    echo "== batch of $B resident"; python tools/batch_bench.py --b $B --passes 3 2>&1 | grep Mpix | cut -c1-150
    echo "== batch of $B u8 in place (default)"; RMD_HIP_INGEST_PROFILE=1 python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | grep "Mpix\|ingest\] batch" | cut -c1-330
    echo "== batch of $B u8, conversion kernel on the copy stream"; RMD_HIP_HOST_FRAMES=inplace_ahead RMD_HIP_INGEST_PROFILE=1 python tools/batch_bench.py --b $B --passes 3 --u8 2>&1 | grep "Mpix\|ingest\] batch" | cut -c1-330
  done
done > $OUT/rates.txt 2>&1
cat $OUT/rates.txt
