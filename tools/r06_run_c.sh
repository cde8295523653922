set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_c; mkdir -p $OUT
timeout 900 python -m pytest tests/test_publish_async.py tests/test_node.py tests/test_host_frame_modes.py tests/test_full_speed.py tests/test_batch.py -m gpu -x -q -rs > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
timeout 300 python tools/live_bench.py --breakdown > $OUT/live.txt 2>&1
bash tools/r06_iter.sh c none big,vga,batch product > /dev/null 2>&1
RMD_HIP_COPY_STREAMS=1 bash tools/r06_iter.sh c_cs1 none big,vga product > /dev/null 2>&1
tail -5 $OUT/pytest.log; cat $OUT/live.txt
