set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r06_f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_publish_async.py tests/test_node.py tests/test_reference_host_sources.py tests/test_concurrency.py tests/test_parity_glibc.py tests/test_hip_parity.py -m gpu -x -q -rs --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
timeout 300 python tools/live_bench.py --breakdown > $OUT/live.txt 2>&1
timeout 300 python tools/live_bench.py > $OUT/live_plain.txt 2>&1
tail -22 $OUT/pytest.log; cat $OUT/live.txt; cat $OUT/live_plain.txt
