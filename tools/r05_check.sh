export TMPDIR=/tmp; ROOT=$(pwd); OUT=$ROOT/gpurun_out/r05_i; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q -rs --durations=12 ) > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log
for T in 2 3 4 2 3 4; do echo "== unit target $T"; apps/bench_main --modes u8,resident --steps 5 --warmup 1 --unit-target $T | cut -c60-130; done > $OUT/unit_target.txt 2>&1; cat $OUT/unit_target.txt
