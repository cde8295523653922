# headline (8-bit host frames) and resident rate of one sequence, a few repetitions; usage: tools/exp_single.sh [label]
for i in 1 2 3; do
python bench.py --steps 5 --warmup 2 --no-extras --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${1:-x} u8 host frames', d['value'], 'Mpix/s', d['config']['us_per_update_wall'], 'us')"
python bench.py --steps 5 --warmup 2 --no-extras --cpu-seconds 0 --resident 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${1:-x} resident      ', d['value'], 'Mpix/s', d['config']['us_per_update_wall'], 'us')"
done
