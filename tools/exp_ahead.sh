for A in 128 64 32 256; do
for i in 1 2; do
RMD_HIP_AHEAD_WGS=$A python bench.py --steps 5 --warmup 2 --no-extras --cpu-seconds 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ahead wgs $A: u8 host frames', d['value'], 'Mpix/s', d['config']['us_per_update_wall'], 'us')"
done; done
