for M in staged_ahead staged inplace inplace_ahead; do
RMD_HIP_HOST_FRAMES=$M RMD_HIP_INGEST_PROFILE=1 python bench.py --size 1920x1080 --steps 1 --warmup 1 --no-extras --cpu-seconds 0 2>/tmp/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$M', '| Mpix/s', d['value'], '| us/update', d['config']['us_per_update_wall'], '| device', d['roofline']['avg_launch_us'])
"; grep "rmd_hip ingest" /tmp/err.txt | tail -1
done
