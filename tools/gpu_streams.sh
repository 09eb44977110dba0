for S in 1 2 4 8; do
 for WL in config2 ref2level; do
 timeout 300 python bench.py --workload $WL --streams $S --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$WL streams', d['config']['streams_per_gpu'], 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],3), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1))
"
 done
done
