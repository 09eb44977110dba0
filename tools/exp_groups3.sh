for G in 2 3 6; do python bench.py --particles 66 --groups $G --steps 60 --warmup 8 --repeats 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('P66 G$G', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'host', d['timed_blocks']['host_enqueue_ms_per_step'])
"; done
