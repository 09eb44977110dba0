for P in 8 16 32 64 128 256 512; do
 timeout 300 python bench.py --particles $P --steps 20 --warmup 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('P', d['config']['particles_per_gpu'], 'value', round(d['value']), 'ms/step', round(d['ms_per_step'],3))
"
done
