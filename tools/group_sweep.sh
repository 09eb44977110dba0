cd $GRAFT_REPO_ROOT
for P in 128 256 512; do for G in 1 2 4; do
python bench.py --particles $P --groups $G --steps 60 --warmup 8 --repeats 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('p$P groups $G', 'ms/step', round(d['ms_per_step'], 4), 'us/ps', round(1e3 * d['ms_per_step'] / $P, 4), {k: v['avg_us'] for k, v in d['stages_probe'].items()})
"
done; done
