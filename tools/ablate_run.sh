# tools/ablate_run.sh <workload> [variants...]: stage times of the timing-only variants built by tools/ablate_build.py (through gpurun)
WL=${1:-config5}; shift
for f in ${@:-$(ls build_abl/*.so)}; do
SLAM2D_LIB=$PWD/$f python bench.py --workload $WL --steps 30 --warmup 6 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('%-24s' % '$f', 'ms/step', round(d['ms_per_step'],4), {k: v['avg_us'] for k, v in d['stages_probe'].items()})
"
done
