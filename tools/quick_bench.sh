# tools/quick_bench.sh [workloads...]: full GPU test suite, then one short bench line per workload (run through gpurun)
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo pytest rc=$?; tail -n 12 gpurun_out/pytest_gpu.log | cut -c1-300
for WL in "${@:-config2}"; do
python bench.py --workload $WL --steps 60 --warmup 8 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$WL', round(d['value']), 'ms/step', round(d['ms_per_step'],4), {k: v['avg_us'] for k, v in d['stages_probe'].items()}, d['roofline']['kernel'], d['fault_flags'])
"
done
