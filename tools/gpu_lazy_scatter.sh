# A/B of the scatter as its own launch behind k_endpoints, placing only cells near needed tiles (round 5)  -- run through gpurun
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out
if [ "$1" = "tests" ]; then
  SLAM2D_MERGE_SCATTER=0 timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "not core_quota and not bench" > gpurun_out/pytest_lazy_scatter.log 2>&1
  echo "tests (SLAM2D_MERGE_SCATTER=0) rc=$?"; tail -n 12 gpurun_out/pytest_lazy_scatter.log | cut -c1-300
fi
for SET in "-" "SLAM2D_MERGE_SCATTER=0" "SLAM2D_MERGE_SCATTER=0 SLAM2D_LAZY_SCATTER=0"; do
  ENVS=$SET; [ "$SET" = "-" ] && ENVS=""
  for WL in config2 ref2level config5; do
    env $ENVS python bench.py --workload $WL --steps 60 --warmup 8 --repeats 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$WL [$SET]', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), {k: v['avg_us'] for k, v in d['stages_probe'].items()}, 'flags', d['fault_flags'])
"
  done
  env $ENVS python bench.py --workload config3 --particles 64 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('config3 closed loop [$SET]', 'scans/s', round(d.get('scans_per_sec', 0), 1), 'ms/step', round(d['ms_per_step'], 4))
"
done
