# A/B of alternative builds (ab_libs/*.so, same ABI) on the bench workloads and the closed loop: tools/gpu_ab_libs.sh <lib|-> ...  (through gpurun)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
cd ${GRAFT_REPO_ROOT:-$PWD}
for LIB in "$@"; do
  ENVS=""; [ "$LIB" != "-" ] && ENVS="SLAM2D_LIB=$PWD/$LIB"
  for WL in ${AB_WL:-config2 ref2level config5}; do
    for PP in ${AB_P:-0}; do PARG=""; [ "$PP" != "0" ] && PARG="--particles $PP"
    env $ENVS python bench.py --workload $WL $PARG --steps ${AB_STEPS:-60} --warmup 8 --repeats 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$WL p$PP [$LIB]', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), {k: v['avg_us'] for k, v in d['stages_probe'].items()}, 'flags', d['fault_flags'])
"
    done
  done
  [ -n "$AB_NO_CLOSED" ] || env $ENVS python bench.py --workload config3 --particles 64 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('config3 closed loop [$LIB]', 'scans/s', round(d.get('scans_per_sec', 0), 1), 'ms/step', round(d['ms_per_step'], 4))
"
done
