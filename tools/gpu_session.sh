#!/bin/bash
# One GPU-box session: smoke, GPU parity tests, a short bench, optional rocprof.
# Usage (from the repo root, through gpurun):  bash tools/gpu_session.sh [quick|full|prof]
MODE=${1:-quick}
mkdir -p gpurun_out
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== mode $MODE  $(date)"
rocm-smi --showproductname 2>/dev/null | grep -m2 -i -E "card series|gfx" || true

timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -n 5 gpurun_out/smoke.log

timeout 1500 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -n 60 gpurun_out/pytest_gpu.log

timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/bench_config2.log 2>&1
echo "bench rc=$?"; tail -n 3 gpurun_out/bench_config2.log

if [ "$MODE" != "quick" ]; then
  timeout 600 python bench.py --workload ref2level --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_ref2level.log 2>&1
  echo "bench ref2level rc=$?"; tail -n 3 gpurun_out/bench_ref2level.log
fi
if [ "$MODE" = "prof" ]; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_config2 -o config2 -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_config2.log 2>&1
  echo "rocprof rc=$?"
  cd $GRAFT_REPO_ROOT
  find gpurun_out/prof_config2 -name "*stats*" | head
fi
echo "== done $(date)"
