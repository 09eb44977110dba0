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
  for WL in config2 ref2level; do
    rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$WL
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$WL -o $WL -- \
        python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$WL.log 2>&1
    echo "rocprof $WL rc=$?"; tail -n 1 $GRAFT_REPO_ROOT/gpurun_out/prof_$WL.log | cut -c1-200
  done
  cd $GRAFT_REPO_ROOT
  find gpurun_out/prof_config2 gpurun_out/prof_ref2level -name "*stats*" | head
  # HBM traffic of the default bench command: separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share one)
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    N=$(echo $C | cut -d' ' -f1)
    rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$N
    timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$N -o pmc -- \
        python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$N.log 2>&1
    echo "pmc $N rc=$?"
  done
  cd $GRAFT_REPO_ROOT
  python tools/summarize_profiles.py > gpurun_out/profile_summary.txt 2>&1; tail -n 30 gpurun_out/profile_summary.txt
  SLAM2D_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_forced_dist.log 2>&1
  echo "forced-dist rc=$?"; grep -c '"metric"' gpurun_out/bench_forced_dist.log
fi
echo "== done $(date)"
