#!/bin/bash
# Round-6 GPU sessions (run through gpurun from the repo root):  bash tools/gpu_r6.sh <stage> [<stage> ...]
#   tests       the whole -m gpu suite
#   newtests    only the tests named in $NEWTESTS (a -k expression)
#   kstatp      rocprofv3 --kernel-trace --stats of `bench.py --workload $WL --particles $P` for P in $PS -> gpurun_out/kstat_<wl>_p<P>
#   pmcp        HBM / SQ / TCP counter passes of the same commands -> gpurun_out/pmc6/<wl>_p<P>/<pass>
#   psweep      bench.py --particles P (no variants) for P in $PS: ms per step, us per particle-scan, stage probes
#   bench       python bench.py (default command) and the driver's short form
#   ab          A/B of environment switches (AB="A=1,B=2;-"), AB_WL workloads, AB_P particles
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out
cd $ROOT
WL=${WL:-config2}
PS=${PS:-256 512}
echo "== $(date) stages: $*"
for ST in "$@"; do
case $ST in
tests)
  timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "tests rc=$?"; tail -n 30 gpurun_out/pytest_gpu.log | cut -c1-400 ;;
newtests)
  timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -k "$NEWTESTS" > gpurun_out/pytest_new.log 2>&1
  echo "newtests rc=$?"; tail -n 40 gpurun_out/pytest_new.log | cut -c1-600 ;;
kstatp)
  for P in $PS; do
    OUT=$ROOT/gpurun_out/kstat_${WL}_p$P${KSTAT_TAG:-}; rm -rf $OUT
    ( cd /tmp && env ${KSTAT_ENV:-} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- \
        python $ROOT/bench.py --workload $WL --particles $P --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-variants > $OUT.log 2>&1 )
    echo "kstat $WL p$P rc=$?"
    python - <<PY
import csv, glob
fs = glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)
for r in (csv.DictReader(open(fs[0])) if fs else []):
    if float(r["Percentage"]) > 0.4:
        print("  %-60s calls %5s avg %8.2f us min %8.2f max %8.2f  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
    grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"particle_groups_per_gpu": [0-9]*' $OUT.log | head -3 | tr '\n' ' '; echo
  done ;;
pmcp)
  for P in $PS; do
    for PASS in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum" \
                "sq:SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
                "sq2:SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" \
                "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
                "ta:TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
      N=${PASS%%:*}; C=${PASS#*:}
      OUT=$ROOT/gpurun_out/pmc6/${WL}_p$P/$N; rm -rf $OUT; mkdir -p $OUT
      ( cd /tmp && env ${PMC_ENV:-} timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc -- \
          python $ROOT/bench.py --workload $WL --particles $P --steps 12 --warmup 6 --repeats 1 --no-cpu-baseline --no-variants > $OUT.log 2>&1 )
      echo "pmc $WL p$P $N rc=$?"
    done
  done ;;
psweep)
  for P in $PS; do
    env ${SWEEP_ENV:-} python bench.py --workload $WL --particles $P --steps ${AB_STEPS:-60} --warmup 8 --repeats 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$WL p$P', 'ms/step', round(d['ms_per_step'], 4), 'us/particle-scan', round(1e3 * d['ms_per_step'] / $P, 4), 'groups', d['config']['particle_groups_per_gpu'], {k: v['avg_us'] for k, v in d['stages_probe'].items()}, 'flags', d['fault_flags'])
"
  done ;;
bench)
  timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
  echo "bench default rc=$?"; cut -c1-1500 gpurun_out/bench_default.json
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err
  echo "bench driver-form rc=$?"; cut -c1-600 gpurun_out/bench_driver.json ;;
ab)
  IFS=';' read -ra SETS <<< "${AB:--}"
  for W in ${AB_WL:-config2}; do
    for P in ${AB_P:-256}; do
      for SET in "${SETS[@]}"; do
        ENVS=$(echo "$SET" | tr ',' ' '); [ "$SET" = "-" ] && ENVS=""
        env $ENVS python bench.py --workload $W --particles $P --steps ${AB_STEPS:-60} --warmup 8 --repeats 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$W p$P [$SET]', 'ms/step', round(d['ms_per_step'], 4), 'us/ps', round(1e3 * d['ms_per_step'] / $P, 4), {k: v['avg_us'] for k, v in d['stages_probe'].items()}, 'flags', d['fault_flags'])
"
      done
    done
  done ;;
*) echo "unknown stage $ST" ;;
esac
done
echo "== done $(date)"
