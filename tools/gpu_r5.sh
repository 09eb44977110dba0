#!/bin/bash
# Round-5 GPU sessions (run through gpurun from the repo root):  bash tools/gpu_r5.sh <stage> [<stage> ...]
#   newtests   the round's new parity tests only
#   tests      the whole -m gpu suite
#   kstat      rocprofv3 --kernel-trace --stats of the three bench workloads -> gpurun_out/kstat_<wl>
#   bench      python bench.py (default command) and the driver's short form
#   clock      in-kernel phase stamps (debug build; LAST: it rebuilds the library)
#   pmc        HBM / SQ / TCP counter passes of the three workloads -> gpurun_out/pmc3/<wl>/<pass>
#   multi      bench.py --gpus 2 on one GPU (gloo dry mode)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out
cd $ROOT
echo "== $(date) stages: $*"
for ST in "$@"; do
case $ST in
newtests)
  timeout 900 python -m pytest $(ls tests/test_gpu_bench_parity.py tests/test_gpu_bench_cli.py tests/test_gpu_export.py tests/test_gpu_wide_counts.py 2>/dev/null) -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_new.log 2>&1
  echo "newtests rc=$?"; tail -n 25 gpurun_out/pytest_new.log | cut -c1-400 ;;
tests)
  timeout 1800 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "tests rc=$?"; tail -n 30 gpurun_out/pytest_gpu.log | cut -c1-400 ;;
kstat)
  for WL in ${KSTAT_WL:-config2 ref2level config5}; do
    OUT=$ROOT/gpurun_out/kstat_$WL${KSTAT_TAG:-}; rm -rf $OUT
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- \
        python $ROOT/bench.py --workload $WL --steps 40 --warmup 5 --no-cpu-baseline --no-variants > $OUT.log 2>&1 )
    echo "kstat $WL rc=$?"
    python - <<PY
import csv, glob
fs = glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)
for r in (csv.DictReader(open(fs[0])) if fs else []):
    if float(r["Percentage"]) > 0.4:
        print("  %-60s calls %5s avg %8.2f us min %8.2f max %8.2f  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
    grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $OUT.log | head -2 | tr '\n' ' '; echo
  done ;;
bench)
  timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
  echo "bench default rc=$?"; cut -c1-1500 gpurun_out/bench_default.json
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.json 2> gpurun_out/bench_driver.err
  echo "bench driver-form rc=$?"; cut -c1-600 gpurun_out/bench_driver.json ;;
multi)
  timeout 600 python bench.py --gpus 2 --backend gloo --share-gpu --steps 20 --warmup 5 --no-variants > gpurun_out/bench_gpus2.json 2> gpurun_out/bench_gpus2.err
  echo "bench --gpus 2 (one GPU shared) rc=$?"; cut -c1-900 gpurun_out/bench_gpus2.json; tail -n 5 gpurun_out/bench_gpus2.err ;;
clock)
  SLAM2D_BENCH_GROUPS=1 timeout 600 python tools/dbg_clock.py config2 64 > gpurun_out/dbg_clock.log 2>&1
  echo "clock rc=$?"; tail -n 12 gpurun_out/dbg_clock.log ;;
pmc)
  for WL in ${PMC_WL:-config2 ref2level config5}; do
    for PASS in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum" \
                "sq:SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
                "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
      N=${PASS%%:*}; C=${PASS#*:}
      OUT=$ROOT/gpurun_out/pmc3/$WL/$N; rm -rf $OUT; mkdir -p $OUT
      ( cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc -- \
          python $ROOT/bench.py --workload $WL --steps 12 --warmup 6 --repeats 1 --no-cpu-baseline --no-variants > $OUT.log 2>&1 )
      echo "pmc $WL $N rc=$?"
    done
  done
  ROUND=${ROUND:-r06} python tools/summarize_profiles.py > gpurun_out/profile_summary.txt 2>&1; tail -n 60 gpurun_out/profile_summary.txt | cut -c1-260 ;;
bnbtests)
  timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "branch_and_bound or benchmarked or config5 or synthetic_shapes or particle_counts" > gpurun_out/pytest_bnb.log 2>&1
  echo "bnbtests rc=$?"; tail -n 15 gpurun_out/pytest_bnb.log | cut -c1-400 ;;
ab)
  # A/B of environment switches: AB="NAME=VAL,NAME2=VAL2;NAME=VAL3;-" (";"-separated settings, "-" = defaults), AB_WL = workloads
  IFS=';' read -ra SETS <<< "${AB:--}"
  for WL in ${AB_WL:-config2}; do
    for SET in "${SETS[@]}"; do
      ENVS=$(echo "$SET" | tr ',' ' '); [ "$SET" = "-" ] && ENVS=""
      env $ENVS python bench.py --workload $WL --steps ${AB_STEPS:-60} --warmup 8 --repeats 3 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$WL [$SET]', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'host', d['timed_blocks']['host_enqueue_ms_per_step'], {k: v['avg_us'] for k, v in d['stages_probe'].items()}, 'flags', d['fault_flags'])
"
    done
  done ;;
fetch)
  # HBM fetch / write of config 2 only (two passes), for A/B of one kernel's traffic: FETCH_ENV="NAME=VAL ..." FETCH_TAG=_x
  for PASS in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    N=${PASS%%:*}; C=${PASS#*:}
    OUT=$ROOT/gpurun_out/pmc4${FETCH_TAG:-}/config2/$N; rm -rf $OUT; mkdir -p $OUT
    ( cd /tmp && env ${FETCH_ENV:-} timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc -- \
        python $ROOT/bench.py --workload config2 --steps 12 --warmup 6 --repeats 1 --no-cpu-baseline --no-variants > $OUT.log 2>&1 )
    echo "fetch${FETCH_TAG:-} $N rc=$?"
    python - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/**/pmc_counter_collection.csv", recursive=True)
acc, cnt = collections.Counter(), collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for k, v in acc.most_common(8):
    print("   %-42s %10.3f per launch (%d launches)  [$C]" % (k, v / cnt[k], cnt[k]))
PY
  done ;;
full)
  ( time timeout 1500 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err ) 2>&1 | tail -n 3
  echo "bench full rc=$?"; python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/bench_full.json") if l.startswith("{")][-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "host", d["timed_blocks"]["host_enqueue_ms_per_step"], "groups", d["config"]["particle_groups_per_gpu"])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "measured_hbm_frac", "avg_launch_us")}, "whole", {k: round(v, 4) if isinstance(v, float) else v for k, v in d["roofline"]["whole_step"].items()})
for k, v in d.get("variants", {}).items():
    if k == "p_sweep":
        print(" p_sweep", {a: (round(b["value"]), round(b["ms_per_step"], 4)) for a, b in v.items()})
    else:
        print(" ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("value", "ms_per_step", "brute_force_ms_per_step", "vs_brute_force", "scans_per_sec", "ms_per_matchScan", "ms_per_updateOccupancyGrid", "resamples")},
              {a: round(b, 4) for a, b in (v.get("tile_stats", {}).get("coarse", {}) or {}).items() if a.startswith("kept")})
print("cpu", {k: v for k, v in d.get("cpu_baseline", {}).items() if k in ("value", "cores", "single_core_value", "spread")})
PY
  ;;
*) echo "unknown stage $ST" ;;
esac
done
echo "== done $(date)"
