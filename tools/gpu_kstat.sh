# per-kernel durations of one bench workload: tools/gpu_kstat.sh <workload> [extra bench args]   (run through gpurun)
WL=${1:-config2}; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstat_$WL
rm -rf $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 100 --warmup 5 --no-cpu-baseline --no-variants "$@" > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.5:
        print("  %-58s calls %4s avg %8.2f us  %5s%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
