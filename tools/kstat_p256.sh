cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstat_p256; rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --particles 256 --steps 30 --warmup 5 --repeats 2 --no-cpu-baseline --no-variants > $OUT.log 2>&1
python - <<PY
import csv, glob
fs = glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(fs[0])):
    if float(r["Percentage"]) > 1.0:
        print("  %-50s calls %5s avg %8.2f us  %5s%%" % (r["Name"].replace("void ","")[:50], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
grep -o '"ms_per_step": [0-9.]*' $OUT.log | head -1
