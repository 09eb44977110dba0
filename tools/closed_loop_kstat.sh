# per-kernel durations of the closed loop (BASELINE config 3: 64 particles, the 910-scan Intel log) -- run through gpurun
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstat_config3
rm -rf $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/tools/host_profile.py > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)[0]
tot = 0
for r in csv.DictReader(open(f)):
    tot += float(r["TotalDurationNs"])
    if float(r["Percentage"]) > 0.8:
        print("  %-58s calls %5s avg %8.2f us total %8.2f ms %5s%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
print("  all kernels: %.1f ms" % (tot / 1e6))
PY
