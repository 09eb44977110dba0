export TMPDIR=/tmp SLAM2D_FORCE_DIST=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstat_dist
rm -rf $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-variants > $OUT.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/k_kernel_trace.csv", recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
prev=None
for r in rows[-26:]:
    st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
    print("%-46s dur %7.2f gap %6.2f" % (r["Kernel_Name"][:46], (en-st)/1e3, (st-prev)/1e3 if prev else 0)); prev=en
PY
