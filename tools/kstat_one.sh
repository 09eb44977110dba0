# tools/kstat_one.sh <workload> <kernel-name-substring> [ENV=VAL ...]: average duration of the matching kernels (single particle group)
WL=$1; PAT=$2; shift 2
export TMPDIR=/tmp SLAM2D_BENCH_GROUPS=1
OUT=/tmp/kstat_one; rm -rf $OUT
( cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 30 --warmup 5 --repeats 2 --no-cpu-baseline --no-variants > $OUT.log 2>&1 )
python - <<PY
import csv, glob
f = glob.glob("$OUT/**/k_kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    if "$PAT" in r["Name"]: out.append("%s %.2f us" % (r["Name"][:34], float(r["AverageNs"]) / 1e3))
import re
ms = re.findall(r'"ms_per_step": ([0-9.]+)', open("$OUT.log").read())
print("$WL [$*]", "; ".join(out), "ms/step", ms[:1])
PY
