# tools/kstat_one.sh <workload> <kernel-name-substring> [ENV=VAL ...]: median (and mean) duration of the matching kernels
# (single particle group) from the rocprofv3 kernel trace -- the mean includes the first builds, the median is the steady state
WL=$1; PAT=$2; shift 2
export TMPDIR=/tmp SLAM2D_BENCH_GROUPS=1
OUT=/tmp/kstat_one; rm -rf $OUT
( cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 30 --warmup 5 --repeats 2 --no-cpu-baseline --no-variants > $OUT.log 2>&1 )
python - <<PY
import csv, glob, collections, statistics, re
f = glob.glob("$OUT/**/k_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if n.replace("void ", "").startswith("k_") and "$PAT" in n:
        d[n.replace("void ", "").split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ms = re.findall(r'"ms_per_step": ([0-9.]+)', open("$OUT.log").read())
print("$WL [$*] ms/step", ms[:1])
tot = 0.0
for n, v in sorted(d.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1])):
    if len(v) < 20: continue
    print("   %-28s n %4d  median %7.2f us  mean %7.2f us" % (n[:28], len(v), statistics.median(v), sum(v) / len(v)))
PY
