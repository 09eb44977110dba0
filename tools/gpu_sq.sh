export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/sq_r5; rm -rf $OUT; mkdir -p $OUT
cd /tmp && timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT -o pmc -- python $ROOT/bench.py --workload ${1:-config2} --steps 12 --warmup 6 --repeats 1 --no-cpu-baseline --no-variants > $OUT.log 2>&1
python - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/**/pmc_counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.Counter()); cnt = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:30]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
for k in acc:
    n=max(cnt[k],1)
    print("%-32s launches %4d " % (k,n), {c: round(v/n) for c,v in acc[k].items()})
PY
