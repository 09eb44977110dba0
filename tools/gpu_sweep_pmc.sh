#!/bin/bash
# L1 / texture-addresser counters of the sweep (separate rocprofv3 passes):  bash tools/gpu_sweep_pmc.sh [workload]
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
WL=${1:-config2}
mkdir -p $ROOT/gpurun_out/sweep_pmc
cd /tmp
pass () {
  name=$1; shift
  rm -rf $ROOT/gpurun_out/sweep_pmc/$name
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/gpurun_out/sweep_pmc/$name -o p -- \
      python $ROOT/bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-variants > $ROOT/gpurun_out/sweep_pmc/$name.log 2>&1
  echo "pass $name rc=$?"
}
# (a pass with TA_TA_BUSY_sum / TA_BUFFER_* hung the run on this pool: left out)
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
pass sq SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU
cd $ROOT
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/sweep_pmc/*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        if k.startswith("k_sweep"):
            print(k, {c: round(sum(v[len(v)//3:]) / max(1, len(v[len(v)//3:])), 1) for c, v in d.items()})
PY
