#!/bin/bash
# PMC counter passes over a short bench run (one rocprofv3 invocation per counter group).
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
WL=${1:-config2}
cd /tmp
rocprofv3 -L > $ROOT/gpurun_out/pmc/counters_list.txt 2>&1
run_pass () {
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc/$name -o $name -- \
      python $ROOT/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/pmc/$name.log 2>&1
  echo "pass $name rc=$?"
}
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run_pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run_pass grbm GRBM_GUI_ACTIVE
cd $ROOT
ls -R gpurun_out/pmc | head -40
python - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob('gpurun_out/pmc/*/')):
    files = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    for f in files:
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row.get('Kernel_Name', '')[:40]
            agg[k][row['Counter_Name']] += float(row['Counter_Value']); cnt[(k, row['Counter_Name'])] += 1
        print('==', f)
        for k in agg:
            if k.startswith(('k_', 'void k_')):
                print(' ', k, {c: round(v / cnt[(k, c)], 1) for c, v in agg[k].items()}, 'n=', max(cnt[(k, c)] for c in agg[k]))
PY
