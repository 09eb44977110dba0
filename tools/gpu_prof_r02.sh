#!/bin/bash
# Round-2 profiles (run through gpurun): rocprofv3 kernel stats of the bench command for three workloads, the HBM-traffic
# PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss, separate invocations) and the SQ / TCP counters of the gather kernels.
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
BENCH="--steps 100 --warmup 10 --no-cpu-baseline --no-variants"
cd /tmp
for WL in config2 ref2level config5; do
  rm -rf $O/prof_$WL
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$WL -o $WL -- python $R/bench.py --workload $WL $BENCH > $O/prof_$WL.log 2>&1
  echo "kernel stats $WL rc=$?"; tail -n 1 $O/prof_$WL.log | cut -c1-160
done
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | cut -d' ' -f1); rm -rf $O/pmc_$N
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -o pmc -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants > $O/pmc_$N.log 2>&1
  echo "pmc $N rc=$?"
done
rm -rf $O/pmc_sq
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_sq -o pmc -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants > $O/pmc_sq.log 2>&1; echo "pmc sq rc=$?"
rm -rf $O/pmc_tcp
timeout 600 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum --kernel-trace --output-format csv -d $O/pmc_tcp -o pmc -- python $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-variants > $O/pmc_tcp.log 2>&1; echo "pmc tcp rc=$?"
cd $R
find gpurun_out -name "*kernel_stats.csv" -newer tools/gpu_prof_r02.sh | head; find gpurun_out -name "*counter_collection.csv" | head
