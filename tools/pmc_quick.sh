#!/bin/bash
# Quick per-kernel counters + standalone durations of one bench command (round 6):
#   PS="512" WL=config2 ENVS="SLAM2D_BOUND_LDS=1" bash tools/pmc_quick.sh  -> gpurun_out/pmcq/<wl>_p<P>/...; prints the table of tools/pmc6_summary.py
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
ROOT=${GRAFT_REPO_ROOT:-$PWD}
WL=${WL:-config2}
for P in ${PS:-512}; do
  for PASS in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum" \
              "sq:SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" \
              "sq2:SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" \
              "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
              "lds:SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
    N=${PASS%%:*}; C=${PASS#*:}
    case " ${PASSES:-fetch write tcc sq sq2 tcp lds} " in *" $N "*) ;; *) continue ;; esac
    OUT=$ROOT/gpurun_out/pmc6/${WL}_p$P${TAG:-}/$N; rm -rf $OUT; mkdir -p $OUT
    ( cd /tmp && env ${ENVS:-} timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o pmc -- \
        python $ROOT/bench.py --workload $WL --particles $P --steps 12 --warmup 6 --repeats 1 --no-cpu-baseline --no-variants > $OUT.log 2>&1 )
    echo "pmc $WL p$P $N rc=$?"
  done
  OUT=$ROOT/gpurun_out/kstat_${WL}_p$P${TAG:-}; rm -rf $OUT
  ( cd /tmp && env ${ENVS:-} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- \
      python $ROOT/bench.py --workload $WL --particles $P --steps 40 --warmup 5 --repeats 3 --no-cpu-baseline --no-variants > $OUT.log 2>&1 )
  cd $ROOT && python tools/pmc6_summary.py ${WL}_p$P${TAG:-} --alone
done
