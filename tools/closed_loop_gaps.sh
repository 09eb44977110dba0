# where the closed loop's stream idles (BASELINE config 3, one group): run through gpurun
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/clgaps${TAG:-}
rm -rf $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o k -- python $GRAFT_REPO_ROOT/tools/closed_loop_groups.py 1 > $OUT.log 2>&1
tail -n 2 $OUT.log
python $GRAFT_REPO_ROOT/tools/trace_gaps.py $(find $OUT -name "k_kernel_trace.csv" | head -1) 0.7 ${SPLIT:-}
