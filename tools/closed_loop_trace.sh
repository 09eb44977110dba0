cd /tmp && export TMPDIR=/tmp
for G in 1 2; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/cltrace$G
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cltrace$G -o k -- python $GRAFT_REPO_ROOT/tools/closed_loop_groups.py $G > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/cltrace$G -name "k_kernel_trace.csv" | head -1)
echo "G=$G"; python $GRAFT_REPO_ROOT/tools/trace_overlap.py $f
done
