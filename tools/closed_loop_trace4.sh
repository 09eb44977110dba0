# kernel trace of the closed loop in G groups (default 4): per-kernel averages, per-queue sums, one scan's timeline
cd /tmp && export TMPDIR=/tmp
for G in ${@:-4}; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/cltrace$G
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cltrace$G -o k -- python $GRAFT_REPO_ROOT/tools/closed_loop_groups.py $G > $GRAFT_REPO_ROOT/gpurun_out/cltrace$G.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/cltrace$G.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/cltrace$G -name "k_kernel_trace.csv" | head -1)
echo "G=$G"; python $GRAFT_REPO_ROOT/tools/trace_scan.py $f > $GRAFT_REPO_ROOT/gpurun_out/cltrace$G.txt; head -3 $GRAFT_REPO_ROOT/gpurun_out/cltrace$G.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/cltrace$G
done
