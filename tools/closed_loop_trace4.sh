# kernel trace of the closed loop in G groups (default 4): per-kernel averages and a 480 us stretch per hardware queue
cd /tmp && export TMPDIR=/tmp
for G in ${@:-4}; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/cltrace$G
REPS=2 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/cltrace$G -o k -- python $GRAFT_REPO_ROOT/tools/closed_loop_groups.py $G > $GRAFT_REPO_ROOT/gpurun_out/cltrace$G.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/cltrace$G.log
f=$(find $GRAFT_REPO_ROOT/gpurun_out/cltrace$G -name "k_kernel_trace.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/trace_scan.py $f > $GRAFT_REPO_ROOT/gpurun_out/cltrace$G.txt; head -2 $GRAFT_REPO_ROOT/gpurun_out/cltrace$G.txt | cut -c1-600
rm -rf $GRAFT_REPO_ROOT/gpurun_out/cltrace$G
done
