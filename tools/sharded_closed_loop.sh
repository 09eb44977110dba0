# the closed loop (bench.py --workload config3, 64 particles x 910 Intel scans) unsharded and as a ONE-rank sharded filter over RCCL:
#   bash tools/sharded_closed_loop.sh      (through gpurun)
cd ${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run () {
  env "$@" python bench.py --workload config3 --particles ${P:-64} --resample-every ${RE:-100} 2>/tmp/err.txt | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$*', '->', round(d['scans_per_sec'],1), 'scans/s;', d['config']['parallelism'], d['filter_stats'])
"; grep -v "amdgpu.ids\|c10d\|^$" /tmp/err.txt | tail -3
}
for i in 1 2; do
  run A=0
  run SLAM2D_FORCE_DIST=1
  run SLAM2D_FORCE_DIST=1 SLAM2D_FILTER_GROUPS=1
  run SLAM2D_FORCE_DIST=1 SLAM2D_FILTER_GROUPED1=0 SLAM2D_FILTER_GROUPS=1
  run SLAM2D_FILTER_GROUPS=2
done
