export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { python bench.py --workload $1 --steps 60 --warmup 8 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$2 $1', round(d['value']), 'ms/step', round(d['ms_per_step'],4), {k: v['avg_us'] for k, v in d['stages_probe'].items()})
"; }
for L in base u2 u8 base; do cp ab_libs/$L.so slam-2d-lidar-scan_amd/libslam2d_hip.so; run config2 $L; done
cp ab_libs/base.so slam-2d-lidar-scan_amd/libslam2d_hip.so
run ref2level auto; SLAM2D_BNB=1 run ref2level bnb1; SLAM2D_BNB=0 run config2 bnb0
for P in 16 128 256; do python bench.py --particles $P --steps 60 --warmup 8 --no-cpu-baseline --no-variants 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('P=$P', round(d['value']), 'ms/step', round(d['ms_per_step'],4))
"; done
