# A/B of alternative builds of the library: tools/gpu_ab.sh <lib1.so> <lib2.so> ...   (run through gpurun)
cp slam-2d-lidar-scan_amd/libslam2d_hip.so /tmp/orig.so
for LIB in "$@"; do
  cp $LIB slam-2d-lidar-scan_amd/libslam2d_hip.so
  for WL in config2 ref2level; do
    python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('$LIB $WL', round(d['value']), 'ms/step', round(d['ms_per_step'],3), {k: v['avg_us'] for k, v in d['stages_probe'].items()})
"
  done
done
cp /tmp/orig.so slam-2d-lidar-scan_amd/libslam2d_hip.so
