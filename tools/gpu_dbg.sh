cp slam-2d-lidar-scan_amd/libslam2d_hip.so /tmp/orig.so
cp slam-2d-lidar-scan_amd/libslam2d_dbg.so slam-2d-lidar-scan_amd/libslam2d_hip.so
for D in 0 1 2; do
  for WL in config2 ref2level; do
    UPD_DBG=$D python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('dbg $D $WL', 'ms/step', round(d['ms_per_step'],3), d['stages_warmup']['k_grid_update'])
"
  done
done
cp /tmp/orig.so slam-2d-lidar-scan_amd/libslam2d_hip.so
