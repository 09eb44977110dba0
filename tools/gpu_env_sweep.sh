# tools/gpu_env_sweep.sh VAR v1 v2 ...  : per-kernel rocprof averages of both workloads for each value of an env knob
VAR=$1; shift
for V in "$@"; do
  export $VAR=$V
  for WL in config2 ref2level; do
    echo "== $VAR=$V $WL"
    bash tools/gpu_kstat.sh $WL 2>&1 | grep -E "k_blur_clamp|k_tile_triage|k_occ_scatter|k_endpoints|k_grid_update|k_sweep" | cut -c1-110
  done
done
