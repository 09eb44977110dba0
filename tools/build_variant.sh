#!/bin/bash
# build an A/B variant of the library into ab_libs/<name>.so:  tools/build_variant.sh <name> [-DMACRO=VALUE ...]
cd "$(dirname "$0")/.." && mkdir -p ab_libs
N=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -pthread "$@" -I include slam-2d-lidar-scan_amd/csrc/slam2d.hip -o ab_libs/$N.so && echo built ab_libs/$N.so
