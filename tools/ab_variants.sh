#!/bin/bash
# Builds tuning variants of libdfx.so into gpurun_build/ (they travel to the GPU box with the snapshot).
# Usage: tools/ab_variants.sh name1:"-DDFX_X=1 ..." name2:"..."
set -e
cd "$(dirname "$0")/../deepfactors_amd/csrc"
mkdir -p ../../gpurun_build
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 $flags -shared \
     dfx_sfm_step.hip dfx_misc_kernels.hip dfx_graph.hip -x hip dfx_api.cpp dfx_comm.cpp -ldl -o ../../gpurun_build/libdfx_$name.so 2>&1 | grep -E "error" || true
  echo "built $name ($flags)"
done
