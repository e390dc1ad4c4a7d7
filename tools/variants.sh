#!/bin/bash
# Builds tuning variants of libdfx.so into gpurun_build/ (they travel to the GPU box with the snapshot; DFX_LIB=gpurun_build/libdfx_<name>.so selects one).
#   tools/variants.sh [--misc] name1:"-DDFX_X=1 ..." name2:"..."
# --misc: the variants differ in dfx_misc_kernels.hip only (the other objects are reused: seconds instead of a minute per variant).
set -e
cd "$(dirname "$0")/../deepfactors_amd/csrc"
mkdir -p ../../gpurun_build
misc=0; [ "${1:-}" = "--misc" ] && { misc=1; shift; }
[ $misc = 1 ] && make -s -j8 >/dev/null
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  if [ $misc = 1 ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $flags -c dfx_misc_kernels.hip -o /tmp/misc_$name.o 2>&1 | grep -E "error" || true
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_build/libdfx_$name.so dfx_sfm_step.o /tmp/misc_$name.o dfx_graph.o dfx_api.o dfx_comm.o -ldl
  else
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 $flags -shared \
       dfx_sfm_step.hip dfx_misc_kernels.hip dfx_graph.hip -x hip dfx_api.cpp dfx_comm.cpp -ldl -o ../../gpurun_build/libdfx_$name.so 2>&1 | grep -E "error" || true
  fi
  echo "built $name ($flags)"
done
