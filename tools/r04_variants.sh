#!/bin/bash
# Builds variants of libdfx.so that differ in dfx_misc_kernels.hip only (the other objects are reused) into gpurun_build/.
# Usage: tools/r04_variants.sh name1:"-DDFX_X=1 ..." name2:"..."
set -e
cd "$(dirname "$0")/../deepfactors_amd/csrc"
make -s -j8 >/dev/null
mkdir -p ../../gpurun_build
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize $flags -c dfx_misc_kernels.hip -o /tmp/misc_$name.o 2>&1 | grep -E "error" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_build/libdfx_$name.so dfx_sfm_step.o /tmp/misc_$name.o dfx_graph.o dfx_api.o dfx_comm.o -ldl
  echo "built $name ($flags)"
done
