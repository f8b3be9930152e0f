#!/bin/bash
# Variant build for A/B runs: the product sources with extra -D flags (EXTRA) as gpurun_exp/libinterflow_hip_$SUFFIX.so;
# load it with IFL_LIB_PATH=gpurun_exp/libinterflow_hip_$SUFFIX.so.  Not the product build.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_exp/obj_$SUFFIX
for f in api layout pointwise conv1 conv1x conv3 conv3w convg rrdb train_glue train_fused; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $EXTRA -c interflow_amd/csrc/$f.hip -o gpurun_exp/obj_$SUFFIX/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libinterflow_hip_${SUFFIX}.so gpurun_exp/obj_$SUFFIX/*.o
rm -rf gpurun_exp/obj_$SUFFIX
