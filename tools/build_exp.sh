#!/bin/bash
# Experiment build: the library with -DIFL_EXP_TIME (per-workgroup timestamps in the bulk kernels) as
# gpurun_exp/libinterflow_hip_exp$SUFFIX.so (EXTRA = more -D flags); load it with IFL_LIB_PATH=gpurun_exp/libinterflow_hip_exp.so.  Not the product build.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_exp
for f in api layout pointwise conv1 conv1x conv3 convg rrdb; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DIFL_EXP_TIME $EXTRA -c interflow_amd/csrc/$f.hip -o gpurun_exp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libinterflow_hip_exp${SUFFIX}.so gpurun_exp/*.o
rm -f gpurun_exp/*.o
