#!/bin/bash
# Variant of the bulk kernel only: conv1x.hip (or FILES="a b") compiled with EXTRA flags, every other object taken from the
# product build (make -C interflow_amd/csrc first) -> gpurun_exp/libinterflow_hip_$SUFFIX.so (IFL_LIB_PATH).  Not product.
set -e
cd "$(dirname "$0")/.."
FILES=${FILES:-conv1x}
mkdir -p gpurun_exp/obj_$SUFFIX
objs=""
for f in api layout pointwise conv1 conv1x conv3 conv3w convg rrdb train_glue train_fused; do
  if [[ " $FILES " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off $EXTRA -c interflow_amd/csrc/$f.hip -o gpurun_exp/obj_$SUFFIX/$f.o &
    objs="$objs gpurun_exp/obj_$SUFFIX/$f.o"
  else
    objs="$objs interflow_amd/csrc/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_exp/libinterflow_hip_${SUFFIX}.so $objs
rm -rf gpurun_exp/obj_$SUFFIX
