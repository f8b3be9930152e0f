#!/bin/bash
# GPU session: ingest-rate variants of the bulk kernel (per-wave loop times) + bare MFMA shapes.  Output: gpurun_out/exp1.log
mkdir -p gpurun_out
L=gpurun_out/exp1.log
: > $L
echo "== bare MFMA shapes" >> $L
timeout -k 10 120 gpurun_exp/mfma_shape >> $L 2>&1 || exit 1
for v in t t_nowin t_now t_nomfma t_nomfma_nowin t_nomfma_now; do
  echo "== variant $v" >> $L
  IFL_LIB_PATH=gpurun_exp/libinterflow_hip_$v.so timeout -k 10 200 python tools/exp_x4_timeline.py >> $L 2>&1 || exit 1
done
echo done >> $L
