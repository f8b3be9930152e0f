#!/bin/bash
# GPU session: parity tests, stress table, bench and per-wave timelines of the bulk-kernel variants.  Output: gpurun_out/exp2*.log
mkdir -p gpurun_out
L=gpurun_out/exp2.log
: > $L
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/exp2_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/exp2_pytest.log >> $L
[ $rc -eq 0 ] || { echo "pytest rc $rc" >> $L; exit 1; }
echo "== stress table" >> $L
timeout -k 10 200 python tools/exp_stress.py >> $L 2>&1 || exit 1
echo "== bench" >> $L
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --breakdown >> $L 2>&1 || exit 1
for v in $VARIANTS; do
  echo "== variant $v" >> $L
  IFL_LIB_PATH=gpurun_exp/libinterflow_hip_$v.so timeout -k 10 200 python tools/exp_x4_timeline.py 2>&1 | grep -v amdgpu.ids | awk 'NR==1||NR==5||NR==9||NR==13' >> $L || exit 1
done
echo done >> $L
