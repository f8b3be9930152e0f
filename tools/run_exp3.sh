#!/bin/bash
# GPU session: quick parity of the product build, bench, then per-wave timelines of bulk-kernel variants.  gpurun_out/exp3.log
mkdir -p gpurun_out
L=gpurun_out/exp3.log
: > $L
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread -k "golden or algorithms or stress or batch32" > gpurun_out/exp3_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/exp3_pytest.log >> $L
[ $rc -eq 0 ] || { echo "pytest rc $rc" >> $L; exit 1; }
echo "== bench" >> $L
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --breakdown 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('value %.1f patches/s  ms/step %.2f  dom %.3f ms frac %.3f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'])); print(r.get('serial_schedule')); print(r.get('kernels_ms_per_step'))
" >> $L || exit 1
for v in $VARIANTS; do
  echo "== variant $v" >> $L
  IFL_LIB_PATH=gpurun_exp/libinterflow_hip_$v.so timeout -k 10 200 python tools/exp_x4_timeline.py 2>&1 | grep -v amdgpu.ids | awk 'NR==1||NR==5||NR==9||NR==13' | cut -c1-260 >> $L || exit 1
done
echo done >> $L
