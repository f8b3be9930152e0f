#!/bin/bash
# GPU session at the end of a block of work: the whole GPU test suite, the default bench line, kernel statistics, the PMC passes
# and the per-kernel roofline table.  Everything under gpurun_out/$TAG/.
TAG=${1:-r03}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q -s --timeout 300 --timeout-method thread > $O/pytest.log 2>&1; rc=$?
tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
grep -h "^stress\|buckets\|relative gradient" $O/pytest.log > $O/stress_parity.txt
timeout -k 10 400 python bench.py > $O/bench_default.log 2>&1 || { tail -5 $O/bench_default.log; exit 1; }
grep '^{' $O/bench_default.log > $O/bench_default.json
python - <<PY
import json; r = json.load(open("$O/bench_default.json"))
print("bench: %.1f patches/s, %.2f ms/step, dominant %.3f ms frac %.3f; reuse %.1f; alone %.3f; cpu %.2f" % (r["value"], r["ms_per_step"], r["roofline"]["avg_launch_ms"], r["roofline"]["frac"], r["cond_reuse"]["value"], r["serial_schedule"]["dominant_frac_alone"], r["cpu_baseline"]["value"]))
PY
bash tools/gpu_profile.sh $TAG > $O/profile.log 2>&1 || { tail -20 $O/profile.log; exit 1; }
timeout -k 10 300 python tools/kernel_roofline.py --out $O/kernel_roofline.json > $O/kernel_roofline.log 2>&1 || { tail -5 $O/kernel_roofline.log; exit 1; }
ls $O
