#!/bin/bash
# A/B runs of bench.py under different environment settings: bash tools/ab.sh "VAR=1 VAR2=3" "VAR=2" ...
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('value %.1f patches/s  ms/step %.2f  dom %.3f ms frac %.3f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac']))
" || exit 1
done
