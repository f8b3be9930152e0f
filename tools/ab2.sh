#!/bin/bash
# as ab.sh, with the per-kernel sums of one step (secondary fields): bash tools/ab2.sh "VAR=1" "VAR=2" ...
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python bench.py --no-cpu-baseline --no-hires --steps 20 --warmup 3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); k = r['kernels_ms_per_step']
        print('value %.1f patches/s  ms/step %.2f  dom %.3f ms frac %.3f | l1 %.2f l2 %.2f conv3f %.2f finish %.2f conv3 %.2f+%.2f anic %.2f | serial %.2f' % (r['value'], r['ms_per_step'], r['roofline']['avg_launch_ms'], r['roofline']['frac'], k['conv1_ft_l1'], k['conv1_ft_l2'], k['conv3_f'], k['conv1_finish'], k['conv3_enc'], k['conv3_dec'], k['anic'], r['serial_schedule']['ms_per_step']))
" || exit 1
done
