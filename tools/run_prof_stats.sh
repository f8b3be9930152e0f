#!/bin/bash
# GPU session: rocprofv3 kernel trace of the bench (40 steps), condensed stats + per-pass timeline.  gpurun_out/$1/
TAG=${1:-r03s}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d $O/raw_stats -- python3 bench.py --no-cpu-baseline --no-secondary --steps 40 > $O/stats.log 2>&1 || { tail -20 $O/stats.log; exit 1; }
grep '^{' $O/stats.log > $O/bench_under_rocprof.json
python3 tools/flow_timeline.py $O/raw_stats 2 > $O/flow_timeline.txt 2>&1
python3 tools/prof_summary.py $O/raw_stats $O/kernel_stats.csv --rm
head -30 $O/kernel_stats.csv | cut -c1-150
cat $O/flow_timeline.txt | head -60
