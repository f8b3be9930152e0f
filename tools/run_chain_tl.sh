#!/bin/bash
# GPU session: kernel trace of the bench, per-queue chain view.  gpurun_out/$1/
TAG=${1:-ctl}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace -f csv -d $O/raw -- python3 bench.py --no-cpu-baseline --no-secondary --steps 12 --warmup 3 > $O/log.txt 2>&1 || { tail -20 $O/log.txt; exit 1; }
grep '^{' $O/log.txt | cut -c1-200
python3 tools/chain_timeline.py $O/raw 46 > $O/chain_timeline.txt 2>&1
python3 tools/chain_timeline.py $O/raw raw 0 13 > $O/raw_listing.txt 2>&1
rm -rf $O/raw
cat $O/chain_timeline.txt
