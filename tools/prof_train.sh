#!/bin/bash
# kernel statistics of the training bench under rocprofv3: tools/prof_train.sh <tag> [bench_train.py flags]; writes gpurun_out/<tag>_stats.txt
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -- python3 $root/tools/bench_train.py "$@" --steps 8 --warmup 4 > $root/gpurun_out/$tag.log 2>&1 || exit 1
cd $root
python tools/trace_timeline.py gpurun_out/$tag > gpurun_out/${tag}_timeline.txt
python - "$tag" <<PY
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(f"gpurun_out/{tag}_stats.txt", "w") as out:
    for r in rows[:70]:
        print(r["Name"][:100].ljust(100), r["Calls"], r["TotalDurationNs"], r["AverageNs"], file=out)
PY
rm -rf gpurun_out/$tag/
