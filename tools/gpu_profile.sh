#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 timing profile of the default bench.py command plus the PMC passes
# (one counter block per run, as rocprofv3 does not split passes), each condensed by tools/prof_summary.py into a
# small CSV under gpurun_out/<tag>/ -- copy the ones to keep into profiles/.
#   usage: [SKIP_STATS=1] bash tools/gpu_profile.sh <tag> [bench args for the PMC passes]
set -o pipefail
TAG=${1:-prof}
shift
PMC_ARGS=${*:---steps 4 --warmup 2 --no-cpu-baseline --no-secondary}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p "$O"

[ -n "$SKIP_STATS" ] || { timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d "$O/raw_stats" -- python3 bench.py --no-cpu-baseline --no-secondary --steps 40 > "$O/stats.log" 2>&1 \
  && python3 tools/prof_summary.py "$O/raw_stats" "$O/kernel_stats.csv" --rm || { tail -20 "$O/stats.log"; exit 1; }
grep '^{' "$O/stats.log" > "$O/bench_under_rocprof.json"; }

pass() {   # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace -f csv -d "$O/raw_$name" -- python3 bench.py $PMC_ARGS > "$O/$name.log" 2>&1 \
    && python3 tools/prof_summary.py "$O/raw_$name" "$O/pmc_$name.csv" --rm || { tail -20 "$O/$name.log"; return 1; }
}
pass fetch FETCH_SIZE \
 && pass write WRITE_SIZE \
 && pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
 && pass grbm GRBM_GUI_ACTIVE \
 && pass tcc TCC_HIT_sum TCC_MISS_sum
rc=$?
rm -f "$O"/*.log.tmp
ls -la "$O"
exit $rc
