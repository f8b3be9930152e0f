set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/r02_train.json; : > $out
for c in "--frozen-rrdb --no-host-sync --prefetch-encoder" "--frozen-rrdb --late-check --prefetch-encoder" "--frozen-rrdb --no-host-sync" "--frozen-rrdb" "" "--no-host-sync" "--frozen-rrdb --consistency" "--consistency"; do
  timeout -k 10 300 python -u tools/bench_train.py $c --steps 30 --warmup 5 2>/dev/null | tail -1 >> $out
  echo "done [$c]"
done
timeout -k 10 100 python tools/host_time_train.py > gpurun_out/r02_train_host_time.txt 2>/dev/null
timeout -k 10 100 python tools/phase_events.py 2>/dev/null > gpurun_out/r02_train_phases.txt
