#!/bin/bash
# sweep of scheduling knobs on the bench workload (diagnostics): usage tools/knob_sweep.sh "<ENV=VAL ...>" ...   (one quoted set per run)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
out=gpurun_out/knob_sweep.log
for set in "$@"; do
  ( for kv in $set; do export "$kv"; done
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-psnr --steps 20 --warmup 5 2>&1 | python tools/bench_line.py "[$set]" ) >> $out 2>&1 || exit 1
done
cat $out
