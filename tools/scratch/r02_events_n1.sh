#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_halves; mkdir -p $O
cd $R
for v in "" "--no-kernel-events" "" "--no-kernel-events"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-alone --no-brick-off --train-steps 300 $v > $O/ev.json 2> $O/ev.err && python tools/bench_line.py "events${v:-_on}" < $O/ev.json
done
