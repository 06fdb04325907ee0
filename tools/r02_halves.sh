#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_halves; mkdir -p $O
cd $R
for pr in 0 1 2 0 1 2; do
  VNR_AMD_PART_PRIORITY=$pr timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-alone --no-brick-off --train-steps 300 > $O/p$pr.json 2> $O/p$pr.err && python tools/bench_line.py prio$pr < $O/p$pr.json; grep "part stream" $O/p$pr.err | head -1
done
