#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_halves; mkdir -p $O
cd $R
for lim in 262144 100000000 262144 100000000; do
  VNR_AMD_COMPACT_SMALL_LIMIT=$lim timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-alone --no-brick-off --train-steps 300 > $O/c$lim.json 2> $O/c$lim.err && python tools/bench_line.py compact_limit_$lim < $O/c$lim.json
done
