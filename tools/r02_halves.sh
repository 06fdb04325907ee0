#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_halves; mkdir -p $O
cd $R
for h in 2 3 4 2 3 4; do
  VNR_AMD_RENDER_HALVES=$h timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-alone --no-brick-off --train-steps 300 > $O/h$h.json 2> $O/h$h.err && python tools/bench_line.py h$h < $O/h$h.json
done
