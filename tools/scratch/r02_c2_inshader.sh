#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_inshader; mkdir -p $O
cd $R
C2="--size 128 --fb 512 --levels 8 --features 8 --log2-hashmap-size 19 --hidden-layers 2 --per-level-scale 2 --train-steps 2000 --no-cpu-baseline --no-psnr --no-alone --no-brick-off"
for k in 0 1 0 1; do
  VNR_AMD_IN_SHADER=$k timeout -k 10 200 python bench.py $C2 --mode 6 > $O/c2_k$k.json 2> $O/c2_k$k.err && python tools/bench_line.py c2_inshader$k < $O/c2_k$k.json || tail -3 $O/c2_k$k.err
done
timeout -k 10 200 python bench.py $C2 --mode 5 > $O/c2_m5.json 2> $O/c2_m5.err && python tools/bench_line.py c2_mode5 < $O/c2_m5.json
