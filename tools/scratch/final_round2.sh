#!/bin/bash
# second half of tools/final_round.sh: C2 line, modes 8 / 11 / 14
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
python bench.py --size 128 --fb 512 --levels 8 --features 8 --log2-hashmap-size 19 --hidden-layers 2 --per-level-scale 2 --train-steps 10000 --no-cpu-baseline > $O/s3_bench_c2.json 2> $O/s3_bench_c2.err && python tools/bench_line.py c2 < $O/s3_bench_c2.json &&
for m in 8 11 14; do python bench.py --mode $m --no-cpu-baseline --no-psnr --no-alone > $O/s3_bench_mode$m.json 2> $O/s3_bench_mode$m.err && python tools/bench_line.py mode$m < $O/s3_bench_mode$m.json || exit 1; done
