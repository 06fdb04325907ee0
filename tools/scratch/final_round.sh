#!/bin/bash
# end-of-session measurements on one box: GPU suite, smoke, default bench line, kernel traces, C2 line, modes 8 / 11 / 14
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
python -m pytest tests -x -q -m gpu > $O/s3_final_gpu_tests.log 2>&1; tail -2 $O/s3_final_gpu_tests.log && grep -q " passed" $O/s3_final_gpu_tests.log && ! grep -q failed $O/s3_final_gpu_tests.log &&
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 &&
python bench.py > $O/r01_l_bench.json 2> $O/r01_l_bench.err && python tools/bench_line.py default < $O/r01_l_bench.json &&
bash tools/run_trace.sh r01_l > $O/s3_trace_l.log 2>&1 && bash tools/run_trace.sh r01_l_one_stream VNR_AMD_RENDER_HALVES=1 > $O/s3_trace_l1.log 2>&1 &&
python bench.py --size 128 --fb 512 --levels 8 --features 8 --log2-hashmap-size 19 --hidden-layers 2 --per-level-scale 2 --train-steps 10000 --no-cpu-baseline > $O/s3_bench_c2.json 2> $O/s3_bench_c2.err && python tools/bench_line.py c2 < $O/s3_bench_c2.json &&
for m in 8 11 14; do python bench.py --mode $m --no-cpu-baseline --no-psnr --no-alone > $O/s3_bench_mode$m.json 2> $O/s3_bench_mode$m.err && python tools/bench_line.py mode$m < $O/s3_bench_mode$m.json || exit 1; done
