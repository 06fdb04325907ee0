#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_ooc; mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_ooc.py tests/test_gpu_dist.py -m gpu -x -q -k "ooc or out_of_core" 2>&1 | tail -2
timeout -k 10 300 python tools/ooc_bench.py 1024 1024 2048 1024 16384 300 1 2>&1 | tee $O/ooc_1024.txt | tail -3
timeout -k 10 300 python tools/ooc_bench.py 1024 1024 2048 128 16384 400 0 2>&1 | tee $O/ooc_128.txt | tail -2
