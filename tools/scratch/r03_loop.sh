#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_loop; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_render.py tests/test_golden.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not in_shader" > $O/tests_${1:-a}.txt 2>&1 || { tail -30 $O/tests_${1:-a}.txt; exit 1; }
tail -3 $O/tests_${1:-a}.txt
SHARE_FRAMES=80 bash tools/r03_ab.sh oldloop newloop 3
