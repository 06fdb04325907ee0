#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_tfn; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_render.py -x -q -m gpu > $O/tests_${1:-a}.txt 2>&1 || { tail -30 $O/tests_${1:-a}.txt; exit 1; }
tail -3 $O/tests_${1:-a}.txt
export SHARE_PIPELINED=1 SHARE_PARTS=1,8 SHARE_FRAMES=60 SHARE_REPS=3
timeout -k 10 300 python tools/share_probe.py 2>&1 | grep share | tee $O/share_${1:-a}.txt
