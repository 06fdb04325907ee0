#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_ranks; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_render.py tests/test_gpu_fullsize.py -x -q -m gpu -k "n_iters_or_tile or unpinned or interleaved_shares or gradient_shading_ground or single_shade_heuristic_ground" > $O/tests_${1:-a}.txt 2>&1 || { tail -30 $O/tests_${1:-a}.txt; exit 1; }
tail -3 $O/tests_${1:-a}.txt
export SHARE_PIPELINED=1 SHARE_PARTS=${SHARE_PARTS:-8,4,2,1} SHARE_FRAMES=60 SHARE_REPS=2
SHARE_CONFIGS="ranks:VNR_AMD_MARCH_RANKS=1;noranks:VNR_AMD_MARCH_RANKS=0" timeout -k 10 500 python tools/share_probe.py 2>&1 | grep share | tee $O/share_${1:-a}.txt
