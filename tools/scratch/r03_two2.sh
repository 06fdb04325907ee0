#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_two; mkdir -p $O
cd $R
{ for q in 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q VNR_AMD_RENDERER_OWN_STREAM=1 timeout -k 10 400 python tools/two_renderers.py 8,1 1,2,3 2>&1 | grep "share 1"
done; } | tee $O/two_${1:-b}.txt
