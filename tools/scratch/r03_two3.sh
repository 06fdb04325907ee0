#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_two; mkdir -p $O
cd $R
{ echo "== the march / packing chain alone (evaluation skipped), GPU_MAX_HW_QUEUES=16"
  GPU_MAX_HW_QUEUES=16 VNR_AMD_DEBUG_SKIP_EVAL=1 VNR_AMD_RENDERER_OWN_STREAM=1 timeout -k 10 400 python tools/two_renderers.py 8,1 1,2,3 2>&1 | grep "share 1"; } | tee $O/two_${1:-c}.txt
