#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_two; mkdir -p $O
cd $R
{ VNR_AMD_RENDERER_OWN_STREAM=1 timeout -k 10 400 python tools/two_renderers.py 1,8 1,2,3 2>&1 | grep "share 1"
  echo "== the 1/8 share on two ray parts per renderer"
  VNR_AMD_RENDERER_OWN_STREAM=1 VNR_AMD_SMALL_SHARE_PARTS=2 timeout -k 10 300 python tools/two_renderers.py 8 1,2,3,4 2>&1 | grep "share 1"; } | tee $O/two_${1:-a}.txt
