#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_tfn; mkdir -p $O
cd $R
export SHARE_PIPELINED=1 SHARE_PARTS=1,8 SHARE_FRAMES=80 SHARE_REPS=3
SHARE_CONFIGS="merged:VNR_AMD_DEBUG_FLAGS=0;separate:VNR_AMD_DEBUG_FLAGS=32" timeout -k 10 400 python tools/share_probe.py 2>&1 | grep share | tee $O/share_ab.txt
