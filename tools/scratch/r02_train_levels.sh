#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_train; mkdir -p $O
cd $R
for lv in 0,1 1,2 2,3 3,4 4,5 5,6 6,7 7,8 8,9 9,10 12,13 15,16 0,16; do
  VNR_AMD_GRID_BWD_LEVELS=$lv timeout -k 10 120 python tools/train_probe.py 200 2>&1 | grep -o "VNR_AMD_GRID_BWD_LEVELS=[0-9,]*\|grid backward [0-9.]*\|optimizer [0-9.]*" | tr '\n' ' '; echo
done | tee $O/levels.log
