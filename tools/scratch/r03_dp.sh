#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_dp; mkdir -p $O
cd $R
{ for rep in 1 2; do
VNR_AMD_DP_SHARDED=0 MASTER_PORT=29701 timeout -k 10 200 python tools/dp_probe.py 2>&1 | grep "C4 model"
VNR_AMD_DP_SHARDED=1 MASTER_PORT=29702 timeout -k 10 200 python tools/dp_probe.py 2>&1 | grep "C4 model"
VNR_AMD_DP_SHARDED=1 VNR_AMD_DP_EMULATE_WORLD=8 MASTER_PORT=29703 timeout -k 10 200 python tools/dp_probe.py 2>&1 | grep "C4 model"
VNR_AMD_DP_SHARDED=1 VNR_AMD_DP_EMULATE_WORLD=2 MASTER_PORT=29704 timeout -k 10 200 python tools/dp_probe.py 2>&1 | grep "C4 model"
done; } | tee $O/dp_probe.txt
