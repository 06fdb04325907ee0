#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_ranks; mkdir -p $O
cd $R
export SHARE_PIPELINED=1 SHARE_PARTS=8 SHARE_FRAMES=60 SHARE_REPS=2
for b in 2 3 4; do
VNR_AMD_INFER_BLOCKS_PER_CU=$b SHARE_CONFIGS="b${b}ranks:VNR_AMD_MARCH_RANKS=1;b${b}noranks:VNR_AMD_MARCH_RANKS=0" timeout -k 10 300 python tools/share_probe.py 2>&1 | grep share
done | tee $O/share_coresident.txt
