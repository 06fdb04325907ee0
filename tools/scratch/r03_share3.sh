#!/bin/bash
# decoupled share with the evaluation kernel's grid capped (room for the walk / compose kernels beside it): one process per cap
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_share; mkdir -p $O
cd $R
export SHARE_PIPELINED=1 SHARE_PARTS=1,8 SHARE_REPS=1
CFG="coupled:VNR_AMD_DECOUPLED=0;a2p1:VNR_AMD_DECOUPLED=1,VNR_AMD_DECOUPLED_AHEAD=2,VNR_AMD_DECOUPLED_PARTS=1;a3p1:VNR_AMD_DECOUPLED=1,VNR_AMD_DECOUPLED_AHEAD=3,VNR_AMD_DECOUPLED_PARTS=1;a5p1:VNR_AMD_DECOUPLED=1,VNR_AMD_DECOUPLED_AHEAD=5,VNR_AMD_DECOUPLED_PARTS=1;a3p2:VNR_AMD_DECOUPLED=1,VNR_AMD_DECOUPLED_AHEAD=3,VNR_AMD_DECOUPLED_PARTS=2"
for b in 0 4 3 2 1; do
  echo "== VNR_AMD_INFER_BLOCKS_PER_CU=$b"
  VNR_AMD_INFER_BLOCKS_PER_CU=$b VNR_AMD_DECOUPLED_PRIO=${PRIO:-0} SHARE_CONFIGS="$CFG" timeout -k 10 200 python tools/share_probe.py 2>&1 | grep "share 1/8"
done | tee $O/share3_${1:-a}.log
