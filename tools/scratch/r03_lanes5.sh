#!/bin/bash
# GPU box: the 1/8 share with evaluation: decoupled loop, eight lanes per ray, grid caps x ray parts x look-ahead, against the coupled loop
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lanes; mkdir -p $O
cd $R
export SHARE_PIPELINED=1 SHARE_PARTS=8 SHARE_FRAMES=60
for b in 512 1024; do
  echo "== blocks $b"
  VNR_AMD_DECOUPLED_BLOCKS=$b SHARE_CONFIGS="coupled:VNR_AMD_DECOUPLED=0;d8p1a3:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_PARTS=1;d8p2a3:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_PARTS=2;d8p4a3:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_PARTS=4;d8p1a2:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_PARTS=1,VNR_AMD_DECOUPLED_AHEAD=2;d8p2a2:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_PARTS=2,VNR_AMD_DECOUPLED_AHEAD=2;d8p1a4:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_PARTS=1,VNR_AMD_DECOUPLED_AHEAD=4" timeout -k 10 300 python tools/share_probe.py 2>&1 | grep share
done | tee $O/share_blocks_${1:-a}.txt
