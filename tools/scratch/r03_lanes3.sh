#!/bin/bash
# GPU box: the chain alone at the 1/8 share with the eight-lane kernels' grid capped (blocks loop over groups): how much of a launch is its per-block atomics
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lanes; mkdir -p $O
cd $R
export SHARE_PIPELINED=1 SHARE_PARTS=8 SHARE_FRAMES=60 VNR_AMD_DECOUPLED=2 VNR_AMD_DEBUG_SKIP_EVAL=1 VNR_AMD_DECOUPLED_LANES=8
for b in 256 512 1024 2048; do
  echo "blocks $b: $(VNR_AMD_DECOUPLED_BLOCKS=$b timeout -k 10 200 python tools/share_probe.py 2>&1 | grep share)" | tee -a $O/blocks_${1:-a}.txt
done
