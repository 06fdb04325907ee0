#!/bin/bash
# round 4: re-sweep of the dense levels' LDS scatter after the z early-out (tile size, tiles per level, blocks per level); C4 model
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "24 64 768" "24 96 768" "24 128 768" "24 256 768" "48 64 768" "48 128 768" "16 128 768" "24 128 1536" "24 64 1536" "24 64 384" "24 64 768"; do
  set -- $cfg
  VNR_AMD_GRID_BWD_LDS_KB=$1 VNR_AMD_GRID_BWD_LDS_TILES=$2 VNR_AMD_GRID_BWD_LDS_BLOCKS=$3 timeout -k 10 120 python tools/train_probe.py 400 2>&1 | grep train_probe | sed 's/loss + MLP.*grid backward/grid backward/' | cut -c1-260
done
