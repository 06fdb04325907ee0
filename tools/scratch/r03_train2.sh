#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_train; mkdir -p $O
cd $R
{ for rep in 1 2; do
VNR_AMD_GRID_BWD_LDS=0 timeout -k 10 120 python tools/train_probe.py 400 2>&1 | grep train_probe | sed 's/phases.*grid backward/gb/; s/, optimizer.*//'
for kb in 16 24 32; do for bl in 384 512 768; do for mt in 32 64 128; do
  VNR_AMD_GRID_BWD_LDS_KB=$kb VNR_AMD_GRID_BWD_LDS_BLOCKS=$bl VNR_AMD_GRID_BWD_LDS_TILES=$mt timeout -k 10 120 python tools/train_probe.py 400 2>&1 | grep train_probe | sed 's/phases.*grid backward/gb/; s/, optimizer.*//'
done; done; done; done; } | tee $O/train_probe_${1:-c}.log
