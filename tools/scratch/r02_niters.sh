#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_niters; mkdir -p $O
cd $R
for n in ${NS:-32 40 48 64 32 40 48}; do
  echo "N_ITERS $n: $(VNR_RM_N_ITERS=$n SHARE_PARTS=8 SHARE_PIPELINED=1 timeout -k 10 200 python tools/share_probe.py 2>&1 | grep 'share 1/')" | tee -a $O/sweep.log
done
