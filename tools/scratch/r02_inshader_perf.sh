#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_inshader; mkdir -p $O
cd $R
for m in 6 9; do for k in 1 0; do
  echo "mode $m in-shader kernel $k"; SHARE_MODE=$m VNR_AMD_IN_SHADER=$k timeout -k 10 300 python tools/share_probe.py 2>&1 | grep "share 1/" | tee -a $O/perf.log
done; done
