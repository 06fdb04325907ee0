#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_pack; mkdir -p $O
cd $R
export SHARE_PIPELINED=1 SHARE_PARTS=1 SHARE_FRAMES=80 SHARE_REPS=2
for rep in 1 2; do
echo "base: $(timeout -k 10 200 python tools/share_probe.py 2>&1 | grep share | tr '\n' ' ')"
echo "pack256: $(VNR_AMD_COMPACT_SMALL_LIMIT=4194304 timeout -k 10 200 python tools/share_probe.py 2>&1 | grep share | tr '\n' ' ')"
echo "fused: $(VNR_AMD_FUSED_PACK=2 timeout -k 10 200 python tools/share_probe.py 2>&1 | grep share | tr '\n' ' ')"
done | tee $O/pack.txt
