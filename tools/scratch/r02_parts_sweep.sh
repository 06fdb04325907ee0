#!/bin/bash
# ray parts of a 1/8 share (VNR_AMD_RENDER_HALVES) after the chain got shorter
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for h in 2 3 4 6 8; do
    echo "parts $h: $(VNR_AMD_RENDER_HALVES=$h SHARE_PIPELINED=1 SHARE_PARTS=8,4 timeout -k 10 120 python tools/share_probe.py 2>&1 | grep 'share 1/' | cut -c1-40 | tr '\n' ' ')"
  done
done
