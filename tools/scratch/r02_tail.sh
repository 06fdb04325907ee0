#!/bin/bash
# the frame's last march without evaluation / packing behind it (VNR_AMD_TAIL_SKIP, default on): tests, then A/B by the switch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 800 python -m pytest tests/test_gpu_render.py tests/test_golden.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  for v in 0 1; do
    echo "tail_skip $v: $(VNR_AMD_TAIL_SKIP=$v SHARE_PIPELINED=1 SHARE_PARTS=1,8 timeout -k 10 120 python tools/share_probe.py 2>&1 | grep 'share 1/' | tr '\n' ' ')"
  done
done
