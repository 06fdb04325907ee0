#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_dist.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2; do echo "default: $(SHARE_PIPELINED=1 timeout -k 10 200 python tools/share_probe.py 2>&1 | grep 'share 1/' | sed 's/ per frame.*//' | tr '\n' ' ')"; done
