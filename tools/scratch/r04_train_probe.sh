#!/bin/bash
# the C4 training step by phase, twice; then the gradient / training parity tests.  usage: r04_train_probe.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for i in 1 2; do timeout -k 10 120 python tools/train_probe.py 600 2>&1 | grep train_probe || exit 1; done
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_network.py -x -q -k "train or accepted or reconfig or gradient" 2>&1 | tail -3
