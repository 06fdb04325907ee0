#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -m gpu -x -q -k "set_model" 2>&1 | tail -15
