#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c4_model_match" 2>&1 | tail -15
