#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_dist.py tests/test_scene.py -m gpu -x -q -k "command_line" 2>&1 | tail -25
