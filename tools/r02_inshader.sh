#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_inshader; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "in_shader" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $O/pytest.log
