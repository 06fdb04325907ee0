#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_suite; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_$1.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_$1.log
