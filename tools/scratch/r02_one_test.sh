#!/bin/bash
# one test file / expression on the GPU box: r02_one_test.sh <pytest args>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 900 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -15
