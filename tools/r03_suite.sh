#!/bin/bash
# round 3: the whole -m gpu suite on a fresh box.   usage: r03_suite.sh <tag> [pytest args]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-a}; shift
O=$R/gpurun_out/r03_suite; mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=15 "$@" > $O/pytest_$T.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 $O/pytest_$T.log
exit $rc
