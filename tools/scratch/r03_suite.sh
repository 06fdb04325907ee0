#!/bin/bash
# round 3: the -m gpu suite (or a part of it) on a fresh box, with a watchdog on host memory: a process of this user that grows beyond
# 48 GB of resident memory is killed by PID (a host that runs out of memory is a lost box).   usage: r03_suite.sh <tag> [pytest args]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-a}; shift
O=$R/gpurun_out/r03_suite; mkdir -p $O
cd $R
( while sleep 2; do
    ps -u "$(id -u)" -o pid=,rss=,comm= | while read pid rss comm; do
      if [ "${rss:-0}" -gt 48000000 ]; then echo "[watchdog] killing $comm pid $pid rss ${rss} kB" | tee -a $O/watchdog_$T.log; kill -9 "$pid"; fi
    done
  done ) &
WD=$!
if [ $# -eq 0 ]; then set -- tests; fi
timeout -k 10 1100 python -m pytest -m gpu -q --durations=15 "$@" > $O/pytest_$T.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -30 $O/pytest_$T.log
kill $WD 2>/dev/null
exit $rc
