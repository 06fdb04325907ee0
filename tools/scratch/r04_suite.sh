#!/bin/bash
# round 4: the -m gpu suite (or a part of it) on a fresh box, with a watchdog on host memory restricted to the DESCENDANTS of the pytest it
# started (not every process of the uid): a child that grows beyond 48 GB of resident memory is killed by PID.
# usage: r04_suite.sh <tag> [pytest args]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-a}; shift
O=$R/gpurun_out/r04_suite; mkdir -p $O
cd $R
if [ $# -eq 0 ]; then set -- tests; fi
timeout -k 10 1100 python -m pytest -m gpu -q --durations=15 "$@" > $O/pytest_$T.log 2>&1 &
PT=$!
descendants() { local p; for p in $(ps -o pid= --ppid "$1"); do echo "$p"; descendants "$p"; done; }
( while sleep 2; do
    for pid in $PT $(descendants $PT); do
      rss=$(ps -o rss= -p "$pid" 2>/dev/null | tr -d ' ')
      if [ "${rss:-0}" -gt 48000000 ]; then echo "[watchdog] killing pid $pid rss ${rss} kB" | tee -a $O/watchdog_$T.log; kill -9 "$pid"; fi
    done
  done ) &
WD=$!
trap 'kill $WD 2>/dev/null' EXIT
wait $PT; rc=$?
echo "pytest rc=$rc"; tail -30 $O/pytest_$T.log
exit $rc
