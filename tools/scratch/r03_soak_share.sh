#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_share; mkdir -p $O
cd $R
( while sleep 2; do ps -u "$(id -u)" -o pid=,rss=,comm= | while read pid rss comm; do
    if [ "${rss:-0}" -gt 48000000 ]; then echo "[watchdog] killing $comm pid $pid rss ${rss} kB" | tee -a $O/watchdog.log; kill -9 "$pid"; fi; done; done ) &
WD=$!
timeout -k 10 280 python tools/pipeline_soak.py 1500 > $O/soak.log 2>&1; echo "soak rc=$?"; tail -3 $O/soak.log
bash tools/r03_share.sh ${1:-a}
kill $WD 2>/dev/null
