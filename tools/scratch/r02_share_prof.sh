#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
echo "events off: $(SHARE_PARTS=8 SHARE_PIPELINED=1 timeout -k 10 200 python tools/share_probe.py 2>&1 | grep 'share 1/' | sed 's/ per frame.*//')"
echo "events on:  $(SHARE_PROFILING=1 SHARE_PARTS=8 SHARE_PIPELINED=1 timeout -k 10 200 python tools/share_probe.py 2>&1 | grep 'share 1/' | sed 's/ per frame.*//')"
done
