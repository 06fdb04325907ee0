#!/bin/bash
# kernel timeline of one steady-state frame of the pipelined 1/8 share
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_share_tl; mkdir -p $O
cd $R
export TMPDIR=/tmp SHARE_PARTS=8 SHARE_PIPELINED=1
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/tools/share_probe.py) > $O/log.txt 2>&1
grep "share 1" $O/log.txt
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 tools/share_timeline.py "$f" > $O/timeline.txt 2>&1; head -80 $O/timeline.txt
find $O -name "*.csv" -size +3M -delete
