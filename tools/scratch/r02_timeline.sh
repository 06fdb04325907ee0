#!/bin/bash
# kernel timeline of one steady frame: whole frame (parts = 1) and a 1/8 share
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_tl; mkdir -p $O
export TMPDIR=/tmp
for parts in 1 8; do
  export SHARE_PARTS=$parts
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/p$parts -o t -- python3 $R/tools/share_probe.py) > $O/p$parts.log 2>&1
  f=$(find $O/p$parts -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/share_timeline.py "$f" > $O/timeline_p$parts.txt 2>&1
  grep "share 1" $O/p$parts.log
  find $O/p$parts -name "*.csv" -size +1M -delete
done
