#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_train; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/tools/train_probe.py 100) > $O/trace.log 2>&1; echo "rc=$?"
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/train_timeline.py "$f" 150 2 | tee $O/timeline.txt
find $O/trace -name "*.csv" -size +1M -delete
