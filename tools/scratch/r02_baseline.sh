#!/bin/bash
# round-2 baseline on a fresh box: GPU tests, the bench line, share probe and a kernel timeline of the 1/8 share
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_base
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" 
tail -3 $O/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 200 python tools/share_probe.py > $O/share.log 2>&1; cat $O/share.log
export TMPDIR=/tmp SHARE_PARTS=8
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/share8 -o t -- python3 $R/tools/share_probe.py) > $O/share8_trace.log 2>&1
f=$(find $O/share8 -name "*kernel_trace.csv" | head -1)
python3 tools/share_timeline.py "$f" > $O/share8_timeline.txt 2>&1
head -60 $O/share8_timeline.txt
find $O/share8 -name "*.csv" -size +1M -delete
