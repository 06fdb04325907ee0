#!/bin/bash
# GPU box: walk8 stamps + kernel timeline of the chain alone at the 1/8 share
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lanes; mkdir -p $O
cd $R
export VNR_AMD_DECOUPLED=2 VNR_AMD_DEBUG_SKIP_EVAL=1 VNR_AMD_DECOUPLED_AHEAD=2 VNR_AMD_DECOUPLED_PARTS=1 VNR_AMD_DECOUPLED_PRIO=0 VNR_AMD_DECOUPLED_LANES=8
VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_stamps.so timeout -k 10 200 python tools/walk_stamps.py 8,64 2>&1 | grep -v "^$" | tee $O/walk8_stamps_${1:-a}.txt
export TMPDIR=/tmp SHARE_PARTS=8 SHARE_PIPELINED=1 SHARE_FRAMES=30
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 $R/tools/share_probe.py) > $O/tl_log.txt 2>&1
grep "share 1" $O/tl_log.txt
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 tools/share_timeline.py "$f" > $O/walk8_timeline_${1:-a}.txt 2>&1; head -40 $O/walk8_timeline_${1:-a}.txt
find $O -name "*.csv" -size +3M -delete
