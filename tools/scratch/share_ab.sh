#!/bin/bash
# A/B of two library builds on a rank's share of the bench frame: tools/share_ab.sh <old.so> [parts]; then a kernel-trace
# timeline of the new build's 1/8 share
R=${GRAFT_REPO_ROOT:-/root/repo}
old=$1; parts=${2:-1,8}
cp $R/instantvnr_amd/libvnr_amd.so /tmp/new.so
export SHARE_PARTS=$parts TMPDIR=/tmp
for i in 1 2; do
  cp /tmp/new.so $R/instantvnr_amd/libvnr_amd.so; echo new; timeout -k 10 150 python3 $R/tools/share_probe.py 2>&1 | grep "share 1" || exit 1
  cp $old $R/instantvnr_amd/libvnr_amd.so; echo old; timeout -k 10 150 python3 $R/tools/share_probe.py 2>&1 | grep "share 1" || exit 1
done
cp /tmp/new.so $R/instantvnr_amd/libvnr_amd.so
O=$R/gpurun_out/trace; mkdir -p $O; rm -rf $O/share_tl
export SHARE_PARTS=8
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/share_tl -o t -- python3 $R/tools/share_probe.py) > $O/share_tl.log 2>&1 || exit 1
f=$(find $O/share_tl -name "*kernel_trace.csv" | head -1)
python3 $R/tools/share_timeline.py $f > $O/share_tl.txt; cat $O/share_tl.txt
find $O/share_tl -name "*.csv" -size +1M -delete
