#!/bin/bash
# round 4 A/B on one box, alternating: the default library against instantvnr_amd/ab/libvnr_amd_<tag>.so on the share probe (whole frame + 1/8)
# usage: r04_fastmath_ab.sh <tag> [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-fast}; reps=${2:-2}
cd $R
export SHARE_PARTS=1,8 TMPDIR=/tmp
for i in $(seq 1 $reps); do
  echo "== base $i"; timeout -k 10 200 python3 tools/share_probe.py 2>&1 | grep "share 1" || exit 1
  echo "== $tag $i"; VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$tag.so timeout -k 10 200 python3 tools/share_probe.py 2>&1 | grep "share 1" || exit 1
done
