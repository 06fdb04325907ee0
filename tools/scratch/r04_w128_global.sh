#!/bin/bash
# round 4: a 128-neuron model on the C4 frame, weight image in LDS (one block of 8 waves per CU) against weights from global memory
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export SHARE_PARTS=1,8 SHARE_NEURONS=128 TMPDIR=/tmp
for i in 1 2; do
  echo "== LDS image $i"; timeout -k 10 200 python3 tools/share_probe.py 2>&1 | grep "share 1" || exit 1
  echo "== global weights $i"; VNR_AMD_WEIGHTS_GLOBAL=1 timeout -k 10 200 python3 tools/share_probe.py 2>&1 | grep "share 1" || exit 1
done
