#!/bin/bash
# knob sweep on a rank's share of the bench frame (tools/share_probe.py): halves x N_ITERS x fused blocks per CU
# usage: share_sweep.sh <parts> "<halves list>" "<n_iters list>" "<blocks per CU list>"
R=${GRAFT_REPO_ROOT:-/root/repo}
export SHARE_PARTS=${1:-8}
for h in ${2:-2 1}; do for n in ${3:-16 24 32 40 48}; do for b in ${4:-0 2 4}; do
  r=$(VNR_AMD_RENDER_HALVES=$h VNR_RM_N_ITERS=$n VNR_AMD_INFER_BLOCKS_PER_CU=$b timeout -k 10 100 python3 $R/tools/share_probe.py 2>&1 | grep "share 1") || exit 1
  echo "halves $h n_iters $n blocks_per_cu $b: $r"
done; done; done
