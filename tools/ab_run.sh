#!/bin/bash
# runs tools/infer_alone.py for every variant given (tags of instantvnr_amd/ab/libvnr_amd_<tag>.so), twice, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for t in "$@"; do
    VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so timeout -k 10 120 python tools/infer_alone.py 20 2>&1 | grep "^\[" || echo "[$t] FAILED"
  done
done
