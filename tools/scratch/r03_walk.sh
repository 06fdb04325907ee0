#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_walk; mkdir -p $O
cd $R
export VNR_AMD_DECOUPLED=2 VNR_AMD_DEBUG_SKIP_EVAL=1 VNR_AMD_DECOUPLED_AHEAD=2 VNR_AMD_DECOUPLED_PARTS=1 VNR_AMD_DECOUPLED_PRIO=0
for t in ${TAGS:-stamps stampsnl}; do echo "== $t"; VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so timeout -k 10 200 python tools/walk_stamps.py ${SHARES:-8,16,64} 2>&1 | grep -v "^$"; done | tee $O/walk_stamps_${1:-a}.txt
