#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lanes; mkdir -p $O
cd $R
export VNR_AMD_DECOUPLED=2 VNR_AMD_DEBUG_SKIP_EVAL=1 VNR_AMD_DECOUPLED_AHEAD=2 VNR_AMD_DECOUPLED_PARTS=1 VNR_AMD_DECOUPLED_PRIO=0 VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_stamps.so
for cfg in "8 8192" "8 512" "1 8192"; do set -- $cfg
  echo "== lanes $1 blocks $2"; VNR_AMD_DECOUPLED_LANES=$1 VNR_AMD_DECOUPLED_BLOCKS=$2 timeout -k 10 200 python tools/wave_records.py 8 2>&1 | grep -v "^$"
done | tee $O/wave_records_a.txt
