#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lanes; mkdir -p $O
cd $R
export VNR_AMD_DECOUPLED=0 VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_stamps.so
timeout -k 10 200 python tools/wave_records.py 8,1 2>&1 | grep -v "^$" | tee $O/wave_records_coupled.txt
