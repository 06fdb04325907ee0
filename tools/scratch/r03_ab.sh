#!/bin/bash
# alternating A/B of library variants (tools/ab_build.sh) on the share probe: r03_ab.sh <tagA> <tagB> [reps]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_ab; mkdir -p $O
cd $R
A=$1; B=$2; N=${3:-2}
export SHARE_PIPELINED=1 SHARE_PARTS=${SHARE_PARTS:-1,8} SHARE_REPS=1
for rep in $(seq 1 $N); do for t in $A $B; do
  echo "== $t"; VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so timeout -k 10 200 python tools/share_probe.py 2>&1 | grep "share 1/"
done; done | tee $O/ab_${A}_${B}.log
