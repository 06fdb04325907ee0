#!/bin/bash
# usage: r02_ab_frames.sh <variant> <variant> ...: render-path parity tests of every variant but the first, then alternating frames (one / two streams) and the pipelined 1/8 share
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
first=$1
for t in "$@"; do
  [ "$t" = "$first" ] && continue
  VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so timeout -k 10 600 python -m pytest tests/test_gpu_render.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -1
done
for rep in 1 2; do
  for t in "$@"; do
    VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so timeout -k 10 120 python tools/infer_alone.py 20 2>&1 | grep "^\[" | grep -v "brick off" || echo "[$t] FAILED"
    echo "[$t] $(VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so SHARE_PIPELINED=1 SHARE_PARTS=8 timeout -k 10 120 python tools/share_probe.py 2>&1 | grep 'share 1/8')"
  done
done
