#!/bin/bash
# result arena in group-interleaved order (variant tarena) against ray-major (base)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_tarena.so timeout -k 10 600 python -m pytest tests/test_gpu_render.py tests/test_golden.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do
  for t in base tarena; do
    VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so timeout -k 10 120 python tools/infer_alone.py 20 2>&1 | grep "^\[" | grep -v "brick off" || echo "[$t] FAILED"
    VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so SHARE_PIPELINED=1 SHARE_PARTS=8 timeout -k 10 120 python tools/share_probe.py 2>&1 | grep "share 1/8" || echo "[$t] share FAILED"
  done
done
