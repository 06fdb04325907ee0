#!/bin/bash
# sample-sort experiment: depth bins x tile quadrants (variant sortq) at several bin depths against the default sort
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
[ -n "$SKIP_TESTS" ] || VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_sortq.so timeout -k 10 400 python -m pytest tests/test_gpu_render.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2; do
  for cfg in ${CFGS:-base:8 sortq:8 sortq:4 sortq:16 base:4 base:16}; do
    t=${cfg%%:*}; d=${cfg##*:}
    echo "== $t bin depth $d"
    VNR_AMD_BIN_DEPTH=$d VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so timeout -k 10 120 python tools/infer_alone.py 20 2>&1 | grep "^\[" | grep -v "brick off" || echo "[$t] FAILED"
  done
done
