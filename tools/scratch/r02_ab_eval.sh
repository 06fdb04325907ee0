#!/bin/bash
# usage: r02_ab_eval.sh <variant> [<variant> ...]: bit-exactness tests of every variant against the oracle, then the alternating A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_ab; mkdir -p $O
cd $R
for t in "$@"; do
  VNR_AMD_LIB_PATH=$R/instantvnr_amd/ab/libvnr_amd_$t.so timeout -k 10 400 python -m pytest tests/test_gpu_network.py tests/test_golden.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_$t.log 2>&1; echo "[$t] parity rc=$? $(tail -1 $O/pytest_$t.log)"
done
bash tools/ab_run.sh "$@" 2>&1 | tee -a $O/ab.log
