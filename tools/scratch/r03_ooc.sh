#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_ooc; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest -m gpu -q tests/test_gpu_ooc.py -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep "asynchronous refresh\|passed\|failed" $O/pytest.log
{ timeout -k 10 300 python tools/ooc_bench.py 1024 1024 2048 1024 16384 300 1
VNR_AMD_OOC_ASYNC=1 timeout -k 10 300 python tools/ooc_bench.py 1024 1024 2048 1024 16384 300 1
VNR_AMD_OOC_ASYNC=1 timeout -k 10 300 python tools/ooc_bench.py 1024 1024 2048 1024 16384 300 0; } 2>&1 | tee $O/ooc_bench.txt
