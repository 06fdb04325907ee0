#!/bin/bash
# training step on one box: tests first, then the phase profile and the kernel timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_train; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_network.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 120 python tools/train_probe.py 2>&1 | tee $O/default.log && bash tools/r02_train_trace.sh | head -13
