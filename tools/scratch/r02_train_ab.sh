#!/bin/bash
# training step on one box: tests first, then the phase profile and the kernel timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_train; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_network.py tests/test_gpu_configs.py tests/test_gpu_dist.py tests/test_gpu_ooc.py tests/test_scene.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 120 python tools/train_probe.py 2>&1 | tee $O/default.log && bash tools/r02_train_trace.sh | head -13
MASTER_PORT=29700 WORLD_SIZE=1 RANK=0 VNR_AMD_DIST_FORCE=1 timeout -k 10 300 python tools/dp_probe.py 2>&1 | grep "C4 model" | tee $O/dp.log
