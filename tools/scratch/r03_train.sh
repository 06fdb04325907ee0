#!/bin/bash
# round 3: training tests + the step by phase with and without the LDS path of the grid backward
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_train; mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest -m gpu -q -x tests/test_gpu_train.py tests/test_gpu_network.py "tests/test_gpu_fullsize.py::test_gradients_of_the_c4_model_match_the_restatement" "tests/test_gpu_fullsize.py::test_adam_step_of_the_c4_model_matches_the_restatement" > $O/pytest_${1:-a}.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_${1:-a}.log
for rep in 1 2; do for v in 0 1; do
  VNR_AMD_GRID_BWD_LDS=$v timeout -k 10 120 python tools/train_probe.py 600 2>&1 | grep train_probe
done; done | tee $O/train_probe_${1:-a}.log
