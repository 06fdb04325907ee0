#!/bin/bash
# GPU box: the training tests, then the C4 step with the per-parameter Adam kernel against the compacting one (alternating)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_adam; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_network.py -x -q -m gpu > $O/tests_${1:-a}.txt 2>&1 || { tail -30 $O/tests_${1:-a}.txt; exit 1; }
tail -3 $O/tests_${1:-a}.txt
{ for rep in 1 2; do for c in 0 1; do VNR_AMD_ADAM_COMPACT=$c timeout -k 10 120 python tools/train_probe.py 400 2>&1 | grep train_probe; done; done; } | tee $O/train_probe_${1:-a}.log
