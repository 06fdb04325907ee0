#!/bin/bash
# round 4, first contact of the width-generic MFMA kernels with the GPU: network + training parity, the probe, a short bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_first
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_network.py tests/test_gpu_train.py -x -q > $O/pytest_net.log 2>&1; echo "pytest net rc=$?"; tail -15 $O/pytest_net.log
timeout -k 10 300 python tools/width_probe.py > $O/width_probe.txt 2>&1; echo "probe rc=$?"; cat $O/width_probe.txt
timeout -k 10 400 python bench.py --no-cpu-baseline --no-brick-table > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/bench_line.py first < $O/bench.json
