#!/bin/bash
# GPU box: the head of a pipelined frame on streams of its own behind the last large evaluation of the frame before it
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_gate; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "pipelined or asynchronous" > $O/tests_${1:-a}.txt 2>&1 || { tail -30 $O/tests_${1:-a}.txt; exit 1; }
tail -3 $O/tests_${1:-a}.txt
export SHARE_PIPELINED=1 SHARE_PARTS=${SHARE_PARTS:-1,8} SHARE_FRAMES=60 SHARE_REPS=2
SHARE_CONFIGS="gate0:VNR_AMD_HEAD_GATE=0;f15:VNR_AMD_HEAD_GATE_FRAC=0.15;f30:VNR_AMD_HEAD_GATE_FRAC=0.3;f60:VNR_AMD_HEAD_GATE_FRAC=0.6;f100:VNR_AMD_HEAD_GATE_FRAC=1.0" timeout -k 10 500 python tools/share_probe.py 2>&1 | grep share | tee $O/share_${1:-a}.txt
