#!/bin/bash
# GPU box: the pipelined renderer's soak (tools/pipeline_soak.py) at 10 000 frames under the default paths and with the optional ones forced on
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_soak2; mkdir -p $O
cd $R
{ echo "== default"; timeout -k 10 300 python tools/pipeline_soak.py 10000 2>&1 | tail -2
  echo "== VNR_AMD_HEAD_GATE=1 VNR_AMD_HEAD_GATE_FRAC=1.0 VNR_AMD_MARCH_RANKS=0"; VNR_AMD_HEAD_GATE=1 VNR_AMD_HEAD_GATE_FRAC=1.0 VNR_AMD_MARCH_RANKS=0 timeout -k 10 300 python tools/pipeline_soak.py 6000 2>&1 | tail -2
  echo "== VNR_AMD_HEAD_GATE=2"; VNR_AMD_HEAD_GATE=2 timeout -k 10 300 python tools/pipeline_soak.py 6000 2>&1 | tail -2
  echo "== VNR_AMD_DECOUPLED=2 VNR_AMD_DECOUPLED_LANES=8 VNR_AMD_DECOUPLED_PARTS=2"; VNR_AMD_DECOUPLED=2 VNR_AMD_DECOUPLED_LANES=8 VNR_AMD_DECOUPLED_PARTS=2 timeout -k 10 300 python tools/pipeline_soak.py 6000 2>&1 | tail -2; } | tee $O/soak.txt
