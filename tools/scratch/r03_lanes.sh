#!/bin/bash
# GPU box: the eight-lanes-per-ray walk / compose kernels: parity tests first, then the walk / compose chain alone and whole shares
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_lanes; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_render.py -x -q -m gpu -k "decoupled or eight_lanes" > $O/tests_${1:-a}.txt 2>&1 || { tail -30 $O/tests_${1:-a}.txt; exit 1; }
tail -3 $O/tests_${1:-a}.txt
export SHARE_PIPELINED=1 SHARE_PARTS=${SHARE_PARTS:-8,16,64} SHARE_FRAMES=60
echo "== chain alone (evaluation skipped)" | tee $O/share_${1:-a}.txt
SHARE_CONFIGS="l1:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_LANES=1,VNR_AMD_DEBUG_SKIP_EVAL=1;l8:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_LANES=8,VNR_AMD_DEBUG_SKIP_EVAL=1" timeout -k 10 300 python tools/share_probe.py 2>&1 | grep share | tee -a $O/share_${1:-a}.txt
echo "== whole share frames" | tee -a $O/share_${1:-a}.txt
SHARE_REPS=2 SHARE_CONFIGS="coupled:VNR_AMD_DECOUPLED=0;dec1:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_LANES=1;dec8:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_LANES=8;dec8a2:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_LANES=8,VNR_AMD_DECOUPLED_AHEAD=2;dec8p2:VNR_AMD_DECOUPLED=2,VNR_AMD_DECOUPLED_LANES=8,VNR_AMD_DECOUPLED_PARTS=2" timeout -k 10 400 python tools/share_probe.py 2>&1 | grep share | tee -a $O/share_${1:-a}.txt
