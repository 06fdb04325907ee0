#!/bin/bash
# round 4: matrix-core counters of the evaluation kernels (north star: "MFMA utilisation on the MLP against the chip's peaks"), one
# rocprofv3 --pmc pass (counters only) on bench.py's short form and one on the width probe.  -> gpurun_out/r04_mfma/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_mfma; mkdir -p $O
cd $R
export TMPDIR=/tmp VNR_AMD_BRICK=1
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
(cd /tmp && timeout -s ABRT -k 10 150 rocprofv3 --pmc $C --output-format csv -d $O/bench -o bench -- python3 -X faulthandler $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --no-brick-table --train-steps 300) > $O/bench.log 2>&1; echo "bench pmc rc=$?"
f=$(ls $O/bench/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_summary.py per-kernel "$f" > $O/bench_mfma_summary.txt && grep -E "fused_infer_kernel<2, 32, 64, 0, false>|march_kernel<false, 0>|weight_grad_mfma_kernel<64, 2>|mlp_backward_kernel<64" $O/bench_mfma_summary.txt | cut -c1-200
unset VNR_AMD_BRICK
(cd /tmp && timeout -s ABRT -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/width -o width -- python3 -X faulthandler $R/tools/width_probe.py) > $O/width.log 2>&1; echo "width pmc rc=$?"
f=$(ls $O/width/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/pmc_summary.py per-kernel "$f" > $O/width_mfma_summary.txt && grep -E "fused_infer_kernel<2, 32, (16|32|64|128), 0, false>" $O/width_mfma_summary.txt | cut -c1-200
find $O -name "*.csv" -size +4M -delete
exit 0
