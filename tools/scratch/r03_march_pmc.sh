#!/bin/bash
# GPU box: SQ counters of march_kernel<false> on the bench frame (rocprofv3 --pmc serialises the kernels: what a march wave waits for when
# it has the GPU to itself).  Two passes of 8 SQ counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03_march_pmc; mkdir -p $O
cd $R
export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU"; do
  i=$((i + 1)); d=$O/p$i
  (cd /tmp && timeout -s ABRT -k 10 150 rocprofv3 --pmc $set --output-format csv -d "$d" -o m -- python3 -X faulthandler $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --no-brick-table --train-steps 300) > "$d.log" 2>&1
  rc=$?; echo "[pmc] pass $i exit $rc"
  f=$(ls "$d"/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py per-kernel "$f" > "$d.summary.txt" && grep "march_kernel<false" "$d.summary.txt" | cut -c1-200
  find $O -name "*.csv" -size +4M -delete
  [ $rc -ne 0 ] && tail -20 "$d.log" && break
done
exit 0
