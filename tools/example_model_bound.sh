#!/bin/bash
# counters of the evaluation kernel of the reference's example model (L8 F8 T2^19 + 4x64, example-model.json) on the bench volume and frame:
# three rocprofv3 --pmc passes on bench.py's short form (the program itself after `--`; counters only).  -> gpurun_out/example_bound/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/example_bound; mkdir -p $O
cd $R
export TMPDIR=/tmp
BENCH="$R/bench.py --steps 3 --warmup 1 --levels 8 --features 8 --log2-hashmap-size 19 --hidden-layers 4 --per-level-scale 2 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --no-brick-table --no-interactive --train-steps 300"
pass() {
  local name=$1; shift
  (cd /tmp && timeout -s ABRT -k 10 170 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o p -- python3 -X faulthandler $BENCH) > $O/$name.log 2>&1
  echo "[example_bound] $name rc=$?"
  local f=$(ls $O/$name/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py per-kernel "$f" > $O/$name.summary.txt && grep -E "fused_infer_kernel<8, 64, 64, 0" $O/$name.summary.txt | cut -c1-200
  find $O -name "*.csv" -size +4M -delete
}
pass sq SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
pass ta TA_TA_BUSY_sum GRBM_GUI_ACTIVE
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
exit 0
