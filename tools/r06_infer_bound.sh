#!/bin/bash
# round 5: what bounds fused_infer_kernel?  rocprofv3 --pmc passes (counters only, one block's counters per pass, the
# program itself after `--`) on the whole bench frame, brick image on (the bench default) and off (the hashed parameter blob).
# The SQ pass ran on bench.py's short form; a TA pass on it hung in the volume generator (the intermittent hang of tools/run_pmc.sh's
# header), so the others go through tools/share_probe.py (same volume, model, camera and frame; 300 training steps), as round 3's did.
#   usage: R06_PASSES="<regex of pass names>" bash tools/r06_infer_bound.sh   -> gpurun_out/r06_bound/<pass>.summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_bound; mkdir -p $O
cd $R
export TMPDIR=/tmp SHARE_PARTS=1 SHARE_FRAMES=3
BENCH="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --no-brick-table --no-interactive --train-steps 300"
pass() {   # name, program (bench / share), brick (1 / 0), counters
  local name=$1 prog=$2 brick=$3; shift 3
  [[ "$name" =~ ${R06_PASSES:-.} ]] || return 0
  export VNR_AMD_BRICK=$brick
  if [ $prog = bench ]; then
    (cd /tmp && timeout -s ABRT -k 10 170 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o p -- python3 -X faulthandler $BENCH) > $O/$name.log 2>&1
  else
    (cd /tmp && timeout -s ABRT -k 10 110 rocprofv3 --pmc "$@" --output-format csv -d $O/$name -o p -- python3 -X faulthandler $R/tools/share_probe.py) > $O/$name.log 2>&1
  fi
  local rc=$?
  echo "[r06_bound] $name rc=$rc"
  local f=$(ls $O/$name/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py per-kernel "$f" > $O/$name.summary.txt && grep -E "fused_infer_kernel<2, 32, 64, 0|grid_backward" $O/$name.summary.txt | cut -c1-200
  find $O -name "*.csv" -size +4M -delete
  return $rc
}
pass sq_on    bench 1 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE || exit 0
pass sq2_on   share 1 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE || exit 0
# (TA passes with four or six TA counters hung twice at the process's first kernel, once under bench.py and once under share_probe.py; TA_TA_BUSY_sum alone
# completes: ta1_on / ta1_off below)
pass tcp_on   share 1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 0
pass tcp2_on  share 1 TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 0
pass tcc_on   share 1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE || exit 0
pass ta1_on   share 1 TA_TA_BUSY_sum GRBM_GUI_ACTIVE || exit 0
pass mfma_on  share 1 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE || exit 0
pass tcc_atomic share 1 TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum GRBM_GUI_ACTIVE || exit 0
pass sq2_off  share 0 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE || exit 0
pass ta1_off  share 0 TA_TA_BUSY_sum GRBM_GUI_ACTIVE || exit 0
pass tcp_off  share 0 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE || exit 0
pass tcc_off  share 0 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE || exit 0
exit 0
