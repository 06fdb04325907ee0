#!/bin/bash
# round 5: does the march kernel gain from a third wave per SIMD?  The same frame at 23 samples per ray and iteration
# with the depth sort's ranks in LDS (59.6 KB per block: two blocks per CU) and without them (53.3 KB: three), against the default 24, each
# under rocprofv3 --kernel-trace --stats (the program itself after `--`).  -> gpurun_out/r05_march/<tag>_kernel_stats.csv, <tag>.log
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_march; mkdir -p $O
cd $R
export TMPDIR=/tmp SHARE_PARTS=1 SHARE_FRAMES=40 SHARE_PIPELINED=1
run() {   # tag, env assignments...
  local tag=$1; shift
  (cd /tmp && env "$@" timeout -k 10 150 python3 $R/tools/share_probe.py) > $O/$tag.plain.log 2>&1; echo "[r05_march] $tag plain rc=$?: $(grep 'share 1/1' $O/$tag.plain.log)"
  for kv in "$@"; do export "$kv"; done
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o p -- python3 $R/tools/share_probe.py) > $O/$tag.log 2>&1
  local rc=$?
  for kv in "$@"; do unset "${kv%%=*}"; done
  local f=$(ls $O/$tag/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv && grep -E "march_kernel|compact_rays|fused_infer_kernel<2, 32, 64, 0" $O/${tag}_kernel_stats.csv | cut -c1-160
  find $O -name "*kernel_trace.csv" -size +1M -delete
  echo "[r05_march] $tag rc=$rc: $(grep 'share 1/1' $O/$tag.log)"
  return $rc
}
run n24        VNR_RM_N_ITERS=24 || exit 0
run n24_nr     VNR_RM_N_ITERS=24 VNR_AMD_MARCH_RANKS=0 || exit 0
run n23        VNR_RM_N_ITERS=23 || exit 0
run n23_nr     VNR_RM_N_ITERS=23 VNR_AMD_MARCH_RANKS=0 || exit 0
run n16        VNR_RM_N_ITERS=16 || exit 0
run n16_nr     VNR_RM_N_ITERS=16 VNR_AMD_MARCH_RANKS=0 || exit 0
exit 0
