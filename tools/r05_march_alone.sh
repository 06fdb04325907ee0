#!/bin/bash
# round 5: the march kernel with the GPU to itself (ONE ray part on one stream: nothing else resident while it runs), two blocks per CU
# against three: 23 samples per ray and iteration with (59.6 KB of LDS per block) and without (53.3 KB) the depth sort's ranks in LDS.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_march; mkdir -p $O
cd $R
export TMPDIR=/tmp SHARE_PARTS=1 SHARE_FRAMES=30 VNR_AMD_RENDER_HALVES=1
for cfg in "h1_n23:VNR_RM_N_ITERS=23:VNR_AMD_MARCH_RANKS=1" "h1_n23_nr:VNR_RM_N_ITERS=23:VNR_AMD_MARCH_RANKS=0" "h1_n24:VNR_RM_N_ITERS=24:VNR_AMD_MARCH_RANKS=1" "h1_n24_nr:VNR_RM_N_ITERS=24:VNR_AMD_MARCH_RANKS=0"; do
  IFS=: read tag a b <<< "$cfg"
  export $a $b
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -o p -- python3 $R/tools/share_probe.py) > $O/$tag.log 2>&1 || { echo "[r05_march_alone] $tag failed"; exit 0; }
  f=$(ls $O/$tag/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp "$f" $O/${tag}_kernel_stats.csv
  find $O -name "*kernel_trace.csv" -size +1M -delete
  echo "[r05_march_alone] $tag: $(grep 'share 1/1' $O/$tag.log)"
done
exit 0
