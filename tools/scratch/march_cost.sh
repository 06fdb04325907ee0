#!/bin/bash
# what the march kernel of a 1/8 share spends its time on: one stream (clean durations), diagnostic flags of RenderParams
# (1: no compose, 2: constant classification, 4: one depth bin); kernel averages from rocprofv3 --stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/trace; mkdir -p $O
export SHARE_PARTS=8 TMPDIR=/tmp VNR_AMD_RENDER_HALVES=1
for f in ${@:-0 1 2 4}; do
  export VNR_AMD_DEBUG_FLAGS=$f
  rm -rf $O/mc_$f
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mc_$f -o t -- python3 $R/tools/share_probe.py) > $O/mc_$f.log 2>&1 || exit 1
  echo "flags $f: $(grep 'share 1' $O/mc_$f.log)"
  python3 - $(find $O/mc_$f -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("march", "compact", "fused_infer")):
        print("   %-46s calls %5s avg %7.1f us  min %7.1f  max %7.1f" % (n.replace("vnr::", "")[:46], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  find $O/mc_$f -name "*.csv" -size +1M -delete
done
