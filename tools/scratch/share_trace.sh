#!/bin/bash
# kernel trace of one rank's 1/N share of the bench frame (tools/share_probe.py): sum of kernel time vs frame time
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-8}
O=$R/gpurun_out/trace
mkdir -p $O
export TMPDIR=/tmp SHARE_PARTS=$N
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/share$N -o t -- python3 $R/tools/share_probe.py) > $O/share$N.log 2>&1
grep "share 1" $O/share$N.log
f=$(find $O/share$N -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
frames = 46
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("march", "compact", "fused_infer_kernel<2, 32, 0>")):
        ms = int(r["TotalDurationNs"]) / 1e6
        tot += ms
        print("%-50s calls %5s per-frame %.3f ms avg %.1f us" % (n[:50], r["Calls"], ms / frames, float(r["AverageNs"]) / 1e3))
print("sum per frame %.3f ms (two streams run side by side)" % (tot / frames))
PY
find $O/share$N -name "*.csv" -size +1M -delete
