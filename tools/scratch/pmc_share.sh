#!/bin/bash
# counters of the evaluation kernel on the whole bench frame through tools/share_probe.py instead of bench.py (whose --pmc passes
# hang intermittently): usage pmc_share.sh "<counter> [<counter> ...]" [halves] [timeout_s] [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc; mkdir -p $O
c=$1; h=${2:-1}; to=${3:-75}; tag=${4:-$(echo $c | cut -d' ' -f1)}
export TMPDIR=/tmp SHARE_PARTS=1 VNR_AMD_RENDER_HALVES=$h VNR_AMD_BRICK=${VNR_AMD_BRICK:-1}
d=$O/share_${tag}_h$h
rm -rf "$d"
(cd /tmp && timeout -k 10 $to rocprofv3 --pmc $c --output-format csv -d "$d" -o p -- python3 $R/tools/share_probe.py) > "$d.log" 2>&1
rc=$?
echo "[pmc_share] $c halves $h exit $rc: $(grep 'share 1' "$d.log")"
f=$(ls "$d"/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 $R/tools/pmc_summary.py per-kernel "$f" > "$d.summary.txt" && grep "fused_infer_kernel<2, 32, 0>\|march_kernel\|compact" "$d.summary.txt" | cut -c1-200
find $O -name "*.csv" -size +4M -delete
exit $rc
