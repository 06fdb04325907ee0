#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (no counters); summary -> gpurun_out/trace/<tag>_kernel_stats.csv
#   usage (on the GPU box): bash tools/run_trace.sh <tag> [env assignments passed to bench, e.g. VNR_AMD_RENDER_HALVES=1]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
O=$R/gpurun_out/trace
mkdir -p "$O"
export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$tag" -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-alone) > "$O/$tag.log" 2>&1
rc=$?
f=$(find "$O/$tag" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$O/${tag}_kernel_stats.csv" && head -12 "$O/${tag}_kernel_stats.csv" | cut -c1-180
tail -1 "$O/$tag.log" | cut -c1-400
find "$O" -name "*.csv" -size +4M -delete
find "$O" -name "*.db" -delete
exit $rc
