#!/bin/bash
# round 3, HEAD: the bench line, then kernel stats + interval digest of the same command (the PMC traffic passes are r03_profiles.sh's: the
# evaluation kernel has not changed since)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-c}
O=$R/gpurun_out/r03_$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/bench_line.py $T < $O/bench.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline --no-psnr --no-brick-table) > $O/stats.log 2>&1; echo "stats rc=$?"
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -14 $O/bench_kernel_stats.csv | cut -c1-170
t=$(find $O/stats -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/union_digest.py "$t" "fused_infer_kernel<2, 32, 0>" $O/bench_infer_intervals.csv > $O/bench_infer_union_digest.txt && cat $O/bench_infer_union_digest.txt
find $O/stats -name "*.csv" -size +2M -delete
exit 0
