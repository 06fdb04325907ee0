#!/bin/bash
# round-2 evidence on a fresh box: kernel stats of the bench command, the bench line, PMC passes on bench.py itself (one counter per
# pass, short timeout, program directly after --).   usage: r02_profiles.sh [tag]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-c}
O=$R/gpurun_out/r02_$T
mkdir -p $O $R/gpurun_out/pmc
cd $R
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_scene.py -m gpu -x -q > $O/pytest_scene.log 2>&1; echo "pytest scene rc=$?"; tail -3 $O/pytest_scene.log
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/bench_line.py $T < $O/bench.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline --no-psnr) > $O/stats.log 2>&1; echo "stats rc=$?"
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -12 $O/bench_kernel_stats.csv | cut -c1-170
t=$(find $O/stats -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/union_digest.py "$t" "fused_infer_kernel<2, 32, 0>" $O/bench_infer_intervals.csv > $O/bench_infer_union_digest.txt && cat $O/bench_infer_union_digest.txt
find $O/stats -name "*.csv" -size +2M -delete
for c in FETCH_SIZE WRITE_SIZE FETCH_SIZE_h1 WRITE_SIZE_h1; do
  d=$O/pmc_bench_$c
  export VNR_AMD_BRICK=1 VNR_AMD_RENDER_HALVES=2; case $c in *_h1) export VNR_AMD_RENDER_HALVES=1;; esac   # the brick image from the first launch: with one stream 24 launches are 3.4 frames
  c=${c%_h1}
  (cd /tmp && timeout -s ABRT -k 10 120 rocprofv3 --pmc $c --output-format csv -d "$d" -o bench -- python3 -X faulthandler $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --train-steps 300) > "$d.log" 2>&1
  rc=$?; echo "[pmc bench.py] $c exit $rc"
  f=$(ls "$d"/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py per-kernel "$f" > "$d.summary.txt" && grep "fused_infer_kernel<2, 32, 0>\|march_kernel" "$d.summary.txt" | cut -c1-200
  find $O -name "*.csv" -size +4M -delete
  [ $rc -ne 0 ] && tail -30 "$d.log" && break
done
unset VNR_AMD_RENDER_HALVES VNR_AMD_BRICK
python3 tools/pmc_traffic.py $O $O/bench.json > $O/pmc_traffic.json && python3 -c "import json; j=json.load(open('$O/pmc_traffic.json')); print({k: round(j[k]['bytes_per_sample'], 1) for k in ('one_stream', 'two_streams') if k in j})"
exit 0
