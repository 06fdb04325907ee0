#!/bin/bash
# round-6 evidence on a fresh box: the bench line, kernel stats + per-leg stats + interval digest of the bench command, PMC traffic passes on
# bench.py itself with the brick image (two streams, one stream) AND without it, then the counter passes of tools/r06_infer_bound.sh and their
# digest (profiles/r06_infer_bound.txt, r06_mfma_pmc.json, r06_train_atomic_pmc.json: stamped with the hash of the sources they describe).
#   usage: r06_profiles.sh [tag]     -> gpurun_out/r06_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-a}
O=$R/gpurun_out/r06_$T
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/bench_line.py $T < $O/bench.json
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python3 $R/bench.py --no-cpu-baseline --no-psnr --no-brick-table --no-interactive) > $O/stats.log 2>&1; echo "stats rc=$?"
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -12 $O/bench_kernel_stats.csv | cut -c1-170
t=$(find $O/stats -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python3 tools/leg_stats.py "$t" $O/bench "fused_infer_kernel<2, 32, 64, 0, false>" > $O/leg_stats.log 2>&1 && head -12 $O/bench_legs.txt
[ -n "$t" ] && python3 tools/union_digest.py "$t" "fused_infer_kernel<2, 32, 64, 0, false>" $O/bench_infer_intervals.csv > $O/bench_infer_union_digest.txt && cat $O/bench_infer_union_digest.txt
find $O/stats -name "*.csv" -size +2M -delete
for leg in on off; do
for c in FETCH_SIZE WRITE_SIZE FETCH_SIZE_h1 WRITE_SIZE_h1; do
  d=$O/$leg/pmc_bench_$c
  mkdir -p $O/$leg
  export VNR_AMD_RENDER_HALVES=2; case $c in *_h1) export VNR_AMD_RENDER_HALVES=1;; esac
  if [ $leg = on ]; then export VNR_AMD_BRICK=1; else export VNR_AMD_BRICK=0; fi   # on: the image from the first launch; off: never
  cc=${c%_h1}
  (cd /tmp && timeout -s ABRT -k 10 120 rocprofv3 --pmc $cc --output-format csv -d "$d" -o bench -- python3 -X faulthandler $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --no-brick-table --no-interactive --train-steps 300) > "$d.log" 2>&1
  rc=$?; echo "[pmc bench.py] brick $leg $c exit $rc"
  f=$(ls "$d"/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tools/pmc_summary.py per-kernel "$f" > "$d.summary.txt" && grep "fused_infer_kernel<2, 32, 64, 0, false>" "$d.summary.txt" | cut -c1-200
  find $O -name "*.csv" -size +4M -delete
  [ $rc -ne 0 ] && tail -30 "$d.log" && break
done
unset VNR_AMD_RENDER_HALVES VNR_AMD_BRICK
python3 tools/pmc_traffic.py $O/$leg $O/bench.json > $O/pmc_traffic_brick_$leg.json && python3 -c "import json; j=json.load(open('$O/pmc_traffic_brick_$leg.json')); print('brick $leg', {k: round(j[k]['bytes_per_sample'], 1) for k in ('one_stream', 'two_streams') if k in j}, j.get('source_sha16'))"
done
rm -rf $R/gpurun_out/r06_bound
R06_PASSES="sq2_on|tcp_on|tcp2_on|tcc_on|ta1_on|mfma_on|tcc_atomic|sq2_off|ta1_off|tcp_off|tcc_off" bash tools/r06_infer_bound.sh
python3 tools/pmc_digest.py $R/gpurun_out/r06_bound $O/digest > $O/digest.log 2>&1; tail -22 $O/digest.log
exit 0
