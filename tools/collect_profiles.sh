#!/bin/bash
# copies what tools/r06_profiles.sh <tag> left under gpurun_out/ into profiles/ under the round's names (only summaries: no raw counter csv)
#   usage: collect_profiles.sh [tag] [round]
R=$(cd "$(dirname "$0")/.." && pwd)
T=${1:-a}; RD=${2:-r06}
O=$R/gpurun_out/${RD}_$T; P=$R/profiles
[ -d "$O" ] || { echo "no $O"; exit 1; }
cp $O/bench.json $P/${RD}_${T}_bench.json
for f in bench_infer_union_digest.txt bench_kernel_stats.csv bench_legs.txt; do [ -f $O/$f ] && cp $O/$f $P/${RD}_${T}_$f; done
for f in $O/bench_leg*_kernel_stats.csv; do [ -f $f ] && cp $f $P/${RD}_${T}_$(basename $f); done
for leg in on off; do for f in $O/$leg/pmc_bench_*.summary.txt; do [ -f $f ] && cp $f $P/${RD}_${T}_pmc_bench_brick_${leg}_$(basename $f .summary.txt | sed 's/pmc_bench_//').txt; done; done
[ -f $O/pmc_traffic_brick_on.json ] && cp $O/pmc_traffic_brick_on.json $P/${RD}_pmc_traffic.json
[ -f $O/pmc_traffic_brick_off.json ] && cp $O/pmc_traffic_brick_off.json $P/${RD}_pmc_traffic_brick_off.json
for f in ${RD}_infer_bound.txt ${RD}_mfma_pmc.json ${RD}_train_atomic_pmc.json; do [ -f $O/digest/$f ] && cp $O/digest/$f $P/$f; done
mkdir -p $P/${RD}_bound_passes && cp $R/gpurun_out/${RD}_bound/*.summary.txt $P/${RD}_bound_passes/ 2>/dev/null
ls $P | grep "^${RD}_" | wc -l
