#!/bin/bash
# bench.py with WORLD ranks on ONE GPU through the host-staged transport (a rehearsal of the N > 1 path, not a measurement),
# usage: r02_bench_ranks.sh [world] [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-2}; shift
O=$R/gpurun_out/r02_ranks
mkdir -p $O
cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 WORLD_SIZE=$W VNR_AMD_DIST_TRANSPORT=shm VNR_AMD_DIST_TIMEOUT=120
pids=()
for r in $(seq 0 $((W-1))); do
  RANK=$r LOCAL_RANK=$r timeout -k 10 300 python bench.py --gpus $W "$@" > $O/w${W}_r$r.out 2> $O/w${W}_r$r.err &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
echo "world $W rc=$rc"; cat $O/w${W}_r0.out | cut -c1-1500; tail -3 $O/w${W}_r0.err
exit $rc
