#!/bin/bash
# round 3: the bench line (and optionally the C4 band test with its printed numbers).  usage: r03_bench.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-a}; shift
O=$R/gpurun_out/r03_bench; mkdir -p $O
cd $R
timeout -k 10 500 python bench.py "$@" > $O/bench_$T.json 2> $O/bench_$T.err; echo "bench rc=$?"
python tools/bench_line.py $T < $O/bench_$T.json
python - <<PY
import json
d = json.load(open("$O/bench_$T.json"))
print("budget table:")
for r in (d["inference_cache"].get("budget_table") or []):
    print("  ", r)
print("train:", d["train_ms_per_step"], d["train_roofline"]["kernels_ms"])
PY
tail -5 $O/bench_$T.err
