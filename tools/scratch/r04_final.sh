#!/bin/bash
# round 4, final evidence on one fresh box: the -m gpu suite, smoke(), then the profiles of tools/scratch/r04_profiles.sh under tag $1,
# then the width probe under rocprofv3 --kernel-trace --stats (per-kernel durations of every width / kind of model)
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-b}
cd $R
bash tools/scratch/r04_suite.sh final_$T || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
bash tools/scratch/r04_profiles.sh $T
O=$R/gpurun_out/r04_$T
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wstats -o w -- python3 $R/tools/width_probe.py) > $O/width_probe_under_rocprof.txt 2>&1; echo "width probe rc=$?"
f=$(find $O/wstats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/width_probe_kernel_stats.csv && head -30 $O/width_probe_kernel_stats.csv | cut -c1-200
find $O/wstats -name "*.csv" -size +2M -delete
exit 0
