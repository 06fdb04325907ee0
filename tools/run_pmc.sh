#!/bin/bash
# ONE rocprofv3 PMC pass (counters only, no trace domain), bounded by its own timeout.
#   usage (on the GPU box): bash tools/run_pmc.sh calib|bench "<counters>" [timeout_s]
#   -> gpurun_out/pmc/<what>_<counters>{.log,.summary.txt,/}
# History (round 1): `rocprofv3 --pmc ... python3 bench.py` hangs INTERMITTENTLY (no output after "HSA version ...
# initialized"): 3 passes finished in ~6 s (FETCH_SIZE x2, WRITE_SIZE x1), 8 hung until killed (WRITE_SIZE x2, RDREQ,
# TCC_HIT/MISS/REQ x3, FETCH_SIZE x2) -- several of them the first profiler process on a fresh box, so it is neither
# counter- nor order-specific.  The two FETCH_SIZE hangs were the only passes run after the renderer went to 2 streams
# (n = 2: no conclusion).  The calib binary never hung (11 runs); bench.py never hung without --pmc.  NOT root-caused.
# Third session: a hung pass stands in vnrAmdSynchronize after bench.py's 1500 training steps; for counters of the
# render kernels use tools/share_probe.py as tools/r06_infer_bound.sh does (8 of 8 passes completed).  Always run under a short timeout;
# chaining four 600-s passes cost ~40 GPU-minutes once.  `-X faulthandler` + SIGABRT on timeout prints where it hangs.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc
mkdir -p "$O"
export TMPDIR=/tmp
what=$1; c=$2; to=${3:-90}
tag=$(echo $c | tr ' ' '+')
d=$O/${what}_$tag
if [ "$what" = calib ]; then
  (cd /tmp && timeout $to rocprofv3 --pmc $c --output-format csv -d "$d" -o calib -- $R/tools/calib/gather_calib) > "$d.log" 2>&1
else
  # SIGABRT (not TERM) on timeout: faulthandler then dumps the Python stack of the hung call into the log; KILL 10 s later
  (cd /tmp && timeout -s ABRT -k 10 $to rocprofv3 --pmc $c --output-format csv -d "$d" -o bench -- python3 -X faulthandler $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-psnr --no-alone) > "$d.log" 2>&1
fi
rc=$?
echo "[run_pmc] $what '$c' exit $rc"
f=$(ls "$d"/*counter_collection.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  python3 $R/tools/pmc_summary.py per-kernel "$f" > "$d.summary.txt"
  grep "fused_infer_kernel<2, 32, 0>\|gather_kernel\|stream16\|march_kernel" "$d.summary.txt" | cut -c1-200
fi
find "$O" -name "*.csv" -size +4M -delete
exit $rc
