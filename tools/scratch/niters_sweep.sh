#!/bin/bash
# N_ITERS sweep of the bench workload (diagnostics)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
out=gpurun_out/niters_sweep.log
: > $out
for rep in 1 2; do
  for n in 16 24 32; do
    echo "== rep=$rep n_iters=$n" >> $out
    VNR_RM_N_ITERS=$n timeout -k 10 120 python bench.py --no-cpu-baseline --no-psnr --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('fps', d['value'], 'ms', d['ms_per_step'], 'kernel_only', d['mlp_msamples_per_s_kernel_only'], 'launch_ms', r['avg_launch_ms'], 'frac', r['frac'], 'frame_frac', r['frame_frac'], 'samples', d['samples_per_frame'], 'iters', d['iterations_per_frame'])" >> $out 2>&1 || exit 1
  done
done
cat $out
