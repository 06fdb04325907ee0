#!/bin/bash
# tile-shape / halves sweep of the bench workload (diagnostics): prints fps, kernel-only rate and avg launch ms
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
out=gpurun_out/tile_sweep.log
: > $out
for halves in 2 1; do
  for w in 8 16 32 64; do
    echo "== halves=$halves tile_w=$w" >> $out
    VNR_AMD_RENDER_HALVES=$halves VNR_AMD_TILE_W=$w timeout -k 10 120 python bench.py --no-cpu-baseline --no-psnr --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']
print('fps', d['value'], 'ms', d['ms_per_step'], 'kernel_only', d['mlp_msamples_per_s_kernel_only'], 'launch_ms', r['avg_launch_ms'], 'frac', r['frac'], 'samples', d['samples_per_frame'], 'iters', d['iterations_per_frame'])" >> $out 2>&1 || exit 1
  done
done
cat $out
