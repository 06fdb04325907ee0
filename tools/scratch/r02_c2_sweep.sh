#!/bin/bash
# C2 (512^2 of a 128^3 volume, L = 8 F = 8, 2 x 64): ray parts and N_ITERS
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r02_c2
for cfg in 4:0 2:0 1:0 4:32 2:32 3:0 4:16; do
  parts=${cfg%%:*}; n=${cfg##*:}
  if [ "$n" = "0" ]; then unset VNR_RM_N_ITERS; else export VNR_RM_N_ITERS=$n; fi
  VNR_AMD_SMALL_SHARE_PARTS=$parts timeout -k 10 200 python bench.py --size 128 --fb 512 --levels 8 --features 8 --log2-hashmap-size 19 --hidden-layers 2 --per-level-scale 2 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --steps 200 --train-steps 300 > gpurun_out/r02_c2/b.json 2>/dev/null
  python3 -c "
import json; j=json.load(open('gpurun_out/r02_c2/b.json')); print('parts $parts N_ITERS $n:', j['value'], 'frames/s', j['ms_per_step'], 'ms', j['iterations_per_frame'], 'iterations', j['samples_per_frame'], 'samples')"
done
