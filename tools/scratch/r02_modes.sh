#!/bin/bash
# frames/s of the other rendering modes on the bench frame
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r02_modes
for m in 6 8 9 11 12 14 15; do
  timeout -k 10 200 python bench.py --mode $m --steps 40 --no-cpu-baseline --no-psnr --no-alone --no-brick-off > gpurun_out/r02_modes/m$m.json 2> gpurun_out/r02_modes/m$m.err
  python3 -c "
import json; j=json.load(open('gpurun_out/r02_modes/m$m.json')); print('mode $m', j['value'], 'frames/s', j['ms_per_step'], 'ms', j['samples_per_frame'], 'samples', j['config'].get('rendering_mode', ''))" 2>&1 | tail -1
done
