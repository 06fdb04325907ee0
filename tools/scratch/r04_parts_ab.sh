#!/bin/bash
# round 4: the whole frame dealt to 2 (default), 3 or 4 ray parts on as many streams, alternating on one box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04_parts
for rep in 1 2; do for h in 2 3 4; do
  VNR_AMD_RENDER_HALVES=$h timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-alone --no-brick-off --no-brick-table --train-steps 600 > gpurun_out/r04_parts/b_${h}_$rep.json 2> gpurun_out/r04_parts/b_${h}_$rep.err || exit 1
  python - "$h" "$rep" <<'P'
import json,sys
d=json.loads(open(f"gpurun_out/r04_parts/b_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(f"parts {sys.argv[1]} rep {sys.argv[2]}: {d['value']:.1f} frames/s  {d['ms_per_step']:.3f} ms  launch {r['avg_launch_ms']:.4f} ms x {r['launches']}  union_frac {r.get('union_frac')}  iterations {d['iterations_per_frame']}")
P
done; done
