#!/bin/bash
# the reference's example model shape (example-model.json: L = 8, F = 8, T = 2^19, base 16, scale 2, 4 hidden layers) on the C4 frame, and C2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_f8; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --levels 8 --features 8 --log2-hashmap-size 19 --hidden-layers 4 --per-level-scale 2 --no-cpu-baseline --steps 40 --train-steps 500 > $O/example_model_c4.json 2> $O/example_model_c4.err; echo "rc=$?"
timeout -k 10 300 python bench.py --size 128 --fb 512 --levels 8 --features 8 --log2-hashmap-size 19 --hidden-layers 2 --per-level-scale 2 --no-cpu-baseline --steps 100 --train-steps 1000 > $O/c2.json 2> $O/c2.err; echo "rc=$?"
python3 - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r02_f8/"
for n in ("example_model_c4","c2"):
    try:
        b=json.loads(open(O+n+".json").read().strip().splitlines()[-1])
        r=b["roofline"]
        print(n, b["value"], "fps", b["ms_per_step"], "ms; samples", b["samples_per_frame"], "frac", r["frac"], "union", r["union"]["frac"], "alone", r.get("alone",{}).get("frac"), r.get("alone",{}).get("msamples_per_s"), "B/sample", r["algorithmic_bytes_per_sample"], "cache", b["inference_cache"].get("brick_image_bytes"), "train", b.get("train_ms_per_step"), "psnr", b.get("psnr_db"))
    except Exception as e:
        print(n, "failed", e); print(open(O+n+".err").read()[-1500:])
PY
