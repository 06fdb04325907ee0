#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_quick; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python tools/bench_line.py q < $O/bench.json; python3 -c "
import json; j=json.load(open('$O/bench.json')); print('train_ms', j['train_ms_per_step'], 'psnr', j['psnr_db'], 'img', j.get('image_vs_ground_truth')); print(j['train_roofline']['kernels_ms'])"
timeout -k 10 300 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -s 2>&1 | grep "C3:"
