#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_inshader; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_render.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for k in 1 0 1 0; do
  VNR_AMD_IN_SHADER=$k timeout -k 10 200 python bench.py --mode 14 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --train-steps 300 --steps 30 > $O/pt14_k$k.json 2> $O/pt14_k$k.err && python tools/bench_line.py pt14_inshader$k < $O/pt14_k$k.json || tail -3 $O/pt14_k$k.err
done
