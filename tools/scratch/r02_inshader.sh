#!/bin/bash
# in-shader kernels: tests, then path-tracing timings (VNR_AMD_PT_TILES_PER_WAVE belonged to the lane-refill experiment, DESIGN.md 7:
# measured slower, code removed; the sweep below is kept as the record of how it was measured)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_inshader; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_render.py -m gpu -x -q -k "path_tracing or in_shader" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for tpw in 0 1 4 16 32; do
  [ $tpw -gt 0 ] && export VNR_AMD_PT_TILES_PER_WAVE=$tpw
  timeout -k 10 200 python bench.py --mode 14 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --train-steps 300 --steps 30 > $O/pt14_t$tpw.json 2> $O/pt14_t$tpw.err && python tools/bench_line.py pt14_tpw$tpw < $O/pt14_t$tpw.json || tail -3 $O/pt14_t$tpw.err
done
unset VNR_AMD_PT_TILES_PER_WAVE
timeout -k 10 200 python bench.py --mode 15 --no-cpu-baseline --no-psnr --no-alone --no-brick-off --train-steps 300 --steps 30 > $O/pt15.json 2> $O/pt15.err && python tools/bench_line.py pt15 < $O/pt15.json
