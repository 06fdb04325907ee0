#!/bin/bash
# RECORD of an experiment: VNR_AMD_STAGGER / VNR_AMD_PART_PRIORITY were switches of experimental builds that ordered the ray
# parts' marches across streams / raised one part's stream priority; both measured slower and the code is gone, so today this script
# only compares identical runs.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_halves; mkdir -p $O
cd $R
for st in 0 1 0 1; do
  VNR_AMD_STAGGER=$st timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-alone --no-brick-off --train-steps 300 > $O/s$st.json 2> $O/s$st.err && python tools/bench_line.py stagger$st < $O/s$st.json
done
for st in 0 1; do echo "stagger $st: $(VNR_AMD_STAGGER=$st SHARE_PARTS=8 SHARE_PIPELINED=1 timeout -k 10 200 python tools/share_probe.py 2>&1 | grep 'share 1/')"; done
