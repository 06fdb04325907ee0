#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: tools/ab/ab.sh <old.so> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
old=$1; shift
cp $R/instantvnr_amd/libvnr_amd.so /tmp/new.so
for i in 1 2 3; do
  cp /tmp/new.so $R/instantvnr_amd/libvnr_amd.so
  timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-psnr --no-alone "$@" 2>&1 | python $R/tools/bench_line.py new
  cp $old $R/instantvnr_amd/libvnr_amd.so
  timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-psnr --no-alone "$@" 2>&1 | python $R/tools/bench_line.py old
done
cp /tmp/new.so $R/instantvnr_amd/libvnr_amd.so
