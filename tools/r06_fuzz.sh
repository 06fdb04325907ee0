#!/bin/bash
# GPU box: every random sweep of tests/test_gpu_fuzz.py at a wider count (round 6: after the four-entry group gathers, fp32 sums for F = 1, the cache tiers, the row-wise image build; round 5: after the
# side-by-side backward pass, the ranks out of LDS).   usage: r06_fuzz.sh <tag> <seed> [model draws] [in-shader draws] [neural frames] [dense scenes]
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-a}; seed=${2:-707}; draws=${3:-1500}; inshader=${4:-300}; frames=${5:-30}; scenes=${6:-600}
mkdir -p gpurun_out/r06_fuzz
export VNR_FUZZ_SEED=$seed VNR_FUZZ_DRAWS=$draws VNR_FUZZ_IN_SHADER=$inshader VNR_FUZZ_FRAMES=$frames VNR_FUZZ_SCENES=$scenes VNR_FUZZ_SHARES=200 VNR_FUZZ_OPTIMIZERS=400 \
       VNR_FUZZ_DAMAGED=600 VNR_FUZZ_ODD=600 VNR_FUZZ_MC=100 VNR_FUZZ_OOC=100 VNR_FUZZ_DECODE=20 VNR_FUZZ_LOG=gpurun_out/r06_fuzz/draws_$tag.log
rm -f "$VNR_FUZZ_LOG"
timeout -k 10 1050 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --durations=20 > gpurun_out/r06_fuzz/pytest_$tag.log 2>&1
rc=$?
echo "pytest rc=$rc"
tail -28 gpurun_out/r06_fuzz/pytest_$tag.log | cut -c1-600
echo "draws ok: $(grep -c ' ok$' "$VNR_FUZZ_LOG"), vacuous: $(grep -c vacuous "$VNR_FUZZ_LOG"), FAIL: $(grep -c FAIL "$VNR_FUZZ_LOG")"
exit $rc
