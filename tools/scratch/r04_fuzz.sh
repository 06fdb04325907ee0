#!/bin/bash
# GPU box: the randomly drawn model sweep of tests/test_gpu_fuzz.py at a wider count.  usage: r04_fuzz.sh <tag> <draws> <seed> [frames] [in-shader draws] [dense scenes] [share draws] [optimizer draws] [damaged descriptions] [odd renderer draws]
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
tag=${1:-a}; draws=${2:-300}; seed=${3:-7}; frames=${4:-12}; inshader=${5:-16}; scenes=${6:-60}; shares=${7:-24}; optimizers=${8:-30}; damaged=${9:-150}; odd=${10:-120}
mkdir -p gpurun_out/r04_fuzz
export VNR_FUZZ_ODD=$odd VNR_FUZZ_DAMAGED=$damaged VNR_FUZZ_OPTIMIZERS=$optimizers VNR_FUZZ_SHARES=$shares VNR_FUZZ_SCENES=$scenes VNR_FUZZ_IN_SHADER=$inshader VNR_FUZZ_FRAMES=$frames VNR_FUZZ_DRAWS=$draws VNR_FUZZ_SEED=$seed VNR_FUZZ_LOG=gpurun_out/r04_fuzz/draws_$tag.log
rm -f "$VNR_FUZZ_LOG"
timeout -k 10 1000 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r04_fuzz/pytest_$tag.log 2>&1
rc=$?
echo "pytest rc=$rc"
tail -5 gpurun_out/r04_fuzz/pytest_$tag.log | cut -c1-2000
grep -c " ok$" "$VNR_FUZZ_LOG"; grep -c vacuous "$VNR_FUZZ_LOG"; grep -c FAIL "$VNR_FUZZ_LOG"
exit $rc
