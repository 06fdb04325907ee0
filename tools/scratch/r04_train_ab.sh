#!/bin/bash
# round 4: the training step of the C4 model with and without the side stream (alternating, one box); extra VAR=value pairs as arguments are
# exported for every run.  usage: r04_train_ab.sh [reps] [VAR=v ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
reps=${1:-2}; shift
for kv in "$@"; do export "$kv"; done
cd $R
for i in $(seq 1 $reps); do
  VNR_AMD_TRAIN_SIDE_STREAM=0 timeout -k 10 120 python tools/train_probe.py 600 2>&1 | grep train_probe || exit 1
  VNR_AMD_TRAIN_SIDE_STREAM=1 timeout -k 10 120 python tools/train_probe.py 600 2>&1 | grep train_probe || exit 1
done
