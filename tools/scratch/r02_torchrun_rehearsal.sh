#!/bin/bash
# the driver's N > 1 command line, verbatim, with the host-staged transport so that the ranks can share this box's one GPU
# (a rehearsal of launcher, rendezvous, sharding and exchange; the host-staged exchange of a 140 MB gradient takes ~0.2 s per step,
# so the full-size configuration is run with 60 training steps)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02_torchrun; mkdir -p $O
cd $R
export VNR_AMD_DIST_TRANSPORT=shm VNR_AMD_DIST_TIMEOUT=120
W=${1:-2}
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29671 bench.py --gpus $W --steps 10 --warmup 2 --size 256 --fb 512 --train-steps 200 --no-cpu-baseline > $O/small.out 2> $O/small.err; echo "small rc=$?"; grep '^{' $O/small.out | cut -c1-300
(timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29672 bench.py --gpus $W --steps 20 --warmup 5 --train-steps 60 --no-cpu-baseline > $O/full.out 2> $O/full.err; echo "full rc=$?" > $O/full.rc) &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 20; echo "... waiting $(date +%T) $(tail -c 200 $O/full.err | tr '\n' ' ' | cut -c1-150)"; done
cat $O/full.rc; grep '^{' $O/full.out | cut -c1-400; tail -3 $O/full.err
