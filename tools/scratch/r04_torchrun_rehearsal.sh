#!/bin/bash
# round 4: the driver's N > 1 command line, verbatim, with the host-staged transport so that the ranks can share this box's one GPU:
# launcher, rendezvous, sharding, the sharded optimizer's exchange and the teardown at HEAD.  2 ranks small + full size, 4 ranks small.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_torchrun; mkdir -p $O
cd $R
export VNR_AMD_DIST_TRANSPORT=shm VNR_AMD_DIST_TIMEOUT=120
for W in 2 4; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29770 + W)) bench.py --gpus $W --steps 10 --warmup 2 --size 256 --fb 512 --train-steps 200 --no-cpu-baseline > $O/small$W.out 2> $O/small$W.err; echo "small W=$W rc=$?"; grep '^{' $O/small$W.out | cut -c1-400
done
(timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29779 bench.py --gpus 2 --steps 20 --warmup 5 --train-steps 60 --no-cpu-baseline > $O/full.out 2> $O/full.err; echo "full rc=$?" > $O/full.rc) &
pid=$!
while kill -0 $pid 2>/dev/null; do sleep 20; echo "... waiting $(date +%T) $(tail -c 200 $O/full.err | tr '\n' ' ' | cut -c1-150)"; done
cat $O/full.rc; grep '^{' $O/full.out | cut -c1-600; tail -3 $O/full.err
