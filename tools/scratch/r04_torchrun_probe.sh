#!/bin/bash
# round 4, late: the driver's N > 1 command line with NO transport chosen, so that bench.py probes RCCL in child processes first.  On this
# box both ranks share one GPU, which RCCL refuses: the rehearsal of a new installation where RCCL does not come up.  Expected: the children
# fail within seconds, the ranks go on over shared memory and the ONE line carries transport_probe.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04_torchrun; mkdir -p $O
cd $R
unset VNR_AMD_DIST_TRANSPORT
export VNR_AMD_DIST_TIMEOUT=120 VNR_BENCH_PROBE_LIMIT=120
for W in 2 4; do
  t0=$(date +%s)
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29870 + W)) bench.py --gpus $W --steps 10 --warmup 2 --size 256 --fb 512 --train-steps 200 --no-cpu-baseline > $O/probe$W.out 2> $O/probe$W.err
  echo "probe W=$W rc=$? ($(( $(date +%s) - t0 )) s)"
  grep '^{' $O/probe$W.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['config']['parallelism'][:120]); print(d['transport_probe'])"
  grep "RCCL probe failed" $O/probe$W.err | head -2 | cut -c1-300
done
