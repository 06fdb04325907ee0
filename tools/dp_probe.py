"""GPU box, one rank: what the data-parallel step costs before any byte crosses xGMI.  A one-rank RCCL communicator runs the real
step (vnrAmdNeuralVolumeTrainDataParallel: pack to fp16 range by range, ncclAllReduce on the communication stream, range-wise Adam) on
the C4 model next to the plain step, and the exchange alone (the 140 MB fp16 payload in the step's ranges, and as one message).
usage: MASTER_PORT=29700 WORLD_SIZE=1 RANK=0 VNR_AMD_DIST_FORCE=1 python tools/dp_probe.py"""
import ctypes as C
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VNR_AMD_DIST_FORCE", "1")
os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("MASTER_PORT", "29700")
from instantvnr_amd import api, dist, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
ctx = dist.init_from_env()
L = lib()
print("transport", ctx.transport, "world", ctx.world, flush=True)
size = int(os.environ.get("SIZE", 1024))
dims = (size,) * 3
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=float(np.exp(np.log(size / 16.0) / 15)))


def timed(fn, steps):
    fn(50)
    check(L.vnrAmdSynchronize())
    t = time.perf_counter()
    fn(steps)
    check(L.vnrAmdSynchronize())
    return (time.perf_counter() - t) * 1e3 / steps


nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
n_params = api.neural_info(nv)["n_params"]
plain = timed(lambda k: api.vnrNeuralVolumeTrain(nv, k, True), 300)
nv2 = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
dp = timed(lambda k: check(L.vnrAmdNeuralVolumeTrainDataParallel(nv2.h, k, 1)), 300)
check(L.vnrAmdNeuralVolumeSyncReplicas(nv2.h))
nv3 = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
hand = timed(lambda k: dist.train_data_parallel_by_hand(ctx, nv3, k), 100)
print(f"[VNR_AMD_DP_SHARDED={os.environ.get('VNR_AMD_DP_SHARDED', '1')} VNR_AMD_DP_EMULATE_WORLD={os.environ.get('VNR_AMD_DP_EMULATE_WORLD', '-')}] "
      f"C4 model, {n_params} parameters: plain step {plain:.3f} ms; data-parallel step on a one-rank communicator {dp:.3f} ms "
      f"(fp16 payload {n_params * 2 / 1e6:.1f} MB per step in ranges, exchange overlapped); by hand (one message, nothing overlapped) {hand:.3f} ms", flush=True)
# the exchange alone
buf = api.DeviceArray((n_params,), np.float16)
buf.zero()
for label, chunks in (("one message", 1), ("9 messages", 9)):
    per = (n_params // chunks) & ~7
    def go(k):
        for _ in range(k):
            for c in range(chunks):
                check(L.vnrAmdDistAllReduce(C.c_void_p(buf.ptr + 2 * c * per), per, dist.F16, dist.SUM))
    ms = timed(go, 50)
    print(f"ncclAllReduce of {per * chunks * 2 / 1e6:.1f} MB fp16 as {label} on one rank (blocking calls): {ms:.3f} ms", flush=True)
dist.finalize()
