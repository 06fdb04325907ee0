"""GPU box: is the gather cost per lane (fixed) or per distinct cache line?"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from instantvnr_amd import api, synthetic as syn  # noqa: E402
L = api.lib(); api.check(L.vnrAmdInit(-1))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=1.3195)
vol = api.vnrCreateNeuralVolume(cfg, (64, 64, 64))
n = 1 << 24
def run(name, c, encode_only=False):
    c = np.ascontiguousarray(c, np.float32)
    dc = api.DeviceArray.from_numpy(c); do = api.DeviceArray((n,), np.float32)
    de = api.DeviceArray((n, 32), np.uint16) if encode_only else None
    def go():
        if encode_only: api.check(L.vnrAmdNeuralVolumeEncode(vol.h, n, dc.ptr, de.ptr, None))
        else: api.check(L.vnrAmdNeuralVolumeInference(vol.h, n, dc.ptr, do.ptr, None))
    for _ in range(3): go()
    api.check(L.vnrAmdSynchronize()); t0 = time.perf_counter()
    for _ in range(10): go()
    api.check(L.vnrAmdSynchronize()); dt = (time.perf_counter() - t0) / 10
    print(f"{name:40s} {dt*1e3:7.3f} ms {n/dt/1e6:8.1f} Msamples/s", flush=True)
rng = np.random.default_rng(0)
same = np.tile(np.array([[0.3, 0.4, 0.5]], np.float32), (n, 1))
run("all samples identical", same)
w = rng.uniform(0, 1, (n // 64, 1, 3)).astype(np.float32)
run("identical within a wave, random across", np.broadcast_to(w, (n // 64, 64, 3)).reshape(n, 3))
# 64 lanes = 64 consecutive x positions at finest spacing, same y,z (best possible coalescing)
base = rng.uniform(0.1, 0.9, (n // 64, 1, 3)).astype(np.float32)
off = np.zeros((1, 64, 3), np.float32); off[0, :, 0] = np.arange(64) / 1024.0
run("x-runs of 64 voxels, random across waves", (base + off).reshape(n, 3))
base2 = np.stack(np.meshgrid(np.arange(16) * 64 / 1024.0, np.arange(128) / 1024.0 * 8, np.arange(128) / 1024.0 * 8, indexing="ij"), -1).reshape(-1, 1, 3)[: n // 64].astype(np.float32)
run("x-runs of 64 voxels, coherent across waves", (base2 + off).reshape(n, 3))
run("x-runs, encode only", (base2 + off).reshape(n, 3), encode_only=True)
run("identical, encode only", same, encode_only=True)
