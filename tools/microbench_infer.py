"""Micro-benchmark of the fused encode+MLP kernel (GPU box): Msamples/s and algorithmic GB/s."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from instantvnr_amd import api, synthetic as syn  # noqa: E402

L = api.lib()
api.check(L.vnrAmdInit(-1))


def run(name, n, coherent, **kw):
    cfg = syn.model_config(**kw)
    vol = api.vnrCreateNeuralVolume(cfg, (64, 64, 64))
    info = api.neural_info(vol)
    rng = np.random.default_rng(0)
    if coherent:  # ray-like: 16 consecutive samples along a ray, neighbouring rays adjacent
        rays = n // 16
        side = int(np.sqrt(rays))
        u, v = np.meshgrid(np.arange(side), np.arange(side))
        o = np.stack([u.ravel() / side, v.ravel() / side, np.zeros(side * side)], 1)
        t = (np.arange(16) / 1024.0 + 0.3)[None, :, None] * np.array([0.05, 0.02, 1.0])[None, None, :]
        c = (o[:, None, :] + t).reshape(-1, 3).astype(np.float32)
        c = np.clip(c, 0, 1)
        n = c.shape[0]
    else:
        c = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    dc = api.DeviceArray.from_numpy(c)
    do = api.DeviceArray((n,), np.float32)
    for _ in range(3):
        api.check(L.vnrAmdNeuralVolumeInference(vol.h, n, dc.ptr, do.ptr, None))
    api.check(L.vnrAmdSynchronize())
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        api.check(L.vnrAmdNeuralVolumeInference(vol.h, n, dc.ptr, do.ptr, None))
    api.check(L.vnrAmdSynchronize())
    dt = (time.perf_counter() - t0) / reps
    bytes_per = 12 + info["n_levels"] * 8 * info["n_features_per_level"] * 2 + 4
    print(f"{name:34s} n={n:9d} {dt*1e3:8.3f} ms  {n/dt/1e6:9.1f} Msamples/s  {n*bytes_per/dt/1e9:8.1f} GB/s algorithmic "
          f"({bytes_per} B/sample)", flush=True)


N = 1 << 24
run("C2 L8F8 T19 H2 random", N, False, n_levels=8, n_features=8, log2_hashmap_size=19, n_hidden_layers=2)
run("C2 L8F8 T19 H2 coherent", N, True, n_levels=8, n_features=8, log2_hashmap_size=19, n_hidden_layers=2)
run("C4 L16F2 T22 H3 s=1.32 random", N, False, n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=1.3195)
run("C4 L16F2 T22 H3 s=1.32 coherent", N, True, n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=1.3195)
run("C4 L16F2 T22 H3 s=2.0 random", N, False, n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3)
run("C4 L16F2 T19 H3 s=1.32 coherent", N, True, n_levels=16, n_features=2, log2_hashmap_size=19, n_hidden_layers=3, per_level_scale=1.3195)
