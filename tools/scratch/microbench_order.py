"""GPU box: how the ORDER of samples in the batch changes the fused kernel's speed (gather coalescing)."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from instantvnr_amd import api, synthetic as syn  # noqa: E402
L = api.lib()
api.check(L.vnrAmdInit(-1))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=1.3195)
vol = api.vnrCreateNeuralVolume(cfg, (64, 64, 64))
W = 1024  # rays per image side
K = 16
d = np.array([0.5, 0.3, 0.8]); d /= np.linalg.norm(d)
u = np.cross(d, [0, 1, 0]); u /= np.linalg.norm(u)
v = np.cross(d, u)
px, py = np.meshgrid(np.arange(W), np.arange(W))  # py rows, px cols
def pos(ix, iy, k, step):
    o = 0.5 + (ix[..., None] - W / 2) / 1024.0 * u + (iy[..., None] - W / 2) / 1024.0 * v
    return o + ((k[..., None] * step + 200) / 1024.0) * d * 0.5
def run(name, c):
    c = np.clip(c.reshape(-1, 3), 0, 1).astype(np.float32)
    n = c.shape[0]
    dc = api.DeviceArray.from_numpy(c); do = api.DeviceArray((n,), np.float32)
    for _ in range(3): api.check(L.vnrAmdNeuralVolumeInference(vol.h, n, dc.ptr, do.ptr, None))
    api.check(L.vnrAmdSynchronize())
    t0 = time.perf_counter()
    for _ in range(10): api.check(L.vnrAmdNeuralVolumeInference(vol.h, n, dc.ptr, do.ptr, None))
    api.check(L.vnrAmdSynchronize())
    dt = (time.perf_counter() - t0) / 10
    print(f"{name:46s} n={n} {dt*1e3:7.3f} ms {n/dt/1e6:8.1f} Msamples/s", flush=True)
for step in (1.0, 8.0):
    ix = px.ravel(); iy = py.ravel()
    k = np.arange(K)
    # A: ray-major (scanline ray order): [ray][k]
    A = pos(ix[:, None].repeat(K, 1), iy[:, None].repeat(K, 1), np.broadcast_to(k, (W * W, K)), step)
    run(f"step {step}: ray-major, scanline rays", A)
    # B: k-major inside groups of 64 consecutive scanline rays: [group][k][64]
    Bc = A.reshape(W * W // 64, 64, K, 3).transpose(0, 2, 1, 3)
    run(f"step {step}: k-major, 64x1 ray groups", Bc)
    # C: 8x8 pixel tiles, k-major
    t = A.reshape(W // 8, 8, W // 8, 8, K, 3).transpose(0, 2, 4, 1, 3, 5)  # [ty][tx][k][ly][lx]
    run(f"step {step}: k-major, 8x8 ray tiles", t)
    # D: 8x8 pixel tiles, ray-major inside the tile
    t2 = A.reshape(W // 8, 8, W // 8, 8, K, 3).transpose(0, 2, 1, 3, 4, 5)
    run(f"step {step}: ray-major, 8x8 ray tiles", t2)
    # E: 4x4 ray tiles x 4 consecutive k per wave
    t3 = A.reshape(W // 4, 4, W // 4, 4, K // 4, 4, 3).transpose(0, 2, 4, 5, 1, 3, 6)
    run(f"step {step}: 4x4 rays x 4 steps bricks", t3)
rng = np.random.default_rng(0)
run("random", rng.uniform(0, 1, (W * W * K, 3)))
