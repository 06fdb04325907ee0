"""GPU box: training throughput from an out-of-core volume (SURVEY 8 a15 / BASELINE C5 in miniature).
Writes a synthetic uint8 volume file under /tmp, opens it with vnrCreateSimpleVolumeOutOfCore and trains the C4-shaped model.
usage: python tools/ooc_bench.py [nx ny nz] [n_concurrent_blocks] [n_blocks] [steps] [keep the file: 0 | 1]
VNR_AMD_OOC_ASYNC=1: asynchronous refresh (a step never waits for the storage; include/vnr_amd.h)"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
nx, ny, nz = (a + [1024, 1024, 2048])[:3] if len(a) >= 3 else (1024, 1024, 2048)
ncb = a[3] if len(a) > 3 else 1024
nb = a[4] if len(a) > 4 else 16384
steps = a[5] if len(a) > 5 else 200
L = lib(); check(L.vnrAmdInit(-1))
path = f"/tmp/ooc_{nx}x{ny}x{nz}.raw"
t0 = time.perf_counter()
if not os.path.exists(path) or os.path.getsize(path) != nx * ny * nz:
    x = np.linspace(0, 1, nx, dtype=np.float32)[None, None, :]
    y = np.linspace(0, 1, ny, dtype=np.float32)[None, :, None]
    with open(path, "wb") as f:
        for z0 in range(0, nz, 16):
            z = (np.arange(z0, min(z0 + 16, nz), dtype=np.float32) / nz)[:, None, None]
            v = 0.5 + 0.5 * np.sin(40 * x + 9 * z) * np.cos(31 * y) * np.sin(23 * z + 5 * x * y)
            f.write((v * 255.0 + 0.5).astype(np.uint8).tobytes())
print(f"file {path}: {nx * ny * nz / 2**30:.2f} GiB written in {time.perf_counter() - t0:.1f} s", flush=True)
t0 = time.perf_counter()
sv = api.vnrCreateSimpleVolumeOutOfCore(path, (nx, ny, nz), np.uint8, (0.0, 255.0), n_concurrent_blocks=ncb, n_blocks=nb)
info = api.out_of_core_info(sv)
t_pre = time.perf_counter() - t0
print(f"preload: {info['bytes_read'] / 2**30:.2f} GiB in {t_pre:.2f} s = {info['bytes_read'] / 2**30 / t_pre:.2f} GiB/s; "
      f"slab {info['block_dims']} -> {info['block_size_aligned']} B; {info['n_blocks']} resident slabs = "
      f"{info['n_blocks'] * info['block_size_aligned'] / 2**30:.2f} GiB of HBM; {info['n_concurrent_blocks']} replaced per step", flush=True)
pls = float(np.exp(np.log(max(nx, ny, nz) / 16.0) / 15))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=True)
api.vnrNeuralVolumeTrain(nv, 10, False)
check(L.vnrAmdSynchronize())
b0 = api.out_of_core_info(sv)["bytes_read"]
t0 = time.perf_counter()
api.vnrNeuralVolumeTrain(nv, steps, False)
check(L.vnrAmdSynchronize())
dt = time.perf_counter() - t0
b1 = api.out_of_core_info(sv)["bytes_read"]
print(f"training from the file: {steps} steps, {dt / steps * 1e3:.3f} ms per step (65536 samples each) = {65536 * steps / dt / 1e6:.1f} M samples/s; "
      f"refresh traffic {(b1 - b0) / steps / 2**20:.1f} MiB per step = {(b1 - b0) / dt / 2**30:.2f} GiB/s from the page cache; loss {api.vnrNeuralVolumeGetTrainingLoss(nv):.4f}", flush=True)
import ctypes as C
n_ref, n_busy = C.c_uint64(), C.c_uint64()
check(L.vnrAmdSimpleVolumeOutOfCoreRefreshStats(sv.h, C.byref(n_ref), C.byref(n_busy)))
print(f"refreshes submitted {n_ref.value} (pre-load {nb // ncb}), steps that ran beside a refresh in flight {n_busy.value}; asynchronous refresh: {os.environ.get('VNR_AMD_OOC_ASYNC', '0')}", flush=True)
if not (len(a) > 6 and a[6]):
    os.remove(path)
