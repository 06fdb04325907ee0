"""GPU box: marginal cost of each hash-grid level in the fused inference kernel, on the REAL sample queue of a C4 frame.

A model with the first k levels of the C4 grid (same base resolution, per-level scale and table size cap) has exactly
the geometry of those levels, so time(k) - time(k-1) on the same coordinates is level k-1's cost (gathers + its share of
the first layer).  Parameters are random: inference has no data-dependent control flow.
    VNR_AMD_RENDER_HALVES=1 python tools/level_cost.py [iteration]      (one stream, so the debug hook sees the whole queue)
"""
import ctypes as C
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["VNR_AMD_RENDER_HALVES"] = "1"
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402

L = lib()
check(L.vnrAmdInit(-1))
size = 1024
dims = (size,) * 3
pls = float(np.exp(np.log(size / 16.0) / 15))
ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def real_queue():
    sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
    cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    api.vnrNeuralVolumeTrain(nv, 300, True)
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetFramebufferSize(ren, (1024, 1024))
    api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
    cam = syn.oblique_camera(dims, distance_scale=1.1)
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    api.vnrRendererSetCamera(ren, camera)
    colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    api.vnrRendererSetTransferFunction(ren, tfn)
    os.environ["VNR_AMD_DEBUG_MAX_ITERS"] = str(ITER)
    api.vnrRender(ren)
    api.vnrRendererMapFrame(ren)
    dc, dn = C.c_void_p(), C.c_void_p()
    ms = (C.c_float * 32)()
    check(L.vnrAmdRendererDebugQueues(ren.h, C.byref(dc), C.byref(dn), ms, 32))
    cnt = np.zeros(16, np.uint32)
    check(L.vnrAmdMemcpyD2H(cnt.ctypes.data_as(C.c_void_p), dn, 64))
    n = int(cnt[2 + ((ITER - 1) & 1)])
    rec = np.empty((n, 4), np.float32)
    check(L.vnrAmdMemcpyD2H(rec.ctypes.data_as(C.c_void_p), dc, n * 16))
    return np.ascontiguousarray(rec[:, :3])


coords = real_queue()
gc.collect()
n = coords.shape[0]
print(f"real queue of march iteration {ITER}: {n} samples (gather order = depth-bin sorted 8x8 tiles)")
d_c = api.DeviceArray.from_numpy(coords)
d_o = api.DeviceArray((n,), np.float32)
shuf = api.DeviceArray.from_numpy(coords[np.random.default_rng(0).permutation(n)])


def timed(nv, d_in):
    for _ in range(3):
        check(L.vnrAmdNeuralVolumeInference(nv.h, n, d_in.ptr, d_o.ptr, None))
    check(L.vnrAmdSynchronize())
    t0 = time.perf_counter()
    for _ in range(10):
        check(L.vnrAmdNeuralVolumeInference(nv.h, n, d_in.ptr, d_o.ptr, None))
    check(L.vnrAmdSynchronize())
    return (time.perf_counter() - t0) / 10 * 1e3


prev = prev_s = None
print(f"{'levels':>6s} {'res of last':>11s} {'entries':>9s} {'hashed':>6s} {'queue ms':>9s} {'+ms':>7s} {'shuffled ms':>11s} {'+ms':>7s}")
for k in (1, 2, 4, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16):
    cfg = syn.model_config(n_levels=k, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
    nv = api.vnrCreateNeuralVolume(cfg, (64, 64, 64))
    scale = np.exp2((k - 1) * np.log2(pls)) * 16 - 1
    res = int(np.ceil(scale)) + 1
    entries = min(((res ** 3 + 7) // 8) * 8, 1 << 22)
    t = timed(nv, d_c)
    ts = timed(nv, shuf)
    print(f"{k:6d} {res:11d} {entries:9d} {str(res ** 3 > entries):>6s} {t:9.3f} {'' if prev is None else f'{t - prev:+7.3f}'} {ts:11.3f} {'' if prev_s is None else f'{ts - prev_s:+7.3f}'}",
          flush=True)
    prev, prev_s = t, ts
    del nv
    gc.collect()
