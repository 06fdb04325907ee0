"""GPU box: take the REAL first-iteration sample queue of a C4 frame and time the fused kernel on it in
different orders (is the in-frame rate limited by the access pattern or by launch structure?)."""
import ctypes as C
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
size = 1024
dims = (size,) * 3
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
pls = float(np.exp(np.log(size / 16.0) / 15))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 300, True)
ren = api.vnrCreateRenderer(nv)
api.vnrRendererSetFramebufferSize(ren, (1024, 1024))
api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
api.vnrRendererSetProfiling(ren, True)
cam = syn.oblique_camera(dims, distance_scale=1.1)
camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
api.vnrRendererSetCamera(ren, camera)
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)
tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
for _ in range(3):
    api.vnrRender(ren); api.vnrRendererMapFrame(ren)
ms = (C.c_float * 32)()
dc, dn = C.c_void_p(), C.c_void_p()
check(L.vnrAmdRendererDebugQueues(ren.h, C.byref(dc), C.byref(dn), ms, 32))
st = api.vnrRendererGetFrameStats(ren)
print("frame stats", st)
print("per-iteration infer ms:", [round(v, 3) for v in ms[:st["n_iterations"] + 1]])
# first-iteration queue
ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 1
os.environ["VNR_AMD_DEBUG_MAX_ITERS"] = str(ITER)
api.vnrRendererResetAccumulation(ren)
api.vnrRender(ren); api.vnrRendererMapFrame(ren)
cnt = np.zeros(16, np.uint32)
check(L.vnrAmdMemcpyD2H(cnt.ctypes.data_as(C.c_void_p), dn, 64))
n = int(cnt[2 + ((ITER - 1) & 1)])
print("iteration", ITER, "counters", cnt[:4], "samples", n)
rec = np.empty((n, 4), np.float32)
check(L.vnrAmdMemcpyD2H(rec.ctypes.data_as(C.c_void_p), dc, n * 16)); coords = np.ascontiguousarray(rec[:, :3])
d = np.linalg.norm(np.diff(coords[:4096], axis=0), axis=1) * size
print("spacing of consecutive queue entries (voxels): median %.2f  p90 %.2f" % (np.median(d), np.quantile(d, 0.9)))

def run(name, c):
    c = np.ascontiguousarray(c, np.float32)
    m = c.shape[0]
    d_c = api.DeviceArray.from_numpy(c); d_o = api.DeviceArray((m,), np.float32)
    for _ in range(3): check(L.vnrAmdNeuralVolumeInference(nv.h, m, d_c.ptr, d_o.ptr, None))
    check(L.vnrAmdSynchronize()); t0 = time.perf_counter()
    for _ in range(10): check(L.vnrAmdNeuralVolumeInference(nv.h, m, d_c.ptr, d_o.ptr, None))
    check(L.vnrAmdSynchronize()); dt = (time.perf_counter() - t0) / 10
    print(f"{name:44s} n={m:9d} {dt*1e3:7.3f} ms {m/dt/1e6:8.1f} Msamples/s", flush=True)

run("real queue order (ray-major)", coords)
q = np.floor(coords * 1024).astype(np.int64).clip(0, 1023)
def morton(q):
    def part(x):
        x = (x | (x << 16)) & 0x030000FF0000FF
        x = (x | (x << 8)) & 0x0300F00F00F00F
        x = (x | (x << 4)) & 0x030C30C30C30C3
        x = (x | (x << 2)) & 0x09249249249249
        return x
    return part(q[:, 0]) | (part(q[:, 1]) << 1) | (part(q[:, 2]) << 2)
run("sorted by Morton code (upper bound)", coords[np.argsort(morton(q), kind="stable")])
run("shuffled (lower bound)", coords[np.random.default_rng(0).permutation(n)])
# k-major inside groups of 64 consecutive rays (queue is ray-major with <=16 per ray; approximate with reshape)
m16 = (n // 1024) * 1024
run("k-major in 64-ray groups (approx)", coords[:m16].reshape(-1, 64, 16, 3).transpose(0, 2, 1, 3).reshape(-1, 3))
