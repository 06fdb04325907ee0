"""GPU box: how the synthetic C4 workload's cost depends on TFN opacity and camera distance."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402

L = lib()
check(L.vnrAmdInit(-1))
size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
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
for dist_scale in (1.6, 1.1):
    cam = syn.oblique_camera(dims, distance_scale=dist_scale)
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    api.vnrRendererSetCamera(ren, camera)
    for osc in (0.03, 0.06, 0.12, 0.25, 0.5, 1.0):
        colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=osc)
        tfn = api.vnrCreateTransferFunction()
        api.vnrTransferFunctionSetColor(tfn, colors)
        api.vnrTransferFunctionSetAlpha(tfn, alphas)
        api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
        api.vnrRendererSetTransferFunction(ren, tfn)
        for _ in range(3):
            api.vnrRender(ren)
        api.vnrRendererMapFrame(ren)
        t0 = time.perf_counter()
        n = 10
        ims = 0.0
        for _ in range(n):
            api.vnrRender(ren)
            api.vnrRendererMapFrame(ren)
            ims += api.vnrRendererGetFrameStats(ren)["infer_kernel_ms"]
        dt = (time.perf_counter() - t0) / n
        st = api.vnrRendererGetFrameStats(ren)
        print(f"cam {dist_scale} opacity {osc:5.2f}: {1/dt:7.1f} fps  {dt*1e3:7.2f} ms  infer {ims/n:6.2f} ms  rays {st['n_rays_hit']:7d} "
              f"samples {st['n_samples']/1e6:7.2f} M  /ray {st['n_samples']/max(st['n_rays_hit'],1):6.1f}  ref slots {st['n_reference_slots']/1e6:7.2f} M "
              f"iters {st['n_iterations']:3d}  kernel {st['n_samples']/max(ims/n,1e-9)/1e3:7.1f} Msamp/s", flush=True)
