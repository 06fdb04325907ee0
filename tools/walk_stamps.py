"""GPU box, diagnostic build (tools/ab_build.sh stamps -DVNR_MARCH_STAMPS): where a trip of walk_kernel<false> (csrc/decoupled.h) spends its
cycles on a share of the bench frame, with the evaluation switched off (VNR_AMD_DEBUG_SKIP_EVAL=1): the walk alone.
usage: VNR_AMD_LIB_PATH=.../libvnr_amd_stamps.so VNR_AMD_DECOUPLED=2 VNR_AMD_DEBUG_SKIP_EVAL=1 python tools/walk_stamps.py [shares]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
size, fb = 1024, 1024
dims = (size,) * 3
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=float(np.exp(np.log(size / 16.0) / 15)))
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 300, True)
cam = syn.oblique_camera(dims, distance_scale=1.1)
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)
names = ["walk (state load + DDA + emit to LDS)", "prefix sum + block claim (2 barriers)", "depth-bin sort", "queue records + dt stores (drained)", "whole trip"]
for parts in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8,16").split(",")]:
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetFramebufferSize(ren, (fb, fb)); api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
    camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"]); api.vnrRendererSetCamera(ren, camera)
    tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
    if parts > 1:
        api.vnrRendererSetPixelInterleave(ren, 8 * fb, parts, 0)
    for _ in range(4):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    out = (C.c_ulonglong * 16)()
    L.vnrAmdDebugMarchStamps(out, 1)
    n = 10
    for _ in range(n):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    check(L.vnrAmdSynchronize())
    L.vnrAmdDebugMarchStamps(out, 1)
    trips = max(out[7], 1)
    if os.environ.get("VNR_AMD_DECOUPLED_LANES", "1") == "8":   # walk8_kernel: a trip is one wave's part (8 rays) of a 64-ray group
        print(f"share 1/{parts}: {trips / n:.0f} wave-trips of walk8_kernel<false> per frame; cycles per trip (s_memtime):")
        rows = [("walk (state load + rounds of 8 cells)", out[0]), ("claim (two block barriers, one atomic)", out[1]), ("depth-bin sort (block)", out[2]),
                ("queue records + dt stores (drained, barrier)", out[3] - out[2]), ("whole trip", out[4]),
                ("  of the first ray's rounds: DDA copies", out[8]), ("  opacity bound fetched", out[9]), ("  rate + count", out[10]),
                ("  scan + ballot", out[11]), ("  emit + advance", out[12]), ("  hand-over (7 shuffles)", out[13])]
        for name, v in rows:
            print(f"   {name:48s} {v / trips:10.0f}")
        print(f"   rounds of the wave's first ray per trip          {out[14] / trips:10.2f}")
    else:
        print(f"share 1/{parts}: {trips / n:.0f} wave-trips of walk_kernel<false> per frame; cycles per trip (s_memtime):")
        for i in range(5):
            print(f"   {names[i]:44s} {out[i] / trips:10.0f}")
    del ren
