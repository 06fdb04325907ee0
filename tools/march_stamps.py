"""GPU box, diagnostic build (tools/ab_build.sh stamps -DVNR_MARCH_STAMPS): where a trip of march_kernel<false> spends its cycles,
for the whole bench frame and for a 1/8 share.  usage: VNR_AMD_LIB_PATH=.../libvnr_amd_stamps.so python tools/march_stamps.py"""
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
names = ["load ray state", "+ compose", "DDA walk + emit to LDS", "compaction + slot claim", "depth-bin sort", "stores", "whole trip", "trips"]
for parts in (1, 8):
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetFramebufferSize(ren, (fb, fb)); api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
    camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"]); api.vnrRendererSetCamera(ren, camera)
    tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
    if parts > 1:
        api.vnrRendererSetPixelInterleave(ren, 8 * fb, parts, 0)
    for _ in range(6):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    out = (C.c_ulonglong * 16)()
    L.vnrAmdDebugMarchStamps(out, 1)
    n = 20
    for _ in range(n):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    check(L.vnrAmdSynchronize())
    L.vnrAmdDebugMarchStamps(out, 1)
    trips = max(out[7], 1)
    print(f"share 1/{parts}: {trips / n:.0f} wave-trips per frame; cycles per trip (s_memtime, 100 MHz ticks x ... as reported):")
    for i in range(7):
        print(f"   {names[i]:28s} {out[i] / trips:10.0f}")
    for i, nm in ((8, "compose: result loads"), (9, "compose: classify"), (10, "compose: blend")):
        print(f"   {nm:28s} {out[i] / trips:10.0f}")
    print(f"   macrocell steps per trip: longest ray of the wave {out[11] / trips:.1f}, mean over its alive rays {out[12] / max(out[13], 1):.2f}; alive rays per trip {out[13] / trips:.1f}")
    del ren
