"""GPU box, experiment: what two frames IN FLIGHT would buy.  N renderers on streams of their own render the same share of the bench frame
in turn with pipelined calls (each accumulates its own buffer: not the product's semantics, but the GPU sees what it would see if frame k + 1
ran beside frame k instead of behind it); frames per second summed over the renderers against one renderer.
usage: VNR_AMD_RENDERER_OWN_STREAM=1 python tools/two_renderers.py [shares=1,8] [renderers=1,2,3]"""
import ctypes as C
import os
import sys
import time
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


def renderer(parts):
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetFramebufferSize(ren, (fb, fb)); api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
    camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"]); api.vnrRendererSetCamera(ren, camera)
    tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
    if parts > 1:
        api.vnrRendererSetPixelInterleave(ren, 8 * fb, parts, 0)
    return ren


for parts in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,8").split(",")]:
    for n_ren in [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3").split(",")]:
        rens = [renderer(parts) for _ in range(n_ren)]
        out = C.c_void_p()
        for _ in range(6):
            for r in rens:
                check(L.vnrAmdRendererRenderPipelined(r.h, C.byref(out)))
        for r in rens:
            check(L.vnrAmdRendererFlushPipeline(r.h, C.byref(out)))
        check(L.vnrAmdSynchronize())
        n = int(os.environ.get("FRAMES", "40"))
        t0 = time.perf_counter()
        for _ in range(n):
            for r in rens:
                check(L.vnrAmdRendererRenderPipelined(r.h, C.byref(out)))
        for r in rens:
            check(L.vnrAmdRendererFlushPipeline(r.h, C.byref(out)))
        check(L.vnrAmdSynchronize())
        dt = (time.perf_counter() - t0) / (n * n_ren)
        print(f"share 1/{parts}, {n_ren} renderer(s) in turn: {dt * 1e3:.3f} ms per frame ({1.0 / dt:.1f} frames/s in all)", flush=True)
        del rens
