"""GPU box: frames/s of the GROUND-TRUTH volume (what the reference's apps show beside the neural one, apps/int_dual_volume.cpp:631-650) on the bench's
1024^3 volume and 1024^2 frame: rendering mode 5 (sample streaming, trilinear sampling kernel) and mode 4 (monolithic marcher).  usage: gt_frame_probe.py [frames]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dims = (1024,) * 3
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
cam = syn.oblique_camera(dims, distance_scale=1.1)
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)
for mode in (5, 4):
    ren = api.vnrCreateRenderer(sv)
    api.vnrRendererSetFramebufferSize(ren, (1024, 1024))
    api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
    camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    api.vnrRendererSetCamera(ren, camera)
    tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
    api.vnrRendererSetMode(ren, mode)
    for _ in range(5):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    check(L.vnrAmdSynchronize())
    t0 = time.perf_counter()
    for _ in range(frames):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    check(L.vnrAmdSynchronize())
    dt = (time.perf_counter() - t0) / frames
    st = api.vnrRendererGetFrameStats(ren)
    print(f"[gt] mode {mode}: {dt * 1e3:.3f} ms per frame ({1 / dt:.1f} frames/s), {st['n_samples'] / 1e6:.1f} M samples, {st['n_iterations']} iterations", flush=True)
    del ren
