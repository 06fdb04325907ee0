"""GPU box: a soak of the pipelined renderer: thousands of vnrAmdRendererRenderPipelined calls with the events an application
produces in between (camera moves, transfer-function edits, mode switches, resizes, flushes, training steps behind a flush), and after
every event the pipelined frame must equal the frame a fresh sequential renderer produces for the same state.
usage: python tools/pipeline_soak.py [frames]"""
import ctypes as C
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(1)
size = 256
sv = api.vnrCreateSimpleVolumePerlin((size,) * 3, seed=42, octaves=4, base_frequency=6.0)
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=19, n_hidden_layers=3, per_level_scale=float(np.exp(np.log(size / 16.0) / 15)))
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 300, True)
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.1)
tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
state = {"fb": (512, 384), "mode": 5, "dist": 1.3}


def camera():
    cam = syn.oblique_camera((size,) * 3, distance_scale=state["dist"])
    c = api.vnrCreateCamera(); api.vnrCameraSet(c, cam["from"], cam["at"], cam["up"], cam["fovy"])
    return c


def fresh_sequential(n_accum):
    r = api.vnrCreateRenderer(nv)
    api.vnrRendererSetTransferFunction(r, tfn); api.vnrRendererSetCamera(r, camera())
    api.vnrRendererSetFramebufferSize(r, state["fb"]); api.vnrRendererSetMode(r, state["mode"])
    for _ in range(n_accum):
        api.vnrRender(r)
    return api.vnrRendererMapFrame(r).copy()


ren = api.vnrCreateRenderer(nv)
api.vnrRendererSetTransferFunction(ren, tfn); api.vnrRendererSetCamera(ren, camera())
api.vnrRendererSetFramebufferSize(ren, state["fb"]); api.vnrRendererSetMode(ren, state["mode"])
api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
out = C.c_void_p()
accum = 0
checks = 0
t0 = time.perf_counter()
for k in range(frames):
    check(L.vnrAmdRendererRenderPipelined(ren.h, C.byref(out)))
    accum += 1
    if rng.random() < 0.02 or k == frames - 1:   # an application event
        check(L.vnrAmdRendererFlushPipeline(ren.h, C.byref(out)))
        w, h = state["fb"]
        got = np.empty((h, w, 4), np.float32)
        check(L.vnrAmdMemcpyD2H(got.ctypes.data_as(C.c_void_p), out, got.nbytes))
        want = fresh_sequential(accum)
        if not np.array_equal(got, want):
            raise SystemExit(f"frame {k}: pipelined frame differs from the sequential one after {accum} accumulated frames "
                             f"(max {np.abs(got - want).max():.3g}, state {state})")
        checks += 1
        ev = rng.integers(0, 6)
        if ev == 0:
            state["dist"] = float(rng.uniform(1.0, 2.0)); api.vnrRendererSetCamera(ren, camera())
        elif ev == 1:
            state["mode"] = int(rng.choice([5, 8, 6])); api.vnrRendererSetMode(ren, state["mode"]); api.vnrRendererResetAccumulation(ren)
        elif ev == 2:
            state["fb"] = (int(rng.choice([256, 512, 640])), int(rng.choice([200, 384, 480]))); api.vnrRendererSetFramebufferSize(ren, state["fb"])
        elif ev == 3:
            api.vnrNeuralVolumeTrain(nv, 3, True); api.vnrRendererResetAccumulation(ren)   # parameters change: the brick image goes, frames restart
        elif ev == 4:
            api.vnrRendererSetTransferFunction(ren, tfn)
        else:
            api.vnrRendererResetAccumulation(ren)
        accum = 0
        if checks % 10 == 0:
            print(f"[soak] {k + 1} frames, {checks} states checked, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"[soak] ok: {frames} pipelined frames, {checks} states equal to a fresh sequential renderer, {time.perf_counter() - t0:.0f} s")
