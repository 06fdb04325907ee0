"""GPU box: how long one rank's share of the bench frame takes (interleaved 8-scanline blocks, part 0 of N) without the
collective: the compute side of the N-GPU strong-scaling curve on one GPU.  usage: python tools/share_probe.py"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
size, fb = 1024, 1024
dims = (size,) * 3
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
pls = float(np.exp(np.log(size / 16.0) / 15))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
if os.environ.get("SHARE_NEURONS"):   # FullyFusedMLP width of the probe's model (default 64)
    cfg["network"]["n_neurons"] = int(os.environ["SHARE_NEURONS"])
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 300, True)
# SHARE_EXTRA_STREAMS=n: n more HIP streams alive in the process before the renderers are created (round 5: a process that owns more streams than
# the runtime has hardware queues -- 4 by default -- runs a small share's ray parts through shared queues)
_extra = []
if os.environ.get("SHARE_EXTRA_STREAMS"):
    import ctypes as _C
    _hip = _C.CDLL("libamdhip64.so")
    for _ in range(int(os.environ["SHARE_EXTRA_STREAMS"])):
        _s = _C.c_void_p()
        assert _hip.hipStreamCreateWithFlags(_C.byref(_s), 1) == 0
        _extra.append(_s)
cam = syn.oblique_camera(dims, distance_scale=1.1)
base = 0.0
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)
import os
# SHARE_CONFIGS="name:VAR=v,VAR=v;name2:..." renders every share once per configuration (the variables are read when a renderer is
# created), in one process on one trained model: an A/B of renderer settings without the set-up in between
configs = [("", {})]
if os.environ.get("SHARE_CONFIGS"):
    configs = []
    for item in os.environ["SHARE_CONFIGS"].split(";"):
        name, _, kv = item.partition(":")
        configs.append((name, dict(x.split("=") for x in kv.split(",") if x)))
reps = int(os.environ.get("SHARE_REPS", "1"))
for rep in range(reps):
  for cname, cenv in configs:
    saved = {k: os.environ.get(k) for k in cenv}
    os.environ.update(cenv)
    for parts in [int(v) for v in os.environ.get("SHARE_PARTS", "1,2,4,8").split(",")]:
        ren = api.vnrCreateRenderer(nv)
        api.vnrRendererSetFramebufferSize(ren, (fb, fb))
        api.vnrRendererSetMode(ren, int(os.environ.get("SHARE_MODE", 5)))   # 6 / 9 / 12: the in-shader kernel (VNR_AMD_IN_SHADER=0: streaming)
        api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
        if os.environ.get("SHARE_PROFILING"):   # the HIP events bench.py records around every evaluation launch
            api.vnrRendererSetProfiling(ren, True)
        camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
        api.vnrRendererSetCamera(ren, camera)
        tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
        api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
        if parts > 1:
            api.vnrRendererSetPixelInterleave(ren, 8 * fb, parts, 0)
        for _ in range(6):
            api.vnrRender(ren); api.vnrRendererMapFrame(ren)
        check(L.vnrAmdSynchronize())
        t0 = time.perf_counter()
        n = int(os.environ.get("SHARE_FRAMES", "40"))
        import ctypes as C
        if os.environ.get("SHARE_PIPELINED"):   # the pipelined calls (head of frame k + 1 before frame k has completed on the host)
            out = C.c_void_p()
            for _ in range(n):
                check(L.vnrAmdRendererRenderPipelined(ren.h, C.byref(out)))
            check(L.vnrAmdRendererFlushPipeline(ren.h, C.byref(out)))
        else:
            for _ in range(n):
                api.vnrRender(ren); api.vnrRendererMapFrame(ren)
        check(L.vnrAmdSynchronize())
        dt = (time.perf_counter() - t0) / n
        st = api.vnrRendererGetFrameStats(ren)
        print(f"{cname + ' ' if cname else ''}share 1/{parts}: {dt * 1e3:.3f} ms per frame, {st['n_samples'] / 1e6:.2f} M samples, {st['n_iterations']} iterations "
              f"-> speed-up {base / dt if parts > 1 and base else 0:.2f}x of {parts}", flush=True)
        if parts == 1 and not base:
            base = dt
        del ren
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
