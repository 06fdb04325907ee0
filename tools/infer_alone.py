"""GPU box: the evaluation kernel on the real sample queues of the bench frame (C4), quickly: kernel-alone rate on one stream and
the frame time on two.  VNR_AMD_LIB_PATH selects the library build (tools/ab_build.sh).  usage: infer_alone.py [frames]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
size, fb = int(os.environ.get("SIZE", 1024)), int(os.environ.get("FB", 1024))
dims = (size,) * 3
os.environ.setdefault("VNR_AMD_INIT_SEED", "20240611")
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
pls = float(np.exp(np.log(size / 16.0) / 15))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 300, True)
cam = syn.oblique_camera(dims, distance_scale=1.1)
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)
tag = os.path.basename(os.environ.get("VNR_AMD_LIB_PATH", "default"))
for halves, brick in ((1, -1), (2, -1), (1, 0)):
    check(L.vnrAmdNeuralVolumeSetBrickImageMode(nv.h, brick))
    os.environ["VNR_AMD_RENDER_HALVES"] = str(halves)
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetFramebufferSize(ren, (fb, fb))
    api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
    api.vnrRendererSetProfiling(ren, True)
    camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    api.vnrRendererSetCamera(ren, camera)
    tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
    for _ in range(5):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    check(L.vnrAmdSynchronize())
    t0 = time.perf_counter(); samples = 0; ms = 0.0; union = 0.0
    for _ in range(frames):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
        st = api.vnrRendererGetFrameStats(ren)
        samples += st["n_samples"]; ms += st["infer_kernel_ms"]; union += st["infer_union_ms"]
    check(L.vnrAmdSynchronize())
    dt = (time.perf_counter() - t0) / frames
    print(f"[{tag}] streams {halves} brick {'on' if brick else 'off'}: frame {dt * 1e3:.3f} ms ({1 / dt:.1f} fps), kernel {samples / (ms * 1e-3) / 1e9:.2f} G samples/s per launch, "
          f"union {samples / (union * 1e-3) / 1e9:.2f} G samples/s, {samples / frames / 1e6:.1f} M samples per frame", flush=True)
    del ren
