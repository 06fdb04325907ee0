"""GPU box: actual error figures of the gradient-shading modes against the oracle (the tests only assert bounds)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402

oracle.build()
psnr = lambda a, b: 10 * np.log10(1.0 / max(float(np.mean((a.astype(np.float64) - b) ** 2)), 1e-30))
vol = syn.analytic_volume(48)
colors, alphas = syn.tfn_ramp_with_bumps()
sv = api.vnrCreateSimpleVolume(vol)
tfn = api.vnrCreateTransferFunction()
api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas); api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
otfn = oracle.TfnHolder(colors, alphas)
cam = syn.oblique_camera((48, 48, 48))
camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])


def render(volume, mode, size):
    r = api.vnrCreateRenderer(volume)
    api.vnrRendererSetTransferFunction(r, tfn); api.vnrRendererSetCamera(r, camera); api.vnrRendererSetFramebufferSize(r, size); api.vnrRendererSetMode(r, mode)
    api.vnrRender(r)
    return api.vnrRendererMapFrame(r).copy()


api.lib().vnrAmdVolumeUpdateMaxOpacity(sv.h, tfn.h)   # the max-opacity table belongs to a TFN: read it only after applying ours
mo = api.volume_macrocell(sv)["max_opacity"]
assert mo.max() > 0
f = lambda c: oracle.sample_volume(vol, c, nodal=True)
for mode in (5, 8, 4, 7):
    sc = oracle.SceneHolder(96, 80, (48, 48, 48), otfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"], shading_mode=1 if mode in (7, 8) else 0)
    want = oracle.render_streaming(sc, f)[0] if mode in (5, 8) else oracle.render_monolithic(sc, vol)[0]
    img = render(sv, mode, (96, 80))
    print(f"dense volume, mode {mode}: max |err| {np.abs(img - want).max():.2e}  PSNR {psnr(img, want):.1f} dB")
L, F, T, base, pls, H = 16, 2, 19, 16, 1.3195, 3
cfg = syn.model_config(n_levels=L, n_features=F, log2_hashmap_size=T, base_resolution=base, n_hidden_layers=H, per_level_scale=pls)
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
info = api.neural_info(nv)
ocfg = oracle.grid_config(L, F, T, base, pls)
params = syn.random_params(info["n_params"], oracle.mlp_n_params(info["padded_width"], 64, H - 1), seed=21)
api.neural_set_params_fp16(nv, params)
net = lambda acc: (lambda c: oracle.network_inference(ocfg, 64, H, params.view(np.uint16), c, acc_mode=acc))
for mode in (5, 8):
    sc = oracle.SceneHolder(64, 56, (48, 48, 48), otfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"], shading_mode=1 if mode == 8 else 0)
    w32 = oracle.render_streaming(sc, net(0))[0]
    w16 = oracle.render_streaming(sc, net(1))[0]
    img = render(nv, mode, (64, 56))
    print(f"neural volume, mode {mode}: PSNR vs oracle(fp32-accumulate MLP) {psnr(img, w32):.1f} dB, vs oracle(fp16-accumulate) {psnr(img, w16):.1f} dB; "
          f"the two oracle variants against each other {psnr(w16, w32):.1f} dB")
