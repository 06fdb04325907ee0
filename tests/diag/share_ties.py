"""diagnostic (GPU box): where do the few pixels come from in which a rank's share of the C4 frame and the oracle's marcher ON THE LIBRARY'S NETWORK VALUES
differ by more than float rounding?  For several differently seeded models: every pixel with an error above 1e-5, with both opacities.  (round 6: the
suspicion to confirm is the saturation tie at 0.9999, tests/test_gpu_fullsize.py assert_renderer_alone.)  usage: share_ties.py [models]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.pop("VNR_RM_N_ITERS", None)
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from oracle import oracle  # noqa: E402
oracle.build()
n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dims = (1024, 1024, 1024)
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
pls = float(np.exp(np.log(1024 / 16.0) / 15))
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)
tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas); api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
cam = syn.oblique_camera(dims, distance_scale=1.1)
camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
block, parts = 8 * 1024, 8
for m in range(n_models):
    os.environ["VNR_AMD_INIT_SEED"] = str(9000 + m)
    nv = api.vnrCreateNeuralVolume(syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls), sv, online_macrocell_construction=False)
    api.vnrNeuralVolumeTrain(nv, 300 + 50 * m, True)
    part = m % parts
    r = api.vnrCreateRenderer(nv)
    api.vnrRendererSetTransferFunction(r, tfn); api.vnrRendererSetCamera(r, camera); api.vnrRendererSetFramebufferSize(r, (1024, 1024)); api.vnrRendererSetMode(r, 5)
    api.vnrRendererSetPixelInterleave(r, block, parts, part)
    api.vnrRender(r)
    img = api.vnrRendererMapFrame(r).reshape(-1, 4).copy()
    mo = api.volume_macrocell(nv)["max_opacity"]
    sc = oracle.SceneHolder(1024, 1024, dims, oracle.TfnHolder(colors, alphas), mo, cam["from"], cam["at"], cam["up"], cam["fovy"], interleave=(block, parts, part))
    ref, _, ost = oracle.render_streaming(sc, lambda c: api.neural_inference(nv, c), n_iters=32)
    ref = ref.reshape(-1, 4)
    err = np.abs(img - ref).max(axis=1)
    bad = np.nonzero(err > 1e-5)[0]
    print(f"[ties] model {m} (seed {9000 + m}, share {part}): max err {err.max():.2e}, pixels above 1e-5: {len(bad)}, above 1e-6: {int((err > 1e-6).sum())}", flush=True)
    for p in bad[:10]:
        print(f"        pixel {p}: err {err[p]:.2e}; opacity library {img[p, 3]:.8f}, oracle {ref[p, 3]:.8f}; rgb diff {np.abs(img[p, :3] - ref[p, :3]).max():.2e}", flush=True)
    del r, nv
