"""GPU box, by hand: forward_backward three times on one batch without an optimizer step in between -- are the gradient blobs equal?
usage: repeat_fb.py <seed> <index>   (a draw of tests/test_gpu_fuzz.py)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as fz  # noqa: E402
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from oracle import oracle as o  # noqa: E402

seed0, idx = int(sys.argv[1]), int(sys.argv[2])
rng0 = np.random.default_rng(seed0)
for i in range(idx + 1):
    d = fz.draw(rng0)
print(d)
L, F, W, H = d["L"], d["F"], d["W"], d["H"]
cfg = syn.model_config(n_levels=L, n_features=F, log2_hashmap_size=d["log2T"], base_resolution=d["base"], n_hidden_layers=H, per_level_scale=d["pls"])
cfg["encoding"]["interpolation"] = d["interp"]; cfg["network"]["n_neurons"] = W
cfg["network"]["activation"] = d["act"]; cfg["network"]["output_activation"] = d["out_act"]
if d["gtype"] != "Hash": cfg["encoding"]["type"] = d["gtype"]
if d["qt"]: cfg["encoding"]["quantize_threshold"] = d["qt"]
if d["max_level"] is not None: cfg["encoding"]["max_level"] = d["max_level"]
sv = api.vnrCreateSimpleVolume(syn.analytic_volume(16))
vol = api.vnrCreateNeuralVolume(cfg, sv)
info = api.neural_info(vol)
n_mlp = o.mlp_n_params(info["padded_width"], W, H - 1)
params = syn.random_params(info["n_params"], n_mlp, seed=seed0 % 1000 + idx, mlp_scale=0.7 if H > 3 else 1.0)
api.neural_set_params_fp16(vol, params)
rng = np.random.default_rng(5)
for B in (320, 311, 4096):
    tc = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    tt = rng.uniform(0, 1, B).astype(np.float32)
    gs = [api.neural_forward_backward(vol, tc, tt).astype(np.float64) for _ in range(3)]
    for k in (1, 2):
        for name, sl in (("mlp", slice(0, n_mlp)), ("grid", slice(n_mlp, None))):
            a, b = gs[0][sl], gs[k][sl]
            nz = np.abs(a) > 1e-3 * np.abs(a).max()
            print("B %d call %d vs call 0, %s: equal %s, rel %.3e, median ratio on the larger entries %.4f" %
                  (B, k, name, np.array_equal(a, b), np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-30), float(np.median(b[nz] / a[nz])) if nz.any() else 0.0))
