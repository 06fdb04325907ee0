"""GPU box, by hand: one draw of the model sweep (seed, index), its gradient batch computed again and again on fresh volumes: is the distance
to the restatement a property of the draw or of the run?  Prints the distribution and, for outliers, the levels that differ.
usage: grad_repeat.py <seed> <index> <repetitions>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as fz
from instantvnr_amd import api, synthetic as syn
from oracle import oracle as o
from oracle import train_oracle as T
o.build()
seed0, index, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed0)
for i in range(index + 1):
    d = fz.draw(rng)
print("draw", d, flush=True)
seed = seed0 % 1000 + index
L, F, W, H = d["L"], d["F"], d["W"], d["H"]
cfg = syn.model_config(n_levels=L, n_features=F, log2_hashmap_size=d["log2T"], base_resolution=d["base"], n_hidden_layers=H, per_level_scale=d["pls"])
cfg["encoding"]["interpolation"] = d["interp"]; cfg["network"]["n_neurons"] = W
cfg["network"]["activation"] = d["act"]; cfg["network"]["output_activation"] = d["out_act"]
if d["gtype"] != "Hash": cfg["encoding"]["type"] = d["gtype"]
if d["qt"]: cfg["encoding"]["quantize_threshold"] = d["qt"]
if d["max_level"] is not None: cfg["encoding"]["max_level"] = d["max_level"]
ocfg = o.grid_config(L, F, d["log2T"], d["base"], d["pls"], fz.INTERP[d["interp"]], d["qt"], 1000.0 if d["max_level"] is None else d["max_level"], d["gtype"])
sv = api.vnrCreateSimpleVolume(syn.analytic_volume(16))
vol = api.vnrCreateNeuralVolume(cfg, sv)
info = api.neural_info(vol)
n_mlp = o.mlp_n_params(info["padded_width"], W, H - 1)
grows = d["act"] in ("Exponential", "Softplus") or d["out_act"] == "Exponential"
params = syn.random_params(info["n_params"], n_mlp, seed=seed, mlp_scale=(0.35 if grows else 1.0) * (0.7 if H > 3 else 1.0))
r2 = np.random.default_rng(seed + 1)
coords = r2.uniform(0, 1, (1025, 3)).astype(np.float32)
code = o.act_code(d["act"], d["out_act"])
B = 320
tc = r2.uniform(0, 1, (2 * B, 3)).astype(np.float32)
tc = tc[fz.away_from_relu_kinks(o, ocfg, W, H, params, n_mlp, tc, d["act"], d["out_act"])][:B]
B = tc.shape[0]
y_tc = o.network_inference(ocfg, W, H, params.view(np.uint16), tc, activation=code)
y_tc = np.where(np.isfinite(y_tc), y_tc, 0).astype(np.float32)
tt = (y_tc + r2.choice([-1.0, 1.0], B) * r2.uniform(0.05, 0.6, B)).astype(np.float32)
ref = T.training_gradients(ocfg, W, H, params.view(np.uint16), tc, tt, loss="L1", activation=d["act"], output_activation=d["out_act"])["grads"]
w = ref[n_mlp:]
lay = o.grid_layout(ocfg)
rels = []
first = None
for k in range(reps):
    v = api.vnrCreateNeuralVolume(cfg, sv)
    api.neural_set_params_fp16(v, params)
    if k % 3 == 0:
        api.neural_encode(v, coords); api.neural_inference(v, coords)          # as the sweep does before the gradient batch
    g = api.neural_forward_backward(v, tc, tt).astype(np.float64)
    if first is None: first = g
    rel = np.linalg.norm(g[n_mlp:] - w) / np.linalg.norm(w)
    rels.append(rel)
    if rel > 0.03 or not np.array_equal(g[:n_mlp], first[:n_mlp]):
        print("rep", k, "grid rel", rel, "mlp equal to the first repetition:", np.array_equal(g[:n_mlp], first[:n_mlp]), flush=True)
        for l in range(L):
            a0, a1 = int(lay["offsets"][l]) * F, int(lay["offsets"][l + 1]) * F
            wl, gl = w[a0:a1], g[n_mlp + a0:n_mlp + a1]
            rl = np.linalg.norm(gl - wl) / max(np.linalg.norm(wl), 1e-30)
            if rl > 0.03: print("    level", l, "rel", rl, "|w|", np.linalg.norm(wl), "|g|", np.linalg.norm(gl), "ratio", np.linalg.norm(gl) / max(np.linalg.norm(wl), 1e-30))
    del v
rels = np.array(rels)
print("repetitions", reps, "grid rel: min %.5f median %.5f p99 %.5f max %.5f; above 3 %%: %d" % (rels.min(), np.median(rels), np.quantile(rels, 0.99), rels.max(), (rels > 0.03).sum()))
