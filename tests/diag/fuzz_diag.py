"""GPU box, by hand: re-runs given draws of tests/test_gpu_fuzz.py (seed, indices) and prints where the HIP path and the oracle differ, per
sample and per gradient block, beside an fp64 evaluation of the same fp16 parameters (no rounding of activations): tells rounding noise of a
deep network (both sides equally far from the fp64 values) from a defect (one side far).  usage: fuzz_diag.py <seed> <index> [<index> ...]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as fz
from instantvnr_amd import api, synthetic as syn
from oracle import oracle as o
from oracle import train_oracle as T


def act64(x, a):
    a = o.ACTIVATIONS.get(a, a)
    if a == 0: return x
    if a == 1: return np.maximum(x, 0)
    if a == 2: return np.exp(x)
    if a == 3: return 1 / (1 + np.exp(-x))
    if a == 4: t = 10 * x; return 0.5 * (t + np.sqrt(t * t + 4)) / 10
    if a == 5: return np.log1p(np.exp(10 * x)) / 10
    raise ValueError(a)


def forward64(ocfg, W, H, params, coords, act, out_act):
    in_w = o.padded_width(ocfg)
    w1, wh, wl, n_mlp = T.split_mlp(params, in_w, W, H - 1)
    feat = o.grid_encode(ocfg, params[n_mlp:].view(np.uint16), coords).view(np.float16).astype(np.float64)
    h = act64(feat @ w1.astype(np.float64).T, act)
    for m in wh:
        h = act64(h @ m.astype(np.float64).T, act)
    return act64(h @ wl[0].astype(np.float64), out_act)


seed0 = int(sys.argv[1]); want_idx = [int(a) for a in sys.argv[2:]]
rng = np.random.default_rng(seed0)
for i in range(max(want_idx) + 1):
    d = fz.draw(rng)
    if i not in want_idx:
        continue
    print("draw", i, d, flush=True)
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
    ocfg = o.grid_config(L, F, d["log2T"], d["base"], d["pls"], fz.INTERP[d["interp"]], d["qt"], 1000.0 if d["max_level"] is None else d["max_level"], d["gtype"])
    n_mlp = o.mlp_n_params(info["padded_width"], W, H - 1)
    grows = d["act"] in ("Exponential", "Softplus") or d["out_act"] == "Exponential"
    seed = seed0 % 1000 + i
    params = syn.random_params(info["n_params"], n_mlp, seed=seed, mlp_scale=(0.35 if grows else 1.0) * (0.7 if H > 3 else 1.0))
    api.neural_set_params_fp16(vol, params)
    r2 = np.random.default_rng(seed + 1)
    coords = r2.uniform(0, 1, (1025, 3)).astype(np.float32)
    coords[0] = (0, 0, 0); coords[1] = (1, 1, 1); coords[2] = (0.5, 0.5, 0.5); coords[3] = (1, 0, 0.999999)
    got = api.neural_inference(vol, coords)
    want = o.network_inference(ocfg, W, H, params.view(np.uint16), coords, activation=o.act_code(d["act"], d["out_act"]))
    exact = forward64(ocfg, W, H, params, coords, d["act"], d["out_act"])
    tol = fz.TOL_ABS * max(1.0, np.abs(want).max())
    e = np.abs(got - want)
    print("  weights_in_lds", info.get("weights_in_lds"), "tol %.5f  |hip-oracle| max %.5f  p99 %.5f  median %.6f  n>tol %d of %d" % (tol, e.max(), np.quantile(e, 0.99), np.median(e), (e > tol).sum(), e.size))
    print("  vs fp64: |hip-exact| max %.5f rms %.6f   |oracle-exact| max %.5f rms %.6f" % (np.abs(got - exact).max(), np.sqrt(((got - exact) ** 2).mean()), np.abs(want - exact).max(), np.sqrt(((want - exact) ** 2).mean())))
    tc = r2.uniform(0, 1, (320, 3)).astype(np.float32); y_tc = o.network_inference(ocfg, W, H, params.view(np.uint16), tc, activation=o.act_code(d['act'], d['out_act']))
    y_tc = np.where(np.isfinite(y_tc), y_tc, 0).astype(np.float32)
    tt = (y_tc + r2.choice([-1.0, 1.0], 320) * r2.uniform(0.05, 0.6, 320)).astype(np.float32)
    grads = api.neural_forward_backward(vol, tc, tt).astype(np.float64)
    ref = T.training_gradients(ocfg, W, H, params.view(np.uint16), tc, tt, loss="L1", activation=d["act"], output_activation=d["out_act"])["grads"]
    for name, sl in [("mlp", slice(0, n_mlp)), ("grid", slice(n_mlp, None))]:
        g, w = grads[sl], ref[sl]
        if np.abs(w).max() == 0: print("  ", name, "all zero"); continue
        print("   %s: rel %.4f  max abs %.5f of %.5f" % (name, np.linalg.norm(g - w) / np.linalg.norm(w), np.abs(g - w).max(), np.abs(w).max()))
    # the same draw at 64 neurons: is the distance a property of the width?
    # fp64 backward (no fp16 rounding anywhere after the encode): distance of both sides from it
    def dact64(x, y, a):
        a = o.ACTIVATIONS.get(a, a)
        if a == 0: return np.ones_like(x)
        if a == 1: return (x > 0).astype(np.float64)
        if a == 2: return y
        if a == 3: return y * (1 - y)
        if a == 4: t = 10 * x; return 0.5 * (1 + t / np.sqrt(t * t + 4))
        if a == 5: return 1 / (1 + np.exp(-10 * x))
    in_w = o.padded_width(ocfg)
    w1, wh, wl, _ = T.split_mlp(params, in_w, W, H - 1)
    feat = o.grid_encode(ocfg, params[n_mlp:].view(np.uint16), tc).view(np.float16).astype(np.float64)
    pre = [feat @ w1.astype(np.float64).T]; post = [act64(pre[0], d["act"])]
    for m in wh:
        pre.append(post[-1] @ m.astype(np.float64).T); post.append(act64(pre[-1], d["act"]))
    zo = post[-1] @ wl[0].astype(np.float64); yo = act64(zo, d["out_act"])
    dy = 128.0 * np.sign(yo - tt) / 320 * dact64(zo, yo, d["out_act"])
    dd = dy[:, None] * wl[0].astype(np.float64)[None, :] * dact64(pre[-1], post[-1], d["act"])
    for l in range(len(wh) - 1, -1, -1):
        dd = (dd @ wh[l].astype(np.float64)) * dact64(pre[l], post[l], d["act"])
    dfeat = dd @ w1.astype(np.float64)
    ex = np.zeros(params.size)
    lay = o.grid_layout(ocfg)
    for l, (idxs, ws) in enumerate(T.corner_indices_and_weights(ocfg, lay, tc)):
        base = n_mlp + int(lay["offsets"][l]) * F
        for f in range(F):
            np.add.at(ex, base + idxs.ravel() * F + f, (ws.astype(np.float64) * dfeat[:, l * F + f][:, None]).ravel())
    gx = ex[n_mlp:]
    if np.abs(gx).max() > 0:
        print("   grid vs fp64: hip rel %.4f   restatement rel %.4f" % (np.linalg.norm(grads[n_mlp:] - gx) / np.linalg.norm(gx), np.linalg.norm(ref[n_mlp:] - gx) / np.linalg.norm(gx)))
    k = int(np.argmax(np.abs(grads[n_mlp:] - ref[n_mlp:])))
    print("   worst grid entry %d: hip %.6f  restatement %.6f  fp64 %.6f   entries past 6%% of max: %d" % (k, grads[n_mlp + k], ref[n_mlp + k], gx[k], (np.abs(grads[n_mlp:] - ref[n_mlp:]) > 0.06 * np.abs(ref[n_mlp:]).max()).sum()))
    # which samples touch it, and how close to zero their hidden pre-activations come (a ReLU mask that can flip)
    for l, (idxs, ws) in enumerate(T.corner_indices_and_weights(ocfg, lay, tc)):
        base = int(lay["offsets"][l]) * F
        if base <= k < base + int(lay["offsets"][l + 1] - lay["offsets"][l]) * F:
            e = (k - base) // F
            smp = np.unique(np.nonzero((idxs == e) & (ws != 0))[0])
            print("   level", l, "samples", smp[:12], "min |pre-activation| per sample", [float(min(np.abs(p_[s_]).min() for p_ in pre)) for s_ in smp[:12]])
            # the fp16 path's own pre-activations (fp16 inputs of each layer as the oracle stored them, exact products summed in fp64):
            # a unit whose sum is within fp32 rounding of zero (ratio to the sum of |terms| ~ 1e-7) gets its sign from the ORDER of the sum
            _, acts16 = o.mlp_forward(params[:n_mlp].view(np.uint16), in_w, W, H - 1, o.grid_encode(ocfg, params[n_mlp:].view(np.uint16), tc),
                                      activation=o.act_code(d["act"], d["out_act"]), want_activations=True)
            acts16 = acts16.view(np.float16).astype(np.float64)
            ins = [feat] + [acts16[j] for j in range(H - 1)]
            mats = [w1.astype(np.float64)] + [m.astype(np.float64) for m in wh]
            for s_ in smp[:12]:
                worst = min((abs(float(ins[j][s_] @ mats[j][u])) / max(1e-30, float(np.abs(ins[j][s_] * mats[j][u]).sum())), j, u) for j in range(len(mats)) for u in range(W))
                print("     sample %d: smallest |sum| / sum|terms| over units = %.2e (layer %d unit %d)" % (s_, worst[0], worst[1], worst[2]))
