"""GPU box, by hand: re-runs the GRADIENT batch of given draws of tests/test_gpu_fuzz.py::test_randomly_drawn_models_equal_the_oracle exactly as
check() draws it (seed, indices) and takes the MLP gradients apart: per layer the HIP path against the numpy restatement and both against an
fp64 evaluation of the same fp16 parameters, then the same batch with wider margins around the ReLU kinks -- a distance that vanishes with the
margin is a unit whose sign follows the order of an fp32 sum (a discontinuity of the function), one that stays is arithmetic.
usage: mlp_grad_diag.py <seed> <index> [<index> ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as fz  # noqa: E402
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from oracle import oracle as o  # noqa: E402
from oracle import train_oracle as T  # noqa: E402

seed0 = int(sys.argv[1])
want_idx = [int(a) for a in sys.argv[2:]]
rng0 = np.random.default_rng(seed0)
for i in range(max(want_idx) + 1):
    d = fz.draw(rng0)
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
    vol = api.vnrCreateNeuralVolume(cfg, sv)     # (a fresh volume per setting below: forward_backward ADDS to the gradient blob until an optimizer step)
    info = api.neural_info(vol)
    ocfg = o.grid_config(L, F, d["log2T"], d["base"], d["pls"], fz.INTERP[d["interp"]], d["qt"], 1000.0 if d["max_level"] is None else d["max_level"], d["gtype"])
    in_w = info["padded_width"]
    n_mlp = o.mlp_n_params(in_w, W, H - 1)
    grows = d["act"] in ("Exponential", "Softplus") or d["out_act"] == "Exponential"
    seed = seed0 % 1000 + i
    params = syn.random_params(info["n_params"], n_mlp, seed=seed, mlp_scale=(0.35 if grows else 1.0) * (0.7 if H > 3 else 1.0))
    api.neural_set_params_fp16(vol, params)
    rng = np.random.default_rng(seed + 1)
    rng.uniform(0, 1, (1025, 3))                     # check()'s inference coordinates: the stream position matters
    B0 = 320
    tc_all = rng.uniform(0, 1, (2 * B0, 3)).astype(np.float32)
    code = o.act_code(d["act"], d["out_act"])
    signs, offs = None, None
    layer_slices = [("first", slice(0, W * in_w))] + [(f"hidden {l}", slice(W * in_w + l * W * W, W * in_w + (l + 1) * W * W)) for l in range(H - 1)] + \
                   [("last", slice(W * in_w + (H - 1) * W * W, W * in_w + (H - 1) * W * W + W))]
    for margin, deep in ((1e-5, 1e-5), (1e-5, 2.0 ** -10), (1e-3, 1e-3)):     # the sweep's margins until round 5, since round 5, and a wide one everywhere
        if margin != 1e-5 or deep != 1e-5:
            vol = api.vnrCreateNeuralVolume(cfg, sv)
            api.neural_set_params_fp16(vol, params)
        keep = fz.away_from_relu_kinks(o, ocfg, W, H, params, n_mlp, tc_all, d["act"], d["out_act"], margin=margin, margin_deep=deep)
        tc = tc_all[keep][:B0]
        B = tc.shape[0]
        if B < 64:
            print("  margins %.0e / %.1e: only %d samples left" % (margin, deep, B)); continue
        y_tc = o.network_inference(ocfg, W, H, params.view(np.uint16), tc, activation=code)
        y_tc = np.where(np.isfinite(y_tc), y_tc, 0).astype(np.float32)
        r2 = np.random.default_rng(seed + 7)         # (check() draws these from the running stream; any targets >= 0.05 off do)
        if deep == 1e-5:
            tt = (y_tc + rng.choice([-1.0, 1.0], B) * (rng.uniform(0.05, 0.6, B) + 2.0 ** -7 * np.abs(y_tc))).astype(np.float32)   # exactly check()'s
        else:
            tt = (y_tc + r2.choice([-1.0, 1.0], B) * (r2.uniform(0.05, 0.6, B) + 2.0 ** -7 * np.abs(y_tc))).astype(np.float32)
        grads = api.neural_forward_backward(vol, tc, tt).astype(np.float64)
        ref = T.training_gradients(ocfg, W, H, params.view(np.uint16), tc, tt, loss="L1", activation=d["act"], output_activation=d["out_act"])
        w_all = ref["grads"]
        y64, x_all = fz.fp64_network(o, ocfg, W, H, params, tc, d["act"], d["out_act"], targets=tt)
        g, w, x = grads[:n_mlp], w_all[:n_mlp], x_all[:n_mlp]
        print("  margins %.0e (first layer) %.1e (deeper layers, output): %d samples of %d; mlp rel hip-restatement %.4f, restatement-fp64 %.4f, hip-fp64 %.4f; |w| %.4g max %.4g"
              % (margin, deep, B, int(keep.sum()), np.linalg.norm(g - w) / np.linalg.norm(w), np.linalg.norm(w - x) / np.linalg.norm(w), np.linalg.norm(g - x) / np.linalg.norm(w),
                 np.linalg.norm(w), np.abs(w).max()))
        for name, sl in layer_slices:
            gl, wl_, xl = g[sl], w[sl], x[sl]
            nw = max(np.linalg.norm(wl_), 1e-30)
            print("      %-9s |w| %.4g  hip-restatement %.4f  restatement-fp64 %.4f  hip-fp64 %.4f   worst entry: hip %.5g restatement %.5g fp64 %.5g"
                  % (name, nw, np.linalg.norm(gl - wl_) / nw, np.linalg.norm(wl_ - xl) / nw, np.linalg.norm(gl - xl) / nw,
                     gl[np.argmax(np.abs(gl - wl_))], wl_[np.argmax(np.abs(gl - wl_))], xl[np.argmax(np.abs(gl - wl_))]))
        if deep == 1e-5:
            # the forward activations and dL/dfeatures the library kept (vnrAmdNeuralVolumeTrainingBuffer), sample by sample against the restatement's
            import ctypes as C
            Lb = api.lib()

            def buf(which):
                pp, nn = C.c_void_p(), C.c_size_t()
                api.check(Lb.vnrAmdNeuralVolumeTrainingBuffer(vol.h, which, C.byref(pp), C.byref(nn)))
                api.check(Lb.vnrAmdSynchronize())
                out = np.empty(nn.value // 2, np.float16)
                api.check(Lb.vnrAmdMemcpyD2H(out.ctypes.data_as(C.c_void_p), pp, nn.value))
                return out
            nh = H - 1
            feat = o.grid_encode(ocfg, params[n_mlp:].view(np.uint16), tc)
            _, acts_o = o.mlp_forward(params[:n_mlp].view(np.uint16), in_w, W, nh, feat, activation=code, want_activations=True)
            acts_o = acts_o.view(np.float16).reshape(nh + 1, B, W)
            acts_h = buf(3)[:(nh + 1) * B * W].reshape(nh + 1, B, W)
            for l in range(nh + 1):
                a, b = acts_h[l].astype(np.float64), acts_o[l].astype(np.float64)
                diff = a != b
                flips = (a == 0) != (b == 0)
                print("      activations of layer %d: %d of %d differ (max |d| %.3e of max %.3e); zero on one side only: %d %s"
                      % (l, int(diff.sum()), diff.size, np.abs(a - b).max(), np.abs(b).max(), int(flips.sum()),
                         [(int(s_), int(u_), float(a[s_, u_]), float(b[s_, u_])) for s_, u_ in zip(*np.nonzero(flips))][:8]))
            dfe_h = buf(1)[:B * in_w].reshape(B, in_w).astype(np.float64)
            dfe_o = ref["dfeat"].astype(np.float64)
            per = np.abs(dfe_h - dfe_o).max(1)
            worst = np.argsort(per)[::-1][:6]
            print("      dL/dfeatures per sample: max |hip - restatement| %.3e of max %.3e; worst samples %s" % (per.max(), np.abs(dfe_o).max(), [(int(w_), float(per[w_])) for w_ in worst]))
            # the output of those samples and their targets (an output ReLU whose mask differs takes the whole sample out of the gradient)
            y_h = api.neural_inference(vol, tc)
            for w_ in worst[:3]:
                print("        sample %d: hip y %.6g  restatement y %.6g  fp64 y %.6g  target %.6g" % (w_, y_h[w_], ref["y"][w_], y64[w_], tt[w_]))
            # how small are the backward activations?  (loss scale 128 / B: a chain of four layers of 0.7-scaled weights can reach the fp16 subnormals)
            dfe = ref["dfeat"]
            print("      dL/dfeatures of the restatement: max %.3e, median |.| %.3e, share of non-zero entries below 6.1e-5 (fp16 subnormal): %.3f"
                  % (np.abs(dfe).max(), np.median(np.abs(dfe[dfe != 0])) if (dfe != 0).any() else 0.0, float((np.abs(dfe[dfe != 0]) < 6.1e-5).mean()) if (dfe != 0).any() else 0.0))
