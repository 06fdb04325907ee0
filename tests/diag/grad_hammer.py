"""GPU box, by hand: hunt for the once-in-28 000 grid-gradient transient of round 4 (profiles/r04_fuzz.txt, last section).

The transient was a gradient batch of the model sweep whose GRID gradients were 13 % (relative L2) from the restatement while the MLP
gradients of the same batch were right.  13 % is what ONE workgroup of grid_backward_kernel (64 samples of one level) lost or doubled
would be for that draw (20 Tiled levels of 512 entries, 5 workgroups a level).  This tool repeats that draw's batch (and its
predecessor's, and a hash-grid model that takes the LDS scatter) and compares every repetition ON THE DEVICE with the first one
(vnrAmdNeuralVolumeGradientDistance: a reduction on the training stream, before any download), so that a miss is caught the moment it
happens and taken apart on the spot:

  * the blob downloaded twice and compared on the host with the device's verdict        -> "download raced the stream"
  * dL/dfeatures downloaded and compared bit for bit with the first repetition's          -> "stale / wrong dfeat" (MLP backward side)
  * the grid backward alone repeated on the stored dL/dfeatures (RescatterGridGradients) -> "scatter lost / double-counted updates"

Three loops:  fast   one volume per model, learning rate 0 (the optimizer step clears the gradients and leaves the parameters), models
                     alternating call by call;
              fresh  a new volume per repetition as the sweep does it (hipMalloc / hipFree churn, first-step allocation and zeroing of
                     the blob, the encode + inference calls before the batch on every third repetition);
              reconf ONE volume re-configured between models with vnrNeuralVolumeSetModel (every configure() transition: scratch keyed by
                     the Network's address, weight-gradient slab, LDS work-item cache, workspace re-allocation).
usage: grad_hammer.py <fast repetitions> <fresh repetitions> <reconf repetitions> [out file]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as fz  # noqa: E402
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from oracle import oracle as o  # noqa: E402

o.build()
n_fast, n_fresh, n_reconf = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join("gpurun_out", "grad_hammer.txt")
BUDGET = [float(x) for x in os.environ.get("VNR_HAMMER_BUDGET_S", "1e9,1e9,1e9").split(",")]   # seconds per loop: whichever ends first
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
LOG = open(out_path, "w")
L_ = api.lib()


def say(*a):
    line = " ".join(str(x) for x in a)
    print(line, flush=True)
    LOG.write(line + "\n"); LOG.flush()


def cfg_of(d, lr=None):
    cfg = syn.model_config(n_levels=d["L"], n_features=d["F"], log2_hashmap_size=d["log2T"], base_resolution=d["base"],
                           n_hidden_layers=d["H"], per_level_scale=d["pls"])
    cfg["encoding"]["interpolation"] = d["interp"]; cfg["network"]["n_neurons"] = d["W"]
    cfg["network"]["activation"] = d["act"]; cfg["network"]["output_activation"] = d["out_act"]
    if d["gtype"] != "Hash": cfg["encoding"]["type"] = d["gtype"]
    if d["qt"]: cfg["encoding"]["quantize_threshold"] = d["qt"]
    if d["max_level"] is not None: cfg["encoding"]["max_level"] = d["max_level"]
    if lr is not None:
        cfg["optimizer"] = {"otype": "Adam", "learning_rate": lr, "beta1": 0.9, "beta2": 0.999, "epsilon": 1e-15, "l2_reg": 0.0}
    return cfg


class Case:
    """a model of the sweep with its parameters and its gradient batch, as tests/test_gpu_fuzz.py::check builds them"""

    def __init__(self, name, d, seed, batch=320):
        self.name, self.d, self.seed = name, d, seed
        L, F, W, H = d["L"], d["F"], d["W"], d["H"]
        self.ocfg = o.grid_config(L, F, d["log2T"], d["base"], d["pls"], fz.INTERP[d["interp"]], d["qt"],
                                  1000.0 if d["max_level"] is None else d["max_level"], d["gtype"])
        self.in_w = o.padded_width(self.ocfg)
        self.n_params = o.n_params(self.ocfg, W, H)
        self.n_mlp = o.mlp_n_params(self.in_w, W, H - 1)
        grows = d["act"] in ("Exponential", "Softplus") or d["out_act"] == "Exponential"
        self.params = syn.random_params(self.n_params, self.n_mlp, seed=seed, mlp_scale=(0.35 if grows else 1.0) * (0.7 if H > 3 else 1.0))
        r2 = np.random.default_rng(seed + 1)
        self.coords = r2.uniform(0, 1, (1025, 3)).astype(np.float32)
        code = o.act_code(d["act"], d["out_act"])
        B = batch
        tc = r2.uniform(0, 1, (2 * B, 3)).astype(np.float32)
        if B <= 1024:
            tc = tc[fz.away_from_relu_kinks(o, self.ocfg, W, H, self.params, self.n_mlp, tc, d["act"], d["out_act"])][:B]
            y = o.network_inference(self.ocfg, W, H, self.params.view(np.uint16), tc, activation=code)
            y = np.where(np.isfinite(y), y, 0).astype(np.float32)
        else:
            tc = tc[:B]; y = np.zeros(B, np.float32)
        self.tc = np.ascontiguousarray(tc)
        self.tt = (y + r2.choice([-1.0, 1.0], tc.shape[0]) * r2.uniform(0.05, 0.6, tc.shape[0])).astype(np.float32)
        self.d_tc = api.DeviceArray.from_numpy(self.tc)
        self.d_tt = api.DeviceArray.from_numpy(self.tt)
        self.B = self.tc.shape[0]
        self.lay = o.grid_layout(self.ocfg)
        self.ref_host = None    # fp16 blob of the first repetition (uint16 bits)
        self.ref_dev = None
        self.ref_dfeat = None
        self.worst = [0.0, 0.0]

    def fb(self, vol):
        api.check(L_.vnrAmdNeuralVolumeForwardBackward(vol.h, self.B, self.d_tc.ptr, self.d_tt.ptr))

    def buffer(self, vol, which):
        p, n = C.c_void_p(), C.c_size_t()
        api.check(L_.vnrAmdNeuralVolumeTrainingBuffer(vol.h, which, C.byref(p), C.byref(n)))
        api.check(L_.vnrAmdSynchronize())
        out = np.empty(n.value // 2, np.uint16)
        if n.value:
            api.check(L_.vnrAmdMemcpyD2H(out.ctypes.data_as(C.c_void_p), p, n.value))
        return out

    def adopt_reference(self, vol):
        self.ref_host = self.buffer(vol, 0)
        self.ref_dev = api.DeviceArray.from_numpy(self.ref_host)
        self.ref_dfeat = self.buffer(vol, 1)

    def distance(self, vol):
        out = (C.c_double * 4)()
        api.check(L_.vnrAmdNeuralVolumeGradientDistance(vol.h, self.ref_dev.ptr, out))
        mlp = float(np.sqrt(out[0] / max(out[1], 1e-300))); grid = float(np.sqrt(out[2] / max(out[3], 1e-300)))
        return mlp, grid

    def host_rel(self, blob):
        g = blob.view(np.float16).astype(np.float64); w = self.ref_host.view(np.float16).astype(np.float64)
        n = self.n_mlp
        return (float(np.linalg.norm(g[:n] - w[:n]) / max(np.linalg.norm(w[:n]), 1e-300)),
                float(np.linalg.norm(g[n:] - w[n:]) / max(np.linalg.norm(w[n:]), 1e-300)))

    def post_mortem(self, vol, where, k, mlp, grid):
        say("EVENT", where, "repetition", k, "model", self.name, "device verdict: mlp rel %.3e grid rel %.3e" % (mlp, grid))
        b1 = self.buffer(vol, 0); b2 = self.buffer(vol, 0)
        say("   blob downloaded twice: equal to each other:", bool(np.array_equal(b1, b2)), " host rel (mlp, grid):", self.host_rel(b1))
        g = b1.view(np.float16).astype(np.float64); w = self.ref_host.view(np.float16).astype(np.float64)
        F = self.d["F"]
        for l in range(self.d["L"]):
            a0, a1 = self.n_mlp + int(self.lay["offsets"][l]) * F, self.n_mlp + int(self.lay["offsets"][l + 1]) * F
            rl = np.linalg.norm(g[a0:a1] - w[a0:a1]) / max(np.linalg.norm(w[a0:a1]), 1e-300)
            if rl > 2e-3:
                bad = np.nonzero(np.abs(g[a0:a1] - w[a0:a1]) > 1e-3 * max(np.abs(w[a0:a1]).max(), 1e-30))[0]
                say("   level", l, "rel %.4f" % rl, "|g| / |w| %.4f" % (np.linalg.norm(g[a0:a1]) / max(np.linalg.norm(w[a0:a1]), 1e-300)),
                    "entries off:", bad.size, "of", a1 - a0, "first", bad[:8].tolist(), "last", bad[-4:].tolist())
        df = self.buffer(vol, 1)
        same = np.array_equal(df, self.ref_dfeat)
        say("   dL/dfeatures equal to the first repetition's, bit for bit:", bool(same))
        if not same:
            rows = np.nonzero((df.reshape(self.B, -1) != self.ref_dfeat.reshape(self.B, -1)).any(1))[0]
            say("   rows of dL/dfeatures that differ:", rows.size, "first", rows[:16].tolist())
        api.check(L_.vnrAmdNeuralVolumeRescatterGridGradients(vol.h, self.B, self.d_tc.ptr))
        m2, g2 = self.distance(vol)
        say("   the grid backward alone, repeated on the stored dL/dfeatures: grid rel %.3e" % g2,
            "->", "the first scatter lost or double-counted updates" if g2 < 5e-3 and same else "the scatter reproduces it (its inputs are off)")
        np.savez(os.path.join(os.path.dirname(out_path) or ".", "grad_hammer_event_%s_%d.npz" % (where, k)), blob=b1, ref=self.ref_host, dfeat=df,
                 ref_dfeat=self.ref_dfeat)

    def check(self, vol, where, k):
        mlp, grid = self.distance(vol)
        self.worst[0] = max(self.worst[0], mlp); self.worst[1] = max(self.worst[1], grid)
        if not (mlp <= 1e-3 and grid <= 5e-3):     # (atomics arrive in any order: the fp16 sums differ by rounding, ~5e-4 of the norm)
            self.post_mortem(vol, where, k, mlp, grid)
            return False
        return True


def sweep_draw(seed0, index):
    rng = np.random.default_rng(seed0)
    for _ in range(index + 1):
        d = fz.draw(rng)
    return d


say("grad_hammer: fast", n_fast, "fresh", n_fresh, "reconf", n_reconf)
d109, d108 = sweep_draw(303, 109), sweep_draw(303, 108)
say("draw 109:", d109); say("draw 108:", d108)
dlds = dict(L=8, F=2, log2T=14, base=4, pls=2.0, H=2, W=64, interp="Linear", act="ReLU", out_act="None", gtype="Hash", qt=0.0, max_level=None)
d4 = dict(L=12, F=4, log2T=12, base=8, pls=1.5, H=3, W=32, interp="Smoothstep", act="ReLU", out_act="None", gtype="Hash", qt=0.0, max_level=None)
cases = [Case("draw109", d109, 303 + 109), Case("draw108", d108, 303 + 108), Case("hash-lds", dlds, 7, batch=320), Case("hash-lds-4096", dlds, 8, batch=4096),
         Case("hash-f4", d4, 9)]
sv = api.vnrCreateSimpleVolume(syn.analytic_volume(16))
events = 0

# ---- fast: one volume per model, learning rate 0 ---------------------------------------------------------------------------
t0 = time.time()
vols = []
for c in cases:
    v = api.vnrCreateNeuralVolume(cfg_of(c.d, lr=0.0), sv)
    api.neural_set_params_fp16(v, c.params)
    c.fb(v); c.adopt_reference(v); api.neural_train_end(v)
    assert np.array_equal(api.neural_get_params_fp16(v).view(np.uint16), c.params.view(np.uint16)), "a step at learning rate 0 moved the parameters"
    vols.append(v)
done = 0
for k in range(n_fast):
    if time.time() - t0 > BUDGET[0]: break
    done = k + 1
    i = k % len(cases)
    c, v = cases[i], vols[i]
    c.fb(v)
    if not c.check(v, "fast", k):
        events += 1
    api.neural_train_end(v)
    if k % 20000 == 0 and k:
        say("fast", k, "repetitions, %.0f s," % (time.time() - t0), "events", events, "worst (mlp, grid):", [tuple("%.2e" % x for x in c.worst) for c in cases])
say("fast done:", done, "repetitions in %.0f s," % (time.time() - t0), "events", events, "worst (mlp, grid):", [(c.name,) + tuple("%.2e" % x for x in c.worst) for c in cases])
for c in cases: c.worst = [0.0, 0.0]
del vols

# ---- fresh: a new volume per repetition, as the sweep does it ----------------------------------------------------------------------
t0 = time.time(); ev0 = events
done = 0
for k in range(n_fresh):
    if time.time() - t0 > BUDGET[1]: break
    done = k + 1
    c = cases[k % 2]
    v = api.vnrCreateNeuralVolume(cfg_of(c.d), sv)
    api.neural_set_params_fp16(v, c.params)
    if k % 3 == 0:
        api.neural_encode(v, c.coords); api.neural_inference(v, c.coords)
    c.fb(v)
    ok = c.check(v, "fresh", k)
    if ok and k % 7 == 0:                          # the sweep's own route: the float copy, downloaded
        g = api.neural_gradients(v)
        hm, hg = c.host_rel(g.astype(np.float16).view(np.uint16))
        if not (hm <= 1e-3 and hg <= 5e-3):
            say("EVENT fresh", k, c.name, "the device's verdict was fine, the DOWNLOADED float copy is not: host rel", hm, hg); ok = False
    if not ok: events += 1
    api.neural_train_end(v); del v
    if k % 5000 == 0 and k:
        say("fresh", k, "repetitions, %.0f s," % (time.time() - t0), "events", events - ev0)
say("fresh done:", done, "repetitions in %.0f s," % (time.time() - t0), "events", events - ev0, "worst (mlp, grid):", [(c.name,) + tuple("%.2e" % x for x in c.worst) for c in cases[:2]])
for c in cases: c.worst = [0.0, 0.0]

# ---- reconf: one volume, re-configured between the models -------------------------------------------------------------------------------
t0 = time.time(); ev0 = events
v = api.vnrCreateNeuralVolume(cfg_of(cases[0].d), sv)
order = [0, 1, 2, 0, 4, 1, 3, 0, 2, 4]
done = 0
for k in range(n_reconf):
    if time.time() - t0 > BUDGET[2]: break
    done = k + 1
    c = cases[order[k % len(order)]]
    api.vnrNeuralVolumeSetModel(v, cfg_of(c.d, lr=0.0))
    api.neural_set_params_fp16(v, c.params)
    for rep in range(1 + k % 3):                   # the step after a re-configuration, and the ones that follow it
        c.fb(v)
        if not c.check(v, "reconf", k): events += 1
        api.neural_train_end(v)
    if k % 5000 == 0 and k:
        say("reconf", k, "re-configurations, %.0f s," % (time.time() - t0), "events", events - ev0)
say("reconf done:", done, "re-configurations in %.0f s," % (time.time() - t0), "events", events - ev0, "worst (mlp, grid):", [(c.name,) + tuple("%.2e" % x for x in c.worst) for c in cases])
say("TOTAL events:", events)
LOG.close()
sys.exit(1 if events else 0)
