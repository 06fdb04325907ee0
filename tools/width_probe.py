"""GPU box: rates of every FullyFusedMLP width / interpolation / activation / grid type on the probe model of round 3
(tools/generic_probe.py: L16 F2 T2^19 3 hidden layers, 256^3 Perlin volume): inference G samples/s on 4 M random coordinates and ms per
training step (batch 65 536).  Since round 4 all of them run on the MFMA kernels (vnrAmdNeuralVolumeGetModelKind says which do not).
usage: python tools/width_probe.py"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
sv = api.vnrCreateSimpleVolumePerlin((256, 256, 256), seed=42, octaves=4, base_frequency=6.0)
coords = api.DeviceArray.from_numpy(np.random.default_rng(0).random((1 << 22, 3), dtype=np.float32))
out = api.DeviceArray((1 << 22,), np.float32)
ROWS = [(64, "Linear", {}), (16, "Linear", {}), (32, "Linear", {}), (128, "Linear", {}), (64, "Nearest", {}), (128, "Nearest", {}),
        (64, "Linear", {"activation": "Sigmoid"}), (64, "Linear", {"activation": "Squareplus", "output_activation": "Exponential"}),
        (64, "Linear", {"type": "Tiled"}), (64, "Linear", {"quantize_threshold": 1e-4}), (128, "Linear", {"n_hidden_layers": 6})]
for W, interp, extra in ROWS:
    cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=19, n_hidden_layers=3, per_level_scale=1.3)
    cfg["network"]["n_neurons"] = W
    cfg["encoding"]["interpolation"] = interp
    for k, v in extra.items():
        cfg["network" if "activation" in k or k == "n_hidden_layers" else "encoding"][k] = v
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    info = api.neural_info(nv)
    for _ in range(2):
        check(L.vnrAmdNeuralVolumeInference(nv.h, 1 << 22, coords.ptr, out.ptr, None))
    check(L.vnrAmdSynchronize())
    t = time.perf_counter()
    for _ in range(5):
        check(L.vnrAmdNeuralVolumeInference(nv.h, 1 << 22, coords.ptr, out.ptr, None))
    check(L.vnrAmdSynchronize())
    inf = 5 * (1 << 22) / (time.perf_counter() - t) / 1e9
    api.vnrNeuralVolumeTrain(nv, 20, True)
    check(L.vnrAmdSynchronize())
    t = time.perf_counter()
    api.vnrNeuralVolumeTrain(nv, 100, True)
    check(L.vnrAmdSynchronize())
    ms = (time.perf_counter() - t) * 10
    kind = "MFMA kernels"
    print(f"n_neurons {W:3d} {interp:8s} {str(extra):62s} ({kind}): inference {inf:6.2f} G samples/s (random coordinates), training {ms:.3f} ms per step, "
          f"loss {api.vnrNeuralVolumeGetTrainingLoss(nv):.4f}", flush=True)
