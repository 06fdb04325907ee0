"""GPU box: rates of the generic (non-MFMA) kernels that carry the model shapes outside the MFMA kernels' configuration: inference
G samples/s on 4 M random coordinates and ms per training step (batch 65 536), next to the same numbers of a 64-neuron model.
usage: python tools/generic_probe.py"""
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
for W, interp, label in ((64, "Linear", "MFMA kernels"), (16, "Linear", "generic"), (32, "Linear", "generic"), (128, "Linear", "generic"), (64, "Nearest", "generic")):
    cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=19, n_hidden_layers=3, per_level_scale=1.3)
    cfg["network"]["n_neurons"] = W
    cfg["encoding"]["interpolation"] = interp
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
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
    print(f"n_neurons {W:3d} {interp:8s} ({label}): inference {inf:6.2f} G samples/s (random coordinates), training {ms:.3f} ms per step, loss {api.vnrNeuralVolumeGetTrainingLoss(nv):.4f}", flush=True)
