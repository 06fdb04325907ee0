"""GPU box, experiment: what a spatially sorted training batch would buy.  The C4 model, 65 536 coordinates drawn uniformly (what the
sampler produces) against the same coordinates in Morton order: forward / backward / optimizer phases of a step from the library's own
HIP-event profile.  usage: python tools/sorted_batch_probe.py"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
from instantvnr_amd.api import DeviceArray  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
size = 1024
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=float(np.exp(np.log(size / 16.0) / 15)))
nv = api.vnrCreateNeuralVolume(cfg, (size,) * 3)
rng = np.random.default_rng(3)
B = 65536


def morton(c, bits=10):
    q = np.minimum((c * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    key = np.zeros(len(c), np.uint64)
    for b in range(bits):
        for k in range(3):
            key |= ((q[:, k] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + k)
    return key


names = ("forward", "loss + MLP backward", "weight gradients", "grid backward", "optimizer")
for order in ("random", "morton", "random", "morton"):
    coords = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    if order == "morton":
        coords = coords[np.argsort(morton(coords))]
    targets = rng.uniform(0, 1, B).astype(np.float32)
    c = DeviceArray.from_numpy(coords); t = DeviceArray.from_numpy(targets)
    for _ in range(20):
        check(L.vnrAmdNeuralVolumeForwardBackward(nv.h, B, c.ptr, t.ptr)); api.neural_train_end(nv)
    check(L.vnrAmdNeuralVolumeSetTrainProfiling(nv.h, 1))
    for _ in range(64):
        check(L.vnrAmdNeuralVolumeForwardBackward(nv.h, B, c.ptr, t.ptr)); api.neural_train_end(nv)
    check(L.vnrAmdSynchronize())
    ph = (C.c_double * 5)(); n = C.c_int()
    check(L.vnrAmdNeuralVolumeGetTrainProfile(nv.h, ph, C.byref(n)))
    check(L.vnrAmdNeuralVolumeSetTrainProfiling(nv.h, 0))
    print(f"{order:7s}: " + ", ".join(f"{a} {b:.4f}" for a, b in zip(names, ph)) + f"; sum {sum(ph):.4f} ms over {n.value} steps", flush=True)
