"""GPU box: the C4 model's training step, the side-by-side backward pass (round 5: weight gradients + the dense levels' LDS scatter on a side
stream beside a persistent atomic scatter) against the one-stream pass, alternating inside ONE process on one trained state
(VNR_AMD_TRAIN_OVERLAP is read per step).  usage: train_ab.py [steps per leg] [repeats]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
size = int(os.environ.get("SIZE", 1024))
dims = (size,) * 3
os.environ.setdefault("VNR_AMD_INIT_SEED", "20240611")
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
pls = float(np.exp(np.log(size / 16.0) / 15))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 200, True)
check(L.vnrAmdSynchronize())
import ctypes as C
for rep in range(reps):
    for overlap in ("0", "1"):
        os.environ["VNR_AMD_TRAIN_OVERLAP"] = overlap
        api.vnrNeuralVolumeTrain(nv, 20, True)
        check(L.vnrAmdSynchronize())
        t0 = time.perf_counter()
        api.vnrNeuralVolumeTrain(nv, steps, True)
        check(L.vnrAmdSynchronize())
        ms = (time.perf_counter() - t0) / steps * 1e3
        check(L.vnrAmdNeuralVolumeSetTrainProfiling(nv.h, 1))
        api.vnrNeuralVolumeTrain(nv, 64, True)
        check(L.vnrAmdSynchronize())
        ph = (C.c_double * 5)(); n = C.c_int()
        check(L.vnrAmdNeuralVolumeGetTrainProfile(nv.h, ph, C.byref(n)))
        check(L.vnrAmdNeuralVolumeSetTrainProfiling(nv.h, 0))
        print(f"overlap {overlap}: {ms:.4f} ms per step (wall, {steps} steps); phases [forward, loss + MLP backward, weight gradients, grid backward (+ join), optimizer] = "
              + ", ".join(f"{ph[i] * 1e3:.1f}" for i in range(5)) + f" us; loss {api.vnrNeuralVolumeGetTrainingLoss(nv):.5f}", flush=True)
print("PSNR after all legs: %.2f dB" % api.vnrNeuralVolumeGetPSNR(nv))
