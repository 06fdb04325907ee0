"""GPU box: the training step of the C4 model by phase (HIP events between the phases of 64 profiled steps), under whatever
diagnostic switches the environment carries (VNR_AMD_GRID_BWD_LEVELS=l0,l1: scatter only those levels).
usage: python tools/train_probe.py [steps]"""
import ctypes as C
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
size = int(os.environ.get("SIZE", 1024))
sv = api.vnrCreateSimpleVolumePerlin((size,) * 3, seed=42, octaves=4, base_frequency=6.0)
cfg = syn.model_config(n_levels=int(os.environ.get("LEVELS", 16)), n_features=int(os.environ.get("FEATURES", 2)), log2_hashmap_size=int(os.environ.get("LOG2T", 22)),
                       n_hidden_layers=int(os.environ.get("HIDDEN", 3)), per_level_scale=float(np.exp(np.log(size / 16.0) / 15)))
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 100, True)
check(L.vnrAmdSynchronize())
t = time.perf_counter()
api.vnrNeuralVolumeTrain(nv, steps, True)
check(L.vnrAmdSynchronize())
wall = (time.perf_counter() - t) * 1e3 / steps
check(L.vnrAmdNeuralVolumeSetTrainProfiling(nv.h, 1))
api.vnrNeuralVolumeTrain(nv, 128, True)
check(L.vnrAmdSynchronize())
ph = (C.c_double * 5)()
n = C.c_int()
check(L.vnrAmdNeuralVolumeGetTrainProfile(nv.h, ph, C.byref(n)))
names = ("forward", "loss + MLP backward", "weight gradients", "grid backward", "optimizer")
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("VNR_AMD_")) or "default"
print(f"[train_probe] {tag}: {wall:.4f} ms per step (wall, {steps} steps), loss {api.vnrNeuralVolumeGetTrainingLoss(nv):.5f}; phases over {n.value} steps: "
      + ", ".join(f"{a} {b:.4f}" for a, b in zip(names, ph)) + f"; sum {sum(ph):.4f} ms", flush=True)
