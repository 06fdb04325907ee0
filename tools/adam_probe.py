"""Decompose the Adam kernel's time: a normal step (touched entries + sweep) vs a step whose batch touches almost no grid
entry (sweep only).  Run under `rocprofv3 --kernel-trace --output-format csv` and read the
alternating adam_kernel durations from the trace (tools/adam_probe.py prints nothing itself but the model size).
C4-sized model (L16 F2 T2^22, 3x64), batch 65 536, small ground-truth volume."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402

pls = float(np.exp(np.log(1024 / 16.0) / 15))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, base_resolution=16, n_hidden_layers=3, per_level_scale=pls)
sv = api.vnrCreateSimpleVolume(syn.analytic_volume(64))
nv = api.vnrCreateNeuralVolume(cfg, sv)
print("n_params", api.neural_info(nv)["n_params"])
api.vnrNeuralVolumeTrain(nv, 20, True)           # warm-up
tiny_x = api.DeviceArray.from_numpy(np.full((256, 3), 0.37, dtype=np.float32))
tiny_y = api.DeviceArray.from_numpy(np.full(256, 0.5, dtype=np.float32))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    api.neural_train_begin(nv)                   # sample + forward + backward: gradients of a 65 536-sample batch
    api.neural_train_end(nv, 1.0, True)          # Adam: touched entries + sweep; clears the gradients
    # (a second train_end is a no-op: nothing pending.)  Sweep only = a step whose batch touches almost nothing:
    # 256 samples at ONE coordinate -> at most 16 levels x 8 corners x 2 features of the 70 M grid parameters
    check(lib().vnrAmdNeuralVolumeForwardBackward(nv.h, 256, tiny_x.ptr, tiny_y.ptr))   # no host copy of the gradients
    api.neural_train_end(nv, 1.0, True)
print("loss", api.vnrNeuralVolumeGetTrainingLoss(nv))
