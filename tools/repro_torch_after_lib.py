"""Does torch see the GPU when it is initialised AFTER the library has been using it in the same process?
usage: python tools/repro_torch_after_lib.py [init-only | render | train]"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
what = sys.argv[1] if len(sys.argv) > 1 else "init-only"
print("env:", {k: v for k, v in os.environ.items() if "VISIBLE" in k or k.startswith("HSA_") or k.startswith("HIP_")})
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402

check(lib().vnrAmdInit(0))
if what in ("render", "train"):
    sv = api.vnrCreateSimpleVolume(syn.analytic_volume(32))
    if what == "train":
        cfg = syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2)
        nv = api.vnrCreateNeuralVolume(cfg, sv)
        api.vnrNeuralVolumeTrain(nv, 5, True)
        print("trained, loss", api.vnrNeuralVolumeGetTrainingLoss(nv))
    else:
        print("volume range", api.vnrVolumeGetValueRange(sv))
print("env after lib use:", {k: v for k, v in os.environ.items() if "VISIBLE" in k})
try:
    import torch
    print("torch", torch.__version__, "device_count", torch.cuda.device_count(), "is_available", torch.cuda.is_available())
    torch.cuda.set_device(0)
    x = torch.zeros(4, device="cuda") + 1
    print("torch on GPU ok:", x.sum().item())
except Exception:
    traceback.print_exc()
    sys.exit(3)
