"""Counterpart of repro_torch_after_lib.py: import torch (optionally initialise its GPU context) BEFORE the library is
loaded, then use the library, then use torch + a one-rank RCCL collective.
usage: python tools/repro_torch_before_lib.py [import-only | cuda-init]"""
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
what = sys.argv[1] if len(sys.argv) > 1 else "import-only"
try:
    import torch
    if what == "cuda-init":
        torch.cuda.set_device(0)
        torch.zeros(1, device="cuda")
    from instantvnr_amd import api, synthetic as syn
    from instantvnr_amd._lib import check, lib
    check(lib().vnrAmdInit(0))
    sv = api.vnrCreateSimpleVolume(syn.analytic_volume(32))
    cfg = syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2)
    nv = api.vnrCreateNeuralVolume(cfg, sv)
    api.vnrNeuralVolumeTrain(nv, 5, True)
    print("library trained, loss", api.vnrNeuralVolumeGetTrainingLoss(nv))
    print("torch is_available", torch.cuda.is_available())
    torch.cuda.set_device(0)
    x = torch.ones(4, device="cuda")
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("torch + RCCL after library use ok:", x.sum().item())
    dist.destroy_process_group()
    # which HIP runtimes are mapped into this process?
    libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime64" in l})
    print("mapped:", libs)
except Exception:
    traceback.print_exc()
    sys.exit(3)
