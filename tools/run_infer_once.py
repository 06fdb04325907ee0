"""GPU box: launches the fused kernel a few times on a ray-coherent batch (for rocprofv3 --pmc passes)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from instantvnr_amd import api, synthetic as syn  # noqa: E402
L = api.lib()
api.check(L.vnrAmdInit(-1))
kind = sys.argv[1] if len(sys.argv) > 1 else "c4"
n = 1 << 24
if kind == "c4":
    cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=1.3195)
else:
    cfg = syn.model_config(n_levels=8, n_features=8, log2_hashmap_size=19, n_hidden_layers=2)
vol = api.vnrCreateNeuralVolume(cfg, (64, 64, 64))
rays = n // 16
side = int(np.sqrt(rays))
u, v = np.meshgrid(np.arange(side), np.arange(side))
o = np.stack([u.ravel() / side, v.ravel() / side, np.zeros(side * side)], 1)
t = (np.arange(16) / 1024.0 + 0.3)[None, :, None] * np.array([0.05, 0.02, 1.0])[None, None, :]
c = np.clip((o[:, None, :] + t).reshape(-1, 3), 0, 1).astype(np.float32)
dc = api.DeviceArray.from_numpy(c)
do = api.DeviceArray((c.shape[0],), np.float32)
for _ in range(5):
    api.check(L.vnrAmdNeuralVolumeInference(vol.h, c.shape[0], dc.ptr, do.ptr, None))
api.check(L.vnrAmdSynchronize())
print("done", c.shape[0])
