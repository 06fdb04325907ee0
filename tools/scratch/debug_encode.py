import sys
import numpy as np
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from instantvnr_amd import api, synthetic as syn
from oracle import oracle
import test_gpu_network as T

for cfg in [(16, 1, 14, 4, 1.5, 2), (16, 2, 14, 4, 1.5, 2), (16, 1, 14, 4, None, 2), (8, 1, 14, 16, None, 2)]:
    L, F, log2T, base, pls, H = cfg
    vol, ocfg, params, n_mlp = T.make(oracle, L, F, log2T, base, pls, H)
    coords = T.coords_for(3000, 1)
    got = api.neural_encode(vol, coords)
    want = oracle.grid_encode(ocfg, params[n_mlp:].view(np.uint16), coords).view(np.float16)
    bad = np.argwhere(got.view(np.uint16) != want.view(np.uint16))
    lay = oracle.grid_layout(ocfg)
    print(cfg, "mismatches", len(bad), "cols", sorted(set(bad[:, 1].tolist())))
    print("  res", lay["resolution"], "sizes", np.diff(lay["offsets"].astype(np.int64)))
    for i, c in bad[:5]:
        print("   sample", i, coords[i], "col", c, "got", got[i, c], "want", want[i, c])
