"""GPU box, by hand: the whole per-draw check of the model sweep (create, set, encode, infer, gradients, params.json round trip, optimizer
steps) on ONE draw again and again in one process, as the sweep's loop does: a transient that depends on what the previous iteration left
behind shows here.  usage: check_repeat.py <seed> <index> <repetitions>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as fz
from oracle import oracle as o
o.build()
seed0, index, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed0)
draws = [fz.draw(rng) for _ in range(index + 1)]
os.environ["VNR_FUZZ_VERBOSE"] = "1"
bad = 0
for k in range(reps):
    for j in (index - 1, index):          # the draw before it, then the draw: what the sweep's loop had just done
        try:
            fz.check(o, draws[j], seed0 % 1000 + j)
        except AssertionError as e:
            bad += 1
            print("repetition", k, "draw", j, "FAILED", repr(e)[:300], flush=True)
    if k % 200 == 0: print("...", k, flush=True)
print("repetitions", reps, "failures", bad)
