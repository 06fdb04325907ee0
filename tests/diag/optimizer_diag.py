"""GPU box, by hand: replays draws of the optimizer sweep of tests/test_gpu_fuzz.py with the history of the worst element printed.
usage: optimizer_diag.py <seed incl. the +71> <index> [<index> ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_fuzz as fz
from oracle import oracle as o
o.build()
for i in sys.argv[2:]:
    try:
        fz.optimizer_draw(o, int(sys.argv[1]), int(i), verbose=True)
        print("draw", i, "ok")
    except AssertionError as e:
        print("draw", i, "FAIL", repr(e)[:200])
