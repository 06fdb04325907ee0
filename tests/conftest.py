import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# PyTorch BEFORE anything loads libvnr_amd.so: the torch wheel bundles its own ROCm runtime, and if the library (linked
# against the system ROCm) is loaded first, torch finds no GPU afterwards (instantvnr_amd/_lib.require_torch_loaded_first).
# tests/test_gpu_dist.py uses torch on the GPU next to the library, and test order must not decide whether it works.
try:
    import torch  # noqa: F401,E402
except ImportError:  # the CPU suite does not need it except for tests/test_dist_cpu.py, which imports it itself
    pass


# The parity tests compare iteration counts and slot statistics with the oracle run at the REFERENCE's batch size
# (N_ITERS = 16, method_raymarching.cu:30-40); the library's own default is 24 (frames agree to the last bits of a few samples
# either way: tests/test_gpu_fullsize.py).
os.environ.setdefault("VNR_RM_N_ITERS", "16")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
