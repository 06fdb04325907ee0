"""The OVR renderer plugin "nnvolume" (ovr_plugin/, replacing /root/reference/device/): its library-facing half is compiled against
include/vnr_amd.h and driven by a test host that plays OVR's main loop on plain data (tests/ovr_plugin_host.cpp); the frames it hands
out must be the frames the vnr* API renders for the same scene.  The OVR-facing half (ovr_plugin/device_nnvolume_amd.cpp) needs the OVR
headers, which the reference tree does not vendor: it is checked for the entry points the reference's plugin has, not compiled."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from instantvnr_amd import api
from instantvnr_amd import synthetic as syn
from instantvnr_amd._lib import check, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_host(tmp_path):
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "ovr_plugin_host")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ovr_plugin"),
                           os.path.join(ROOT, "tests", "ovr_plugin_host.cpp"), "-o", exe, "-L", os.path.join(ROOT, "instantvnr_amd"), "-lvnr_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "instantvnr_amd")])
    return exe


def test_plugin_adapter_compiles_and_has_the_reference_plugin_s_entry_points(tmp_path):
    build_host(tmp_path)
    src = open(os.path.join(ROOT, "ovr_plugin", "device_nnvolume_amd.cpp")).read()
    for name in ("void init(int", "void swap() override", "void commit() override", "void render() override", "void mapframe(FrameBufferData* fb) override",
                 "OVR_REGISTER_OBJECT(ovr::MainRenderer, renderer, ovr::nnvolume::DeviceNNVolume, nnvolume)", "CrossDeviceBuffer::DEVICE_CUDA"):
        assert name in src, name
    # every vnrAmd* call of the adapter is declared in the C-ABI header
    hdr = open(os.path.join(ROOT, "include", "vnr_amd.h")).read()
    used = set(re.findall(r"\b(vnrAmd[A-Za-z0-9]+)\(", open(os.path.join(ROOT, "ovr_plugin", "device_nnvolume_amd.h")).read()))
    assert used and all(re.search(r"\b%s\s*\(" % u, hdr) for u in used), used


@pytest.mark.gpu
def test_plugin_frames_equal_the_api_s_frames(tmp_path):
    """init (volume from memory with its type, object -> world map from origin / spacing, transfer function from nodes in data units),
    commit (resize, camera, a new transfer function, sampling rate), render, mapframe (device pixels) through the adapter, against the
    same scene set up through the Python mirror of api.h"""
    exe = build_host(tmp_path)
    n = (40, 32, 24)
    vol = (syn.analytic_volume(40)[:24, :32, :] * 60000.0 + 2000.0).astype(np.uint16)      # [z, y, x]
    assert vol.shape == (24, 32, 40)
    vol.tofile(tmp_path / "v.raw")
    out = tmp_path / "frames.raw"
    rc = subprocess.call([exe, str(tmp_path / "v.raw"), "40", "32", "24", str(out)])
    assert rc == 0
    W, H = 160, 120
    got = np.fromfile(out, np.float32).reshape(2, H, W, 4)
    # the same through the API
    sv = api.vnrCreateSimpleVolume(vol)            # normalised by its own min / max
    lo, hi = float(vol.min()), float(vol.max())
    m = (C_float12)(40, 0, 0, 0, 32, 0, 0, 0, 24, -20.0, -16.0, -12.0)
    check(lib().vnrAmdVolumeSetTransform(sv.h, m))
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, [[0, 0, 1], [0, 1, 0], [1, 1, 0], [1, 0, 0]])
    api.vnrTransferFunctionSetAlpha(tfn, [[0.0, 0.0], [0.25, 0.02], [0.5, 0.1], [0.75, 0.3], [1.0, 0.6]])
    api.vnrTransferFunctionSetValueRange(tfn, ((0.0 - lo) / (hi - lo), (65535.0 - lo) / (hi - lo)))
    cam = api.vnrCreateCamera()
    api.vnrCameraSet(cam, (1.6 * 40, 1.1 * 32, -1.9 * 24), (0, 0, 0), (0, 1, 0))
    r = api.vnrCreateRenderer(sv)
    api.vnrRendererSetTransferFunction(r, tfn)
    api.vnrRendererSetCamera(r, cam)
    api.vnrRendererSetFramebufferSize(r, (W, H))
    api.vnrRendererSetMode(r, 5)
    for _ in range(3):
        api.vnrRender(r)
        want0 = api.vnrRendererMapFrame(r).copy()
    assert (want0[..., 3] > 0).mean() > 0.1
    assert np.array_equal(got[0], want0)
    api.vnrTransferFunctionSetColor(tfn, [[1, 1, 1], [1, 0.5, 0]])
    api.vnrTransferFunctionSetAlpha(tfn, [[0.0, 0.0], [0.5, 0.05], [1.0, 0.8]])
    w = np.float32(1.0) / (np.float32(hi) - np.float32(lo))      # the adapter maps the range in single precision, in this order
    api.vnrTransferFunctionSetValueRange(tfn, (float((np.float32(8000.0) - np.float32(lo)) * w), float((np.float32(60000.0) - np.float32(lo)) * w)))
    api.vnrRendererSetTransferFunction(r, tfn)
    api.vnrRendererSetVolumeSamplingRate(r, 2.0)
    api.vnrRender(r)
    want1 = api.vnrRendererMapFrame(r).copy()
    assert np.array_equal(got[1], want1) and not np.array_equal(want1, want0)


import ctypes as _C   # noqa: E402
C_float12 = _C.c_float * 12


def test_the_uncompiled_half_uses_only_names_the_reference_s_plugin_uses():
    """tools/ovr_adapter_vs_reference.py, where the reference's device/ sources are present: the OVR-facing half of the adapter cannot be
    compiled in this image, so every OVR-side identifier in it is at least looked up in the reference's own plugin"""
    import os
    import subprocess
    import sys
    import pytest
    if not os.path.isdir("/root/reference/device"):
        pytest.skip("the reference's sources are not on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "ovr_adapter_vs_reference.py")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "unknown" not in out.stdout, out.stdout
