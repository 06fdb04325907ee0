"""GPU box, diagnostic build (tools/ab_build.sh stamps -DVNR_MARCH_STAMPS): one record per wave-trip of the SECOND walk launch of a frame
(walk_kernel / walk8_kernel, csrc/decoupled.h) on a share of the bench frame: when the wave started and ended on the device's 100 MHz clock
(s_memrealtime: the launch's dispatch profile) and the s_memtime cycles of its phases.  Plain stores, no atomics.
usage: VNR_AMD_LIB_PATH=.../libvnr_amd_stamps.so VNR_AMD_DECOUPLED=2 [VNR_AMD_DEBUG_SKIP_EVAL=1] [VNR_AMD_DECOUPLED_LANES=1|8] python tools/wave_records.py [shares]"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantvnr_amd import api, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
L = lib(); check(L.vnrAmdInit(-1))
size, fb = 1024, 1024
dims = (size,) * 3
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=float(np.exp(np.log(size / 16.0) / 15)))
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 300, True)
cam = syn.oblique_camera(dims, distance_scale=1.1)
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)
lanes = int(os.environ.get("VNR_AMD_DECOUPLED_LANES", "1"))
for parts in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8").split(",")]:
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetFramebufferSize(ren, (fb, fb)); api.vnrRendererSetOutputAsDeviceFramebuffer(ren, True)
    camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"]); api.vnrRendererSetCamera(ren, camera)
    tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
    if parts > 1:
        api.vnrRendererSetPixelInterleave(ren, 8 * fb, parts, 0)
    for _ in range(6):
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    check(L.vnrAmdSynchronize())
    if os.environ.get("VNR_AMD_DECOUPLED", "1") == "0":   # the coupled loop: the LONGEST trip of every 64-ray group of march_kernel<false> over a frame
        out = np.zeros((65536, 8), np.uint64)
        L.vnrAmdDebugWaveRecords(out.ctypes.data_as(C.c_void_p), 65536, 1)
        api.vnrRender(ren); api.vnrRendererMapFrame(ren)
        check(L.vnrAmdSynchronize())
        L.vnrAmdDebugWaveRecords(out.ctypes.data_as(C.c_void_p), 65536, 0)
        r = out[out[:, 7] > 0].astype(np.int64)
        dur = (r[:, 1] - r[:, 0]) / 100.0
        order = np.argsort(-r[:, 7])
        print(f"share 1/{parts}, coupled loop: {len(r)} groups; the longest trip of a group lasts (us, percentiles 0 10 50 90 99 100): "
              + " ".join(f"{np.percentile(dur, q):.1f}" for q in (0, 10, 50, 90, 99, 100)))
        names = ["state load + compose", "walk + emit to LDS", "compaction + claim", "depth-bin sort", "stores", "whole trip"]
        for label, sel in (("all groups", order), ("the longest tenth", order[:max(len(order) // 10, 1)]), ("the longest hundredth", order[:max(len(order) // 100, 1)])):
            print(f"   {label}: cycles, mean per phase")
            for k, name in enumerate(names):
                print(f"      {name:24s} {r[sel, 2 + k].mean():9.0f}")
        del ren
        continue
    n_rays = fb * fb // parts
    n_rec = n_rays // 64 * (8 if lanes == 8 else 1)
    out = np.zeros((n_rec, 8), np.uint64)
    L.vnrAmdDebugWaveRecords(out.ctypes.data_as(C.c_void_p), n_rec, 1)
    api.vnrRender(ren); api.vnrRendererMapFrame(ren)
    check(L.vnrAmdSynchronize())
    L.vnrAmdDebugWaveRecords(out.ctypes.data_as(C.c_void_p), n_rec, 0)
    r = out[out[:, 0] > 0].astype(np.int64)
    t0 = r[:, 0].min()
    start, end = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0   # us
    pct = lambda a: " ".join(f"{np.percentile(a, q):8.1f}" for q in (0, 10, 50, 90, 99, 100))
    print(f"share 1/{parts}, {lanes} lane(s) per ray: {len(r)} wave-trips recorded of {n_rec}; first start to last end {end.max():.1f} us")
    print(f"   percentiles                 0       10       50       90       99      100")
    print(f"   trip starts at (us)  {pct(start)}")
    print(f"   trip ends at (us)    {pct(end)}")
    print(f"   trip lasts (us)      {pct(end - start)}")
    names = ["walk", "claim (barriers + atomic)", "sort", "stores (drained)"]
    for k, name in enumerate(names):
        c = r[:, 2 + k]
        print(f"   {name:28s} cycles: mean {c.mean():9.0f}  median {np.median(c):9.0f}  p99 {np.percentile(c, 99):9.0f}  max {c.max():9.0f}")
    whole = r[:, 2:6].sum(1)
    live = (end - start) > 0
    print(f"   cycles per us over the trips: {np.median(whole[live] / np.maximum(end - start, 0.01)[live]):.0f}")
    if lanes == 8:
        rounds = r[:, 7] >> 48
        print(f"   of the walk: opacity fetches {r[:, 6].mean():9.0f} (median {np.median(r[:, 6]):.0f}), the rest of the rounds {(r[:, 7] & 0xffffffffffff).mean():9.0f}; rounds of the wave's first ray: mean {rounds.mean():.2f} max {rounds.max()}")
    del ren
