"""GPU box: one rank's frame loop of an 8-GPU run, on ONE GPU: a 1/8 share of the bench frame rendered by the real kernels and
gathered through dist.ShardedRenderer, with RCCL reduced to a one-rank group (the all_gather moves this rank's 2 MB into
slot 0: same host path and launch, no xGMI transfer).  Compares the pipelined object (render k | gather k - 1) with the plain
sequence render, map, gather.  usage: python tools/gather_probe.py [parts]"""
import os
import sys
import time
import socket
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (before the library: INTEGRATION.md 4)
import torch.distributed as tdist  # noqa: E402
from instantvnr_amd import api, dist as vdist, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.cuda.set_device(0)
L = lib(); check(L.vnrAmdInit(0))
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
tdist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
size, fb = 1024, 1024
dims = (size,) * 3
sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
pls = float(np.exp(np.log(size / 16.0) / 15))
cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
api.vnrNeuralVolumeTrain(nv, 300, True)
cam = syn.oblique_camera(dims, distance_scale=1.1)
colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)


class Rank0Of(vdist.Context):
    distributed = True


real_all_gather = tdist.all_gather_into_tensor


def one_rank_all_gather(out, inp, *a, **k):
    return real_all_gather(out[: inp.numel()], inp, *a, **k)


tdist.all_gather_into_tensor = one_rank_all_gather


def make():
    ren = api.vnrCreateRenderer(nv)
    api.vnrRendererSetFramebufferSize(ren, (fb, fb))
    camera = api.vnrCreateCamera(); api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    api.vnrRendererSetCamera(ren, camera)
    tfn = api.vnrCreateTransferFunction(); api.vnrTransferFunctionSetColor(tfn, colors); api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1)); api.vnrRendererSetTransferFunction(ren, tfn)
    return vdist.ShardedRenderer(Rank0Of(0, parts, 0, "nccl"), ren, fb, fb)


def timed(fn, flush, n=60):
    for _ in range(8):
        fn()
    flush(); check(L.vnrAmdSynchronize()); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    flush(); check(L.vnrAmdSynchronize()); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


sr = make()
ms_pipe = timed(sr.render, sr.flush)
sr2 = make()
check(L.vnrAmdRendererSetAsync(sr2.r.h, 0))


def plain():
    api.vnrRender(sr2.r)
    sr2._gather(api.vnrRendererMapFrame(sr2.r))


ms_plain = timed(plain, lambda: None)
sr3 = make()
check(L.vnrAmdRendererSetAsync(sr3.r.h, 0))


def render_only():
    api.vnrRender(sr3.r); api.vnrRendererMapFrame(sr3.r)


ms_render = timed(render_only, lambda: None)
print(f"1/{parts} share: render only {ms_render:.3f} ms, render + gather in sequence {ms_plain:.3f} ms, pipelined {ms_pipe:.3f} ms per frame", flush=True)
tdist.destroy_process_group()
