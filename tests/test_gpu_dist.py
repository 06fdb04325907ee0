"""The device side of the multi-GPU path on ONE GPU: a one-rank `nccl` (= RCCL) process group.

What cannot be tested on a 1-GPU box is more than one rank; what can, and is the part most likely to break on the
driver's multi-GPU run, is everything a rank does with its own GPU: torch aliasing the library's device buffers through
`__cuda_array_interface__` (no copies), the library-stream -> torch-stream hand-off, RCCL initialisation on this image
(HSA_ENABLE_IPC_MODE_LEGACY=0), `all_gather_into_tensor` of the frame shares and `all_reduce` of the gradient blob.
With one rank both collectives are the identity, so results must equal the undistributed path.
`tests/test_dist_cpu.py` covers world size 2 (gloo) with a CPU stand-in for the renderer;
`tests/test_gpu_render.py::test_interleaved_shares_assemble_to_the_unsharded_frame` covers the sharded rendering itself."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from instantvnr_amd import api
from instantvnr_amd import dist as vdist
from instantvnr_amd import synthetic as syn
from instantvnr_amd._lib import check, check_ptr, lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def group():
    import torch
    import torch.distributed as dist
    check(lib().vnrAmdInit(0))
    torch.cuda.set_device(0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def test_frame_share_roundtrip_through_rccl(group):
    import torch
    size = (96, 80)
    n_pixels, block = size[0] * size[1], 8 * size[0]
    vol = syn.analytic_volume(48)
    sv = api.vnrCreateSimpleVolume(vol)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    cam = syn.oblique_camera((48, 48, 48), distance_scale=0.95)
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])

    def renderer(device_output):
        r = api.vnrCreateRenderer(sv)
        api.vnrRendererSetTransferFunction(r, tfn)
        api.vnrRendererSetCamera(r, camera)
        api.vnrRendererSetFramebufferSize(r, size)
        api.vnrRendererSetMode(r, 5)
        api.vnrRendererSetOutputAsDeviceFramebuffer(r, device_output)
        return r

    r_host = renderer(False)
    api.vnrRender(r_host)
    want = api.vnrRendererMapFrame(r_host).reshape(-1, 4).copy()
    assert (want[:, 3] > 0).mean() > 0.2

    r_dev = renderer(True)
    api.vnrRender(r_dev)
    ptr = api.vnrRendererMapFrame(r_dev)                       # device pointer; MapFrame synced the render streams
    frame = vdist.as_torch(ptr, (n_pixels, 4))                # alias, no copy
    assert frame.is_cuda and frame.data_ptr() == int(ptr)
    share = vdist.pack_share(frame, block, 1, 0, n_pixels)
    _, _, n_local = vdist.interleave_layout(n_pixels, block, 1)
    gathered = torch.empty((1, n_local, 4), dtype=torch.float32, device="cuda")
    group.all_gather_into_tensor(gathered.view(-1), share.view(-1))
    full = vdist.assemble_shares(gathered, block, 1, n_pixels)
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy(), want)


def test_sharded_renderer_object_on_a_one_rank_group(group):
    """dist.ShardedRenderer as bench.py drives it with WORLD_SIZE > 1, on the one-rank group: the even-division path (strided
    view of the share made once per framebuffer, all_gather, one strided copy into the frame) over several frames, so both of
    the renderer's framebuffers are aliased; the frames must equal the undistributed renderer's"""
    import torch

    class OneRank(vdist.Context):
        distributed = True

    size = (64, 48)
    vol = syn.analytic_volume(32)
    sv = api.vnrCreateSimpleVolume(vol)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    cam = syn.oblique_camera((32, 32, 32), distance_scale=0.95)
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])

    def renderer():
        r = api.vnrCreateRenderer(sv)
        api.vnrRendererSetTransferFunction(r, tfn)
        api.vnrRendererSetCamera(r, camera)
        api.vnrRendererSetFramebufferSize(r, size)
        return r

    plain = renderer()
    sharded = vdist.ShardedRenderer(OneRank(0, 1, 0, "nccl"), renderer(), size[0], size[1])
    assert sharded.even
    # the object pipelines: render() enqueues frame k (asynchronous frames), gathers frame k - 1 and returns it
    wants = []
    for k in range(5):   # accumulation over frames, alternating framebuffers
        api.vnrRender(plain)
        wants.append(api.vnrRendererMapFrame(plain).reshape(-1, 4).copy())
        full = sharded.render()
        if k == 0:
            assert full is None
        else:
            torch.cuda.synchronize()
            assert np.array_equal(full.cpu().numpy(), wants[k - 1])
    full = sharded.flush()
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy(), wants[-1])
    assert sharded.flush() is None
    assert len(sharded._views) == 2
    # frame statistics complete a pending frame; a render after a flush starts the pipeline again
    assert sharded.render() is None
    st = api.vnrRendererGetFrameStats(sharded.r)
    assert st["n_samples"] > 0 and st["n_iterations"] > 0
    api.vnrRender(plain)
    want = api.vnrRendererMapFrame(plain).reshape(-1, 4).copy()
    full = sharded.flush()
    torch.cuda.synchronize()
    assert np.array_equal(full.cpu().numpy(), want)


def test_asynchronous_frames_equal_synchronous_ones(group, monkeypatch):
    """vnrAmdRendererSetAsync: vnrRender returns after enqueueing the iterations the previous frame needed; MapFrame completes the
    frame.  Frames must equal the synchronous renderer's, also when a frame needs MORE iterations than the previous one (the
    camera moves from far to near and the sampling rate rises: the prediction is too short and MapFrame has to launch the rest) and when
    it needs fewer."""
    import torch
    size = (96, 64)
    n_pixels = size[0] * size[1]
    vol = syn.analytic_volume(48)
    sv = api.vnrCreateSimpleVolume(vol)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    monkeypatch.setenv("VNR_RM_N_ITERS", "4")   # read when a renderer is created: many iterations per frame on a small volume

    def renderer(asynchronous):
        r = api.vnrCreateRenderer(sv)
        api.vnrRendererSetTransferFunction(r, tfn)
        api.vnrRendererSetFramebufferSize(r, size)
        api.vnrRendererSetMode(r, 5)
        api.vnrRendererSetOutputAsDeviceFramebuffer(r, True)
        check(lib().vnrAmdRendererSetAsync(r.h, 1 if asynchronous else 0))
        return r
    r_sync, r_async = renderer(False), renderer(True)
    iterations = []
    for distance, rate in ((3.0, 1.0), (3.0, 1.0), (0.9, 4.0), (0.9, 4.0), (0.9, 1.0), (3.0, 0.5), (1.5, 2.0)):
        cam = syn.oblique_camera((48, 48, 48), distance_scale=distance)
        frames = []
        for r in (r_sync, r_async):
            camera = api.vnrCreateCamera()
            api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
            api.vnrRendererSetCamera(r, camera)
            api.vnrRendererSetVolumeSamplingRate(r, rate)
            api.vnrRender(r)
            ptr = api.vnrRendererMapFrame(r)
            frames.append(vdist.as_torch(ptr, (n_pixels, 4)).cpu().numpy().copy())
        assert np.array_equal(frames[0], frames[1])
        a, b = api.vnrRendererGetFrameStats(r_sync), api.vnrRendererGetFrameStats(r_async)
        assert a["n_samples"] == b["n_samples"] and a["n_iterations"] == b["n_iterations"] and a["n_rays_hit"] == b["n_rays_hit"]
        iterations.append(a["n_iterations"])
    assert max(iterations) > min(iterations) + 1   # the scene really changed its iteration count
    # statistics before MapFrame complete the frame too
    api.vnrRender(r_async)
    st = api.vnrRendererGetFrameStats(r_async)
    assert st["n_iterations"] == iterations[-1]


def test_gradient_allreduce_step_equals_plain_step(group):
    import torch
    os.environ["VNR_AMD_INIT_SEED"] = "77"
    data = syn.analytic_volume(32)
    cfg = syn.model_config(n_levels=4, n_features=2, log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2)
    L = lib()
    res = []
    for through_rccl in (False, True):
        sv = api.vnrCreateSimpleVolume(data)
        nv = api.vnrCreateNeuralVolume(cfg, sv)
        if through_rccl:
            grads = None
            for _ in range(20):
                check(L.vnrAmdNeuralVolumeTrainBegin(nv.h))
                if grads is None:
                    n = C.c_size_t()
                    p = check_ptr(L.vnrAmdNeuralVolumeGradients(nv.h, C.byref(n)))
                    grads = vdist.as_torch(p, (n.value,))
                    assert grads.numel() == api.neural_info(nv)["n_params"]
                check(L.vnrAmdSynchronize())                   # library stream -> torch stream hand-off
                before = float(grads.abs().sum())
                group.all_reduce(grads, op=group.ReduceOp.SUM)  # one rank: identity, in place on the library's buffer
                torch.cuda.current_stream().synchronize()
                assert before > 0 and float(grads.abs().sum()) == before
                check(L.vnrAmdNeuralVolumeTrainEnd(nv.h, 1.0, 1))
        else:
            api.vnrNeuralVolumeTrain(nv, 20, True)
        res.append((api.vnrNeuralVolumeGetTrainingLoss(nv), api.neural_get_params_fp16(nv).astype(np.float32)))
    # same bar as test_split_training_step_equals_train: float atomics make training not bitwise reproducible
    assert abs(res[0][0] - res[1][0]) < 0.1 * res[0][0]
    assert np.mean(np.abs(res[0][1] - res[1][1])) < 1e-3
