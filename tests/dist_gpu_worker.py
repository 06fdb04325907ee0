"""One rank of a multi-process GPU test (tests/test_gpu_dist.py starts WORLD_SIZE of these on the ONE GPU of the test box, with
the host-staged "shm" transport, or one of them with the "rccl" transport).  usage: dist_gpu_worker.py <scenario> <out.npz>
Everything goes through the C-ABI entry points an application would use; results are written per rank for the parent to judge."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from instantvnr_amd import api, dist as vdist, synthetic as syn  # noqa: E402
from instantvnr_amd._lib import check, lib  # noqa: E402

SMALL_MODEL = dict(n_levels=4, n_features=int(os.environ.get("TEST_FEATURES", "2")), log2_hashmap_size=12, base_resolution=4, n_hidden_layers=2)


def make_scene_objects(dims):
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    cam = syn.oblique_camera(dims, distance_scale=0.95)
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    return tfn, camera


def make_renderer(volume, tfn, camera, size, mode):
    r = api.vnrCreateRenderer(volume)
    api.vnrRendererSetTransferFunction(r, tfn)
    api.vnrRendererSetCamera(r, camera)
    api.vnrRendererSetFramebufferSize(r, size)
    api.vnrRendererSetMode(r, mode)
    return r


def scenario_frames(ctx, out):
    """the assembled frame of every rank equals the unsharded frame, bit for bit: synchronous API (vnrRender + vnrRendererMapFrame,
    host frames), pipelined API (device frames), accumulation over frames, ragged and even tile-row counts, dense and neural volumes"""
    L = lib()
    vol = syn.analytic_volume(48)
    sv = api.vnrCreateSimpleVolume(vol)
    os.environ["VNR_AMD_INIT_SEED"] = "4242"   # identical networks on every rank without any exchange
    nv = api.vnrCreateNeuralVolume(syn.model_config(**SMALL_MODEL), sv, online_macrocell_construction=False)
    api.vnrNeuralVolumeTrain(nv, 30, True)     # every rank trains the same 30 steps on the same stream: identical up to atomics order ...
    check(L.vnrAmdNeuralVolumeSyncReplicas(nv.h))   # ... and exactly identical after this
    tfn, camera = make_scene_objects((48, 48, 48))
    n_frames = 3
    for case, (volume, size, mode) in enumerate([(sv, (96, 80), 5), (sv, (64, 64), 5), (nv, (96, 80), 5), (sv, (72, 40), 4), (nv, (64, 64), 8)]):
        plain = make_renderer(volume, tfn, camera, size, mode)
        want = []
        for _ in range(n_frames):
            api.vnrRender(plain)
            want.append(api.vnrRendererMapFrame(plain).copy())
        # synchronous: the calls an unchanged application makes
        r_sync = make_renderer(volume, tfn, camera, size, mode)
        check(L.vnrAmdRendererSetDistributed(r_sync.h, 1))
        ok_sync = True
        for k in range(n_frames):
            api.vnrRender(r_sync)
            ok_sync = ok_sync and bool(np.array_equal(api.vnrRendererMapFrame(r_sync), want[k]))
        # pipelined: frame k - 1 is gathered while frame k renders
        sr = vdist.ShardedRenderer(ctx, make_renderer(volume, tfn, camera, size, mode), size[0], size[1])
        got = []
        for k in range(n_frames):
            f = sr.render()
            if k == 0:
                assert f is None
            else:
                got.append(sr.download(f))
        got.append(sr.download(sr.flush()))
        assert sr.flush() is None
        ok_pipe = all(np.array_equal(g, w) for g, w in zip(got, want))
        st = api.vnrRendererGetFrameStats(sr.r)
        out[f"case{case}_sync"] = ok_sync
        out[f"case{case}_pipe"] = ok_pipe
        out[f"case{case}_coverage"] = float((want[-1][..., 3] > 0).mean())
        out[f"case{case}_samples"] = int(st["n_samples"])
        vdist.barrier()


def scenario_frames_c4(ctx, out):
    """BASELINE C4 at its full size on every rank: 1024^2 frame of a 1024^3 volume, L = 16 / F = 2 / T = 2^22 hash grid + 3 x 64 MLP with its
    de-hashed image (finest level beyond 4 GiB): the frame assembled from the ranks' pipelined shares equals the unsharded frame, bit for bit
    (N_ITERS pinned by the caller: a share of at most 262 144 pixels would otherwise march 32 samples per iteration)"""
    L = lib()
    dims = (1024, 1024, 1024)
    sv = api.vnrCreateSimpleVolumePerlin(dims, seed=42, octaves=4, base_frequency=6.0)
    os.environ["VNR_AMD_INIT_SEED"] = "4242"
    pls = float(np.exp(np.log(1024 / 16.0) / 15))
    cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    api.vnrNeuralVolumeTrain(nv, 100, True)
    check(L.vnrAmdNeuralVolumeSyncReplicas(nv.h))
    colors, alphas = syn.tfn_ramp_with_bumps(opacity_scale=0.06)
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    cam = syn.oblique_camera(dims, distance_scale=1.1)
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    size = (1024, 1024)
    plain = make_renderer(nv, tfn, camera, size, 5)
    n_frames = 4   # the image is built after 24 launches: the later frames read it
    want = []
    for _ in range(n_frames):
        api.vnrRender(plain)
        want.append(api.vnrRendererMapFrame(plain).copy())
    out["brick_in_use"] = bool(api.neural_brick_image(nv)["in_use"])
    sr = vdist.ShardedRenderer(ctx, make_renderer(nv, tfn, camera, size, 5), size[0], size[1])
    got = []
    for k in range(n_frames):
        f = sr.render()
        if k:
            got.append(sr.download(f))
    got.append(sr.download(sr.flush()))
    out["equal"] = np.array([bool(np.array_equal(g, w)) for g, w in zip(got, want)])
    out["max_diff"] = float(max(np.abs(g - w).max() for g, w in zip(got, want)))
    out["differing_pixels"] = float(np.mean(np.any(got[-1] != want[-1], axis=-1)))
    out["coverage"] = float((want[-1][..., 3] > 0).mean())
    out["samples"] = int(api.vnrRendererGetFrameStats(sr.r)["n_samples"])


def scenario_train(ctx, out):
    """data-parallel training: replicas start from DIFFERENT seeds and must be identical after the first call; parameters stay
    identical over the steps; the by-hand form (TrainBegin / AllReduceGradients / TrainEnd) follows the same trajectory"""
    L = lib()
    vol = syn.analytic_volume(32)
    os.environ["VNR_AMD_INIT_SEED"] = str(100 + ctx.rank)
    steps = int(os.environ.get("TEST_STEPS", "20"))
    sv = api.vnrCreateSimpleVolume(vol)
    nv = api.vnrCreateNeuralVolume(syn.model_config(**SMALL_MODEL), sv, online_macrocell_construction=False)
    before = vdist.params_checksum(nv)
    vdist.train_data_parallel(ctx, nv, steps)
    out["checksum_before"] = before
    out["checksum"] = vdist.params_checksum(nv)
    out["loss"] = api.vnrNeuralVolumeGetTrainingLoss(nv)
    out["step"] = api.vnrNeuralVolumeGetTrainingStep(nv)
    out["params"] = api.neural_get_params_fp16(nv).view(np.uint16)
    sv2 = api.vnrCreateSimpleVolume(vol)
    nv2 = api.vnrCreateNeuralVolume(syn.model_config(**SMALL_MODEL), sv2, online_macrocell_construction=False)
    vdist.train_data_parallel_by_hand(ctx, nv2, steps)
    out["checksum_by_hand"] = vdist.params_checksum(nv2)
    out["params_by_hand"] = api.neural_get_params_fp16(nv2).view(np.uint16)
    out["psnr"] = api.vnrNeuralVolumeGetPSNR(nv)


def scenario_frames_unpinned(ctx, out):
    """what a rank of the driver's 8-GPU run executes: VNR_RM_N_ITERS NOT set, a share of at most 262 144 pixels, hence 32 samples per ray
    and iteration, 4 ray parts and the packing fused into the evaluation kernel (render.hip Renderer::render / render_streaming /
    launch_tail).  The frame assembled from such shares must equal, bit for bit, the unsharded frame rendered with the batch size
    pinned to 32; and differ from the unsharded default (24) by no more than the resume rounding."""
    L = lib()
    assert "VNR_RM_N_ITERS" not in os.environ
    dims = (160, 160, 160)
    sv = api.vnrCreateSimpleVolume(syn.analytic_volume(160))
    os.environ["VNR_AMD_INIT_SEED"] = "4243"
    cfg = syn.model_config(n_levels=12, n_features=2, log2_hashmap_size=19, n_hidden_layers=3, per_level_scale=1.4)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    api.vnrNeuralVolumeTrain(nv, 150, True)
    check(L.vnrAmdNeuralVolumeSyncReplicas(nv.h))
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    cam = syn.oblique_camera(dims, distance_scale=0.8)
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])
    size = (1024, 192 * ctx.world)             # 24 tile rows = 196 608 pixels per rank at any world size
    n_frames = 4
    frames = {}
    for pin in ("32", "24"):
        os.environ["VNR_RM_N_ITERS"] = pin      # read when a renderer is created
        plain = make_renderer(nv, tfn, camera, size, 5)
        frames[pin] = []
        for _ in range(n_frames):
            api.vnrRender(plain)
            frames[pin].append(api.vnrRendererMapFrame(plain).copy())
        out["iterations_" + pin] = int(api.vnrRendererGetFrameStats(plain)["n_iterations"])
    del os.environ["VNR_RM_N_ITERS"]
    out["brick_in_use"] = bool(api.neural_brick_image(nv)["in_use"])
    sr = vdist.ShardedRenderer(ctx, make_renderer(nv, tfn, camera, size, 5), size[0], size[1])
    got = []
    for k in range(n_frames):
        f = sr.render()
        if k:
            got.append(sr.download(f))
    got.append(sr.download(sr.flush()))
    r_sync = make_renderer(nv, tfn, camera, size, 5)
    check(L.vnrAmdRendererSetDistributed(r_sync.h, 1))
    ok_sync = True
    for k in range(n_frames):
        api.vnrRender(r_sync)
        ok_sync = ok_sync and bool(np.array_equal(api.vnrRendererMapFrame(r_sync), frames["32"][k]))
    out["equal_32"] = np.array([bool(np.array_equal(g, w)) for g, w in zip(got, frames["32"])])
    out["equal_32_sync"] = ok_sync
    out["equal_24"] = np.array([bool(np.array_equal(g, w)) for g, w in zip(got, frames["24"])])
    out["max_diff_24"] = float(max(np.abs(g - w).max() for g, w in zip(got, frames["24"])))
    out["coverage"] = float((frames["32"][-1][..., 3] > 0).mean())
    out["samples"] = int(api.vnrRendererGetFrameStats(sr.r)["n_samples"])
    out["iterations"] = int(api.vnrRendererGetFrameStats(sr.r)["n_iterations"])


def scenario_sharded_optimizer(ctx, out):
    """the sharded step (reduce-scatter, Adam on the rank's 1/world of every range, all-gather of the fp16 parameters) against the
    replicated one (all-reduce, Adam on everything) on the SAME gradients: Adam is per parameter and both exchange with the same
    reduction, so the parameters must agree bit for bit, step after step; then the optimizer state is gathered (SyncReplicas) and a
    replicated step on both volumes must still agree (the gathered moments and step counts are the replicated ones)."""
    L = lib()
    import ctypes as C
    os.environ["VNR_AMD_INIT_SEED"] = "77"
    sv = api.vnrCreateSimpleVolume(syn.analytic_volume(32))
    model = dict(SMALL_MODEL, n_levels=6, log2_hashmap_size=14)
    cfg = syn.model_config(**model)
    if os.environ.get("TEST_MODEL_JSON"):      # another shape: {"model": {model_config arguments}, "encoding": {...}, "network": {...}}
        import json
        o = json.loads(os.environ["TEST_MODEL_JSON"])
        cfg = syn.model_config(**o.get("model", model))
        cfg["encoding"].update(o.get("encoding", {}))
        cfg["network"].update(o.get("network", {}))
    a = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    b = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    check(L.vnrAmdNeuralVolumeSyncReplicas(a.h))
    check(L.vnrAmdNeuralVolumeSyncReplicas(b.h))
    n = api.neural_info(a)["n_params"]
    out["n_params"] = n
    assert np.array_equal(api.neural_get_params_fp16(a).view(np.uint16), api.neural_get_params_fp16(b).view(np.uint16))

    def gradient(step):
        rng = np.random.default_rng(1000 * step + ctx.rank)
        g = (rng.normal(size=n) * 0.5).astype(np.float32)
        g[rng.random(n) < 0.6] = 0.0     # most hash-grid entries are untouched in a step: their parameters must not move
        return g

    def set_grad(v, g):
        check(L.vnrAmdNeuralVolumeSetGradients(v.h, g.ctypes.data_as(C.POINTER(C.c_float)), g.size))

    equal, moved = [], []
    before = api.neural_get_params_fp16(a).view(np.uint16).copy()
    for step in range(5):
        g = gradient(step)
        set_grad(a, g); set_grad(b, g)
        check(L.vnrAmdNeuralVolumeTrainEndDataParallel(a.h, 1, 0))
        check(L.vnrAmdNeuralVolumeTrainEndDataParallel(b.h, 1, 1))
        pa, pb = api.neural_get_params_fp16(a).view(np.uint16), api.neural_get_params_fp16(b).view(np.uint16)
        equal.append(bool(np.array_equal(pa, pb)))
        moved.append(float((pa != before).mean()))
        assert np.all(api.neural_gradients(a) == 0) and np.all(api.neural_gradients(b) == 0)   # both shapes leave a clear blob
    out["equal"] = np.array(equal)
    out["moved"] = np.array(moved)
    out["checksum"] = vdist.params_checksum(b)
    # a full step on one rank would read stale state of the other ranks' slices: refused until the state has been gathered
    set_grad(b, gradient(99))
    out["refused"] = L.vnrAmdNeuralVolumeTrainEnd(b.h, 1.0, 1) != 0 if ctx.world > 1 else True
    check(L.vnrAmdNeuralVolumeSyncReplicas(b.h))
    g = gradient(5)
    set_grad(a, g); set_grad(b, g)
    check(L.vnrAmdNeuralVolumeTrainEndDataParallel(a.h, 1, 0))
    check(L.vnrAmdNeuralVolumeTrainEndDataParallel(b.h, 1, 0))
    out["equal_after_gather"] = bool(np.array_equal(api.neural_get_params_fp16(a).view(np.uint16), api.neural_get_params_fp16(b).view(np.uint16)))
    out["step"] = api.vnrNeuralVolumeGetTrainingStep(b)


def scenario_resync(ctx, out):
    """a replica whose parameters change outside an optimizer step (SetParams on ONE rank, a re-initialised model) is found by the
    next data-parallel call (one control-plane round trip) and the replicas are synchronised again"""
    os.environ["VNR_AMD_INIT_SEED"] = str(500 + ctx.rank)
    sv = api.vnrCreateSimpleVolume(syn.analytic_volume(32))
    nv = api.vnrCreateNeuralVolume(syn.model_config(**SMALL_MODEL), sv, online_macrocell_construction=False)
    vdist.train_data_parallel(ctx, nv, 3)
    out["checksum_a"] = vdist.params_checksum(nv)
    if ctx.rank == ctx.world - 1:       # one rank loads other parameters
        p = api.neural_get_params_fp16(nv)
        api.neural_set_params_fp16(nv, (p.astype(np.float32) * 0.5).astype(np.float16))
    out["checksum_changed"] = vdist.params_checksum(nv)
    vdist.train_data_parallel(ctx, nv, 2)
    out["checksum_b"] = vdist.params_checksum(nv)
    if ctx.rank == 0:                   # one rank re-creates its model (a fresh seed)
        os.environ["VNR_AMD_INIT_SEED"] = "999"
        api.vnrNeuralVolumeSetModel(nv, syn.model_config(**SMALL_MODEL))
    vdist.train_data_parallel(ctx, nv, 2)
    out["checksum_c"] = vdist.params_checksum(nv)
    out["step"] = api.vnrNeuralVolumeGetTrainingStep(nv)


def scenario_train_c4(ctx, out):
    """the C4 model (70 M parameters, 140 MB fp16 gradient blob exchanged range by range under the backward pass) trained data-parallel:
    replicas that start from different seeds are identical after every call, and the loss falls"""
    os.environ["VNR_AMD_INIT_SEED"] = str(300 + ctx.rank)
    sv = api.vnrCreateSimpleVolumePerlin((256, 256, 256), seed=7, octaves=4, base_frequency=6.0)
    pls = float(np.exp(np.log(1024 / 16.0) / 15))
    cfg = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    out["n_params"] = api.neural_info(nv)["n_params"]
    out["checksum_before"] = vdist.params_checksum(nv)
    vdist.train_data_parallel(ctx, nv, 2)
    out["loss_first"] = api.vnrNeuralVolumeGetTrainingLoss(nv)
    out["checksum_2"] = vdist.params_checksum(nv)
    vdist.train_data_parallel(ctx, nv, 38)
    out["loss_last"] = api.vnrNeuralVolumeGetTrainingLoss(nv)
    out["checksum_40"] = vdist.params_checksum(nv)
    out["step"] = api.vnrNeuralVolumeGetTrainingStep(nv)


def scenario_macrocell(ctx, out):
    """online macrocell construction under data-parallel training: every rank ends with the min / max over ALL ranks' samples"""
    vol = syn.analytic_volume(32)
    os.environ["VNR_AMD_INIT_SEED"] = "9"
    sv = api.vnrCreateSimpleVolume(vol)
    nv = api.vnrCreateNeuralVolume(syn.model_config(**SMALL_MODEL), sv, online_macrocell_construction=True)
    vdist.train_data_parallel(ctx, nv, 3, fast_mode=False)
    mc = api.volume_macrocell(nv)
    out["value_range"] = mc["value_range"]
    out["max_opacity"] = mc["max_opacity"]


def scenario_selftest(ctx, out):
    """vnrAmdDistSelfTest (what bench.py --gpus N runs before its timed region): every collective of the sharded paths on patterned buffers"""
    ok, text = vdist.self_test(deadline_s=60.0)
    out["ok"] = ok
    out["report"] = text
    out["transport"] = ctx.transport


def scenario_ooc(ctx, out):
    """BASELINE C5 in small (and, with TEST_OOC_DIMS / TEST_OOC_MODEL=c4, at 2 GiB): out-of-core volume, every rank its own slab set,
    gradients exchanged every step"""
    path = os.environ["TEST_OOC_FILE"]
    if "TEST_OOC_DIMS" in os.environ:
        dims = tuple(int(v) for v in os.environ["TEST_OOC_DIMS"].split(","))
    else:
        dims = (int(os.environ["TEST_OOC_SIZE"]),) * 3
    ncb, nb = (int(v) for v in os.environ.get("TEST_OOC_BLOCKS", "8,64").split(","))
    os.environ["VNR_AMD_INIT_SEED"] = "11"
    sv = api.vnrCreateSimpleVolumeOutOfCore(path, dims, "uint8", (0.0, 255.0), n_concurrent_blocks=ncb, n_blocks=nb)
    out["slabs"] = np.asarray(api.out_of_core_blocks(sv))
    if os.environ.get("TEST_OOC_MODEL") == "c4":
        model = syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=float(np.exp(np.log(max(dims) / 16.0) / 15)))
    else:
        model = syn.model_config(**SMALL_MODEL)
    nv = api.vnrCreateNeuralVolume(model, sv, online_macrocell_construction=True)
    vdist.train_data_parallel(ctx, nv, 5)
    out["loss_first"] = api.vnrNeuralVolumeGetTrainingLoss(nv)
    vdist.train_data_parallel(ctx, nv, int(os.environ.get("TEST_STEPS", "300")) - 5)
    out["checksum"] = vdist.params_checksum(nv)
    out["psnr"] = -1.0 if os.environ.get("TEST_OOC_NO_PSNR") else api.vnrNeuralVolumeGetPSNR(nv)
    out["loss"] = api.vnrNeuralVolumeGetTrainingLoss(nv)
    out["value_range"] = api.volume_macrocell(nv)["value_range"]


def main():
    scenario, out_path = sys.argv[1], sys.argv[2]
    ctx = vdist.init_from_env()
    out = {"rank": ctx.rank, "world": ctx.world, "transport": ctx.transport or "none", "rccl_ranks_seen": int(api.lib().vnrAmdDistRcclRanksSeen())}
    {"frames": scenario_frames, "frames_c4": scenario_frames_c4, "frames_unpinned": scenario_frames_unpinned, "train": scenario_train,
     "train_c4": scenario_train_c4, "sharded_optimizer": scenario_sharded_optimizer, "resync": scenario_resync, "macrocell": scenario_macrocell,
     "ooc": scenario_ooc, "selftest": scenario_selftest}[scenario](ctx, out)
    vdist.barrier()
    np.savez(out_path, **out)
    vdist.finalize()


if __name__ == "__main__":
    main()
