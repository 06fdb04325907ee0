"""GPU parity of decoding: vnrNeuralVolumeDecodeProgressive / DecodeInference / DecodeReference (api.h:137-140,
network.cu:290-405) and the decoding rendering modes 4 and 7 on a neural volume, which march the decoded dense volume."""
import numpy as np
import pytest

from instantvnr_amd import api
from instantvnr_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL_NET = 2.0 ** -8   # per-sample network tolerance, as in test_gpu_network.py


def neural_volume(oracle, sv, seed=31):
    L, F, log2T, base, H = 8, 4, 14, 4, 2
    cfg = syn.model_config(n_levels=L, n_features=F, log2_hashmap_size=log2T, base_resolution=base, n_hidden_layers=H)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)
    info = api.neural_info(nv)
    ocfg = oracle.grid_config(L, F, log2T, base)
    params = syn.random_params(info["n_params"], oracle.mlp_n_params(info["padded_width"], 64, H - 1), seed=seed)
    api.neural_set_params_fp16(nv, params)
    return nv, (lambda c: oracle.network_inference(ocfg, 64, H, params.view(np.uint16), c))


def test_progressive_decode_blob_by_blob_and_decoding_modes(oracle):
    n = 48
    vol = syn.analytic_volume(n)
    sv = api.vnrCreateSimpleVolume(vol)
    nv, net = neural_volume(oracle, sv)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = api.vnrCreateTransferFunction()
    api.vnrTransferFunctionSetColor(tfn, colors)
    api.vnrTransferFunctionSetAlpha(tfn, alphas)
    api.vnrTransferFunctionSetValueRange(tfn, (0, 1))
    cam = syn.oblique_camera((n, n, n))
    camera = api.vnrCreateCamera()
    api.vnrCameraSet(camera, cam["from"], cam["at"], cam["up"], cam["fovy"])

    def renderer(mode):
        r = api.vnrCreateRenderer(nv)
        api.vnrRendererSetTransferFunction(r, tfn)
        api.vnrRendererSetCamera(r, camera)
        api.vnrRendererSetFramebufferSize(r, (96, 80))
        api.vnrRendererSetMode(r, mode)
        return r

    # before any decode the decoding modes have nothing to march: fail with an explanation
    assert api.neural_decoded_volume(nv, (n, n, n)) is None
    with pytest.raises(api.VnrAmdError, match="DecodeProgressive"):
        api.vnrRender(renderer(4))

    # generate_coords (network.cu:51-68): voxel centres, x fastest
    want = net(oracle.grid_coords((0, 0, 0), (n, n, n), (1.0 / n,) * 3)).reshape(n, n, n)
    assert api.vnrNeuralVolumeGetNumberOfBlobs(nv) == 3            # 16 z-slices per blob
    api.vnrNeuralVolumeDecodeProgressive(nv)
    dec = api.neural_decoded_volume(nv, (n, n, n))
    assert np.abs(dec[:16] - want[:16]).max() <= TOL_NET and np.all(dec[16:] == 0)      # one blob decoded, the rest still zero
    api.vnrNeuralVolumeDecodeProgressive(nv)
    api.vnrNeuralVolumeDecodeProgressive(nv)
    dec = api.neural_decoded_volume(nv, (n, n, n))
    assert np.abs(dec - want).max() <= TOL_NET
    assert dec.std() > 0.05
    api.vnrNeuralVolumeDecodeProgressive(nv)                          # wraps around to blob 0: same values again
    assert np.array_equal(api.neural_decoded_volume(nv, (n, n, n)), dec)

    # modes 4 / 7 march exactly this decoded array: compare with the oracle's marcher on the same data (tight), so that the
    # network tolerance is not mixed into the renderer's
    r4 = renderer(4)
    api.vnrRender(r4)
    img4 = api.vnrRendererMapFrame(r4).copy()
    r7 = renderer(7)
    api.vnrRender(r7)
    img7 = api.vnrRendererMapFrame(r7).copy()
    otfn = oracle.TfnHolder(colors, alphas)
    mo = api.volume_macrocell(nv)["max_opacity"]
    sc4 = oracle.SceneHolder(96, 80, (n, n, n), otfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"])
    sc7 = oracle.SceneHolder(96, 80, (n, n, n), otfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"], shading_mode=1)
    want4, _ = oracle.render_monolithic(sc4, dec)
    want7, _ = oracle.render_monolithic(sc7, dec)
    assert (want4[..., 3] > 0).mean() > 0.05
    assert np.abs(img4 - want4).max() < 2e-5, np.abs(img4 - want4).max()
    assert np.abs(img7 - want7).max() < 2e-5, np.abs(img7 - want7).max()
    assert np.abs(want7[..., :3] - want4[..., :3]).mean() > 1e-3      # and the two modes do differ


def test_decode_to_files_with_padded_slices(oracle, tmp_path):
    """50 x 37 slices = 1850 values, padded to 2048 per slice: DecodeInference fills the padding with the network's values
    at the coordinates generate_coords yields for the overflow indices (the start of slice z + 1); DecodeReference writes
    the normalised reference at the voxel centres (exact) and, here, zeros as padding."""
    nx, ny, nz = 50, 37, 21
    z, y, x = np.meshgrid(np.linspace(0, 1, nz), np.linspace(0, 1, ny), np.linspace(0, 1, nx), indexing="ij")
    raw = (0.5 + 0.5 * np.sin(5.0 * x + 1.0) * np.cos(3.0 * y) * np.sin(2.0 * z + 0.5)).astype(np.float32)
    sv = api.vnrCreateSimpleVolume(raw)
    lo, hi = np.float32(raw.min()), np.float32(raw.max())
    ref = np.clip((raw - lo) / (hi - lo), np.float32(0), np.float32(1)).astype(np.float32)   # convert_volume, neural_sampler.cpp:176-210
    nv, net = neural_volume(oracle, sv, seed=32)
    xy, count = nx * ny, 2048
    assert api.vnrNeuralVolumeGetNumberOfBlobs(nv) == 2

    f_inf = str(tmp_path / "inference.bin")
    api.vnrNeuralVolumeDecodeInference(nv, f_inf)
    got = np.fromfile(f_inf, dtype=np.float32)
    assert got.size == count * nz
    got = got.reshape(nz, count)
    # slice z of the file = the network at `count` consecutive grid indices starting at slice z (generate_coords with
    # size (nx, ny, 1): x = i % nx, y = (i % (nx ny)) / nx, z = z0 + i / (nx ny))
    i = np.arange(count)
    for zz in (0, 7, nz - 1):
        c = np.stack([((i % nx) + 0.5) / nx, (((i % xy) // nx) + 0.5) / ny, ((zz + i // xy) + 0.5) / nz], axis=1).astype(np.float32)
        assert np.abs(got[zz] - net(c)).max() <= TOL_NET, zz
    assert np.abs(got[3, xy:] - got[4, :count - xy]).max() <= 2 * TOL_NET    # padding of slice 3 = start of slice 4 (same coordinates)

    f_ref = str(tmp_path / "reference_volume.bin")
    api.vnrNeuralVolumeDecodeReference(nv, f_ref)
    got = np.fromfile(f_ref, dtype=np.float32).reshape(nz, count)
    file_ref = got[:, :xy].reshape(nz, ny, nx)
    # take_samples_grid (neural_sampler.cu:166-198) = the cell-centred trilinear lookup at (i + 0.5) * (1 / n): bit-exact against
    # the oracle's restatement of it; those coordinates are voxel centres only up to fp32 rounding, so against the voxels
    # themselves the dump agrees to rounding, not bit for bit
    gc = oracle.grid_coords((0, 0, 0), (nx, ny, nz), (1.0 / nx, 1.0 / ny, 1.0 / nz))
    assert np.array_equal(file_ref, oracle.sample_volume(ref, gc, nodal=False).reshape(nz, ny, nx))
    assert np.abs(file_ref - ref).max() < 1e-5, np.abs(file_ref - ref).max()
    assert np.all(got[:, xy:] == 0)


def test_ssim_matches_numpy_restatement_and_rises_with_training(oracle):
    """vnrNeuralVolumeGetSSIM (api.h:130, network.cu:474-549).  The library's value against oracle.mssim on the SAME data: the
    volume the library decodes at the voxel centres and the reference sampled there, so only the SSIM arithmetic and the
    halo-block bookkeeping are compared (fp32 window sums vs float64: stated tolerance 2e-4).  Non-cubic volume, several
    blocks in y and z."""
    nx, ny, nz = 50, 37, 21
    z, y, x = np.meshgrid(np.linspace(0, 1, nz), np.linspace(0, 1, ny), np.linspace(0, 1, nx), indexing="ij")
    raw = (0.5 + 0.5 * np.sin(5.0 * x + 1.0) * np.cos(3.0 * y) * np.sin(2.0 * z + 0.5)).astype(np.float32)
    sv = api.vnrCreateSimpleVolume(raw)
    lo, hi = np.float32(raw.min()), np.float32(raw.max())
    ref = np.clip((raw - lo) / (hi - lo), np.float32(0), np.float32(1)).astype(np.float32)
    gc = oracle.grid_coords((0, 0, 0), (nx, ny, nz), (1.0 / nx, 1.0 / ny, 1.0 / nz))
    ref_at_centres = oracle.sample_volume(ref, gc, nodal=False).reshape(nz, ny, nx)

    def decoded(nv):
        for _ in range(api.vnrNeuralVolumeGetNumberOfBlobs(nv)):
            api.vnrNeuralVolumeDecodeProgressive(nv)
        return api.neural_decoded_volume(nv, (nx, ny, nz))

    # (1) random parameters: a low but well-defined SSIM
    nv, _ = neural_volume(oracle, sv, seed=33)
    got = api.vnrNeuralVolumeGetSSIM(nv)
    want = oracle.mssim(decoded(nv), ref_at_centres)
    assert abs(got - want) < 2e-4, (got, want)
    assert -1.0 < got < 0.5

    # (2) a trained network: much higher, and still the same number as the restatement
    import os
    os.environ["VNR_AMD_INIT_SEED"] = "5"
    cfg = syn.model_config(n_levels=6, n_features=4, log2_hashmap_size=14, base_resolution=4, n_hidden_layers=2)
    tv = api.vnrCreateNeuralVolume(cfg, sv)
    before = api.vnrNeuralVolumeGetSSIM(tv)
    api.vnrNeuralVolumeTrain(tv, 400, True)
    after = api.vnrNeuralVolumeGetSSIM(tv)
    assert abs(after - oracle.mssim(decoded(tv), ref_at_centres)) < 2e-4
    assert after > 0.9 and after > before + 0.3, (before, after)
