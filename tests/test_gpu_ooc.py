"""GPU parity of the out-of-core training sampler (vnrCreateSimpleVolume(scene, "OUT_OF_CORE")) through the
C-ABI against the oracle's restatement of core/samplers/neural_sampler.cpp:377-668, 1043-1127.

The slab CHOICE draws from the host's std::mt19937 / uniform_int_distribution exactly as the reference does, which the C
oracle does not restate (libstdc++ internals); so the product's slot table is read back, checked for validity, and handed to
the oracle, which loads the same slabs from the same file contents and samples them with the same pcg32 numbers: coordinates
and values must be bit-identical."""
import numpy as np
import pytest

from instantvnr_amd import api
from instantvnr_amd import synthetic as syn

pytestmark = pytest.mark.gpu

SEED, STREAM = 1337, 0xda3e39cb94b95bdb  # neural_sampler.cu:36 / tcnn default sequence


def make_volume(tmp_path, shape, dtype, seed=0, header=0):
    rng = np.random.default_rng(seed)
    if np.issubdtype(dtype, np.integer):
        info = np.iinfo(dtype)
        vol = rng.integers(info.min, info.max, shape, dtype=dtype)
        vr = (float(info.min) + 2.0, float(info.max) - 7.0)
    else:
        vol = rng.normal(0, 1, shape).astype(dtype)
        vr = (-1.25, 1.75)
    path = tmp_path / f"vol_{np.dtype(dtype).name}.raw"
    with open(path, "wb") as f:
        f.write(b"\xab" * header)
        f.write(vol.tobytes())
    return vol, vr, path


def oracle_batch(oracle, vol, vr, blocks, n, offset, lower=(0, 0, 0), upper=(1, 1, 1)):
    r = oracle.pcg32_floats(5 * n, offset, SEED, STREAM)
    s = oracle.OocSlabSet(vol, blocks)
    return s.sample(vr, r[:3 * n].reshape(n, 3), r[3 * n:4 * n], r[4 * n:], lower, upper)


@pytest.mark.parametrize("dtype,shape", [(np.uint8, (7, 300, 520)), (np.int16, (5, 70, 300)), (np.float32, (4, 50, 200)),
                                         (np.float64, (3, 20, 90)), (np.uint16, (1, 9, 40))])
def test_batches_are_bit_identical_to_the_oracle(oracle, tmp_path, dtype, shape):
    vol, vr, path = make_volume(tmp_path, shape, dtype, seed=1, header=24)
    dims = shape[::-1]
    sv = api.vnrCreateSimpleVolumeOutOfCore(path, dims, dtype, vr, offset=24, n_concurrent_blocks=8, n_blocks=40)
    info = api.out_of_core_info(sv)
    g = oracle.ooc_geometry(dims, dtype)
    assert info["block_dims"] == tuple(g.block_dims) and info["block_index_space"] == tuple(g.index_space)
    assert info["block_size_aligned"] == g.block_size_aligned and info["n_blocks"] == 40 and info["n_concurrent_blocks"] == 8
    assert api.vnrVolumeGetDims(sv) == tuple(min(1024, d) for d in dims)
    offset, seen = 0, []
    for step, n in enumerate([4096, 1000, 1, 2500]):   # ragged batches; the slot table changes between calls
        blocks = api.out_of_core_blocks(sv)
        assert blocks[:, 0].min() >= 0 and blocks[:, 0].max() < g.index_space[1]
        assert blocks[:, 1].min() >= 0 and blocks[:, 1].max() < g.index_space[2]
        seen.append(blocks.copy())
        lower, upper = ((0, 0, 0), (1, 1, 1)) if step != 1 else ((0.1, 0.2, 0.0), (0.9, 0.7, 0.5))
        c, v = api.simple_volume_take_samples(sv, n, lower, upper)
        wc, wv, bad = oracle_batch(oracle, vol, vr, blocks, n, offset, lower, upper)
        assert bad == 0
        assert np.array_equal(c, wc)
        assert np.array_equal(v, wv)
        offset += 5 * n
    # every call replaced n_concurrent_blocks consecutive slots (wrapping) and nothing else
    for a, b in zip(seen[:-1], seen[1:]):
        changed = np.flatnonzero((a != b).any(axis=1))
        assert len(changed) <= 8
        if len(changed):
            span = [(changed - s) % 40 for s in range(40)]
            assert min(int(x.max()) for x in span) < 8
    assert api.out_of_core_info(sv)["bytes_read"] > 0


def test_grid_samples_and_missing_ground_truth(oracle, tmp_path):
    vol, vr, path = make_volume(tmp_path, (9, 40, 130), np.uint8, seed=2)
    sv = api.vnrCreateSimpleVolumeOutOfCore(path, (130, 40, 9), np.uint8, vr, n_concurrent_blocks=4, n_blocks=16)
    c, v = api.simple_volume_take_samples_grid(sv, (3, 5, 1), (120, 30, 7))
    want = oracle.ooc_sample_grid(vol, vr, (3, 5, 1), (120, 30, 7), (1 / 130, 1 / 40, 1 / 9))
    assert np.array_equal(v, want)
    assert np.array_equal(c, oracle.grid_coords((3, 5, 1), (120, 30, 7), (1 / 130, 1 / 40, 1 / 9)))
    # no resident ground truth: no point lookups, no macrocell, no rendering of the volume itself
    with pytest.raises(api.VnrAmdError, match="no resident ground truth"):
        api.simple_volume_sample(sv, np.zeros((4, 3), np.float32), nodal=False)
    ren = api.vnrCreateRenderer(sv)
    api.vnrRendererSetFramebufferSize(ren, (16, 16))
    with pytest.raises(api.VnrAmdError, match="macrocell"):
        api.vnrRender(ren)


def test_errors(tmp_path):
    vol, vr, path = make_volume(tmp_path, (4, 30, 100), np.float32)
    with pytest.raises(api.VnrAmdError, match="cannot open"):
        api.vnrCreateSimpleVolumeOutOfCore(tmp_path / "missing.raw", (100, 30, 4), np.float32, vr, n_concurrent_blocks=2, n_blocks=4)
    with pytest.raises(api.VnrAmdError, match="too short"):
        api.vnrCreateSimpleVolumeOutOfCore(path, (100, 30, 5), np.float32, vr, n_concurrent_blocks=2, n_blocks=4)
    sv = api.vnrCreateSimpleVolumeOutOfCore(path, (100, 30, 4), np.float32, (1.0, 0.0), n_concurrent_blocks=2, n_blocks=4)
    with pytest.raises(api.VnrAmdError, match="valid value range"):   # neural_sampler.cpp:1069-1071
        api.simple_volume_take_samples(sv, 16)


def test_training_from_an_out_of_core_volume(oracle, tmp_path):
    """vnr_cmd_train --training-mode OUT_OF_CORE in miniature: the network only ever sees slab samples, the macrocell is built
    online from the training batches (no ground-truth texture), PSNR streams the reference from the file"""
    vol = (syn.analytic_volume(64) * 255.0 + 0.5).astype(np.uint8)  # [z, y, x]
    path = tmp_path / "analytic_u8.raw"
    vol.tofile(path)
    # 64-B rows: a slab is the whole 64-row slice; 48 of the 64 possible slabs are resident, 16 replaced per step
    sv = api.vnrCreateSimpleVolumeOutOfCore(path, (64, 64, 64), np.uint8, (0.0, 255.0), n_concurrent_blocks=16, n_blocks=48)
    cfg = syn.model_config(n_levels=6, n_features=4, log2_hashmap_size=14, base_resolution=4, n_hidden_layers=2)
    nv = api.vnrCreateNeuralVolume(cfg, sv, online_macrocell_construction=False)  # no ground-truth macrocell: falls back to online
    api.vnrNeuralVolumeTrain(nv, 400, False)
    assert api.vnrNeuralVolumeGetTrainingStep(nv) == 400
    psnr = api.vnrNeuralVolumeGetPSNR(nv)
    assert psnr > 30.0, psnr
    mc = api.volume_macrocell(nv)
    vr = mc["value_range"].reshape(-1, 2)
    # online macrocell: cells that received samples hold (min - 1, max + 1) of what was seen (macrocell.cu:35-39)
    touched = vr[:, 1] > vr[:, 0]
    assert touched.mean() > 0.9
    lo, hi = vr[touched, 0] + 1.0, vr[touched, 1] - 1.0
    assert lo.min() >= 0.0 and hi.max() <= 1.0 and (hi >= lo).all()
    assert api.out_of_core_info(sv)["bytes_read"] >= 400 * 16 * 64 * 64


def test_asynchronous_refresh_never_samples_a_slab_that_is_being_replaced(oracle, tmp_path):
    """vnrAmdSimpleVolumeOutOfCoreSetAsyncRefresh: steps do not wait for the storage; while a refresh is in flight the batch comes from the
    slabs that are not being replaced.  Whatever the timing, every sample must be the file's own trilinear value at its coordinate (a
    slab read while it is half overwritten would not be), and the refreshes must go on (the slab set keeps changing)"""
    import ctypes as C
    from instantvnr_amd._lib import check, lib
    shape = (24, 200, 256)
    rng = np.random.default_rng(5)
    vol = rng.integers(0, 255, shape, dtype=np.uint8)
    path = tmp_path / "async.raw"
    vol.tofile(path)
    dims = shape[::-1]
    sv = api.vnrCreateSimpleVolumeOutOfCore(str(path), dims, np.uint8, (0.0, 255.0), n_concurrent_blocks=16, n_blocks=96)
    check(lib().vnrAmdSimpleVolumeOutOfCoreSetAsyncRefresh(sv.h, 1))
    norm = np.clip(vol.astype(np.float32) / np.float32(255.0), 0, 1)
    first = np.asarray(api.out_of_core_blocks(sv)).copy()
    worst = 0.0
    for step in range(300):
        c, v = api.simple_volume_take_samples(sv, 4096)
        # trilinear_vkl at clamp(p, 0.5, dims - 0.5) over the normalised voxels (neural_sampler.cpp:302-329), in double precision here
        p = c.astype(np.float64) * np.array(dims, np.float64)
        q = np.clip(p, 0.5, np.array(dims, np.float64) - 0.5) - 0.5
        i0 = np.floor(q).astype(np.int64)
        w = q - i0
        i0 = np.minimum(i0, np.array(dims) - 1)
        i1 = np.minimum(i0 + 1, np.array(dims) - 1)
        acc = np.zeros(len(c))
        for dz, wz in ((0, 1 - w[:, 2]), (1, w[:, 2])):
            for dy, wy in ((0, 1 - w[:, 1]), (1, w[:, 1])):
                for dx, wx in ((0, 1 - w[:, 0]), (1, w[:, 0])):
                    zi = (i1 if dz else i0)[:, 2]; yi = (i1 if dy else i0)[:, 1]; xi = (i1 if dx else i0)[:, 0]
                    acc += wz * wy * wx * norm[zi, yi, xi]
        worst = max(worst, float(np.abs(acc - v).max()))
    assert worst < 2e-5, worst
    n_ref, n_busy = C.c_uint64(), C.c_uint64()
    check(lib().vnrAmdSimpleVolumeOutOfCoreRefreshStats(sv.h, C.byref(n_ref), C.byref(n_busy)))
    assert n_ref.value > 96 // 16 + 5                              # beyond the pre-load: refreshes keep coming
    assert not np.array_equal(np.asarray(api.out_of_core_blocks(sv)), first)
    print(f"asynchronous refresh: {n_ref.value} refreshes, {n_busy.value} of 300 steps ran beside one, max |value - file| {worst:.2e}")
