"""GPU parity of the out-of-core training sampler (vnrCreateSimpleVolume(scene, "OUT_OF_CORE")) through the
C-ABI against the oracle's restatement of core/samplers/neural_sampler.cpp:377-668, 1043-1127.

The slab CHOICE draws from the host's std::mt19937 / uniform_int_distribution exactly as the reference does, which the C
oracle does not restate (libstdc++ internals); so the product's slot table is read back, checked for validity, and handed to
the oracle, which loads the same slabs from the same file contents and samples them with the same pcg32 numbers: coordinates
and values must be bit-identical."""
import os

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


# ------------------------------------------------------------------------------------------------ BASELINE C5 at a size the driver can see
C5_DIMS = (1024, 1024, 2048)   # x, y, z: 2 GiB of uint8, larger than the resident slab set by construction (a slab is 1024 x 32 x 1 voxels + ghosts)


@pytest.fixture(scope="module")
def c5_file(tmp_path_factory):
    """a 2 GiB uint8 volume written slab-wise (about 8 s); the same analytic field as tools/ooc_bench.py"""
    nx, ny, nz = C5_DIMS
    path = tmp_path_factory.mktemp("c5") / f"c5_{nx}x{ny}x{nz}.raw"
    x = np.linspace(0, 1, nx, dtype=np.float32)[None, None, :]
    y = np.linspace(0, 1, ny, dtype=np.float32)[None, :, None]
    with open(path, "wb") as f:
        for z0 in range(0, nz, 16):
            z = (np.arange(z0, min(z0 + 16, nz), dtype=np.float32) / nz)[:, None, None]
            v = 0.5 + 0.5 * np.sin(40 * x + 9 * z) * np.cos(31 * y) * np.sin(23 * z + 5 * x * y)
            f.write((v * 255.0 + 0.5).astype(np.uint8).tobytes())
    yield path
    try:
        path.unlink()
    except OSError:
        pass


def c5_model():
    pls = float(np.exp(np.log(max(C5_DIMS) / 16.0) / 15))
    return syn.model_config(n_levels=16, n_features=2, log2_hashmap_size=22, n_hidden_layers=3, per_level_scale=pls)


def test_c5_batches_of_the_2gib_file_are_bit_identical_to_the_oracle(oracle, c5_file):
    """the sampler on a file larger than its resident set (2 GiB, 2 048 resident slabs of 1024 x 32 x 1 voxels + ghost rows,
    128 replaced per call): for the slab set the library reports, coordinates and values of whole training batches (65 536 samples) equal
    the oracle's restatement of neural_sampler.cpp:488-668 bit for bit, across refreshes"""
    nx, ny, nz = C5_DIMS
    vol = np.memmap(c5_file, dtype=np.uint8, mode="r", shape=(nz, ny, nx))
    sv = api.vnrCreateSimpleVolumeOutOfCore(str(c5_file), C5_DIMS, np.uint8, (0.0, 255.0), n_concurrent_blocks=128, n_blocks=2048)
    info = api.out_of_core_info(sv)
    assert info["block_dims"] == (1024, 32, 1) and info["n_blocks"] == 2048 and info["file_dims"] == C5_DIMS
    offset = 0
    for step in range(3):
        blocks = api.out_of_core_blocks(sv)
        c, v = api.simple_volume_take_samples(sv, 65536)
        wc, wv, bad = oracle_batch(oracle, vol, (0.0, 255.0), blocks, 65536, offset)
        assert bad == 0 and np.array_equal(c, wc) and np.array_equal(v, wv)
        offset += 5 * 65536
    assert api.out_of_core_info(sv)["bytes_read"] >= (2048 + 2 * 128) * info["block_size_aligned"]


def test_c5_training_from_the_2gib_file_synchronous_and_asynchronous_refresh(c5_file):
    """BASELINE C5's step where the driver can see it: the C4-shaped model (L16 F2 T2^22 + 3x64, 70 M parameters) trained from the 2 GiB
    file with 16 384 resident slabs (1.6 GiB of HBM) and 1 024 slabs replaced per refresh (neural_sampler.cpp:1043-1127: the reference's
    NUM_BLOCKS = 64 x 1024 at a quarter, its refresh count per step in full).  Default (synchronous: every step waits for its refresh, the
    reference's semantics) and asynchronous refresh (opt-in: a step never waits for the storage): the loss falls, the refresh counters
    behave, and the asynchronous step takes < 1.2 ms"""
    import ctypes as C
    import time
    from instantvnr_amd._lib import check, lib
    L = lib()
    sv = api.vnrCreateSimpleVolumeOutOfCore(str(c5_file), C5_DIMS, np.uint8, (0.0, 255.0), n_concurrent_blocks=1024, n_blocks=16384)
    info = api.out_of_core_info(sv)
    nv = api.vnrCreateNeuralVolume(c5_model(), sv, online_macrocell_construction=True)
    api.vnrNeuralVolumeTrain(nv, 10, False)
    first = api.vnrNeuralVolumeGetTrainingLoss(nv)

    def leg(steps):
        check(L.vnrAmdSynchronize())
        b0 = api.out_of_core_info(sv)["bytes_read"]
        r0, y0 = C.c_uint64(), C.c_uint64()
        check(L.vnrAmdSimpleVolumeOutOfCoreRefreshStats(sv.h, C.byref(r0), C.byref(y0)))
        t0 = time.perf_counter()
        api.vnrNeuralVolumeTrain(nv, steps, False)
        check(L.vnrAmdSynchronize())
        dt = time.perf_counter() - t0
        r1, y1 = C.c_uint64(), C.c_uint64()
        check(L.vnrAmdSimpleVolumeOutOfCoreRefreshStats(sv.h, C.byref(r1), C.byref(y1)))
        return dt / steps * 1e3, (api.out_of_core_info(sv)["bytes_read"] - b0) / dt / 2**30, r1.value - r0.value, y1.value - y0.value

    ms_sync, gib_sync, ref_sync, busy_sync = leg(150)
    loss_sync = api.vnrNeuralVolumeGetTrainingLoss(nv)
    assert np.isfinite(loss_sync) and loss_sync < 0.7 * first, (first, loss_sync)
    assert ref_sync == 150 and busy_sync == 0                     # one refresh per step, none in flight while a batch is drawn
    check(L.vnrAmdSimpleVolumeOutOfCoreSetAsyncRefresh(sv.h, 1))
    api.vnrNeuralVolumeTrain(nv, 10, False)
    ms_async, gib_async, ref_async, busy_async = leg(300)
    loss_async = api.vnrNeuralVolumeGetTrainingLoss(nv)
    print(f"\\nC5 at 2 GiB: synchronous refresh {ms_sync:.3f} ms per step ({gib_sync:.1f} GiB/s turnover, {ref_sync} refreshes in 150 steps), "
          f"asynchronous {ms_async:.3f} ms per step ({gib_async:.1f} GiB/s, {ref_async} refreshes in 300 steps, {busy_async} steps beside one); "
          f"loss {first:.4f} -> {loss_sync:.4f} -> {loss_async:.4f}; slab {info['block_size_aligned']} B")
    assert np.isfinite(loss_async) and loss_async < 0.1 * first        # (the loss of ONE batch: it wanders by tens of percent from step to step)
    assert 0 < ref_async <= 300 and busy_async > 0               # refreshes go on, and steps run beside them instead of waiting
    assert ms_async < ms_sync                                      # structural: a step that never waits for the storage is the faster one
    if os.environ.get("VNR_TEST_TIMING") == "1":                  # the wall-clock bar is a perf run's: measured 0.62 (synchronous 2.81)
        assert ms_async < 1.2, ms_async


def test_c5_two_ranks_train_from_the_2gib_file(c5_file, tmp_path):
    """the sharded form (tests/dist_gpu_worker.py::scenario_ooc; both ranks on this box's one GPU, host-staged collectives): every rank
    its own slab set of the same 2 GiB file, one model afterwards, the loss falls"""
    from test_gpu_dist import run_ranks
    res = run_ranks("ooc", 2, tmp_path, timeout=600,
                    extra_env={"TEST_OOC_FILE": str(c5_file), "TEST_OOC_DIMS": "%d,%d,%d" % C5_DIMS, "TEST_STEPS": "60", "TEST_OOC_BLOCKS": "256,4096",
                               "TEST_OOC_MODEL": "c4", "TEST_OOC_NO_PSNR": "1"})
    assert not np.array_equal(res[0]["slabs"], res[1]["slabs"])
    assert int(res[0]["checksum"]) == int(res[1]["checksum"])
    assert float(res[0]["loss"]) < 0.8 * float(res[0]["loss_first"]), (float(res[0]["loss_first"]), float(res[0]["loss"]))


# ------------------------------------------------------------------------------------------------ BASELINE C5 beyond 2^32 bytes (missing 5)
# A SPARSE 4096^3 uint8 file (64 GiB apparent, a few hundred MB written): the slab choice is a default-seeded std::mt19937
# (neural_sampler.cpp:53,66-75), so a first pass over the empty file tells which slabs the sampler will hold at every point of the test,
# seeded random bytes are then written exactly there (ghost rows and slices included), and a second sampler -- same seed, same slab
# sequence -- reads real data at byte offsets up to 2^36.  94 % of a uniformly drawn slab set lies beyond 2^32 bytes, half beyond 2^35.
C5_BIG = [(4096, 4096, 4096), (4096, 4096, 512)]   # 64 GiB; 8 GiB where the temporary file system refuses the first
BIG_SLOTS, BIG_REPLACED = 512, 64


@pytest.fixture(scope="module")
def sparse_file(tmp_path_factory):
    import os
    d = tmp_path_factory.mktemp("c5big")
    for dims in C5_BIG:
        path = d / ("sparse_%dx%dx%d.raw" % dims)
        try:
            with open(path, "wb") as f:
                f.truncate(dims[0] * dims[1] * dims[2])
            if os.stat(path).st_size == dims[0] * dims[1] * dims[2]:
                yield path, dims
                break
        except OSError:
            pass
        try:
            path.unlink()
        except OSError:
            pass
    else:
        pytest.fail("the temporary directory holds neither a 64 GiB nor an 8 GiB sparse file")
    try:
        path.unlink()
    except OSError:
        pass


def _write_slabs(path, dims, slabs, seed):
    """seeded random bytes over the ghost region of every slab (y, z): rows y0 - 1 .. y1 + 1 of slices z - 1 .. z + 1, whole rows"""
    nx, ny, nz = dims
    rows = 8 if nx == 4096 else None
    assert rows, "slab height for this width"
    rng = np.random.default_rng(seed)
    written = set()
    with open(path, "r+b") as f:
        for iy, iz in sorted(set((int(a), int(b)) for a, b in slabs)):
            y0, y1 = max(iy * rows - 1, 0), min((iy + 1) * rows + 1, ny)
            for z in range(max(iz - 1, 0), min(iz + 2, nz)):
                if (y0, y1, z) in written:
                    continue
                written.add((y0, y1, z))
                f.seek((z * ny + y0) * nx)
                f.write(rng.integers(1, 256, (y1 - y0) * nx, dtype=np.uint8).tobytes())
    return len(written)


def _c5_big_sequence(oracle, path, dims, vol, check_values):
    """the calls of the test in their order; -> the slot tables at its checkpoints (and, with check_values, the parity checks themselves)"""
    sv = api.vnrCreateSimpleVolumeOutOfCore(str(path), dims, np.uint8, (0.0, 255.0), n_concurrent_blocks=BIG_REPLACED, n_blocks=BIG_SLOTS)
    info = api.out_of_core_info(sv)
    assert info["block_dims"] == (4096, 8, 1) and info["file_dims"] == tuple(dims) and info["n_blocks"] == BIG_SLOTS
    tables, offset, nonzero = [], 0, []

    def batch(n):
        nonlocal offset
        blocks = api.out_of_core_blocks(sv)
        tables.append(blocks.copy())
        c, v = api.simple_volume_take_samples(sv, n)
        if check_values:
            wc, wv, bad = oracle_batch(oracle, vol, (0.0, 255.0), blocks, n, offset)
            assert bad == 0 and np.array_equal(c, wc) and np.array_equal(v, wv)
            nonzero.append(float((v > 0).mean()))
        offset += 5 * n
        return blocks

    for _ in range(4):                                   # whole batches across four refreshes
        batch(16384)
    nv = api.vnrCreateNeuralVolume(syn.model_config(n_levels=8, n_features=2, log2_hashmap_size=15, base_resolution=8, n_hidden_layers=2), sv,
                                   online_macrocell_construction=False)
    api.vnrNeuralVolumeTrain(nv, 60, True)               # 60 steps, one synchronous refresh each (64 slabs): the table moves on by 60 x 64 slots
    loss = api.vnrNeuralVolumeGetTrainingLoss(nv)
    offset += 60 * 5 * 65536                             # (the training batches drew from the same pcg32 stream)
    batch(16384)
    return sv, nv, tables, nonzero, loss, info, offset


def test_c5_byte_offsets_beyond_4gib(oracle, sparse_file):
    """64-bit block arithmetic end to end (neural_sampler.cpp:377-668): whole batches drawn from slabs whose bytes lie between 0 and 2^36
    equal the oracle's, bit for bit, before and after 60 training steps with synchronous refresh; the asynchronous refresh (timing
    dependent: its slab sequence cannot be written ahead) is then checked on whatever table it ends with"""
    import ctypes as C
    from instantvnr_amd._lib import check, lib
    path, dims = sparse_file
    nx, ny, nz = dims
    vol = np.memmap(path, dtype=np.uint8, mode="r", shape=(nz, ny, nx))
    # pass 1 over the empty file: which slabs will be resident at the checkpoints
    sv, nv, tables, _, _, info, _ = _c5_big_sequence(oracle, path, dims, vol, check_values=False)
    del nv, sv
    slabs = np.concatenate(tables)
    n_regions = _write_slabs(path, dims, slabs, seed=99)
    off = (slabs[:, 1].astype(np.int64) * ny + slabs[:, 0].astype(np.int64) * 8) * nx
    assert (off >= 2**32).mean() > 0.4 and off.max() >= min(2**35, nx * ny * nz // 2)
    # pass 2: the same calls on the written file
    sv, nv, tables2, nonzero, loss, _, offset = _c5_big_sequence(oracle, path, dims, vol, check_values=True)
    assert all(np.array_equal(a, b) for a, b in zip(tables, tables2)), "the slab sequence is not reproducible"
    assert min(nonzero) > 0.99, nonzero                                  # the batches did read the written bytes (values 1 .. 255), not holes
    assert np.isfinite(loss)
    # asynchronous refresh on the same file: 60 more steps, then one more batch against the oracle on the table the sampler then holds
    check(lib().vnrAmdSimpleVolumeOutOfCoreSetAsyncRefresh(sv.h, 1))
    r0, y0 = C.c_uint64(), C.c_uint64()
    check(lib().vnrAmdSimpleVolumeOutOfCoreRefreshStats(sv.h, C.byref(r0), C.byref(y0)))
    api.vnrNeuralVolumeTrain(nv, 60, True)
    check(lib().vnrAmdSynchronize())
    r1, y1 = C.c_uint64(), C.c_uint64()
    check(lib().vnrAmdSimpleVolumeOutOfCoreRefreshStats(sv.h, C.byref(r1), C.byref(y1)))
    assert r1.value > r0.value and np.isfinite(api.vnrNeuralVolumeGetTrainingLoss(nv))
    check(lib().vnrAmdSimpleVolumeOutOfCoreSetAsyncRefresh(sv.h, 0))
    blocks = api.out_of_core_blocks(sv)
    c, v = api.simple_volume_take_samples(sv, 8192)
    state = api.out_of_core_info(sv)
    # a step draws its 5 x 65 536 numbers whether or not a refresh is in flight, so the stream position is known; the table is whatever the
    # asynchronous refreshes left (mostly holes of the sparse file by now: zeros, at 64-bit offsets all the same)
    wc, wv, bad = oracle_batch(oracle, vol, (0.0, 255.0), blocks, 8192, offset + 60 * 5 * 65536)
    assert bad == 0 and np.array_equal(c, wc) and np.array_equal(v, wv)
    print(f"\nC5 sparse {nx}x{ny}x{nz}: {n_regions} slab regions written, offsets up to 2^{np.log2(off.max()):.1f}, {(off >= 2**32).mean() * 100:.0f} % of the slabs "
          f"beyond 2^32 bytes; nonzero samples {min(nonzero):.3f}; loss {loss:.4f}; asynchronous refreshes {r1.value - r0.value}; bytes read {state['bytes_read']}")
