"""CPU checks of the oracle's restatement of the out-of-core training sampler (core/samplers/neural_sampler.cpp:377-668,
1043-1127): slab geometry against hand-computed cases, the slab copies against numpy slicing, and the sampled values against
an independent route to the same number (normalise the whole volume, then the cell-centred trilinear lookup of the oracle's
GPU-sampler restatement, which is a different function with a different arithmetic order)."""
import numpy as np
import pytest


def test_geometry_hand_cases(oracle):
    # 4096^3 uint8 (BASELINE C5): rows of 4096 B -> 8 rows per 32 KiB slab, one slice; ghosts +2 rows, +2 slices
    g = oracle.ooc_geometry((4096, 4096, 4096), np.uint8)
    assert tuple(g.block_dims) == (4096, 8, 1) and tuple(g.ghost_dims) == (4096, 10, 3)
    assert tuple(g.index_space) == (1, 512, 4096) and g.block_size_aligned == 4096 * 10 * 3
    # float32, 1000 x 300 x 7: ceil(32768 / 4000) = 9 rows; 1000 * 11 * 3 * 4 = 132000 -> next multiple of 512
    g = oracle.ooc_geometry((1000, 300, 7), np.float32)
    assert tuple(g.block_dims) == (1000, 9, 1) and tuple(g.ghost_dims) == (1000, 11, 3)
    assert tuple(g.index_space) == (1, 34, 7) and g.block_size_aligned == 132096
    # a volume smaller than one stream: the slab is the whole slice, ghosts are capped by the volume
    g = oracle.ooc_geometry((16, 8, 2), np.uint16)
    assert tuple(g.block_dims) == (16, 8, 1) and tuple(g.ghost_dims) == (16, 8, 2) and tuple(g.index_space) == (1, 1, 2)
    with pytest.raises(KeyError):
        oracle.ooc_geometry((8, 8, 8), np.int64)


def test_slab_copies_match_numpy_slices(oracle):
    rng = np.random.default_rng(0)
    vol = rng.integers(0, 65535, (5, 70, 300), dtype=np.uint16)  # [z, y, x]; 600-B rows -> 55 rows per slab
    g = oracle.ooc_geometry((300, 70, 5), np.uint16)
    assert tuple(g.block_dims) == (300, 55, 1) and tuple(g.index_space) == (1, 2, 5)
    idx = [(0, 0), (1, 0), (0, 2), (1, 4), (1, 3)]
    s = oracle.OocSlabSet(vol, idx)
    for i, (by, bz) in enumerate(idx):
        b = s.blocks[i]
        y0, y1 = by * 55, min(by * 55 + 55, 70)
        assert (tuple(b.bounds_lo), tuple(b.bounds_hi)) == ((0, y0, bz), (300, y1, bz + 1))
        gy0, gy1, gz0, gz1 = max(y0 - 1, 0), min(y1 + 1, 70), max(bz - 1, 0), min(bz + 2, 5)
        assert (tuple(b.ghost_lo), tuple(b.ghost_hi)) == ((0, gy0, gz0), (300, gy1, gz1))
        assert b.offset == y0 * 300 + bz * 70 * 300 and b.length == 300 * (y1 - y0)
        want = vol[gz0:gz1, gy0:gy1, :].reshape(-1)
        got = s.data[i * g.block_size_aligned:i * g.block_size_aligned + want.size * 2].view(np.uint16)
        assert np.array_equal(got, want)
    with pytest.raises(ValueError):
        oracle.OocSlabSet(vol, [(2, 0)])


@pytest.mark.parametrize("dtype", [np.uint8, np.int16, np.float32, np.float64])
def test_sample_agrees_with_an_independent_trilinear_lookup(oracle, dtype):
    rng = np.random.default_rng(1)
    shape = (6, 40, 260)  # [z, y, x]
    if np.issubdtype(dtype, np.integer):
        info = np.iinfo(dtype)
        vol = rng.integers(info.min, info.max, shape, dtype=dtype)
        vr = (float(info.min) + 3.0, float(info.max) - 5.0)   # clamping on both ends is exercised
    else:
        vol = rng.normal(0, 1, shape).astype(dtype)
        vr = (-1.5, 2.0)
    g = oracle.ooc_geometry(shape[::-1], dtype)
    n_slots = 24
    idx = np.stack([rng.integers(0, g.index_space[1], n_slots), rng.integers(0, g.index_space[2], n_slots)], axis=1)
    s = oracle.OocSlabSet(vol, idx)
    n = 3000
    rc, rb, rv = rng.random((n, 3), np.float32), rng.random(n, np.float32), rng.random(n, np.float32)
    coords, values, bad = s.sample(vr, rc, rb, rv)
    assert bad == 0 and coords.min() >= 0.0 and coords.max() <= 1.0
    # every sample lies inside the slab proper of the slot it picked
    slot = (rb * np.float32(n_slots)).astype(np.int64)
    vox = np.floor(coords.astype(np.float64) * np.array(shape[::-1])).astype(np.int64)
    by = np.minimum(vox[:, 1] // g.block_dims[1], g.index_space[1] - 1)
    inside = (by == idx[slot, 0]) & (vox[:, 2] == idx[slot, 1])
    assert inside.mean() > 0.999  # float rounding of p / dims can move a point on a voxel boundary
    # independent route: normalise the volume first, cell-centred trilinear lookup with clamp addressing
    norm = np.clip((vol.astype(np.float32) - np.float32(vr[0])) * (np.float32(1.0) / np.float32(vr[1] - vr[0])), 0, 1).astype(np.float32)
    want = oracle.sample_volume(norm, coords, nodal=False)
    assert np.abs(values - want).max() < 2e-5
    # lower / upper remap the coordinates only
    c2, v2, _ = s.sample(vr, rc, rb, rv, lower=(0.25, 0.0, 0.5), upper=(0.75, 1.0, 1.0))
    assert np.array_equal(v2, values)
    assert np.allclose(c2, coords * np.array([0.5, 1.0, 0.5], np.float32) + np.array([0.25, 0.0, 0.5], np.float32), atol=1e-6)


def test_largest_random_float_never_leaves_its_range(oracle):
    # the reference throws "[aio] invalid block index" when u64(r * N) == N.  r < 1 is a multiple of 2^-24 at most
    # 1 - 2^-24, and r * N then rounds to at most N - ulp, for every N: the throw (a clamp here, counted) is unreachable
    vol = np.arange(2 * 40 * 260, dtype=np.float32).reshape(2, 40, 260)
    r = np.array([np.nextafter(np.float32(1), np.float32(0))], np.float32)
    for n_slots in (1, 3, 5, 7, 24, 63, 100):
        s = oracle.OocSlabSet(vol, [(0, i % 2) for i in range(n_slots)])
        _, _, bad = s.sample((0.0, 1.0), np.zeros((1, 3), np.float32), r, r)
        assert bad == 0


def test_sample_grid_is_the_normalised_voxel(oracle):
    rng = np.random.default_rng(3)
    vol = rng.integers(0, 255, (9, 20, 33), dtype=np.uint8)
    vr = (10.0, 200.0)
    got = oracle.ooc_sample_grid(vol, vr, (2, 3, 1), (30, 11, 7), (1 / 33, 1 / 20, 1 / 9))
    want = np.clip((vol[1:8, 3:14, 2:32].astype(np.float32) - np.float32(10)) * (np.float32(1) / np.float32(190)), 0, 1)
    assert np.array_equal(got.reshape(7, 11, 30), want.astype(np.float32))
