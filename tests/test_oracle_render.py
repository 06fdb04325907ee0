"""Oracle known-answer / property tests for the march path: trilinear sampling, TFN,
macrocells, DDA traversal (SURVEY §8c vi), and the streaming-vs-monolithic marchers."""
import numpy as np
import pytest

from instantvnr_amd import synthetic as syn


def test_tex3d_voxel_centres_and_clamp(oracle):
    rng = np.random.default_rng(0)
    vol = rng.uniform(0, 1, (5, 6, 7)).astype(np.float32)  # z,y,x
    dz, dy, dx = vol.shape
    zz, yy, xx = np.meshgrid(np.arange(dz), np.arange(dy), np.arange(dx), indexing="ij")
    c = np.stack([(xx + 0.5) / dx, (yy + 0.5) / dy, (zz + 0.5) / dz], -1).reshape(-1, 3).astype(np.float32)
    v = oracle.sample_volume(vol, c, nodal=False)
    assert np.allclose(v, vol.ravel(), atol=1e-6)          # cell-centred: exact at voxel centres
    # nodal (renderer's sampleVolume): p = i/(N-1) hits voxel i exactly
    c = np.stack([xx / (dx - 1), yy / (dy - 1), zz / (dz - 1)], -1).reshape(-1, 3).astype(np.float32)
    v = oracle.sample_volume(vol, c, nodal=True)
    assert np.allclose(v, vol.ravel(), atol=2e-6)
    # clamp addressing outside [0,1]
    v = oracle.sample_volume(vol, np.array([[-1, -1, -1], [2, 2, 2]], np.float32), nodal=False)
    assert v[0] == vol[0, 0, 0] and v[1] == vol[-1, -1, -1]
    # midpoint between two voxels along x
    p = np.array([[(1.0) / dx, 0.5 / dy, 0.5 / dz]], np.float32)
    v = oracle.sample_volume(vol, p, nodal=False)
    assert np.isclose(v[0], 0.5 * (vol[0, 0, 0] + vol[0, 0, 1]), atol=1e-6)


def test_tfn_nodal_lookup(oracle):
    colors = np.array([[0, 0, 0], [1, 0.5, 0.25], [0.5, 1, 0]], np.float32)
    alphas = np.array([0.0, 1.0, 0.5, 0.25, 0.0], np.float32)
    tfn = oracle.TfnHolder(colors, alphas, 0.0, 1.0)
    out = oracle.tfn_sample(tfn, [0.0, 0.5, 1.0, 0.25, -3.0, 7.0, 0.125])
    assert np.allclose(out[0], [0, 0, 0, 0], atol=1e-6)
    assert np.allclose(out[1], [1, 0.5, 0.25, 0.5], atol=1e-6)        # node 1 of 3 colours; node 2 of 5 alphas
    assert np.allclose(out[2], [0.5, 1, 0, 0], atol=1e-6)
    assert np.allclose(out[3], [0.5, 0.25, 0.125, 1.0], atol=1e-6)    # halfway colour 0->1; alpha node 1
    assert np.allclose(out[4], out[0]) and np.allclose(out[5], out[2])  # clamped to the range
    assert np.isclose(out[6, 3], 0.5, atol=1e-6)                       # halfway between alpha nodes 0 and 1
    # value range remap
    tfn2 = oracle.TfnHolder(colors, alphas, 0.25, 0.75)
    assert np.allclose(oracle.tfn_sample(tfn2, [0.5])[0], out[1], atol=1e-6)


def test_macrocell_ranges_and_max_opacity(oracle):
    vol = syn.analytic_volume(40)   # 40^3 -> 3^3 macrocells, last one partial
    vr = oracle.macrocell_compute_implicit(vol)
    assert vr.shape == (3, 3, 3, 2)
    lo, hi = vr[..., 0] + 1.0, vr[..., 1] - 1.0
    # brute force: cell c covers voxels [16c-1, 16c+16] (one voxel apron from the +-1 neighbour rule)
    for cz in range(3):
        for cy in range(3):
            for cx in range(3):
                sl = [slice(max(16 * c - 1, 0), min(16 * c + 17, 40)) for c in (cz, cy, cx)]
                blk = vol[sl[0], sl[1], sl[2]]
                assert np.isclose(lo[cz, cy, cx], blk.min(), atol=2e-6)
                assert np.isclose(hi[cz, cy, cx], blk.max(), atol=2e-6)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = oracle.TfnHolder(colors, alphas)
    mo = oracle.macrocell_max_opacity(tfn, vr)
    n = alphas.shape[0]
    for idx in np.ndindex(3, 3, 3):
        il = int(np.clip(np.floor(lo[idx] * (n - 1) + 0.5) - 1, 0, n - 1))
        iu = int(np.clip(np.floor(hi[idx] * (n - 1) + 0.5) + 1, 0, n - 1))
        assert np.isclose(mo[idx], alphas[il:iu + 1].max())
    # explicit update with the voxel-centre samples reproduces the implicit build
    dz, dy, dx = vol.shape
    zz, yy, xx = np.meshgrid(np.arange(dz), np.arange(dy), np.arange(dx), indexing="ij")
    c = np.stack([(xx + 0.5) / dx, (yy + 0.5) / dy, (zz + 0.5) / dz], -1).reshape(-1, 3).astype(np.float32)
    vals = oracle.sample_volume(vol, c, nodal=False)
    vr2 = oracle.macrocell_update_explicit(np.zeros_like(vr), (dx, dy, dz), c, vals)
    assert np.array_equal(vr, vr2)


def test_dda_axis_aligned(oracle):
    cells, ts = oracle.dda_trace((-1.0, 1.5, 2.5), (1.0, 0.0, 0.0), 1.0, 5.0, (4, 4, 4))
    assert cells.tolist() == [[0, 1, 2], [1, 1, 2], [2, 1, 2], [3, 1, 2]]
    assert np.allclose(ts, [[1, 2], [2, 3], [3, 4], [4, 5]])


def test_dda_negative_direction_and_zero_components(oracle):
    cells, ts = oracle.dda_trace((5.0, 0.5, 3.5), (-1.0, 0.0, 0.0), 1.0, 5.0, (4, 4, 4))
    assert cells.tolist() == [[3, 0, 3], [2, 0, 3], [1, 0, 3], [0, 0, 3]]
    assert np.allclose(ts[:, 0], [1, 2, 3, 4])


def test_dda_diagonal_visits_connected_cells(oracle):
    d = np.array([1.0, 0.7, 0.4], np.float32)
    cells, ts = oracle.dda_trace((0.1, 0.2, 0.3), d, 0.0, 3.5, (4, 4, 4))
    assert cells[0].tolist() == [0, 0, 0]
    step = np.abs(np.diff(cells, axis=0)).sum(1)
    assert np.all(step >= 1) and np.all(step <= 3)
    assert np.allclose(ts[1:, 0], ts[:-1, 1])            # intervals tile the ray
    assert np.all(ts[:, 1] > ts[:, 0])
    # each interval's midpoint lies in its cell
    mid = 0.5 * (ts[:, 0] + ts[:, 1])
    pts = np.array([0.1, 0.2, 0.3]) + mid[:, None] * d
    assert np.array_equal(np.floor(pts).astype(int), cells)


def test_dda_exact_diagonal_steps_all_axes_together(oracle):
    cells, _ = oracle.dda_trace((0.0, 0.0, 0.0), (1.0, 1.0, 1.0), 0.0, 4.0, (4, 4, 4))
    assert cells.tolist() == [[0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]]


def test_dda_grazing_ray_stays_in_grid(oracle):
    cells, ts = oracle.dda_trace((0.0, 3.9999, 0.5), (1.0, 1e-6, 0.0), 0.0, 4.0, (4, 4, 4))
    assert np.all((cells >= 0) & (cells < 4))
    assert cells[:, 0].tolist() == [0, 1, 2, 3]


@pytest.fixture(scope="module")
def small_scene(oracle):
    vol = syn.analytic_volume(32)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = oracle.TfnHolder(colors, alphas)
    vr = oracle.macrocell_compute_implicit(vol)
    mo = oracle.macrocell_max_opacity(tfn, vr)
    cam = syn.oblique_camera((32, 32, 32))
    sc = oracle.SceneHolder(48, 40, (32, 32, 32), tfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"])
    return vol, sc


def test_streaming_is_independent_of_n_iters(oracle, small_scene):
    vol, sc = small_scene
    f = lambda c: oracle.sample_volume(vol, c, nodal=True)
    img16, _, st16 = oracle.render_streaming(sc, f, n_iters=16)
    img3, _, st3 = oracle.render_streaming(sc, f, n_iters=3)
    img64, _, st64 = oracle.render_streaming(sc, f, n_iters=64)
    assert st16["n_rays_hit"] > 0 and img16[..., 3].max() > 0.5
    assert np.abs(img16 - img3).max() < 2e-4
    assert np.abs(img16 - img64).max() < 2e-4
    # emitted samples grow with n_iters only through the tail emitted after a ray saturates mid-batch
    assert st3["n_samples"] <= st16["n_samples"] <= st64["n_samples"] <= 1.2 * st3["n_samples"]
    assert st16["n_slots"] >= st16["n_samples"]
    assert st3["n_iterations"] > st16["n_iterations"] > st64["n_iterations"]


def test_streaming_close_to_monolithic(oracle, small_scene):
    vol, sc = small_scene
    f = lambda c: oracle.sample_volume(vol, c, nodal=True)
    a, _, _ = oracle.render_streaming(sc, f)
    b, _ = oracle.render_monolithic(sc, vol)
    c, _ = oracle.render_monolithic(sc, vol, n_threads=3)
    assert np.array_equal(b, c)
    mse = float(np.mean((a - b) ** 2))
    assert 10 * np.log10(1.0 / mse) > 30.0   # same maths, different step equalisation
    assert np.array_equal(a[..., 3] == 0, b[..., 3] == 0) or np.mean((a[..., 3] == 0) != (b[..., 3] == 0)) < 0.01


def test_tiles_compose_exactly(oracle, small_scene):
    vol, sc = small_scene
    f = lambda c: oracle.sample_volume(vol, c, nodal=True)
    full, _, _ = oracle.render_streaming(sc, f)
    npx = sc.c.width * sc.c.height
    out = np.zeros_like(full).reshape(-1, 4)
    for lo, hi in [(0, 700), (700, 1500), (1500, npx)]:
        sc.c.pixel_lo, sc.c.pixel_hi = lo, hi
        img, _, _ = oracle.render_streaming(sc, f)
        out[lo:hi] = img.reshape(-1, 4)[lo:hi]
    sc.c.pixel_lo, sc.c.pixel_hi = 0, npx
    assert np.array_equal(out.reshape(full.shape), full)


def test_interleaved_shares_compose_exactly(oracle, small_scene):
    """a rank's share of a tile-sharded frame (vnro_scene il_block / il_parts / il_part): the shares of 3 ranks in blocks of 40
    pixels hold exactly the whole frame's pixels (global pixel indices keep the random sequences), touch nobody else's pixels, and their ray
    counts add up; a share's iteration count is at most the frame's"""
    vol, sc = small_scene
    f = lambda c: oracle.sample_volume(vol, c, nodal=True)
    full, _, st = oracle.render_streaming(sc, f)
    npx = sc.c.width * sc.c.height
    out = np.zeros_like(full).reshape(-1, 4)
    rays = 0
    iters = []
    for part in range(3):
        sc.c.il_block, sc.c.il_parts, sc.c.il_part = 40, 3, part
        img, _, s = oracle.render_streaming(sc, f)
        mine = (np.arange(npx) // 40) % 3 == part
        assert not img.reshape(-1, 4)[~mine].any()
        out[mine] = img.reshape(-1, 4)[mine]
        rays += s["n_rays_hit"]
        iters.append(s["n_iterations"])
    sc.c.il_block, sc.c.il_parts, sc.c.il_part = 0, 0, 0
    assert np.array_equal(out.reshape(full.shape), full)
    assert rays == st["n_rays_hit"] and max(iters) == st["n_iterations"]


def test_accumulation_divides_by_frame_index(oracle, small_scene):
    vol, sc = small_scene
    f = lambda c: oracle.sample_volume(vol, c, nodal=True)
    img1, acc, _ = oracle.render_streaming(sc, f)
    sc.c.frame_index = 2
    img2, acc, _ = oracle.render_streaming(sc, f, accumulation=acc)
    sc.c.frame_index = 1
    assert np.allclose(acc.reshape(img1.shape), img2 * 2, atol=1e-6)
    assert np.abs(img2 - img1).max() < 0.2   # different jitter, same picture


def test_psnr_definition(oracle):
    ref = np.linspace(0, 2, 1000).astype(np.float32)
    pred = ref + 0.02
    assert np.isclose(oracle.psnr(pred, ref), 10 * np.log10(4.0 / 0.0004), atol=1e-3)
    c = oracle.grid_coords((1, 2, 3), (2, 2, 2), (0.1, 0.1, 0.1))
    assert np.allclose(c[0], [0.15, 0.25, 0.35]) and np.allclose(c[-1], [0.25, 0.35, 0.45])


# --------------------------------------------------------------------------- gradient shading (modes 7 / 8)
def test_shade_scivis_light_hand_cases(oracle):
    """shade_scivis_light (raytracing.h:224-246) with mat {.6, .9, .4, 40}: view, normal and light aligned ->
    0.5 * albedo * 1.0 + 0.5 * (0.6 a + 0.9 a + 0.4) = 1.25 a + 0.2; light behind the surface -> 0.5 a + 0.5 * 0.6 a = 0.8 a."""
    got = oracle.shade_scivis_light((0, 0, 1), (0, 0, -2), (0.2, 0.4, 0.8), (0, 0, -3))
    assert np.allclose(got, [0.45, 0.7, 1.2], atol=1e-6)
    assert np.allclose(oracle.shade_scivis_light((0, 0, 1), (0, 0, -1), (0.5, 0.5, 0.5), (0, 0, 1)), 0.4, atol=1e-6)
    assert np.array_equal(oracle.shade_scivis_light((0, 0, 1), (0, 0, 0), (0.5, 0.5, 0.5), (0, 0, -1)), [0, 0, 0])   # no gradient
    # normal perpendicular to the view, light along the normal: simple term 0.2 a, scivis 0.6 a + 0.9 a + 0.4 * cos(45 deg)^40
    got = oracle.shade_scivis_light((0, 0, 1), (1, 0, 0), (0.5, 0.5, 0.5), (1, 0, 0))
    assert np.allclose(got, 0.5 * 0.2 * 0.5 + 0.5 * (1.5 * 0.5 + 0.4 * np.cos(np.pi / 4) ** 40), atol=1e-6)


def test_light_is_flipped_towards_the_viewer(oracle):
    assert oracle.flipped_light_dir((0, 0, 100), (0, 0, 0)) == pytest.approx(oracle.DEFAULT_LIGHT_DIR)     # view along -z: dot < 0
    assert oracle.flipped_light_dir((0, 0, -100), (0, 0, 0)) == pytest.approx([-v for v in oracle.DEFAULT_LIGHT_DIR])


def test_gradient_shading_changes_colour_only_and_modes_agree(oracle):
    """modes 8 (streaming) and 7 (monolithic) agree as closely as 5 and 4 do; alpha and the sample statistics are those of
    the unshaded modes"""
    vol = syn.analytic_volume(32)
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = oracle.TfnHolder(colors, alphas)
    mo = oracle.macrocell_max_opacity(tfn, oracle.macrocell_compute_implicit(vol))
    cam = syn.oblique_camera((32, 32, 32))
    f = lambda c: oracle.sample_volume(vol, c, nodal=True)
    sc0 = oracle.SceneHolder(48, 40, (32, 32, 32), tfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"])
    sc1 = oracle.SceneHolder(48, 40, (32, 32, 32), tfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"], shading_mode=1)
    a0, _, st0 = oracle.render_streaming(sc0, f)
    a1, _, st1 = oracle.render_streaming(sc1, f)
    m1, _ = oracle.render_monolithic(sc1, vol)
    assert st0 == st1 and np.array_equal(a0[..., 3], a1[..., 3])
    assert np.abs(a1[..., :3] - a0[..., :3]).mean() > 1e-3
    assert np.abs(a1 - m1).mean() < 1e-3


def test_single_shade_heuristic_properties(oracle, small_scene):
    """SINGLE_SHADE_HEURISTIC (method_raymarching.cu:455-484, 789-833, 877-900) has no reference fixture; what must hold by its
    definition: alpha is the unshaded alpha; a colour is 0.05 x unshaded + 0.95 x (a transfer-function colour) x alpha x T with
    T in [0, 1]; the streaming result does not depend on the batch size; streaming and monolithic agree up to the monolithic
    marcher's step equalisation; a transparent volume stays black"""
    vol, sc0 = small_scene
    f = lambda c: oracle.sample_volume(vol, c, nodal=True)
    cam = syn.oblique_camera((32, 32, 32))
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = oracle.TfnHolder(colors, alphas)
    mo = oracle.macrocell_max_opacity(tfn, oracle.macrocell_compute_implicit(vol))
    sc = oracle.SceneHolder(48, 40, (32, 32, 32), tfn, mo, cam["from"], cam["at"], cam["up"], cam["fovy"], shading_mode=2)
    plain, _, _ = oracle.render_streaming(sc0, f)
    ssh, _, st = oracle.render_streaming(sc, f)
    assert np.array_equal(ssh[..., 3], plain[..., 3])
    lo = 0.05 * plain[..., :3]
    hi = lo + 0.95 * np.asarray(colors, np.float32).max() * plain[..., 3:4]
    assert (ssh[..., :3] >= lo - 1e-6).all() and (ssh[..., :3] <= hi + 1e-6).all()
    assert np.abs(ssh[..., :3] - plain[..., :3]).mean() > 2e-3
    assert st["n_iterations"] > oracle.render_streaming(sc0, f)[2]["n_iterations"]   # the shadow pass adds its own
    ssh5, _, _ = oracle.render_streaming(sc, f, n_iters=5)
    assert np.abs(ssh - ssh5).max() < 2e-4
    mono, _ = oracle.render_monolithic(sc, vol)
    assert np.abs(mono - ssh).mean() < 0.01
    # nothing opaque: no strongest sample, no shadow ray, black pixels
    clear = oracle.TfnHolder(colors, np.zeros_like(np.asarray(alphas, np.float32)))
    mo0 = oracle.macrocell_max_opacity(clear, oracle.macrocell_compute_implicit(vol))
    sc_clear = oracle.SceneHolder(48, 40, (32, 32, 32), clear, mo0, cam["from"], cam["at"], cam["up"], cam["fovy"], shading_mode=2)
    assert not oracle.render_streaming(sc_clear, f)[0].any() and not oracle.render_monolithic(sc_clear, vol)[0].any()


def test_path_tracing_properties(oracle, small_scene):
    """the oracle's path tracer (method_pathtracing.cu:532-813) has no reference fixture either; by its definition: alpha is 1
    everywhere, radiance is non-negative and bounded by ambient light x the largest possible throughput, a frame is a function of
    (frame index, pixel) only, primary rays that leave without scattering stay black, a transparent volume stays black, and the
    mean over many frames settles (it is a Monte-Carlo estimate: two disjoint sets of frames agree within their noise)"""
    vol, sc0 = small_scene
    f = lambda c: oracle.sample_volume(vol, c, nodal=True)
    cam = syn.oblique_camera((32, 32, 32))
    colors, alphas = syn.tfn_ramp_with_bumps()
    tfn = oracle.TfnHolder(colors, alphas)
    mo = oracle.macrocell_max_opacity(tfn, oracle.macrocell_compute_implicit(vol))
    mk = lambda k, ds=4.0, t=tfn, m=mo: oracle.SceneHolder(48, 40, (32, 32, 32), t, m, cam["from"], cam["at"], cam["up"], cam["fovy"],
                                                           frame_index=k, density_scale=ds)
    a, _, st = oracle.render_pathtracing(mk(1), f)
    b, _, _ = oracle.render_pathtracing(mk(1), f)
    c, _, _ = oracle.render_pathtracing(mk(2), f)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert (a[..., 3] == 1.0).all() and (a[..., :3] >= 0).all()
    assert a[..., :3].max() <= 1.5 * 1.0 / 0.95 ** 8 + 1e-6          # ambient 1.5 x throughput (<= 1 before roulette, / q after)
    assert st["n_rays_hit"] > 0 and st["n_samples"] > st["n_rays_hit"] and st["n_iterations"] > 4
    clear = oracle.TfnHolder(colors, np.zeros_like(np.asarray(alphas, np.float32)))
    mo0 = oracle.macrocell_max_opacity(clear, oracle.macrocell_compute_implicit(vol))
    z, _, zst = oracle.render_pathtracing(mk(1, 4.0, clear, mo0), f)
    assert not z[..., :3].any() and (z[..., 3] == 1.0).all() and zst["n_samples"] == 0
    means = []
    for lo in (1, 41):
        acc = None
        for k in range(lo, lo + 40):
            _, acc, _ = oracle.render_pathtracing(mk(k), f, accumulation=acc)
        means.append(float(acc.reshape(-1, 4)[:, :3].mean()) / 40.0)   # the accumulation buffer holds the sum of the frames
    assert means[0] > 0.01 and abs(means[0] - means[1]) < 0.1 * means[0]
