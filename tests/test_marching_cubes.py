"""Isosurface extraction (vnrMarchingCube / vnrSaveTriangles, core/marching_cube.cuh:6-8): the derived case table, the numpy restatement
of the reference's rules, and (-m gpu) the HIP kernels against it through the C-ABI."""
import os

import numpy as np
import pytest

from oracle import mc_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_table_is_what_the_derivation_makes():
    assert open(os.path.join(ROOT, "instantvnr_amd", "csrc", "mc_table.h")).read() == mc_oracle.gen.header()
    t = mc_oracle.TABLE
    assert t.shape == (256, 16) and (t[0] < 0).all() and (t[255] < 0).all()
    # complementary cases cut the same edges (apart from ambiguous faces the same polygons, reversed)
    for c in range(256):
        assert set(t[c][t[c] >= 0]) == set(t[255 - c][t[255 - c] >= 0])
        assert (t[c] >= 0).sum() % 3 == 0


def smooth_field(n, seed):
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*[np.linspace(-1, 1, n, dtype=np.float32)] * 3, indexing="ij")
    f = np.zeros((n, n, n), np.float32)
    for _ in range(6):
        c = rng.uniform(-0.6, 0.6, 3)
        f += np.exp(-(((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / rng.uniform(0.05, 0.2))).astype(np.float32) * rng.uniform(0.5, 1.0)
    return f


def test_surface_is_closed_and_oriented_on_smooth_and_on_ambiguous_data():
    """every directed edge of a triangle has its reverse in exactly one other triangle: watertight and consistently oriented, also across
    the ambiguous faces a checkerboard-like field is full of (the face rule depends on the face only, so neighbouring cells agree)"""
    rng = np.random.default_rng(0)
    fields = [smooth_field(20, 1), rng.random((9, 10, 11), dtype=np.float32)]
    for f in fields:
        g = np.pad(f, 1, constant_values=0.0)        # outside everywhere on the boundary: the surface cannot leave the volume
        verts = mc_oracle.marching_cubes(g, 0.5)
        assert verts.shape[0] > 300 and verts.shape[0] % 3 == 0
        keys = np.round(verts * 4096).astype(np.int64)
        tri = keys.reshape(-1, 3, 3)
        edges = {}
        for t in tri:
            for k in range(3):
                a, b = tuple(t[k]), tuple(t[(k + 1) % 3])
                if a == b:
                    continue                          # a triangle collapsed by the 0.001 guard
                edges[(a, b)] = edges.get((a, b), 0) + 1
        bad = [e for e, n in edges.items() if n != 1 or edges.get((e[1], e[0]), 0) != 1]
        assert len(bad) <= 0.002 * len(edges), (len(bad), len(edges))     # (vertices welded by the rounding key on nearly degenerate cells)


def test_sphere_vertices_lie_on_the_sphere():
    n = 33
    z, y, x = np.meshgrid(*[np.arange(n, dtype=np.float32)] * 3, indexing="ij")
    r = np.sqrt((x - 16) ** 2 + (y - 16) ** 2 + (z - 16) ** 2)
    verts = mc_oracle.marching_cubes(r, 10.3)          # inside = r <= 10.3 (no node lies exactly on the surface)
    d = np.sqrt(((verts - 0.5 - 16) ** 2).sum(1))      # the reference offsets vertices by + 0.5 (core/marching_cube.cu:245)
    assert verts.shape[0] > 3000 and np.abs(d - 10.3).max() < 0.08
    # the reference's winding (its case table, core/marching_cube_constants.cuh, compared case by case in tools/mc_table_vs_reference.py): the
    # right-hand normal of (v0, v1, v2) points from the outside to the inside (<= iso), here towards the sphere's centre
    tri = verts.reshape(-1, 3, 3)
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    ctr = tri.mean(1) - 0.5 - 16
    assert ((nrm * ctr).sum(1) < 0).mean() > 0.999 and not ((nrm * ctr).sum(1) > 0).any()


def test_the_derived_case_table_is_the_reference_s_surface_case_by_case():
    """tools/mc_table_vs_reference.py: where the reference's sources are present (this container, not the GPU box) its hand-made case table
    is read as text and compared with the table this library derives: in all 256 cases the triangles meet the cell's faces in the same
    directed segments (same topology, same resolution of the 120 ambiguous cases, same winding) and are equally many"""
    import os
    import subprocess
    import sys
    if not os.path.exists("/root/reference/core/marching_cube_constants.cuh"):
        pytest.skip("the reference's sources are not on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "mc_table_vs_reference.py")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "256 of 256 cases: the same directed face segments and the same number of triangles" in out.stdout, out.stdout[-800:]
    assert "reference [False, False, False, False, False, False, False, False], here [False, False, False, False, False, False, False, False]" in out.stdout


@pytest.mark.gpu
def test_gpu_marching_cubes_equals_the_restatement(tmp_path):
    from instantvnr_amd import api
    from instantvnr_amd import synthetic as syn
    f = smooth_field(40, 3)[:28, :34, :]
    f = ((f - f.min()) / (f.max() - f.min())).astype(np.float32)     # the library normalises a volume by its min / max: already [0, 1]
    sv = api.vnrCreateSimpleVolume(f)
    got = api.vnrMarchingCube(sv, 0.4)
    want = mc_oracle.marching_cubes(f, 0.4)
    assert want.shape[0] > 2000 and got.shape == want.shape
    assert np.array_equal(got, want)                                   # same cells, same order, same bits
    assert api.vnrMarchingCube(sv, 2.0).shape == (0, 3)                # no surface: an empty array
    # vnrSaveTriangles: "v x y z" per vertex (%f), "f a b c" per triangle, 1-based
    api.vnrSaveTriangles(tmp_path / "iso.obj", got)
    lines = open(tmp_path / "iso.obj").read().splitlines()
    n = got.shape[0]
    assert len(lines) == n + n // 3 and lines[0] == "v %f %f %f" % tuple(got[0]) and lines[n] == "f 1 2 3" and lines[-1] == "f %d %d %d" % (n - 2, n - 1, n)
    # a neural volume is evaluated at the grid nodes index / dims (core/marching_cube.cu:117-122)
    nv = api.vnrCreateNeuralVolume(syn.model_config(n_levels=6, n_features=2, log2_hashmap_size=14, base_resolution=4, n_hidden_layers=2), sv,
                                   online_macrocell_construction=False)
    api.vnrNeuralVolumeTrain(nv, 200, True)
    nz, ny, nx = f.shape
    zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    nodes = np.stack([xx.ravel() / np.float32(nx), yy.ravel() / np.float32(ny), zz.ravel() / np.float32(nz)], axis=1).astype(np.float32)
    values = api.neural_inference(nv, nodes).reshape(nz, ny, nx)
    got_n = api.vnrMarchingCube(nv, 0.4)
    assert got_n.shape[0] > 1000 and np.array_equal(got_n, mc_oracle.marching_cubes(values, 0.4))
