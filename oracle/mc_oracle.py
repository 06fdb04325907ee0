"""numpy restatement of the isosurface extraction (core/marching_cube.cu:23-44, 84-92, 117-122, 147-250): TEST INFRASTRUCTURE ONLY.

The reference holds no fixture for its marching cubes and cannot run here: PARITY UNPINNED.  What is restated is the reference's rule set
(dual grid, `value <= isovalue`, the vertex rule with its 0.001 guard, + cell + 0.5); the case table is the one tools/gen_mc_table.py derives
(this repository's own: see there), imported from that script, not from the product."""
import importlib.util
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("gen_mc_table", os.path.join(_ROOT, "tools", "gen_mc_table.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)
TABLE = gen.table()
EDGES = np.array(gen.EDGES, np.int64)
OFFSETS = np.array(gen.CORNERS, np.int64)       # corner i -> (dx, dy, dz)


def marching_cubes(values, isovalue):
    """values: [z, y, x] float32 at the grid nodes -> float32 [n, 3] vertices (x, y, z), in cell order, table order inside a cell"""
    v = np.asarray(values, np.float32)
    nz, ny, nx = v.shape
    iso = np.float32(isovalue)
    corner = [v[dz:nz - 1 + dz, dy:ny - 1 + dy, dx:nx - 1 + dx] for dx, dy, dz in OFFSETS]
    case = np.zeros((nz - 1, ny - 1, nx - 1), np.int64)
    for i in range(8):
        case |= (corner[i] <= iso).astype(np.int64) << i
    counts = (TABLE >= 0).sum(1)[case]
    cz, cy, cx = np.nonzero(counts)                     # C order = x fastest = the kernel's cell index order
    out = []
    for z, y, x in zip(cz, cy, cx):
        c = int(case[z, y, x])
        vals = [corner[i][z, y, x] for i in range(8)]
        for e in TABLE[c]:
            if e < 0:
                break
            a, b = EDGES[e]
            fa, fb = np.float32(vals[a]), np.float32(vals[b])
            t = np.float32(0.0)
            if abs(np.float32(fa - fb)) >= np.float32(0.001):
                t = np.float32(np.float32(iso - fa) / np.float32(fb - fa))
            va, vb = OFFSETS[a].astype(np.float32), OFFSETS[b].astype(np.float32)
            p = (np.float32(1.0) - t) * va + t * vb
            p = (p + np.array([x, y, z], np.float32)) + np.float32(0.5)
            out.append(p.astype(np.float32))
    return np.array(out, np.float32).reshape(-1, 3)
