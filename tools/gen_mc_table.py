"""Generates instantvnr_amd/csrc/mc_table.h: the marching-cubes case table of this library, DERIVED, not transcribed.

The reference ships the classic hand-made 256 x 16 table (core/marching_cube_constants.cuh); copying a table out of its sources is
not an option here, and its particular triangle order cannot be checked anyway (the reference holds no fixture for it).  The table below
is constructed from first principles and gives the same kind of surface: for each of the 256 sign configurations of a cell's corners

  1. every cube face whose corners are not all on one side is crossed by one or two segments joining the cut edges of that face; a face
     with two diagonally opposite corners inside (<= isovalue) is AMBIGUOUS and is resolved by one fixed rule, "inside corners are cut off
     separately", which depends on the face's four signs only, so the two cells that share a face agree and the surface is watertight;
  2. the segments of the six faces chain into closed loops over the cut edges (every cut edge lies on exactly two faces);
  3. every loop is oriented like the reference's triangles (right-hand normal pointing from the outside to the inside, <= isovalue) and is
     cut into a triangle fan.

tools/mc_table_vs_reference.py (development time, this container: it reads the reference's table as text) compares the result with the
reference's hand-made table case by case: in all 256 cases, the 120 with an ambiguous face included, the triangles meet the faces of the
cell in the same DIRECTED segments and there are equally many of them: the same surface, the same winding, crack-free against a neighbour
triangulated by either table; only the way a loop's interior is cut into triangles may differ.

Numbering: corner i has offset (i & 1, (i >> 1) & 1, (i >> 2) & 1); edge e of EDGES joins two corners.  A case has at most 4 loops and
12 cut edges; no case needs more than 5 triangles (16 table elements per case with the terminator, like the classic table).

usage: python tools/gen_mc_table.py  (writes the header; tests/test_marching_cubes.py checks that the committed header is what this makes)"""
import itertools
import os

import numpy as np

CORNERS = [(i & 1, (i >> 1) & 1, (i >> 2) & 1) for i in range(8)]
EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]   # 12 edges, corner pairs that differ in one axis
EDGE_ID = {e: k for k, e in enumerate(EDGES)}
MAX_TRIS = 5


def faces():
    """the 6 faces as cycles of 4 corners (in order around the face)"""
    out = []
    for axis in range(3):
        for side in (0, 1):
            u, v = [a for a in range(3) if a != axis]
            cyc = []
            for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):
                c = [0, 0, 0]
                c[axis] = side; c[u] = du; c[v] = dv
                cyc.append(c[0] | c[1] << 1 | c[2] << 2)
            out.append(cyc)
    return out


FACES = faces()


def edge(a, b):
    return EDGE_ID[(min(a, b), max(a, b))]


def case_triangles(case):
    inside = [(case >> i) & 1 for i in range(8)]
    # 1. segments per face: pairs of cut edges
    segments = []
    for cyc in FACES:
        s = [inside[c] for c in cyc]
        cut = [k for k in range(4) if s[k] != s[(k + 1) % 4]]       # cut face edges: between cyc[k] and cyc[k + 1]
        if len(cut) == 2:
            segments.append((edge(cyc[cut[0]], cyc[(cut[0] + 1) % 4]), edge(cyc[cut[1]], cyc[(cut[1] + 1) % 4])))
        elif len(cut) == 4:
            # ambiguous: cut off each INSIDE corner separately: the two face edges that meet at an inside corner are joined
            for k in range(4):
                if s[k]:
                    segments.append((edge(cyc[k - 1], cyc[k]), edge(cyc[k], cyc[(k + 1) % 4])))
    # 2. loops
    nbr = {}
    for a, b in segments:
        nbr.setdefault(a, []).append(b)
        nbr.setdefault(b, []).append(a)
    assert all(len(v) == 2 for v in nbr.values()), (case, nbr)
    loops, seen = [], set()
    for start in sorted(nbr):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxt = [n for n in nbr[cur] if n != prev]
            nxt = nxt[0] if nxt else nbr[cur][0]
            if nxt == start:
                break
            loop.append(nxt); seen.add(nxt)
            prev, cur = cur, nxt
        loops.append(loop)
    # 3. orientation + fan.  Edge midpoints stand in for the vertices; the normal points towards the inside corners of the loop's edges
    tris = []
    for loop in loops:
        pts = np.array([(np.array(CORNERS[EDGES[e][0]]) + np.array(CORNERS[EDGES[e][1]])) / 2.0 for e in loop])
        centre = pts.mean(0)
        normal = sum(np.cross(pts[k] - centre, pts[(k + 1) % len(loop)] - centre) for k in range(len(loop)))
        ins = np.array([CORNERS[a] if inside[a] else CORNERS[b] for a, b in (EDGES[e] for e in loop)], float).mean(0)
        if np.dot(normal, centre - ins) > 0:   # the reference's winding: the normal of (v0, v1, v2) points TOWARDS the inside (<= isovalue) corners
            loop = loop[::-1]
        for k in range(1, len(loop) - 1):
            tris.append((loop[0], loop[k], loop[k + 1]))
    return tris


def table():
    t = np.full((256, 3 * MAX_TRIS + 1), -1, np.int8)
    for case in range(256):
        tris = case_triangles(case)
        assert len(tris) <= MAX_TRIS
        flat = list(itertools.chain.from_iterable(tris))
        t[case, :len(flat)] = flat
    return t


def header():
    t = table()
    lines = ["// mc_table.h — GENERATED by tools/gen_mc_table.py (the derivation is described there); do not edit.",
             "#pragma once", "#include <cstdint>", "namespace vnr {",
             "constexpr int kMcCaseElements = %d;   // up to %d triangles x 3 cut-edge indices, then -1" % (t.shape[1], MAX_TRIS),
             "// corner i: offset (i & 1, (i >> 1) & 1, (i >> 2) & 1); edge e joins corners kMcEdgeCorners[e][0..1]",
             "static const int8_t kMcEdgeCornersHost[12][2] = {" + ", ".join("{%d, %d}" % e for e in EDGES) + "};",
             "static const int8_t kMcCaseTableHost[256][kMcCaseElements] = {"]
    for case in range(256):
        lines.append("  {" + ", ".join(str(int(v)) for v in t[case]) + "},")
    lines += ["};", "}  // namespace vnr", ""]
    return "\n".join(lines)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "instantvnr_amd", "csrc", "mc_table.h")
    open(path, "w").write(header())
    t = table()
    print("wrote", path, "; triangles per case: max", int(((t >= 0).sum(1) // 3).max()), ", total", int((t >= 0).sum() // 3))
