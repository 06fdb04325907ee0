"""Development-time check (this container only: it reads /root/reference as text): is the marching-cubes case table this library DERIVES
(tools/gen_mc_table.py) the same surface, case by case, as the hand-made table the reference ships (core/marching_cube_constants.cuh)?
Compared is what decides the surface's topology, its continuity across cells and its winding: the DIRECTED segments in which a case's
triangles meet the six faces of the cell (pairs of cut edges), after mapping the reference's corner and edge numbering to this library's; the triangulation of a
loop's interior is free.  usage: python tools/mc_table_vs_reference.py"""
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_mc_table as G  # noqa: E402

path = "/root/reference/core/marching_cube_constants.cuh"
if not os.path.exists(path):
    print("the reference is not here"); sys.exit(0)
text = open(path).read()


def table(name):
    body = re.search(name + r"\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}", text).group(1)
    return [int(v) for v in re.findall(r"-?\d+", body)]


cases, ref_corner, ref_edge = table("MC_CASE_TABLE"), table("INDEX_TO_VERTEX"), table("EDGE_VERTICES")
to_mine = [ref_corner[3 * r] | ref_corner[3 * r + 1] << 1 | ref_corner[3 * r + 2] << 2 for r in range(8)]   # reference corner -> this library's
edge_to_mine = [G.edge(to_mine[ref_edge[2 * e]], to_mine[ref_edge[2 * e + 1]]) for e in range(12)]
face_of_edges = {}
for f, cyc in enumerate(G.FACES):
    for k in range(4):
        face_of_edges.setdefault(G.edge(cyc[k], cyc[(k + 1) % 4]), set()).add(f)


def face_segments(tris):
    """DIRECTED triangle sides that lie in a face of the cell and whose reverse is no side of another triangle: where the oriented patch
    meets the cell's boundary, and which way round"""
    sides = set()
    for t in tris:
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            sides.add((a, b))
    return {(a, b) for a, b in sides if (b, a) not in sides and face_of_edges[a] & face_of_edges[b]}


same, differ, ambiguous = 0, [], 0
for c_ref in range(256):
    row = cases[16 * c_ref:16 * c_ref + 16]
    row = row[:row.index(-1)]
    ref_tris = [tuple(edge_to_mine[e] for e in row[k:k + 3]) for k in range(0, len(row), 3)]
    c_mine = sum(1 << to_mine[r] for r in range(8) if (c_ref >> r) & 1)
    inside = [(c_mine >> i) & 1 for i in range(8)]
    amb = any(sum(inside[c] != inside[cyc[(k + 1) % 4]] for k, c in enumerate(cyc)) == 4 for cyc in G.FACES)
    ambiguous += amb
    mine = G.case_triangles(c_mine)
    if face_segments(ref_tris) == face_segments(mine) and len(ref_tris) == len(mine):
        same += 1
    else:
        differ.append((c_ref, c_mine, amb, len(ref_tris), len(mine)))
print(f"{same} of 256 cases: the same directed face segments and the same number of triangles as the reference's table; {ambiguous} cases have an ambiguous face")
for c_ref, c_mine, amb, nr, nm in differ:
    print(f"   reference case {c_ref:3d} (here {c_mine:3d}): {'ambiguous face' if amb else 'NOT ambiguous'}; triangles {nr} there, {nm} here")

# winding: for the eight cases with ONE corner inside, does a triangle's normal (right-hand rule over its vertex order, vertices at the cut
# edges' midpoints) point away from that corner?
import numpy as np  # noqa: E402
mid = [(np.array(G.CORNERS[a], float) + np.array(G.CORNERS[b], float)) / 2 for a, b in G.EDGES]


def outward(tri, corner):
    p = [mid[e] for e in tri]
    n = np.cross(p[1] - p[0], p[2] - p[0])
    return float(np.dot(n, (p[0] + p[1] + p[2]) / 3 - np.array(G.CORNERS[corner], float))) > 0


ref_out, mine_out = [], []
for r in range(8):
    row = cases[16 * (1 << r):16 * (1 << r) + 3]
    ref_out.append(outward(tuple(edge_to_mine[e] for e in row), to_mine[r]))
    mine_out.append(outward(G.case_triangles(1 << to_mine[r])[0], to_mine[r]))
print(f"winding of the one-corner cases (normal away from the inside corner): reference {ref_out}, here {mine_out}")
