"""oracle/marching_cubes.py (the restatement of PyMCubes' published algorithm, reference renderer.py:36) checked for internal
consistency on the CPU: the 256-case tables, watertightness on lattices that hit every case, and the vertex bookkeeping."""
import numpy as np
import pytest

from oracle import marching_cubes as omc


def test_published_edge_table_values():
    # the head, the middle and the tail of the published `edge_table` (marchingcubes.cpp / Bourke)
    want = {0: 0x0, 1: 0x109, 2: 0x203, 3: 0x30a, 4: 0x406, 5: 0x50f, 6: 0x605, 7: 0x70c, 8: 0x80c, 9: 0x905, 10: 0xa0f, 11: 0xb06,
            12: 0xc0a, 13: 0xd03, 14: 0xe09, 15: 0xf00, 16: 0x190, 17: 0x99, 127: 0x8c0, 128: 0x8c0, 254: 0x109, 255: 0x0}
    for c, v in want.items():
        assert omc.EDGE_TABLE[c] == v, (c, hex(omc.EDGE_TABLE[c]), hex(v))


def test_every_case_uses_exactly_its_sign_changing_edges_and_patches_close():
    """Per case: the triangles touch exactly the edges whose endpoints differ; every triangle side that does not lie in a cube face
    is shared by two triangles of the case with opposite direction (the patch has no holes inside the cell)."""
    face_of_edge = {}
    for e, (a, b) in enumerate(omc.EDGES):
        ca, cb = omc.CORNERS[a], omc.CORNERS[b]
        faces = set()
        for d in range(3):
            if ca[d] == cb[d]:
                faces.add((d, ca[d]))
        face_of_edge[e] = faces
    for c in range(256):
        tt = omc.TRIANGLE_TABLE[c]
        assert len(tt) % 3 == 0 and len(tt) <= 15
        used = 0
        for e in tt:
            used |= 1 << e
        assert used == omc.EDGE_TABLE[c], (c, hex(used), hex(omc.EDGE_TABLE[c]))
        directed = {}
        for t in range(0, len(tt), 3):
            tri = tt[t:t + 3]
            assert len(set(tri)) == 3, (c, tri)
            for s in range(3):
                p, q = tri[s], tri[(s + 1) % 3]
                directed[(p, q)] = directed.get((p, q), 0) + 1
        for (p, q), n in directed.items():
            assert n == 1, (c, p, q)
            on_face = bool(face_of_edge[p] & face_of_edge[q])
            if not on_face:
                assert directed.get((q, p), 0) == 1, ("open side inside the cell", c, p, q)
            else:
                assert directed.get((q, p), 0) <= 1


def _closed_and_oriented(V, F, interior_only=None):
    d = {}
    for a, b, c in F:
        for p, q in ((a, b), (b, c), (c, a)):
            d[(p, q)] = d.get((p, q), 0) + 1
    bad = 0
    for (p, q), n in d.items():
        if interior_only is not None and not (interior_only[p] and interior_only[q]):
            continue
        if n != 1 or d.get((q, p), 0) != 1:
            bad += 1
    return bad


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_noise_lattice_is_watertight(seed):
    """White noise around the level: all 256 cases and every ambiguous face configuration between neighbouring cells. Away from the
    lattice boundary every triangle side must be shared by exactly one opposite side."""
    rng = np.random.default_rng(seed)
    R = 16
    u = rng.standard_normal((R, R, R)).astype(np.float32)
    V, F = omc.marching_cubes(u, 0.0)
    assert len(F) > 500
    seen = set()
    for i in range(R - 1):
        for j in range(R - 1):
            for k in range(R - 1):
                c = 0
                for m, (dx, dy, dz) in enumerate(omc.CORNERS):
                    if u[i + dx, j + dy, k + dz] <= 0.0:
                        c |= 1 << m
                seen.add(c)
    assert len(seen) == 256
    inner = np.all((V > 0.0) & (V < R - 1.0), axis=1)
    assert _closed_and_oriented(V, F, inner) == 0
    # one vertex per cut lattice edge: no two vertices coincide
    assert len(np.unique(np.round(V, 12), axis=0)) == len(V)
    # every vertex lies on a lattice edge: two integer coordinates
    assert np.all(np.sum(V == np.round(V), axis=1) >= 2)


def test_sphere_orientation_count_and_level():
    R = 20
    g = np.linspace(-1.2, 1.2, R)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    u = (0.8 - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32)          # u = -sdf of a sphere, as extract_fields returns it
    V, F = omc.marching_cubes(u, 0.0)
    assert _closed_and_oriented(V, F) == 0
    assert len(V) - len(F) * 3 // 2 + len(F) == 2                            # Euler characteristic of a sphere
    P = V / (R - 1.0) * 2.4 - 1.2
    assert np.abs(np.linalg.norm(P, axis=1) - 0.8).max() < 0.01
    # orientation as the tables give it with corners set where u <= isovalue (outside the object): normals point OUT of the object
    n = np.cross(P[F[:, 1]] - P[F[:, 0]], P[F[:, 2]] - P[F[:, 0]])
    cen = P[F].mean(axis=1)
    assert np.all(np.sum(n * cen, axis=1) > 0)
    # vertices are numbered in creation order: the first one belongs to the first cut cell in x-major order
    assert np.all(np.diff(np.lexsort((V[:, 2], V[:, 1], V[:, 0]))[:1]) >= 0)


def test_device_table_literal_equals_the_oracle_table():
    """csrc/mc_tables.h (what vdn_mesh_mc_* index) holds the same 256 rows."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "vdn-nerf_amd", "csrc", "mc_tables.h")).read()
    rows = re.findall(r"^\s*\{([-0-9, ]+)\},", text, flags=re.M)
    assert len(rows) == 256
    for c, row in enumerate(rows):
        vals = [int(x) for x in row.split(",")]
        assert len(vals) == 16
        used = vals[:vals.index(-1)] if -1 in vals else vals
        assert used == omc.TRIANGLE_TABLE[c] and all(v == -1 for v in vals[len(used):]), c
