"""The marching-tetrahedra restatement (oracle/marching_tets.py) that pins vdn_mesh_count / vdn_mesh_emit: no reference
vectors exist for this step (PyMCubes is third-party and absent), so the oracle itself is held to the
properties of a correct iso-surface."""
from collections import Counter

import numpy as np
import pytest

from oracle import marching_tets as mt


def _edges(F):
    d = Counter()
    for f in F:
        for a, b in ((f[0], f[1]), (f[1], f[2]), (f[2], f[0])):
            d[(int(a), int(b))] += 1
    und = Counter()
    for (a, b), c in d.items():
        und[(min(a, b), max(a, b))] += c
    return d, und


@pytest.mark.parametrize("thr", [0.0, 0.1])
def test_sphere_is_closed_oriented_and_on_the_level_set(thr):
    R = 12
    g = np.linspace(-1, 1, R)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    u = (0.7 - np.sqrt(X * X + Y * Y + Z * Z)).astype(np.float32)
    pos, key = mt.marching_tets(u, thr)
    V, F = mt.weld(pos, key)
    d, und = _edges(F)
    assert max(d.values()) == 1 and set(und.values()) == {2}
    assert len(V) - len(und) + len(F) == 2
    h = 2.0 / (R - 1)
    P = V * h - 1.0
    assert np.abs(np.linalg.norm(P, axis=1) - (0.7 - thr)).max() < h * h
    n = np.cross(P[F[:, 1]] - P[F[:, 0]], P[F[:, 2]] - P[F[:, 0]])
    assert (np.sum(n * P[F].mean(1), axis=1) > 0).all()


def test_empty_and_full_lattices_give_no_triangles():
    u = np.ones((5, 5, 5), np.float32)
    for s in (1.0, -1.0):
        pos, key = mt.marching_tets(s * u, 0.0)
        assert pos.shape == (0, 3, 3) and key.shape == (0, 3)


def test_plane_cut_is_flat_and_shared_vertices_are_identical():
    R = 6
    g = np.arange(R, dtype=np.float32)
    X, _, _ = np.meshgrid(g, g, g, indexing="ij")
    pos, key = mt.marching_tets(2.4 - X, 0.0)
    assert len(pos) > 0
    assert np.allclose(pos[..., 0], 2.4, atol=1e-6)
    flat_k, flat_p = key.reshape(-1), pos.reshape(-1, 3)
    order = np.argsort(flat_k, kind="stable")
    same = flat_k[order][1:] == flat_k[order][:-1]
    assert np.array_equal(flat_p[order][1:][same], flat_p[order][:-1][same])       # bit-identical duplicates
