"""TEST INFRASTRUCTURE ONLY - CPU restatement (plain numpy loops, small lattices) of the marching-tetrahedra iso-surface
that vdn_mesh_count / vdn_mesh_emit compute on the device (include/vdn_render.h). It stands where the reference calls
PyMCubes (renderer.py:36); PyMCubes is third-party and absent from the reference tree, so the triangulation itself is
"parity unpinned" (SURVEY.md 8c): this oracle pins the kernel to the algorithm as specified, and the tests add the
geometric properties any correct iso-surface has (closed, consistently oriented, on the level set)."""
import numpy as np

TETS = [(0, 1, 3, 7), (0, 1, 5, 7), (0, 2, 3, 7), (0, 2, 6, 7), (0, 4, 5, 7), (0, 4, 6, 7)]   # corner bit 0 = +x, 1 = +y, 2 = +z


def marching_tets(u, threshold=0.0):
    """-> (tri_pos [n,3,3] float32 lattice coordinates, tri_key [n,3] int64), cube-major / tet-major order like the kernel."""
    u = np.asarray(u, dtype=np.float32)
    R = u.shape[0]
    thr = np.float32(threshold)
    R3 = R ** 3
    pos, keys = [], []

    def cut(ci, cj, vid, val, xyz):
        if vid[ci] > vid[cj]:
            ci, cj = cj, ci
        t = (thr - val[ci]) / (val[cj] - val[ci])
        p = xyz[ci] + t * (xyz[cj] - xyz[ci])
        return p.astype(np.float32), int(vid[ci]) * R3 + int(vid[cj])

    def put(p, k, ipt, opt):
        n = np.cross(p[1] - p[0], p[2] - p[0])
        if float(np.dot(n, opt - ipt)) < 0.0:
            p, k = [p[0], p[2], p[1]], [k[0], k[2], k[1]]
        pos.append(np.stack(p))
        keys.append(k)

    for x in range(R - 1):
        for y in range(R - 1):
            for z in range(R - 1):
                xyz = np.array([[x + (c & 1), y + ((c >> 1) & 1), z + ((c >> 2) & 1)] for c in range(8)], dtype=np.float32)
                vid = [(int(q[0]) * R + int(q[1])) * R + int(q[2]) for q in xyz]
                val = [u[int(q[0]), int(q[1]), int(q[2])] for q in xyz]
                for tet in TETS:
                    ins = [c for c in tet if val[c] > thr]
                    outs = [c for c in tet if not val[c] > thr]
                    if len(ins) in (0, 4):
                        continue
                    ipt, opt = xyz[ins[0]], xyz[outs[0]]
                    if len(ins) in (1, 3):
                        lone = ins[0] if len(ins) == 1 else outs[0]
                        others = outs if len(ins) == 1 else ins
                        pk = [cut(lone, o, vid, val, xyz) for o in others]
                        put([q[0] for q in pk], [q[1] for q in pk], ipt, opt)
                    else:
                        pk = [cut(ins[0], outs[0], vid, val, xyz), cut(ins[0], outs[1], vid, val, xyz),
                              cut(ins[1], outs[1], vid, val, xyz), cut(ins[1], outs[0], vid, val, xyz)]
                        put([pk[0][0], pk[1][0], pk[2][0]], [pk[0][1], pk[1][1], pk[2][1]], ipt, opt)
                        put([pk[0][0], pk[2][0], pk[3][0]], [pk[0][1], pk[2][1], pk[3][1]], ipt, opt)
    if not pos:
        return np.zeros((0, 3, 3), np.float32), np.zeros((0, 3), np.int64)
    return np.stack(pos).astype(np.float32), np.asarray(keys, dtype=np.int64)


def weld(tri_pos, tri_key):
    """One vertex per cut edge: -> (vertices [V,3], triangles [F,3]) with vertices ordered by key."""
    uniq, first, inv = np.unique(tri_key.reshape(-1), return_index=True, return_inverse=True)
    return tri_pos.reshape(-1, 3)[first], inv.reshape(-1, 3)
