"""Iso-surface of a device lattice (reference renderer.py:36 calls mcubes.marching_cubes there), two forms:
  marching_cubes  - vdn_mesh_mc_count / vdn_mesh_mc_emit: the classic 256-case marching cubes with PyMCubes' own vertex and
                    triangle numbering (include/vdn_render.h; the default of extract_geometry since round 6);
  marching_tets   - vdn_mesh_count / vdn_mesh_emit: marching tetrahedra on the Kuhn decomposition (rounds 3-5), kept as an option.
The prefix sums (and the tetrahedra form's vertex welding) are torch ops on the device; nothing runs on the host."""
import torch

from . import lib


def marching_tets(u, threshold=0.0):
    """u [R,R,R] fp32 CUDA tensor -> (vertices [V,3] fp32 in lattice index coordinates, triangles [F,3] int64)."""
    if not (torch.is_tensor(u) and u.is_cuda and u.dim() == 3 and u.shape[0] == u.shape[1] == u.shape[2]):
        raise ValueError("marching_tets needs a cubic [R,R,R] CUDA tensor")
    u = u.contiguous().float()
    R = u.shape[0]
    if R < 2:
        raise ValueError("lattice resolution must be at least 2")
    st = torch.cuda.current_stream().cuda_stream
    n = (R - 1) ** 3
    counts = torch.empty(n, dtype=torch.int32, device=u.device)
    a = lib.VdnMeshArgs()
    a.u, a.threshold, a.R, a.counts = u.data_ptr(), float(threshold), R, counts.data_ptr()
    lib.call("vdn_mesh_count", a, st)
    incl = torch.cumsum(counts, 0, dtype=torch.int64)
    n_tri = int(incl[-1].item())
    if n_tri == 0:
        return torch.zeros(0, 3, device=u.device), torch.zeros(0, 3, dtype=torch.int64, device=u.device)
    offsets = (incl - counts).contiguous()
    pos = torch.empty(n_tri, 3, 3, dtype=torch.float32, device=u.device)
    key = torch.empty(n_tri, 3, dtype=torch.int64, device=u.device)
    a.offsets, a.tri_pos, a.tri_key = offsets.data_ptr(), pos.data_ptr(), key.data_ptr()
    lib.call("vdn_mesh_emit", a, st)
    # weld: one vertex per cut lattice edge
    uniq, inv = torch.unique(key.reshape(-1), return_inverse=True)
    first = torch.full((uniq.numel(),), n_tri * 3, dtype=torch.int64, device=u.device)
    first.scatter_reduce_(0, inv, torch.arange(n_tri * 3, device=u.device), reduce="amin")
    vertices = pos.reshape(-1, 3)[first]
    return vertices, inv.reshape(n_tri, 3)


def marching_cubes(u, threshold=0.0):
    """u [R,R,R] fp32 CUDA tensor -> (vertices [V,3] float64 in lattice index coordinates, triangles [F,3] int64): the two arrays
    `mcubes.marching_cubes(u, threshold)` returns (PyMCubes 0.1.2, as restated in oracle/marching_cubes.py), element for element."""
    if not (torch.is_tensor(u) and u.is_cuda and u.dim() == 3 and u.shape[0] == u.shape[1] == u.shape[2]):
        raise ValueError("marching_cubes needs a cubic [R,R,R] CUDA tensor")
    u = u.contiguous().float()
    R = u.shape[0]
    if R < 2:
        raise ValueError("lattice resolution must be at least 2")
    st = torch.cuda.current_stream().cuda_stream
    n, dev = (R - 1) ** 3, u.device
    case = torch.empty(n, dtype=torch.uint8, device=dev)
    nv, nt = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    a = lib.VdnMeshMcArgs()
    # (the library's array entry point takes the level as a C float - `_mcubes.pyx`: `float isovalue` - before its C++ code widens it)
    a.u, a.isovalue, a.R = u.data_ptr(), float(torch.tensor(float(threshold), dtype=torch.float32).item()), R
    a.cube_case, a.n_verts, a.n_tris = case.data_ptr(), nv.data_ptr(), nt.data_ptr()
    lib.call("vdn_mesh_mc_count", a, st)
    iv, it = torch.cumsum(nv, 0, dtype=torch.int64), torch.cumsum(nt, 0, dtype=torch.int64)
    totals = torch.stack([iv[-1], it[-1]]).tolist()          # one host read: the sizes of the two outputs
    V, F = int(totals[0]), int(totals[1])
    vertices = torch.empty(V, 3, dtype=torch.float64, device=dev)
    triangles = torch.empty(F, 3, dtype=torch.int64, device=dev)
    if F == 0:
        return vertices, triangles
    vo, to = (iv - nv).contiguous(), (it - nt).contiguous()
    a.vert_offsets, a.tri_offsets, a.vertices, a.triangles = vo.data_ptr(), to.data_ptr(), vertices.data_ptr(), triangles.data_ptr()
    lib.call("vdn_mesh_mc_emit", a, st)
    return vertices, triangles
