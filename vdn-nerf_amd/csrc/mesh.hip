// Device iso-surface extraction for NeuSRenderer.extract_geometry (reference renderer.py:33-41, where the
// triangulation is PyMCubes' marching cubes): marching tetrahedra on the Kuhn decomposition of every lattice cube.
// HBM-bound integer / float work: one thread per cube, 8 corner loads (neighbouring threads share 4 of them through L1).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "vdn_render.h"
#include "mc_tables.h"

namespace vdn {

// corner c of a cube: bit 0 = +x, bit 1 = +y, bit 2 = +z. Tetrahedron i follows one of the 6 axis orders (a, b, c):
// corners 0, 1<<a, (1<<a)|(1<<b), 7.
__device__ __constant__ unsigned char kTet[6][4] = {
    {0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7}, {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};

struct Cube {
    float u[8];
    long vid[8];
    float px[8], py[8], pz[8];
};

__device__ inline bool load_cube(const VdnMeshArgs& a, long cube, Cube& q) {
    const long n = a.R - 1;
    if (cube >= n * n * n) return false;
    const int z = (int)(cube % n), y = (int)((cube / n) % n), x = (int)(cube / (n * n));
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int xx = x + (c & 1), yy = y + ((c >> 1) & 1), zz = z + ((c >> 2) & 1);
        const long id = ((long)xx * a.R + yy) * a.R + zz;
        q.vid[c] = id;
        q.u[c] = a.u[id];
        q.px[c] = (float)xx; q.py[c] = (float)yy; q.pz[c] = (float)zz;
    }
    return true;
}

// number of triangles of a tetrahedron from its inside mask
__device__ inline int tet_tris(int mask) {
    const int pc = __popc(mask);
    return pc == 0 || pc == 4 ? 0 : (pc == 2 ? 2 : 1);
}

__global__ void mesh_count_kernel(VdnMeshArgs a) {
    const long cube = (long)blockIdx.x * blockDim.x + threadIdx.x;
    Cube q;
    if (!load_cube(a, cube, q)) return;
    int n = 0;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        int mask = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) mask |= (q.u[kTet[t][k]] > a.threshold ? 1 : 0) << k;
        n += tet_tris(mask);
    }
    a.counts[cube] = n;
}

// the cut point of lattice edge (i, j) of the cube, always evaluated from the lower lattice id to the higher one so that
// every tetrahedron (and every neighbouring cube) produces bit-identical coordinates for the same edge
__device__ inline void cut(const VdnMeshArgs& a, const Cube& q, int i, int j, float* pos, int64_t* key) {
    if (q.vid[i] > q.vid[j]) { const int t = i; i = j; j = t; }
    const float ui = q.u[i], uj = q.u[j];
    const float t = (a.threshold - ui) / (uj - ui);
    pos[0] = q.px[i] + t * (q.px[j] - q.px[i]);
    pos[1] = q.py[i] + t * (q.py[j] - q.py[i]);
    pos[2] = q.pz[i] + t * (q.pz[j] - q.pz[i]);
    const long R3 = (long)a.R * a.R * a.R;
    *key = q.vid[i] * R3 + q.vid[j];
}

__device__ inline void put_tri(const VdnMeshArgs& a, long slot, float (*p)[3], int64_t* k, const float* in_pt, const float* out_pt) {
    // wind so that the normal points from the inside corner towards the outside corner
    const float e1[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]};
    const float e2[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
    const float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const float dir = n[0] * (out_pt[0] - in_pt[0]) + n[1] * (out_pt[1] - in_pt[1]) + n[2] * (out_pt[2] - in_pt[2]);
    const int o1 = dir < 0.0f ? 2 : 1, o2 = dir < 0.0f ? 1 : 2;
    const int order[3] = {0, o1, o2};
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const int s = order[v];
        a.tri_key[slot * 3 + v] = k[s];
#pragma unroll
        for (int d = 0; d < 3; ++d) a.tri_pos[(slot * 3 + v) * 3 + d] = p[s][d];
    }
}

__global__ void mesh_emit_kernel(VdnMeshArgs a) {
    const long cube = (long)blockIdx.x * blockDim.x + threadIdx.x;
    Cube q;
    if (!load_cube(a, cube, q)) return;
    long slot = a.offsets[cube];
    for (int t = 0; t < 6; ++t) {
        int mask = 0, in[4], out[4], ni = 0, no = 0;
        for (int k = 0; k < 4; ++k) {
            const int c = kTet[t][k];
            if (q.u[c] > a.threshold) { mask |= 1 << k; in[ni++] = c; } else { out[no++] = c; }
        }
        if (ni == 0 || ni == 4) continue;
        float p[4][3];
        int64_t key[4];
        const float ipt[3] = {q.px[in[0]], q.py[in[0]], q.pz[in[0]]};
        const float opt[3] = {q.px[out[0]], q.py[out[0]], q.pz[out[0]]};
        if (ni == 1 || ni == 3) {
            // one corner alone on its side: the triangle cuts its three edges
            const int lone = ni == 1 ? in[0] : out[0];
            const int* others = ni == 1 ? out : in;
            for (int e = 0; e < 3; ++e) cut(a, q, lone, others[e], p[e], &key[e]);
            put_tri(a, slot++, p, key, ipt, opt);
        } else {
            // two against two: quad (in0-out0, in0-out1, in1-out1, in1-out0), split along the first diagonal
            cut(a, q, in[0], out[0], p[0], &key[0]);
            cut(a, q, in[0], out[1], p[1], &key[1]);
            cut(a, q, in[1], out[1], p[2], &key[2]);
            cut(a, q, in[1], out[0], p[3], &key[3]);
            put_tri(a, slot++, p, key, ipt, opt);
            float p2[3][3];
            int64_t k2[3] = {key[0], key[2], key[3]};
            for (int d = 0; d < 3; ++d) { p2[0][d] = p[0][d]; p2[1][d] = p[2][d]; p2[2][d] = p[3][d]; }
            put_tri(a, slot++, p2, k2, ipt, opt);
        }
    }
}

// ---- marching cubes with PyMCubes' vertex and triangle numbering (vdn_mesh_mc_*: include/vdn_render.h) ------------------------
// The library the reference calls (renderer.py:36) walks the cells sequentially, x-major with z innermost, gives every cut lattice
// edge ONE vertex - created by the first visited cell that contains the edge - and numbers vertices in creation order. In parallel
// form: the count pass stores each cell's case, the number of vertices it creates and its triangle count; the caller's exclusive
// prefix sums over cells in visiting order ARE the sequential numbering; the emit pass writes each cell's vertices at its offset in
// the library's in-cell creation order and resolves a triangle corner on edge e through the cell that owns e: offset[owner] + the
// rank of e among the vertices the owner creates.
__device__ __constant__ unsigned char kMcCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
__device__ __constant__ unsigned char kMcEdge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};
// in-cell creation order (marchingcubes.h: 0x040, 0x020, 0x400, then the shared edges where no earlier cell exists)
__device__ __constant__ unsigned char kMcOrder[12] = {6, 5, 10, 0, 1, 2, 3, 4, 7, 8, 9, 11};

// does cell (i, j, k) create the vertex of its edge e (i.e. is it the first visited cell that contains that lattice edge)?
__device__ inline bool mc_creates(int e, int i, int j, int k) {
    switch (e) {
        case 6: case 5: case 10: return true;
        case 0: return j == 0 && k == 0;
        case 1: case 2: return k == 0;
        case 3: return i == 0 && k == 0;
        case 4: case 9: return j == 0;
        case 7: case 11: return i == 0;
        default: return i == 0 && j == 0;      // 8
    }
}
__device__ inline int mc_edge_mask(int cube) {
    int m = 0;
#pragma unroll
    for (int e = 0; e < 12; ++e)
        if (((cube >> kMcEdge[e][0]) ^ (cube >> kMcEdge[e][1])) & 1) m |= 1 << e;
    return m;
}
__device__ inline int mc_case(const VdnMeshMcArgs& a, int i, int j, int k, double* v) {
    int cube = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const float f = a.u[((long)(i + kMcCorner[m][0]) * a.R + (j + kMcCorner[m][1])) * a.R + (k + kMcCorner[m][2])];
        if (v != nullptr) v[m] = (double)f;
        if ((double)f <= a.isovalue) cube |= 1 << m;          // marchingcubes.h: `if(v[m] <= isovalue) cubeindex |= 1<<m`, in double
    }
    return cube;
}

__global__ void mesh_mc_count_kernel(VdnMeshMcArgs a) {
    const long n = a.R - 1;
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= n * n * n) return;
    const int k = (int)(cell % n), j = (int)((cell / n) % n), i = (int)(cell / (n * n));
    const int cube = mc_case(a, i, j, k, nullptr);
    const int edges = mc_edge_mask(cube);
    int nv = 0, nt = 0;
#pragma unroll
    for (int e = 0; e < 12; ++e)
        if (((edges >> e) & 1) && mc_creates(e, i, j, k)) ++nv;
    for (int t = 0; t < 16 && kMcTri[cube][t] >= 0; t += 3) ++nt;
    a.cube_case[cell] = (unsigned char)cube;
    a.n_verts[cell] = nv;
    a.n_tris[cell] = nt;
}

__global__ void mesh_mc_emit_kernel(VdnMeshMcArgs a) {
    const long n = a.R - 1;
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= n * n * n) return;
    const int cube = a.cube_case[cell];
    const int edges = mc_edge_mask(cube);
    if (edges == 0) return;
    const int k = (int)(cell % n), j = (int)((cell / n) % n), i = (int)(cell / (n * n));
    double v[8];
    mc_case(a, i, j, k, v);
    // this cell's own vertices, in the library's creation order; interpolated FROM corner a TO corner b of the edge, in double:
    // (x_b - x_a) * (isovalue - f_a) / (f_b - f_a) + x_a, the midpoint when f_a == f_b  (mc_isovalue_interpolation)
    long vo = a.vert_offsets[cell];
    for (int o = 0; o < 12; ++o) {
        const int e = kMcOrder[o];
        if (!((edges >> e) & 1) || !mc_creates(e, i, j, k)) continue;
        const int ca = kMcEdge[e][0], cb = kMcEdge[e][1];
        double p[3] = {(double)(i + kMcCorner[ca][0]), (double)(j + kMcCorner[ca][1]), (double)(k + kMcCorner[ca][2])};
        const int ax = kMcCorner[ca][0] != kMcCorner[cb][0] ? 0 : (kMcCorner[ca][1] != kMcCorner[cb][1] ? 1 : 2);
        const double x1 = p[ax], x2 = (double)((ax == 0 ? i : (ax == 1 ? j : k)) + kMcCorner[cb][ax]);
        const double f1 = v[ca], f2 = v[cb];
        p[ax] = f2 == f1 ? (x2 + x1) / 2.0 : (x2 - x1) * (a.isovalue - f1) / (f2 - f1) + x1;
        a.vertices[vo * 3 + 0] = p[0]; a.vertices[vo * 3 + 1] = p[1]; a.vertices[vo * 3 + 2] = p[2];
        ++vo;
    }
    // triangles: the vertex of edge e lives with the first visited cell that contains the lattice edge
    long to = a.tri_offsets[cell];
    for (int t = 0; t < 16 && kMcTri[cube][t] >= 0; ++t) {
        const int e = kMcTri[cube][t];
        const int ca = kMcEdge[e][0], cb = kMcEdge[e][1];
        const int ax = kMcCorner[ca][0] != kMcCorner[cb][0] ? 0 : (kMcCorner[ca][1] != kMcCorner[cb][1] ? 1 : 2);
        // lattice position of the edge's lower end
        const int lx = i + min(kMcCorner[ca][0], kMcCorner[cb][0]), ly = j + min(kMcCorner[ca][1], kMcCorner[cb][1]), lz = k + min(kMcCorner[ca][2], kMcCorner[cb][2]);
        // owner: along the edge's axis the cell index is fixed; across it, the lower neighbour where one exists
        const int oi = ax == 0 ? lx : max(lx - 1, 0), oj = ax == 1 ? ly : max(ly - 1, 0), ok = ax == 2 ? lz : max(lz - 1, 0);
        const int dx = lx - oi, dy = ly - oj, dz = lz - ok;
        // the owner's number for that edge
        int oe;
        if (ax == 0) oe = dy == 0 ? (dz == 0 ? 0 : 4) : (dz == 0 ? 2 : 6);
        else if (ax == 1) oe = dx == 0 ? (dz == 0 ? 3 : 7) : (dz == 0 ? 1 : 5);
        else oe = dx == 0 ? (dy == 0 ? 8 : 11) : (dy == 0 ? 9 : 10);
        const long ocell = ((long)oi * n + oj) * n + ok;
        const int oedges = mc_edge_mask(a.cube_case[ocell]);
        int rank = 0;
        for (int o = 0; o < 12 && kMcOrder[o] != oe; ++o) {
            const int e2 = kMcOrder[o];
            if (((oedges >> e2) & 1) && mc_creates(e2, oi, oj, ok)) ++rank;
        }
        a.triangles[to * 3 + t] = a.vert_offsets[ocell] + rank;
    }
}

}  // namespace vdn

static int mesh_check(const VdnMeshArgs* a) {
    if (a == nullptr || a->u == nullptr || a->R < 2) return -1;
    return 0;
}

extern "C" int vdn_mesh_count(const VdnMeshArgs* a, void* stream) {
    if (mesh_check(a) != 0 || a->counts == nullptr) return -1;
    const long n = (long)(a->R - 1) * (a->R - 1) * (a->R - 1);
    hipLaunchKernelGGL(vdn::mesh_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int vdn_mesh_mc_count(const VdnMeshMcArgs* a, void* stream) {
    if (a == nullptr || a->u == nullptr || a->R < 2 || a->cube_case == nullptr || a->n_verts == nullptr || a->n_tris == nullptr) return -1;
    const long n = (long)(a->R - 1) * (a->R - 1) * (a->R - 1);
    hipLaunchKernelGGL(vdn::mesh_mc_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int vdn_mesh_mc_emit(const VdnMeshMcArgs* a, void* stream) {
    if (a == nullptr || a->u == nullptr || a->R < 2 || a->cube_case == nullptr || a->vert_offsets == nullptr || a->tri_offsets == nullptr ||
        a->vertices == nullptr || a->triangles == nullptr) return -1;
    const long n = (long)(a->R - 1) * (a->R - 1) * (a->R - 1);
    hipLaunchKernelGGL(vdn::mesh_mc_emit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int vdn_mesh_emit(const VdnMeshArgs* a, void* stream) {
    if (mesh_check(a) != 0 || a->offsets == nullptr || a->tri_pos == nullptr || a->tri_key == nullptr) return -1;
    const long n = (long)(a->R - 1) * (a->R - 1) * (a->R - 1);
    hipLaunchKernelGGL(vdn::mesh_emit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}
