// Weight-gradient GEMM for the training step on gfx950, fp32 MFMA (v_mfma_f32_32x32x2_f32):
// dW[m,n] = sum over points of A[p,m] * B[p,n]. The contraction runs over 65 536+ points, the
// output is at most 288 x 352, so the K dimension is split across workgroups (deterministic partial
// slabs + a finalize pass that also maps image coordinates back to the parameter layout and
// applies weight-norm's backward). Replaces autograd's mm-backward of reference fields.py.
#include "vdn_common.h"
#include "vdn_kernels.h"

namespace vdn {

// workgroup = 4 waves = 128 x 128 outputs (wave: 2 x 2 MFMA tiles of 32 x 32) over one K split
__global__ __launch_bounds__(256) void dw_gemm_f32_kernel(const DwDesc* descs, int n_desc) {
    const int wg = blockIdx.x;
    int di = 0;
    while (di + 1 < n_desc && descs[di + 1].wg_begin <= wg) ++di;
    const DwDesc d = descs[di];
    const int local = wg - d.wg_begin;
    const int mt4 = (d.m_tiles + 3) / 4, nt4 = max((d.n_tiles + 3) / 4, 1);
    const int split = local / (mt4 * nt4);
    const int tile = local % (mt4 * nt4);
    const int tm = tile / nt4, tn = tile % nt4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = tm * 4 + wm * 2, n0 = tn * 4 + wn * 2;      // first MFMA tile of this wave
    const bool mv0 = m0 < d.m_tiles, mv1 = m0 + 1 < d.m_tiles;
    const bool nv0 = n0 < d.n_tiles, nv1 = n0 + 1 < d.n_tiles;
    const long P = d.P;
    const long Ktot = d.A2 != nullptr ? 2 * P : P;
    long per = (Ktot + d.splits - 1) / d.splits;
    per = (per + 1) & ~1L;
    const long k_begin = (long)split * per, k_end = min(k_begin + per, Ktot);

    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float cs0 = 0.0f, cs1 = 0.0f;
    const bool do_colsum = d.colsum != nullptr && tn == 0 && wn == 0;
#pragma unroll 4
    for (long kk = k_begin; kk < k_end; kk += 2) {
        const long pp = kk + h;
        const bool valid = pp < k_end;
        const bool seg2 = pp >= P;
        const long idx = seg2 ? pp - P : pp;
        const float* A = reinterpret_cast<const float*>(seg2 ? d.A2 : d.A1);
        const float* Bm = reinterpret_cast<const float*>(seg2 ? d.B2 : d.B1);
        const int lda = seg2 ? d.lda2 : d.lda1, ldb = seg2 ? d.ldb2 : d.ldb1;
        float a0 = 0.0f, a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
        if (valid) {
            if (mv0) a0 = A[idx * lda + m0 * 32 + c];
            if (mv1) a1 = A[idx * lda + (m0 + 1) * 32 + c];
            if (nv0) b0 = Bm[idx * ldb + n0 * 32 + c];
            if (nv1) b1 = Bm[idx * ldb + (n0 + 1) * 32 + c];
        }
        if (do_colsum && !seg2) {
            cs0 += a0;
            cs1 += a1;
        }
        if (nv0) {
            acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
        }
        if (nv1) {
            acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
    }
    const int M = d.m_tiles * 32, N = d.n_tiles * 32;
    auto put = [&](const f32x16& acc, int mt, int nt) {
        float* base = d.slab + ((long)split * M + mt * 32) * N + nt * 32 + c;
#pragma unroll
        for (int t = 0; t < 16; ++t) base[(long)rho(t, h) * N] = acc[t];
    };
    if (mv0 && nv0) put(acc00, m0, n0);
    if (mv0 && nv1) put(acc01, m0, n0 + 1);
    if (mv1 && nv0) put(acc10, m0 + 1, n0);
    if (mv1 && nv1) put(acc11, m0 + 1, n0 + 1);
    if (do_colsum) {
        cs0 += __shfl_xor(cs0, 32);
        cs1 += __shfl_xor(cs1, 32);
        if (h == 0) {
            if (mv0) d.colsum[(long)split * M + m0 * 32 + c] = cs0;
            if (mv1) d.colsum[(long)split * M + (m0 + 1) * 32 + c] = cs1;
        }
    }
}

// one block per (descriptor, image row)
__global__ void dw_finalize_kernel(const DwFinalizeDesc* descs, int phase) {
    const DwFinalizeDesc d = descs[blockIdx.x];
    if (d.accumulate != phase) return;
    const int i = blockIdx.y;
    if (i >= d.M) return;
    const int r = d.rmap[i];
    if (r < 0) return;
    if (d.target != nullptr) {
        for (int j = threadIdx.x; j < d.N; j += blockDim.x) {
            const int cc = d.cmap[j];
            if (cc < 0) continue;
            float v = 0.0f;
            for (int s = 0; s < d.splits; ++s) v += d.slab[((long)s * d.M + i) * d.N + j];
            float* t = d.target + (long)r * d.t_stride + cc;
            *t = d.accumulate ? *t + d.scale * v : d.scale * v;
        }
    }
    if (d.btarget != nullptr && threadIdx.x == 0) {
        float v = 0.0f;
        for (int s = 0; s < d.splits; ++s) v += d.colsum[(long)s * d.M + i];
        float* t = d.btarget + r;
        *t = d.accumulate ? *t + d.bscale * v : d.bscale * v;
    }
}

// one wave per row
__global__ void weightnorm_bwd_kernel(const WeightNormBwdDesc* descs) {
    const WeightNormBwdDesc d = descs[blockIdx.x];
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= d.rows) return;
    const float* v = d.v + (long)row * d.cols;
    const float* dw = d.dw_eff + (long)row * d.cols;
    float dot = 0.0f;
    for (int c = lane; c < d.cols; c += 64) dot += dw[c] * v[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    const float inv = d.inv_norm[row], g = d.g[row];
    float* dv = d.dv + (long)row * d.cols;
    const float k1 = g * inv, k2 = dot * inv * inv;
    for (int c = lane; c < d.cols; c += 64) dv[c] = k1 * (dw[c] - k2 * v[c]);
    if (lane == 0) d.dg[row] = dot * inv;
}

}  // namespace vdn

extern "C" int vdn_dw_gemm_f32(const VdnDwDesc* descs_dev, int n_desc, int total_wgs, void* stream) {
    if (!descs_dev || n_desc <= 0 || total_wgs <= 0) return -1;
    hipLaunchKernelGGL(vdn::dw_gemm_f32_kernel, dim3(total_wgs), dim3(256), 0, (hipStream_t)stream, descs_dev, n_desc);
    return (int)hipGetLastError();
}

extern "C" int vdn_dw_finalize(const VdnDwFinalizeDesc* descs_dev, int n_desc, int max_M, int phase, void* stream) {
    if (!descs_dev || n_desc <= 0 || max_M <= 0 || phase < 0 || phase > 1) return -1;
    hipLaunchKernelGGL(vdn::dw_finalize_kernel, dim3(n_desc, max_M), dim3(128), 0, (hipStream_t)stream, descs_dev, phase);
    return (int)hipGetLastError();
}

extern "C" int vdn_weightnorm_bwd(const VdnWeightNormBwdDesc* descs_dev, int n_layers, int max_rows, void* stream) {
    if (!descs_dev || n_layers <= 0 || max_rows <= 0) return -1;
    hipLaunchKernelGGL(vdn::weightnorm_bwd_kernel, dim3(n_layers, (max_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, descs_dev);
    return (int)hipGetLastError();
}
