// Weight-gradient GEMM for the training step on gfx950, fp32 MFMA (v_mfma_f32_32x32x2_f32):
// dW[m,n] = sum over points of A[p,m] * B[p,n]. The contraction runs over 65 536+ points, the
// output is at most 288 x 352, so the K dimension is split across workgroups (deterministic partial
// slabs + a finalize pass that also maps image coordinates back to the parameter layout and
// applies weight-norm's backward). Replaces autograd's mm-backward of reference fields.py.
#include "vdn_common.h"
#include "vdn_kernels.h"

namespace vdn {

// workgroup = 4 waves = 128 x 128 outputs (wave: 2 x 2 MFMA tiles of 32 x 32) over one K split.
// Operand panels ([32 points][128 features] fp32, row-major like the planes) are copied global -> LDS with the
// async 16-byte-per-lane DMA (one wave instruction = two 512-byte rows), double-buffered: the next stage is in flight
// while the current one is multiplied; a lane's MFMA operand (k = h, feature = c) is one ds_read_b32.
constexpr int kF32StagePts = 32;
constexpr int kF32PanelBytes = kF32StagePts * 128 * 4;          // 16 KiB per operand per stage

__global__ __launch_bounds__(256, 2) void dw_gemm_f32_kernel(const DwDesc* descs, int n_desc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [buffer(2)][operand(2)][32][128] fp32
    const int wg = blockIdx.x;
    int di = 0;
    while (di + 1 < n_desc && descs[di + 1].wg_begin <= wg) ++di;
    const DwDesc d = descs[di];
    const int local = wg - d.wg_begin;
    const int mt4 = (d.m_tiles + 3) / 4, nt4 = max((d.n_tiles + 3) / 4, 1);
    // XCD-aware id space: workgroup ids are dealt round-robin over the 8 XCDs (observed; used for speed only), so the
    // tiles of one K split - which read the same A/B panels - get ids that are equal mod 8 and adjacent in time: the
    // second reader of an operand panel finds it in that XCD's L2. wg_begin is a multiple of 8 (vdn_dw_entry_wgs_f32).
    const int ntile = mt4 * nt4;
    const int slot = local & 7, round = local >> 3;
    const int split = slot + 8 * (round / ntile);
    const int tile = round % ntile;
    if (split >= d.splits) return;
    const int tm = tile / nt4, tn = tile % nt4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = tm * 4 + wm * 2, n0 = tn * 4 + wn * 2;      // first MFMA tile of this wave
    const bool mv0 = m0 < d.m_tiles, mv1 = m0 + 1 < d.m_tiles;
    const bool nv0 = n0 < d.n_tiles, nv1 = n0 + 1 < d.n_tiles;
    const bool two = d.A2 != nullptr;
    const int seg_splits = two ? d.splits / 2 : d.splits;       // two segments: first half of the splits = segment 1
    const bool seg2 = two && split >= seg_splits;
    const int s_in = seg2 ? split - seg_splits : split;
    const long P = d.P_dev != nullptr ? min((long)d.P, (long)*d.P_dev) : (long)d.P;
    long per = (P + seg_splits - 1) / seg_splits;
    per = (per + kF32StagePts - 1) / kF32StagePts * kF32StagePts;
    const long k_begin = (long)s_in * per, k_end = min(k_begin + per, P);
    const float* A = reinterpret_cast<const float*>(seg2 ? d.A2 : d.A1);
    const float* Bm = reinterpret_cast<const float*>(seg2 ? d.B2 : d.B1);
    const int lda = seg2 ? d.lda2 : d.lda1, ldb = seg2 ? d.ldb2 : d.ldb1;

    // loader role: waves 0,1 copy operand A (16 rows each), waves 2,3 operand B. Columns beyond the operand's last tile
    // are redirected to column 0 (their products are never stored), rows beyond P to row P-1 (masked at the MFMA).
    const bool load_b = wave >= 2;
    const float* src = load_b ? Bm : A;
    const int ld = load_b ? ldb : lda;
    const int ncols = (load_b ? d.n_tiles : d.m_tiles) * 32;
    int col = (load_b ? tn : tm) * 128 + c * 4;
    if (col >= ncols) col = 0;
    const bool have_src = src != nullptr && ncols > 0;
    auto issue = [&](long kbase, int buf) VDN_INL {
        if (!have_src) return;
        char* dst = smem + buf * 2 * kF32PanelBytes + (load_b ? kF32PanelBytes : 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r0 = (wave & 1) * 16 + 2 * i;                   // this instruction copies rows r0, r0 + 1
            const long row = min(kbase + r0 + h, P - 1);
            glds16(src + row * ld + col, dst + r0 * 512);
        }
    };
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float cs0 = 0.0f, cs1 = 0.0f;
    const bool do_colsum = d.colsum != nullptr && tn == 0 && wn == 0 && !seg2;
    const long n_stages = (k_end - k_begin + kF32StagePts - 1) / kF32StagePts;
    if (n_stages > 0) issue(k_begin, 0);
    for (long t = 0; t < n_stages; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                  // stage t landed for every wave; stage t-1 fully consumed
        if (t + 1 < n_stages) issue(k_begin + (t + 1) * kF32StagePts, (t + 1) & 1);
        const float* pa = reinterpret_cast<const float*>(smem + (t & 1) * 2 * kF32PanelBytes);
        const float* pb = pa + kF32PanelBytes / 4;
        const long kb = k_begin + t * kF32StagePts;
#pragma unroll
        for (int ks = 0; ks < kF32StagePts / 2; ++ks) {
            const int r = 2 * ks + h;
            const bool valid = kb + r < k_end;
            float a0 = pa[r * 128 + wm * 64 + c], a1 = pa[r * 128 + wm * 64 + 32 + c];
            float b0 = pb[r * 128 + wn * 64 + c], b1 = pb[r * 128 + wn * 64 + 32 + c];
            if (!valid) { a0 = 0.0f; a1 = 0.0f; b0 = 0.0f; b1 = 0.0f; }
            if (do_colsum) {
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
    if (d.colsum != nullptr && tn == 0 && wn == 0) {
        cs0 += __shfl_xor(cs0, 32);
        cs1 += __shfl_xor(cs1, 32);
        if (h == 0) {
            if (mv0) d.colsum[(long)split * M + m0 * 32 + c] = cs0;      // zero for segment-2 splits
            if (mv1) d.colsum[(long)split * M + (m0 + 1) * 32 + c] = cs1;
        }
    }
}

// Reduce the K splits and scatter to the parameter layout: one block = 4 image rows of one descriptor, one thread = 4
// consecutive image columns (16-byte loads), 8 splits in flight per thread. The slabs are ~150 MB per step: this is a
// bandwidth kernel, and what it needs is loads in flight (the first version, one 4-byte load chain per thread and 128-thread
// blocks, reached 2.9 TB/s). The summation order over s is fixed (8 interleaved partial sums, then a fixed tree): deterministic.
__global__ __launch_bounds__(256) void dw_finalize_kernel(const DwFinalizeDesc* descs, int phase) {
    const DwFinalizeDesc d = descs[blockIdx.x];
    if (d.accumulate != phase) return;
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= d.M) return;
    const int r = d.rmap[i];
    if (r < 0) return;
    const int lane = threadIdx.x & 63;
    if (d.target != nullptr) {
        const long step = (long)d.M * d.N;
        for (int j = 4 * lane; j < d.N; j += 256) {          // N is a multiple of 32
            const float* col = d.slab + (long)i * d.N + j;
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            int s = 0;
            for (; s + 8 <= d.splits; s += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += *reinterpret_cast<const f32x4*>(col + (s + k) * step);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (s + k < d.splits) v[k] += *reinterpret_cast<const f32x4*>(col + (s + k) * step);
            const f32x4 t4 = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cc = d.cmap[j + e];
                if (cc < 0) continue;
                float val = d.scale * t4[e];
                if (d.xsum != nullptr && r == d.xrow) {
                    float x = 0.0f;
                    for (int s2 = 0; s2 < d.xsplits; ++s2) x += d.xsum[(long)s2 * d.xM + cc];
                    val += d.xscale * x;
                }
                float* t = d.target + (long)r * d.t_stride + cc;
                *t = d.accumulate ? *t + val : val;
            }
        }
    }
    if (d.btarget != nullptr) {
        // the row's wave sums the splits (lanes strided over s, then a fixed shuffle tree: deterministic); the variance gradient
        // arrives here as a 512-"split" column sum, which one lane alone would stretch the whole launch by
        float v = 0.0f;
        for (int s = lane; s < d.splits; s += 64) v += d.colsum[(long)s * d.M + i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) {
            float* t = d.btarget + r;
            *t = d.accumulate ? *t + d.bscale * v : d.bscale * v;
        }
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

extern "C" int vdn_dw_entry_wgs_f32(int m_tiles, int n_tiles, int splits) {
    const int mt4 = (m_tiles + 3) / 4, nt4 = n_tiles > 0 ? (n_tiles + 3) / 4 : 1;
    return 8 * ((splits + 7) / 8) * mt4 * nt4;          // a multiple of 8: see the id space in dw_gemm_f32_kernel
}

extern "C" int vdn_dw_gemm_f32(const VdnDwDesc* descs_dev, int n_desc, int total_wgs, void* stream) {
    if (!descs_dev || n_desc <= 0 || total_wgs <= 0) return -1;
    static bool once = (vdn::allow_big_lds(vdn::dw_gemm_f32_kernel, 4 * vdn::kF32PanelBytes), true);
    (void)once;
    hipLaunchKernelGGL(vdn::dw_gemm_f32_kernel, dim3(total_wgs), dim3(256), 4 * vdn::kF32PanelBytes, (hipStream_t)stream, descs_dev, n_desc);
    return (int)hipGetLastError();
}

extern "C" int vdn_dw_finalize(const VdnDwFinalizeDesc* descs_dev, int n_desc, int max_M, int phase, void* stream) {
    if (!descs_dev || n_desc <= 0 || max_M <= 0 || phase < 0 || phase > 1) return -1;
    hipLaunchKernelGGL(vdn::dw_finalize_kernel, dim3(n_desc, (max_M + 3) / 4), dim3(256), 0, (hipStream_t)stream, descs_dev, phase);
    return (int)hipGetLastError();
}

extern "C" int vdn_weightnorm_bwd(const VdnWeightNormBwdDesc* descs_dev, int n_layers, int max_rows, void* stream) {
    if (!descs_dev || n_layers <= 0 || max_rows <= 0) return -1;
    hipLaunchKernelGGL(vdn::weightnorm_bwd_kernel, dim3(n_layers, (max_rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, descs_dev);
    return (int)hipGetLastError();
}
