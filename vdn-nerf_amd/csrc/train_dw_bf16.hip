// Weight-gradient GEMM on gfx950 with bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
// dW[m,n] = sum over points of A[p,m] * B[p,n], A/B = bf16 activation planes in the tile-blocked PT32 layout
// written by the backward chains (mlp_engine.h: points in blocks of 32; within a (block, 32-feature tile):
// [q(4)][hh(2)][point(32)][e(4)], feature = 32*tile + 8q + 4hh + e).
//
// The contraction index (points) is the slow index of both operands, while an MFMA lane needs 8 k-values of ONE
// feature. Loader: one wave-instruction reads 1 KiB contiguous = half a (block, tile): lane (q2, hh, cpair) gets
// 2 points x 4 features; the same lane position in 4 consecutive blocks gives 8 points x 4 features, which a
// 16-bit interleave turns into four ready MFMA fragments (k order = (block, point) - identical for A and B, so
// the contraction is unchanged). Fragments go to LDS as [k-group(16)][feature(128)][16 B] (+16 B pad per group); k-group = cpair.
// Workgroup = 4 waves = 128 x 128 outputs (wave: 2 x 2 tiles of 32 x 32); stage = 128 points = 8 k-steps; the
// next stage's global loads are in flight (registers) while the current stage is multiplied out of LDS.
// K is split across workgroups; partial slabs are reduced by vdn_dw_finalize (train_dw_f32.hip), deterministically.
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kDwStagePts = 128;
// one operand: [16 k-groups][128 features][8 x bf16]; each k-group is padded by one 16-byte slot: a ds_write_b128 is
// served in groups of 8 lanes = 8 consecutive k-groups, which at a 2 KiB stride would all hit the same banks
constexpr int kDwGroupBytes = 128 * 16 + 16;
constexpr int kDwPanelBytes = 16 * kDwGroupBytes;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void dw_gemm_bf16_kernel(const DwDesc* descs, int n_desc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wg = blockIdx.x;
    int di = 0;
    while (di + 1 < n_desc && descs[di + 1].wg_begin <= wg) ++di;
    const DwDesc d = descs[di];
    const int local = wg - d.wg_begin;
    const int mt4 = (d.m_tiles + 3) / 4, nt4 = max((d.n_tiles + 3) / 4, 1);
    // XCD-aware mapping: workgroup ids are dealt round-robin over the 8 XCDs (observed; used for speed only), so
    // the tiles of one K split - which read the same A/B panels - are given ids that are equal mod 8 and adjacent in
    // time: the second reader finds the panel in that XCD's L2 instead of HBM. wg_begin is a multiple of 8.
    const int ntile = mt4 * nt4;
    const int slot = local & 7, round = local >> 3;
    const int split = slot + 8 * (round / ntile);
    const int tile = round % ntile;
    if (split >= d.splits) return;
    const int tm = tile / nt4, tn = tile % nt4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = tm * 4 + wm * 2, n0 = tn * 4 + wn * 2;
    const bool mv0 = m0 < d.m_tiles, mv1 = m0 + 1 < d.m_tiles;
    const bool nv0 = n0 < d.n_tiles, nv1 = n0 + 1 < d.n_tiles;
    // segment / K range of this split: with two segments the first half of the splits covers segment 1
    const bool two = d.A2 != nullptr;
    const int seg_splits = two ? d.splits / 2 : d.splits;
    const bool seg2 = two && split >= seg_splits;
    const int s_in = seg2 ? split - seg_splits : split;
    const long P = d.P_dev != nullptr ? min((long)d.P, (long)*d.P_dev) : (long)d.P;
    long per = (P + seg_splits - 1) / seg_splits;
    per = (per + kDwStagePts - 1) / kDwStagePts * kDwStagePts;
    const long k_begin = (long)s_in * per, k_end = min(k_begin + per, P);
    const unsigned short* A = reinterpret_cast<const unsigned short*>(seg2 ? d.A2 : d.A1);
    const unsigned short* Bm = reinterpret_cast<const unsigned short*>(seg2 ? d.B2 : d.B1);
    const int lda = seg2 ? d.lda2 : d.lda1, ldb = seg2 ? d.ldb2 : d.ldb1;

    // loader role: waves 0,1 -> operand A feature tiles {0,1},{2,3} of the 128-wide panel; waves 2,3 -> operand B
    const bool load_b = wave >= 2;
    const unsigned short* src = load_b ? Bm : A;
    const int ld = load_b ? ldb : lda;
    const int ntiles = load_b ? d.n_tiles : d.m_tiles;
    const int panel_t0 = (load_b ? tn : tm) * 4;
    const int q2 = lane >> 5, hh = (lane >> 4) & 1, cpair = lane & 15;
    u32x4 regs[2][2][4];       // [tile-in-pair][Q][block]
    auto load_stage = [&](long kbase) VDN_INL {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int t = panel_t0 + (wave & 1) * 2 + tl;
#pragma unroll
            for (int Q = 0; Q < 2; ++Q) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const long p0 = kbase + 32 * b + 2 * cpair;        // this lane's two points
                    u32x4 v = {0u, 0u, 0u, 0u};
                    if (t < ntiles && p0 < k_end) {
                        v = *reinterpret_cast<const u32x4*>(src + ((kbase >> 5) + b) * (32L * ld) + t * 1024 + (2 * Q + q2) * 256 + hh * 128 + cpair * 8);
                        if (p0 + 1 >= k_end) { v[2] = 0u; v[3] = 0u; }   // second point beyond the range (padding rows hold garbage)
                    }
                    regs[tl][Q][b] = v;
                }
            }
        }
    };
    auto store_stage = [&]() VDN_INL {
        char* panel = smem + (load_b ? kDwPanelBytes : 0);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
#pragma unroll
            for (int Q = 0; Q < 2; ++Q) {
                // dwords of a load: d0 = (pt0: e0,e1) d1 = (pt0: e2,e3) d2 = (pt1: e0,e1) d3 = (pt1: e2,e3)
                u32x4 f0, f1, f2, f3;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const u32x4 v = regs[tl][Q][b];
                    f0[b] = (v[0] & 0xFFFFu) | (v[2] << 16);
                    f1[b] = (v[0] >> 16) | (v[2] & 0xFFFF0000u);
                    f2[b] = (v[1] & 0xFFFFu) | (v[3] << 16);
                    f3[b] = (v[1] >> 16) | (v[3] & 0xFFFF0000u);
                }
                const int fl = ((wave & 1) * 2 + tl) * 32 + (2 * Q + q2) * 8 + hh * 4;     // panel-local feature of e = 0
                char* dst = panel + cpair * kDwGroupBytes + fl * 16;
                *reinterpret_cast<u32x4*>(dst) = f0;
                *reinterpret_cast<u32x4*>(dst + 16) = f1;
                *reinterpret_cast<u32x4*>(dst + 32) = f2;
                *reinterpret_cast<u32x4*>(dst + 48) = f3;
            }
        }
    };
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float cs0 = 0.0f, cs1 = 0.0f;
    const bool do_colsum = d.colsum != nullptr && tn == 0 && wn == 0 && !seg2;
    auto frag_sum = [](const bf16x8& f) VDN_INL {
        const u32x4 u = __builtin_bit_cast(u32x4, f);
        float s = 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m) s += bf16_lo(u[m]) + bf16_hi(u[m]);
        return s;
    };
    const long n_stages = (k_end - k_begin + kDwStagePts - 1) / kDwStagePts;
    if (n_stages > 0) load_stage(k_begin);
    for (long t = 0; t < n_stages; ++t) {
        __syncthreads();                         // previous stage fully consumed
        store_stage();
        __syncthreads();
        if (t + 1 < n_stages) load_stage(k_begin + (t + 1) * kDwStagePts);
        const char* pa = smem;
        const char* pb = smem + kDwPanelBytes;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int offa = (2 * ks + h) * kDwGroupBytes + (wm * 64 + c) * 16;
            const int offb = (2 * ks + h) * kDwGroupBytes + (wn * 64 + c) * 16;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(pa + offa);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(pa + offa + 32 * 16);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(pb + offb);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(pb + offb + 32 * 16);
            if (do_colsum) {
                cs0 += frag_sum(a0);
                cs1 += frag_sum(a1);
            }
            acc00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc11, 0, 0, 0);
        }
    }
    const int M = d.m_tiles * 32, N = d.n_tiles * 32;
    auto put = [&](const f32x16& acc, int mt, int nt) VDN_INL {
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

}  // namespace vdn

extern "C" int vdn_dw_gemm_bf16(const VdnDwDesc* descs_dev, int n_desc, int total_wgs, void* stream) {
    using namespace vdn;
    if (!descs_dev || n_desc <= 0 || total_wgs <= 0) return -1;
    static bool once = (allow_big_lds(dw_gemm_bf16_kernel, 2 * kDwPanelBytes), true);
    (void)once;
    hipLaunchKernelGGL(dw_gemm_bf16_kernel, dim3(total_wgs), dim3(256), 2 * kDwPanelBytes, (hipStream_t)stream, descs_dev, n_desc);
    return (int)hipGetLastError();
}
