// Weight-gradient GEMM on gfx950 with bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
// dW[m,n] = sum over points of A[p,m] * B[p,n], A/B = bf16 row-major activations written by the
// backward chains. The contraction index (points) is the ROW index of both operands, while an MFMA
// lane needs 8 consecutive k for one feature, so every tile is transposed on its way into LDS:
//   global (8 rows x 4 features per lane, 8-byte loads, 256 B contiguous per half-wave)
//   -> registers -> 16-bit interleave -> ds_write_b128 into [k-half][feature][8 x bf16]
//   -> ds_read_b128 = one ready MFMA fragment per lane.
// Workgroup = 4 waves = 128 x 128 outputs (wave: 2 x 2 tiles of 32 x 32), 64 points per LDS stage,
// two stages (global loads of stage t+1 are in flight while stage t is multiplied). K is split across
// workgroups; partial slabs are reduced by vdn_dw_finalize (train_dw_f32.hip), deterministically.
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kDwStagePts = 64;                 // points per stage = 4 k-steps of 16
constexpr int kDwPanelBytes = 4 * 4096;         // one operand, one stage: [4 k-steps][2 halves][128 features][16 B]
constexpr int kDwStageBytes = 2 * kDwPanelBytes;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void dw_gemm_bf16_kernel(const DwDesc* descs, int n_desc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    const int m0 = tm * 4 + wm * 2, n0 = tn * 4 + wn * 2;
    const bool mv0 = m0 < d.m_tiles, mv1 = m0 + 1 < d.m_tiles;
    const bool nv0 = n0 < d.n_tiles, nv1 = n0 + 1 < d.n_tiles;
    // segment / K range of this split: with two segments the first half of the splits covers segment 1
    const bool two = d.A2 != nullptr;
    const int seg_splits = two ? d.splits / 2 : d.splits;
    const bool seg2 = two && split >= seg_splits;
    const int s_in = seg2 ? split - seg_splits : split;
    const long P = d.P;
    long per = (P + seg_splits - 1) / seg_splits;
    per = (per + kDwStagePts - 1) / kDwStagePts * kDwStagePts;
    const long k_begin = (long)s_in * per, k_end = min(k_begin + per, P);
    const unsigned short* A = reinterpret_cast<const unsigned short*>(seg2 ? d.A2 : d.A1);
    const unsigned short* Bm = reinterpret_cast<const unsigned short*>(seg2 ? d.B2 : d.B1);
    const int lda = seg2 ? d.lda2 : d.lda1, ldb = seg2 ? d.ldb2 : d.ldb1;

    // loader role: waves 0,1 -> operand A, k-step pairs {0,1},{2,3}; waves 2,3 -> operand B likewise
    const bool load_b = wave >= 2;
    const int ks0 = (wave & 1) * 2;
    const unsigned short* src = load_b ? Bm : A;
    const int ld = load_b ? ldb : lda;
    const int col0 = (load_b ? tn : tm) * 128 + 4 * c;           // this lane's 4 features
    const bool col_ok = col0 < (load_b ? d.n_tiles : d.m_tiles) * 32 && (load_b ? d.n_tiles > 0 : true);
    uint2 regs[2][8];
    auto load_stage = [&](long kbase) VDN_INL {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const long row = kbase + (ks0 + b) * 16 + 8 * h + j;
                uint2 v = make_uint2(0u, 0u);
                if (col_ok && row < k_end) v = *reinterpret_cast<const uint2*>(src + row * ld + col0);
                regs[b][j] = v;
            }
        }
    };
    auto store_stage = [&](int buf) VDN_INL {
        char* panel = smem + buf * kDwStageBytes + (load_b ? kDwPanelBytes : 0);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            // features f0..f3 of this lane, 8 points each -> four 16-byte fragments
            u32x4 f0, f1, f2, f3;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const uint2 ra = regs[b][2 * m], rb = regs[b][2 * m + 1];
                f0[m] = (ra.x & 0xFFFFu) | (rb.x << 16);
                f1[m] = (ra.x >> 16) | (rb.x & 0xFFFF0000u);
                f2[m] = (ra.y & 0xFFFFu) | (rb.y << 16);
                f3[m] = (ra.y >> 16) | (rb.y & 0xFFFF0000u);
            }
            char* dst = panel + (ks0 + b) * 4096 + (h * 128 + 4 * c) * 16;
            *reinterpret_cast<u32x4*>(dst) = f0;
            *reinterpret_cast<u32x4*>(dst + 16) = f1;
            *reinterpret_cast<u32x4*>(dst + 32) = f2;
            *reinterpret_cast<u32x4*>(dst + 48) = f3;
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
    if (n_stages > 0) {
        load_stage(k_begin);
        store_stage(0);
    }
    __syncthreads();
    for (long t = 0; t < n_stages; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < n_stages) load_stage(k_begin + (t + 1) * kDwStagePts);
        const char* pa = smem + buf * kDwStageBytes;
        const char* pb = pa + kDwPanelBytes;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int offa = ks * 4096 + (h * 128 + wm * 64 + c) * 16;
            const int offb = ks * 4096 + (h * 128 + wn * 64 + c) * 16;
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
        if (t + 1 < n_stages) store_stage(buf ^ 1);
        __syncthreads();
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
    static bool once = (allow_big_lds(dw_gemm_bf16_kernel, 2 * kDwStageBytes), true);
    (void)once;
    hipLaunchKernelGGL(dw_gemm_bf16_kernel, dim3(total_wgs), dim3(256), 2 * kDwStageBytes, (hipStream_t)stream, descs_dev, n_desc);
    return (int)hipGetLastError();
}
