// Weight-gradient GEMM on gfx950 with bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
// dW[m,n] = sum over points of A[p,m] * B[p,n], A/B = bf16 activation planes in the tile-blocked PT32 layout
// written by the forward / backward chains (mlp_engine.h: points in blocks of 32; within a (block, 32-feature tile):
// [k(2)][h(2)][point(32)][16 B], the 16-byte unit (k, h, point) = two 8-byte granules of 4 consecutive features of that
// point: feature quads fq = 4k + h and fq = 4k + h + 2, feature = 32*tile + 4*fq + e).
//
// The op is HBM-bound: 2*256*256 flop per 1 KiB of operand rows = 128 flop/B, i.e. ~0.8 PFLOP/s at the achievable
// 6.3 TB/s - a third of the MFMA peak. So the kernel is built around reading every operand byte exactly once and doing
// nothing else per byte:
//  * one workgroup (8 waves) owns a K split of a whole 256 x 256 output block (waves 4(M) x 2(N), 64 x 128 each, 128
//    accumulator registers): no second reader of any panel, no dependence on L2 hits;
//  * planes go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no registers, no VALU): a stage is one block of 32
//    points x up to 16 feature tiles (32 KiB), NBUF stages form a ring with ONE barrier per stage;
//  * the contraction index (points) is the slow index of both operands while an MFMA lane wants 8 k-values of one
//    feature: ds_read_b64_tr_b16 does that transpose inside the LDS read. A DMA lane writes LDS at base + 16*lane but
//    chooses its own global address, so the LDS image of a tile is re-ordered to [point quad(8)][k(2)][h(2)][point in quad(4)]
//    [16 B]: the 4 points x 32 features a half-wave's transposing read touches are then 256 contiguous bytes
//    (conflict-free), while a lane's 16-byte global piece is one unit of the plane layout (4 lanes = 4 consecutive
//    points = 64 contiguous bytes). The k order inside a fragment is the same for A and B, so the contraction is unchanged;
//  * bias gradients (column sums of A) come from the matrix core too: one extra MFMA per m-tile against a fragment of
//    ones instead of unpacking bf16 pairs on the VALU.
// K splits are reduced by vdn_dw_finalize (train_dw_f32.hip), deterministically.
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kDwTileBytes = 2048;                 // 32 points x 32 features
constexpr int kDwStageBytes = 16 * kDwTileBytes;   // 8 A tiles, then 8 B tiles
#ifndef VDN_DW_BUF
#define VDN_DW_BUF 4
#endif
constexpr int kDwBuf = VDN_DW_BUF;                 // stages in the ring (3 in flight while one is multiplied; 5 = all 160 KiB of LDS: A/B arm)
constexpr int kDwWaves = 8;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// the compiler cannot prove a pointer computed from a descriptor it looked up in a loop is wave-uniform: tell it
VDN_DEV const char* dw_uniform(const char* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}

VDN_DEV void dw_glds16(const char* base_uniform, unsigned lane_off, unsigned lds_wave_base) {
#if VDN_DW_LD_NT == 2           // (development A/B: system-coherent as well)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 sc1 nt" ::"v"(lane_off), "s"(lds_wave_base), "s"(base_uniform) : "memory", "m0");
#elif VDN_DW_LD_NT              // non-temporal: the operand planes are read exactly once (vdn_common.h)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 nt" ::"v"(lane_off), "s"(lds_wave_base), "s"(base_uniform) : "memory", "m0");
#else
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(lane_off), "s"(lds_wave_base), "s"(base_uniform) : "memory", "m0");
#endif
}

// s_waitcnt vmcnt(n) with a wave-uniform runtime n (the immediate must be a constant)
VDN_DEV void dw_wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// 8 k-values (points 8h .. 8h+7 of a 16-point k-step) of this lane's feature: two transposing reads of 4 points each
VDN_DEV bf16x8 dw_frag(const lds_char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 256));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

VDN_DEV int dw_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

__global__ __launch_bounds__(512, 1) void dw_gemm_bf16_kernel(const DwDesc* descs, int n_desc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wg = blockIdx.x;
    int di = 0;
    while (di + 1 < n_desc && descs[di + 1].wg_begin <= wg) ++di;
    const DwDesc& d = descs[dw_uni(di)];
    // everything that steers control flow is made explicitly wave-uniform (scalar registers, scalar branches): the
    // transposing LDS reads need EXEC all ones, and the compiler cannot see that values loaded through the descriptor
    // (and through P_dev) are the same in every lane
    const int m_tiles = dw_uni(d.m_tiles), n_tiles = dw_uni(d.n_tiles), splits = dw_uni(d.splits);
    const int local = wg - dw_uni(d.wg_begin);
    const int mg = (m_tiles + 7) / 8, ng = max((n_tiles + 7) / 8, 1);
    const int split = local / (mg * ng), blk = local % (mg * ng);
    if (split >= splits) return;
    const int tm = blk / ng, tn = blk % ng;
    const int wave = dw_uni(threadIdx.x >> 6), lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // this block's tiles of the operands
    const int a_t0 = tm * 8, b_t0 = tn * 8;
    const int nA = min(8, m_tiles - a_t0), nB = max(0, min(8, n_tiles - b_t0));
    // segment / K range of this split: with two segments the first half of the splits covers segment 1
    const bool two = d.A2 != nullptr;
    const int seg_splits = two ? splits / 2 : splits;
    const bool seg2 = two && split >= seg_splits;
    const int s_in = seg2 ? split - seg_splits : split;
    int P = dw_uni(d.P);
    if (d.P_dev != nullptr) P = min(P, dw_uni(*d.P_dev));
    int per = (P + seg_splits - 1) / seg_splits;
    per = (per + 127) / 128 * 128;
    const int k_begin = s_in * per, k_end = min(k_begin + per, P);
    const char* A = dw_uniform(reinterpret_cast<const char*>(seg2 ? d.A2 : d.A1));
    const char* Bm = dw_uniform(reinterpret_cast<const char*>(seg2 ? d.B2 : d.B1));
    const long lda = dw_uni(seg2 ? d.lda2 : d.lda1), ldb = dw_uni(seg2 ? d.ldb2 : d.ldb1);
    const int n_stages = k_end > k_begin ? (k_end - k_begin + 31) >> 5 : 0;
    float* slab = d.slab;
    float* colsum = d.colsum;

    // ---- DMA: piece u = (tile u>>1, half u&1) of the stage, 1 KiB each; wave w moves pieces w, w+8, w+16, w+24
    const int n_pieces = 2 * (nA + nB);
    const int npw = (n_pieces - wave + kDwWaves - 1) / kDwWaves;             // pieces of this wave per stage (wave-uniform)
    // lane -> unit of the plane tile: point quad (lane >> 4) of this half tile, k = bit 3, h = bit 2, point in quad = bits 1:0
    const unsigned lane_off = ((lane >> 3) & 1) * 1024 + ((lane >> 2) & 1) * 512 + ((lane >> 4) * 4 + (lane & 3)) * 16;
    const unsigned lds0 = (unsigned)(size_t)(lds_char*)smem;
    auto issue_stage = [&](int s) VDN_INL {
        const long blk32 = (k_begin >> 5) + s;
        const unsigned buf = lds0 + (s % kDwBuf) * kDwStageBytes;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = wave + i * kDwWaves;
            if (u < n_pieces) {
                const int t = u >> 1, half = u & 1;
                const bool isA = t < nA;
                const char* src = isA ? A + blk32 * (64 * lda) + (long)(a_t0 + t) * 2048
                                      : Bm + blk32 * (64 * ldb) + (long)(b_t0 + t - nA) * 2048;
                const int slot = isA ? t : 8 + (t - nA);
                dw_glds16(src + half * 256, lane_off, buf + slot * kDwTileBytes + half * 1024);      // half = points 16 half ..
            }
        }
    };

    // ---- fragment addressing: group g = lane>>4 (j = g&1: feature half, h = g>>1: k half), i = lane&15 = 4q'+p
    const int gi = lane & 15, qp = gi >> 2, pp = gi & 3, jj = (lane >> 4) & 1;
    // granule (point 8h + qp, feature quad fq = 4jj + pp) of a 16-point k-step: point quad 2h (the second read: 2h + 1), unit
    // (k = jj, h' = pp & 1, point in quad = qp), half of the unit pp >> 1
    const unsigned frag_off = (2 * h) * 256 + (jj * 8 + (pp & 1) * 4 + qp) * 16 + (pp >> 1) * 8;
    const lds_char* lbase = (lds_char*)smem + frag_off;

    f32x16 acc[2][4];
    f32x16 cs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        cs[i] = f32x16{0};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x16{0};
    }
    const bool do_colsum = colsum != nullptr && tn == 0 && wn == 0 && !seg2;
    // valid tiles of this wave (bit i: m-tile 2wm+i, bit j: n-tile 4wn+j)
    const int mvm = (2 * wm < nA ? 1 : 0) | (2 * wm + 1 < nA ? 2 : 0);
    const int nvm = (4 * wn < nB ? 1 : 0) | (4 * wn + 1 < nB ? 2 : 0) | (4 * wn + 2 < nB ? 4 : 0) | (4 * wn + 3 < nB ? 8 : 0);
    const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

    // FULL: all 2 x 4 tiles of this wave exist (the common 256 x 256 block): no per-tile branches in the k-loop
    auto run = [&](auto full_c) VDN_INL {
        constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
        for (int s = 0; s < kDwBuf - 1; ++s)
            if (s < n_stages) issue_stage(s);
        for (int s = 0; s < n_stages; ++s) {
            // this wave's pieces of stage s have landed once at most the pieces of the stages issued after it are outstanding
            const int later = min(n_stages - 1 - s, kDwBuf - 2);
            dw_wait_vm(npw * later);
            asm volatile("s_barrier" ::: "memory");   // everyone's pieces of stage s are in LDS; stage s-1 is fully consumed
            if (s + kDwBuf - 1 < n_stages) issue_stage(s + kDwBuf - 1);       // into the buffer stage s-1 used
            if (s == n_stages - 1 && (k_end & 31)) {
                // the last block is partial: rows from k_end on hold whatever the producers' padding left there
                const int valid = k_end & 31;
                char* buf = smem + (s % kDwBuf) * kDwStageBytes;
                for (int u = threadIdx.x; u < 16 * 256; u += 512) {          // 8-byte granules: [tile][point quad(8)][unit(16)][2]
                    const int pt = ((u >> 5) & 7) * 4 + ((u >> 1) & 3);
                    if (pt >= valid) *reinterpret_cast<unsigned long long*>(buf + u * 8) = 0ull;
                }
                __syncthreads();
            }
            const lds_char* st = lbase + (s % kDwBuf) * kDwStageBytes;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[2], b[4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    if (FULL || (mvm >> i & 1)) a[i] = dw_frag(st + (2 * wm + i) * kDwTileBytes + ks * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (FULL || (nvm >> j & 1)) b[j] = dw_frag(st + (8 + 4 * wn + j) * kDwTileBytes + ks * 1024);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (!FULL && !(mvm >> i & 1)) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (FULL || (nvm >> j & 1)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
                    if (do_colsum) cs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], ones, cs[i], 0, 0, 0);
                }
            }
        }
    };
    if (mvm == 3 && nvm == 15) run(std::true_type{});
    else run(std::false_type{});

    const int M = m_tiles * 32, N = n_tiles * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!(mvm >> i & 1)) continue;
        const int mt = a_t0 + 2 * wm + i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!(nvm >> j & 1)) continue;
            float* base = slab + ((long)split * M + mt * 32) * N + (b_t0 + 4 * wn + j) * 32 + c;
#pragma unroll
            for (int t = 0; t < 16; ++t) base[(long)rho(t, h) * N] = acc[i][j][t];
        }
        // every column of the ones-product holds the same sums: lane column 0 of each half writes its 16 rows
        if (colsum != nullptr && tn == 0 && wn == 0 && c == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t) colsum[(long)split * M + mt * 32 + rho(t, h)] = cs[i][t];      // zero for segment-2 splits
        }
    }
}

}  // namespace vdn

extern "C" int vdn_dw_entry_wgs_bf16(int m_tiles, int n_tiles, int splits) {
    const int mg = (m_tiles + 7) / 8, ng = n_tiles > 0 ? (n_tiles + 7) / 8 : 1;
    return splits * mg * ng;
}

extern "C" int vdn_dw_gemm_bf16(const VdnDwDesc* descs_dev, int n_desc, int total_wgs, void* stream) {
    using namespace vdn;
    if (!descs_dev || n_desc <= 0 || total_wgs <= 0) return -1;
    static bool once = (allow_big_lds(dw_gemm_bf16_kernel, kDwBuf * kDwStageBytes), true);
    (void)once;
    hipLaunchKernelGGL(dw_gemm_bf16_kernel, dim3(total_wgs), dim3(512), kDwBuf * kDwStageBytes, (hipStream_t)stream, descs_dev, n_desc);
    return (int)hipGetLastError();
}
