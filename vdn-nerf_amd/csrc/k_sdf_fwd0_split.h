// SDF value only (reference dpt_models/fields.py:72-105, SDFNetwork.sdf) for SMALL point sets on gfx950: the up-sampling
// passes of the hierarchical sampler (renderer.py:352-372) evaluate 512 rays x 16 new samples = 8 192 points per pass,
// three passes in a row, each waiting for the previous one.
//
// k_sdf_fwd2.h's MODE 0 gives every wave 32 points and ALL 256 output features of every layer: a workgroup's latency is
// 8 hidden layers x 128 MFMAs per wave (~31 us as measured, issue-bound) no matter how few workgroups there are, and at
// 8 192 points only 64 of the 256 CUs have one. This kernel splits the FEATURES over the waves instead: one workgroup =
// 32 points, 8 waves, wave w computes output tile w (32 features) of every layer - 16 MFMAs per wave per layer - and the
// activations meet in LDS between layers (one barrier per layer, ping-pong buffers). 8 192 points = 256 workgroups = one
// per CU.
//
//  * weights never touch LDS: the chunk of (layer, tile) is read by exactly one wave, whose lane (i,h) wants the chunk's
//    16 bytes [k-step][lane] as they lie (mlp_engine.h, BF16 chunk format) - plain coalesced global loads into registers,
//    two layers ahead (two register sets: even / odd layers; a set is refilled as soon as its layer's MFMAs are issued).
//  * the same weight stream ("sdf2", vdn_hip/images.py) and the same arithmetic as MODE 0, operation for operation: bias
//    into the accumulator, k-steps in order, softplus in scaled units, bf16 packing of the hidden activations, and the
//    last layer's sdf row as ONE f32 fma chain over the unrounded activations of layer 7 in MODE 0's order (wave 0 runs
//    it from LDS). The two kernels return bit-identical values (tests/test_gpu_parity.py).
//
// ROUNDS = true appends the rest of an up-sampling round (renderer.py:372-386) to the pass: with 16 new samples per ray a
// workgroup's 32 points are exactly two rays, so waves 0 and 1 go on to merge their ray's new samples (z and the sdf values
// just computed, handed over in LDS) into its sorted row and to draw the next round's samples from it - vdn_merge_upsample's
// work (k_ray_rows.h: the same device functions, the same bits) without its launch and its trip through HBM.
#pragma once
#include "k_sdf_fwd2.h"
#include "k_ray_rows.h"

namespace vdn {
namespace sdf0s {

constexpr int kWaves = 8;
constexpr int kPeb = 0;                         // encoded input, 4 k-steps x 1 KiB  (layer 0's input; k-steps 14..17 of layer 4)
constexpr int kBuf0 = 4 * 1024;                 // hidden activations, 16 k-steps x 1 KiB each, ping-pong
constexpr int kBuf1 = kBuf0 + 16 * 1024;
constexpr int kW8 = kBuf1 + 16 * 1024;          // row 0 of the last layer, 256 f32
constexpr int kG = kW8 + 1024;                  // layer 7's activations in f32: [tile][q][lane] x 16 B
constexpr int kLds = kG + 8 * 4 * 1024;

using PG = sdf2::Prog<0>;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int L>
struct LayerIO {
    static constexpr int kt = PG::layer(L).kt, nt = PG::layer(L).nt, ns = 2 * kt;
    // LDS byte offset of k-step s of layer L's input
    static constexpr int in_off(int s) {
        if (L == 0) return kPeb + s * 1024;
        if (L == 4 && s >= 14) return kPeb + (s - 14) * 1024;
        return ((L & 1) ? kBuf0 : kBuf1) + s * 1024;      // layer L-1 wrote buffer (L-1) & 1
    }
    static constexpr int out_base = (L & 1) ? kBuf1 : kBuf0;
};

struct WSet {
    bf16x8 w[18];
    f32x4 b[4];
};

// the chunk of (layer L, this wave's tile) -> registers
template <int L>
VDN_DEV void load_weights(WSet& W, const char* blob, int tile, int lane) {
    constexpr int nt = PG::layer(L).nt, kt = PG::layer(L).kt;
    const int t = tile < nt ? tile : nt - 1;            // layer 3 has 7 tiles: wave 7 recomputes tile 6 and drops it
    const char* ch = blob + (long)(PG::first_chunk(L) + t) * sdf2::kStride;
    const bf16x8* wa = reinterpret_cast<const bf16x8*>(ch) + lane;
    static_for<2 * kt>([&](auto s_c) VDN_INL { W.w[decltype(s_c)::value] = wa[decltype(s_c)::value * 64]; });
    const f32x4* bb = reinterpret_cast<const f32x4*>(ch + kt * 2048);
#pragma unroll
    for (int q = 0; q < 4; ++q) W.b[q] = bb[2 * q + (lane >> 5)];
}

VDN_DEV void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int L>
VDN_DEV f32x16 layer_mma(const WSet& W, const char* smem, int lane) {
    using IO = LayerIO<L>;
    constexpr int NS = IO::ns;
    constexpr int PRE = NS < 6 ? NS : 6;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        acc[4 * q + 0] = W.b[q][0]; acc[4 * q + 1] = W.b[q][1]; acc[4 * q + 2] = W.b[q][2]; acc[4 * q + 3] = W.b[q][3];
    }
    bf16x8 x[NS];
    static_for<NS>([&](auto s_c) VDN_INL {
        constexpr int s = decltype(s_c)::value;
        x[s] = *reinterpret_cast<const bf16x8*>(smem + IO::in_off(s) + lane * 16);
    });
    static_for<NS>([&](auto s_c) VDN_INL {
        constexpr int s = decltype(s_c)::value;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.w[s], x[s], acc, 0, 0, 0);
    });
    // PRE fragment reads up front, then one read per MFMA (mlp_engine.h, BF16::mma)
    __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
    static_for<NS - PRE>([&](auto) VDN_INL {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    });
    __builtin_amdgcn_sched_group_barrier(0x008, PRE, 0);
    return acc;
}

constexpr int kRows = kG;                       // ROUNDS: per-ray scratch rows (4 x kMaxT floats per ray) where layer 7's activations were
constexpr int kNewSdf = kW8;                    // ROUNDS: the 32 new sdf values, where the last layer's row was

template <bool ROUNDS = false, int VID = 0>
__global__ __launch_bounds__(kWaves * 64, 1) void sdf_fwd0_split_kernel(SdfArgs a, MergeArgs mg, UpsampleArgs up) {
    using P = BF16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const WorkRow wr = work_row(a.active_idx, a.n_active, a.P, 1, 0, c);
    if (wr.none) return;
    if (a.cold_start) {         // (vdn_common.h; wave-uniform condition: every wave takes the barrier)
        warm_l2_issue(a.blob, PG::total * sdf2::kStride, wr.n_wg, 256, smem + wave * 1024);
        warm_l2_sync();
    }

    WSet WA, WB;                                    // even / odd layers
    load_weights<0>(WA, a.blob, wave, lane);
    load_weights<1>(WB, a.blob, wave, lane);
    __builtin_amdgcn_sched_barrier(0);

    const long pd = wr.point;
    long sdf_idx = pd;
    float b0 = 0.0f;
    if (wave < 2) {                                 // the encoded input: wave 0 writes tile 0, wave 1 tile 1
        float xin[3];
        if (a.pts != nullptr) {
#pragma unroll
            for (int d = 0; d < 3; ++d) xin[d] = a.pts[pd * 3 + d] * a.scale;
        } else {
            const long r = pd / a.n_per_ray;
            const long sidx = pd - r * a.n_per_ray;
            const float z = a.z[r * a.z_ld + sidx];
            sdf_idx = r * a.sdf_ld + sidx;
#pragma unroll
            for (int d = 0; d < 3; ++d) xin[d] = (a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z) * a.scale;
        }
        float pe39[39], pe[64];
        posenc<3, 6, false>(xin, pe39);
        // scaled units and the bf16 residue slots, exactly as k_sdf_fwd2.h forms them
#pragma unroll
        for (int i = 0; i < 39; ++i) pe[i] = pe39[i] * sdf2::kC1;
#pragma unroll
        for (int i = 0; i < 25; ++i) pe[39 + i] = fmaf(pe39[i], sdf2::kC1, -bf16_lo(pack_bf16x2(pe[i], 0.0f)));      // (the residue of the exact product)
        typename P::template Act<1> X;
        X.set(0, wave == 0 ? vals_tile<64>(pe, h, 0) : vals_tile<64>(pe, h, 1));
        *reinterpret_cast<bf16x8*>(smem + kPeb + (2 * wave) * 1024 + lane * 16) = X.r[0];
        *reinterpret_cast<bf16x8*>(smem + kPeb + (2 * wave + 1) * 1024 + lane * 16) = X.r[1];
        if (wave == 0) b0 = *reinterpret_cast<const float*>(a.blob + (long)PG::first_chunk(8) * sdf2::kStride + 8 * 2048);
    } else if (wave == 2) {                         // row 0 of W8 (f32, in every chunk's tail) -> LDS
        const f32x4 v = *(reinterpret_cast<const f32x4*>(a.blob + sdf2::kTail) + lane);
        *reinterpret_cast<f32x4*>(smem + kW8 + lane * 16) = v;
    }
    lds_barrier();

    auto hidden = [&](auto l_c, WSet& W) VDN_INL {
        constexpr int L = decltype(l_c)::value;
        using IO = LayerIO<L>;
        const f32x16 acc = layer_mma<L>(W, smem, lane);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (L + 2 <= 7) load_weights<L + 2>(W, a.blob, wave, lane);      // this set is free again
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (L < 7) {
            u32x4 o[2];
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                const float g0 = sdf2::softplus_sigma(acc[2 * pr]).g, g1 = sdf2::softplus_sigma(acc[2 * pr + 1]).g;
                o[pr >> 2][pr & 3] = pack_bf16x2(g0, g1);
            }
            if (wave < IO::nt) {
                *reinterpret_cast<u32x4*>(smem + IO::out_base + (2 * wave) * 1024 + lane * 16) = o[0];
                *reinterpret_cast<u32x4*>(smem + IO::out_base + (2 * wave + 1) * 1024 + lane * 16) = o[1];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 g;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = sdf2::softplus_sigma(acc[4 * q + j]).g;
                *reinterpret_cast<f32x4*>(smem + kG + (wave * 4 + q) * 1024 + lane * 16) = g;
            }
        }
        lds_barrier();
    };
    hidden(std::integral_constant<int, 0>{}, WA);
    hidden(std::integral_constant<int, 1>{}, WB);
    hidden(std::integral_constant<int, 2>{}, WA);
    hidden(std::integral_constant<int, 3>{}, WB);
    hidden(std::integral_constant<int, 4>{}, WA);
    hidden(std::integral_constant<int, 5>{}, WB);
    hidden(std::integral_constant<int, 6>{}, WA);
    hidden(std::integral_constant<int, 7>{}, WB);

    if (wave > (ROUNDS ? 1 : 0)) return;
    if (wave == 0) {
    // sdf = W8[0,:] . h8 + b8[0]: MODE 0's f32 chain (k_sdf_fwd2.h, layer 7's epilogue), tile by tile, pair by pair
    float sdf_dot = 0.0f;
#pragma unroll
    for (int T = 0; T < 8; ++T) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w8 = *reinterpret_cast<const f32x4*>(smem + kW8 + (8 * T + 2 * q + h) * 16);
            const f32x4 g = *reinterpret_cast<const f32x4*>(smem + kG + (T * 4 + q) * 1024 + lane * 16);
            sdf_dot = fmaf(g[0], w8[0], fmaf(g[1], w8[1], sdf_dot));
            sdf_dot = fmaf(g[2], w8[2], fmaf(g[3], w8[3], sdf_dot));
        }
    }
    const float dot = sdf_dot + __shfl_xor(sdf_dot, 32);
    const float sdf = fmaf(dot, 1.0f / sdf2::kC1, b0) * (1.0f / a.scale);
    if (wr.ok && h == 0) a.sdf[sdf_idx] = sdf;
    if constexpr (ROUNDS) {
        if (h == 0) *reinterpret_cast<float*>(smem + kNewSdf + c * 4) = sdf;
    }
    }
    if constexpr (ROUNDS) {
        lds_barrier();                              // waves 0 and 1 (the others have left): the new sdf values are in LDS, G is free
        const int r = blockIdx.x * 2 + wave;        // point c of this workgroup = ray c / 16, new sample c % 16
        if (r >= mg.B) return;
        float* rows = reinterpret_cast<float*>(smem + kRows) + wave * 4 * kMaxT;
        float *za = rows, *zb = rows + kMaxT, *lz = rows + 2 * kMaxT, *ls = rows + 3 * kMaxT;
        merge_row(mg, r, lane, za, zb, lz, ls, nullptr, reinterpret_cast<const float*>(smem + kNewSdf) + wave * 16);
        __builtin_amdgcn_wave_barrier();
        upsample_row(up, r, lane, mg.M + mg.K, lz, ls, za);     // the old row's scratch serves as the cdf row
    }
}

template <bool ROUNDS = false, int VID = 0>
int launch(const VdnSdfArgs* args, hipStream_t stream, const VdnMergeArgs* mg = nullptr, const VdnUpsampleArgs* up = nullptr) {
    static bool once = (allow_big_lds(sdf_fwd0_split_kernel<ROUNDS, VID>, kLds), true);
    (void)once;
    static_assert(kRows + 2 * 4 * kMaxT * 4 <= kLds && 32 * 4 <= 1024, "ROUNDS scratch fits the regions it reuses");
    const int grid = (args->P + 31) / 32;
    hipLaunchKernelGGL((sdf_fwd0_split_kernel<ROUNDS, VID>), dim3(grid), dim3(kWaves * 64), kLds, stream, *args,
                       mg != nullptr ? *mg : VdnMergeArgs{}, up != nullptr ? *up : VdnUpsampleArgs{});
    return (int)hipGetLastError();
}

}  // namespace sdf0s
}  // namespace vdn
