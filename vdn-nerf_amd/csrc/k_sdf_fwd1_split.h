// SDF network forward + analytic input-gradient sweep WITH the training saves (reference dpt_models/fields.py:72-108), for the
// TAIL of a training step's foreground work list, on gfx950.
//
// k_sdf_fwd2.h's MODE 1 runs one 128-point workgroup per CU (its 8-bit softplus' store takes the whole register file and LDS), so
// a launch costs whole rounds of 256 workgroups: the bench step's ~37 K rows are 290 workgroups = one full round + 34 workgroups
// that cost a second full round (~75 us) for 12 % of the work. This kernel takes the rows behind the last full round with the
// FEATURES split over the waves, as k_sdf_fwd0_split.h does for the sampler's passes: one workgroup = 32 rows, 8 waves, wave w
// computes output tile w of every one of the 17 layer steps (16 MFMAs per step), activations meet in LDS between steps (B-fragment
// order, ping-pong buffers, one barrier per step), weights go straight from L2 into registers two steps ahead. A pass is a latency
// chain of 17 short steps (~1/3 of the large kernel's pass), and 4 K rows are 128 workgroups on 128 CUs.
//
//  * softplus' of tile w of layer l is produced by wave w in the forward half and consumed by wave w in the sweep (the sweep's
//    output tile w of W_{l+1}^T v_{l+1} is multiplied by sigma_l tile w): it never leaves the wave - 8 x 16 B per lane in a
//    wave-private LDS strip;
//  * the same weight stream ("full"), the same arithmetic as MODE 1 operation for operation - accumulator from the bias (forward)
//    or zero (sweep), k-steps in order, softplus / sigma in scaled units with the same 8-bit packing, the f32 sdf row as one fma
//    chain in MODE 1's order (wave 0, from LDS), the encoding's adjoint by one wave in MODE 1's order - so every plane (H, V, PE,
//    feature), sdf and normal comes out bit-identical to the large kernel's (tests/test_gpu_parity.py).
#pragma once
#include "k_sdf_fwd2.h"

namespace vdn {
namespace sdf1s {

constexpr int kWaves = 8;
constexpr int kPeb = 0;                         // encoded input, 4 k-steps x 1 KiB (layer 0's input; k-steps 14..17 of layer 4)
constexpr int kBuf0 = 4 * 1024;                 // activations, 16 k-steps x 1 KiB each, ping-pong
constexpr int kBuf1 = kBuf0 + 16 * 1024;
constexpr int kW8 = kBuf1 + 16 * 1024;          // row 0 of the last layer, 256 f32
constexpr int kG = kW8 + 1024;                  // layer 7's activations in f32: [tile][q][lane] x 16 B (the f32 sdf row)
constexpr int kSig = kG + 8 * 4 * 1024;         // 255 sigma: [layer 8][wave 8][lane] x 16 B, wave-private
constexpr int kLds = kSig + 8 * 8 * 1024;       // 133 KiB

using PG = sdf2::Prog<1>;
using sdf2::kStride;
using sdf2::kTail;
using sdf2::kC1;
using sdf2::kVSave;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LI>
struct StepIO {
    static constexpr sdf2::LayerDesc L = PG::layer(LI);
    static constexpr int kt = L.kt, nt = L.nt, ns = 2 * L.kt;
    // LDS byte offset of k-step s of step LI's input
    static constexpr int in_off(int s) {
        if (LI == 0) return kPeb + s * 1024;
        if (LI == 4 && s >= 14) return kPeb + (s - 14) * 1024;
        return (((LI - 1) & 1) ? kBuf1 : kBuf0) + s * 1024;     // step LI-1 wrote buffer (LI-1) & 1
    }
    static constexpr int out_base = (LI & 1) ? kBuf1 : kBuf0;
    static constexpr bool bias = L.kind <= sdf2::LAST;
};

struct WSet {
    bf16x8 w[18];
    f32x4 b[4];
};

// the chunk of (step LI, tile) -> registers
template <int LI>
VDN_DEV void load_weights(WSet& W, const char* blob, int tile, int lane) {
    using IO = StepIO<LI>;
    const char* ch = blob + (long)(PG::first_chunk(LI) + tile) * kStride;
    const bf16x8* wa = reinterpret_cast<const bf16x8*>(ch) + lane;
    static_for<IO::ns>([&](auto s_c) VDN_INL { W.w[decltype(s_c)::value] = wa[decltype(s_c)::value * 64]; });
    if constexpr (IO::bias) {
        const f32x4* bb = reinterpret_cast<const f32x4*>(ch + IO::kt * 2048);
#pragma unroll
        for (int q = 0; q < 4; ++q) W.b[q] = bb[2 * q + (lane >> 5)];
    }
}
// the tile wave w computes in step LI: its own where the step has it (layer 3 has 7: wave 7 recomputes tile 6 and drops it);
// the two encoding tiles of the skip layer's sweep (7, then 8) and of W0^T (0, then 1) are wave 7's
template <int LI>
VDN_DEV int tile_of_wave(int wave) {
    constexpr int nt = PG::layer(LI).nt;
    if constexpr (PG::layer(LI).kind == sdf2::SWEEP_PE) return 0;
    if constexpr (PG::layer(LI).kind == sdf2::SWEEP_SKIP) return wave;           // 0..7 (tile 8 follows on wave 7)
    return wave < nt ? wave : nt - 1;
}

VDN_DEV void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int LI>
VDN_DEV f32x16 step_mma(const WSet& W, const char* smem, int lane) {
    using IO = StepIO<LI>;
    constexpr int NS = IO::ns;
    constexpr int PRE = NS < 6 ? NS : 6;
    f32x16 acc;
    if constexpr (IO::bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[4 * q + 0] = W.b[q][0]; acc[4 * q + 1] = W.b[q][1]; acc[4 * q + 2] = W.b[q][2]; acc[4 * q + 3] = W.b[q][3];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
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

// rows [row0 + 32 b, row0 + 32 b + 32) of the work list, b = blockIdx.x - only when the list ends within max_rows behind row0
// (then sdf2::sdf_fwd2_kernel leaves those rows alone: VdnSdfArgs.tail_row0 / tail_max_rows)
template <bool SAVE>
__global__ __launch_bounds__(kWaves * 64, 1) void sdf_fwd1_split_kernel(SdfArgs a) {
    using P = BF16;
    using ST = unsigned short;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const long n_rows = a.active_idx != nullptr ? (long)*a.n_active : (long)a.P;
    const long first = (long)a.tail_row0 + 32L * blockIdx.x;
    if (n_rows <= a.tail_row0 || n_rows - a.tail_row0 > a.tail_max_rows || first >= n_rows) return;
    const long raw = first + c;
    const bool ok = raw < n_rows;
    const long p = ok ? raw : n_rows - 1;                                   // row of the saves (compact)
    const long pd = a.active_idx != nullptr ? (long)a.active_idx[p] : p;    // dense point id

    WSet WA, WB;                                    // even / odd steps
    load_weights<0>(WA, a.blob, tile_of_wave<0>(wave), lane);
    load_weights<1>(WB, a.blob, tile_of_wave<1>(wave), lane);
    __builtin_amdgcn_sched_barrier(0);

    float xin[3];
    long sdf_idx = pd;
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
    ST* Hs = reinterpret_cast<ST*>(a.H);
    ST* Vs = reinterpret_cast<ST*>(a.V);
    ST* feat = reinterpret_cast<ST*>(a.feat);
    const long PS = P::plane(a.P, 256);
    const long prow = (p >> 5) * (32L * 256) + h * 256 + (p & 31) * 8;       // PT32 offset of this lane's 16-byte pieces (mlp_engine.h)
    const float inv_scale = 1.0f / a.scale;
    float b0 = 0.0f;
    char* sig = smem + kSig + wave * 1024 + lane * 16;          // + l * 8 KiB: this lane's 16 bytes of 255 sigma_l (tile = wave)

    if (wave < 2) {                                 // the encoded input: wave 0 writes tile 0, wave 1 tile 1
        float pe39[39], pe[64];
        posenc<3, 6, false>(xin, pe39);
        // scaled units and the bf16 residue slots, exactly as k_sdf_fwd2.h forms them
#pragma unroll
        for (int i = 0; i < 39; ++i) pe[i] = pe39[i] * kC1;
#pragma unroll
        for (int i = 0; i < 25; ++i) pe[39 + i] = fmaf(pe39[i], kC1, -bf16_lo(pack_bf16x2(pe[i], 0.0f)));      // (the residue of the exact product)
        typename P::template Act<1> X;
        X.set(0, wave == 0 ? vals_tile<64>(pe, h, 0) : vals_tile<64>(pe, h, 1));
        *reinterpret_cast<bf16x8*>(smem + kPeb + (2 * wave) * 1024 + lane * 16) = X.r[0];
        *reinterpret_cast<bf16x8*>(smem + kPeb + (2 * wave + 1) * 1024 + lane * 16) = X.r[1];
        if constexpr (SAVE) {
            if (a.PE != nullptr) {                  // the residue slots are saved as zeros (the weight-gradient GEMM contracts over the 39 encoded values)
#pragma unroll
                for (int i = 39; i < 64; ++i) pe[i] = 0.0f;
                P::store_tile(reinterpret_cast<ST*>(a.PE), p, 64, wave, h, wave == 0 ? vals_tile<64>(pe, h, 0) : vals_tile<64>(pe, h, 1), true);
            }
        }
        // the sdf row's bias: row 0 of the last layer's 9th chunk (k_sdf_fwd2.h reads it from that chunk's bias block)
        if (wave == 0) b0 = *reinterpret_cast<const float*>(a.blob + (long)(PG::first_chunk(8) + 8) * kStride + 8 * 2048);
    } else if (wave == 2) {                         // row 0 of W8 (f32, in every chunk's tail) -> LDS
        const f32x4 v = *(reinterpret_cast<const f32x4*>(a.blob + kTail) + lane);
        *reinterpret_cast<f32x4*>(smem + kW8 + lane * 16) = v;
    }
    lds_barrier();

    f32x16 UPE[2];              // wave 7: d sdf / d(PE) tiles (W4^T rows 7, 8 and W0^T)
    float n[3] = {0.0f, 0.0f, 0.0f};
    auto pe_backward = [&]() VDN_INL {              // n += J_PE^T u  (k_sdf_fwd2.h: the same expressions in the same order)
        float u[39];
        tiles_vals<39, 2>(UPE, h, u);
#pragma unroll
        for (int d = 0; d < 3; ++d) n[d] += u[d];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float f = (float)(1 << k);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float sn, co;
                sincos_pe<false>(xin[d] * f, sn, co);
                n[d] += f * (co * u[3 + 6 * k + d] - sn * u[3 + 6 * k + 3 + d]);
            }
        }
    };

    auto step = [&](auto li_c, WSet& W) VDN_INL {
        constexpr int LI = decltype(li_c)::value;
        using IO = StepIO<LI>;
        constexpr sdf2::LayerDesc L = IO::L;
        constexpr bool has_next2 = LI + 2 < PG::NL;
        const int T = tile_of_wave<LI>(wave);
        // the last step is wave 7's alone (both tiles of W0^T)
        if constexpr (L.kind == sdf2::SWEEP_PE) {
            if (wave != 7) return;
        }
        f32x16 acc = step_mma<LI>(W, smem, lane);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (L.kind == sdf2::SWEEP_SKIP || L.kind == sdf2::SWEEP_PE) {
            if (wave == 7) {                         // the second encoding tile with the same register set
                UPE[0] = acc;
                load_weights<LI>(W, a.blob, L.kind == sdf2::SWEEP_SKIP ? 8 : 1, lane);
                __builtin_amdgcn_sched_barrier(0);
                UPE[1] = step_mma<LI>(W, smem, lane);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (has_next2) load_weights<LI + 2>(W, a.blob, tile_of_wave<LI + 2>(wave), lane);      // this set is free again
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (L.kind == sdf2::HID) {
            u32x4 o[2], sq;
            float e_hold0 = 0.0f, e_hold1 = 0.0f;
            f32x4 gq;
#pragma unroll
            for (int pr = 0; pr < 8; ++pr) {
                const sdf2::SpE sp0 = sdf2::softplus_sigma(acc[2 * pr]), sp1 = sdf2::softplus_sigma(acc[2 * pr + 1]);
                o[pr >> 2][pr & 3] = pack_bf16x2(sp0.g, sp1.g);
                if constexpr (L.l == 7) {
                    gq[2 * (pr & 1)] = sp0.g;
                    gq[2 * (pr & 1) + 1] = sp1.g;
                    if (pr & 1) *reinterpret_cast<f32x4*>(smem + kG + (wave * 4 + (pr >> 1)) * 1024 + lane * 16) = gq;
                }
                if ((pr & 1) == 0) {
                    e_hold0 = sp0.e;
                    e_hold1 = sp1.e;
                } else {
                    sq[pr >> 1] = sdf2::sigma255_pack(e_hold0, e_hold1, sp0.e, sp1.e);
                }
            }
            if (wave < IO::nt) {
                *reinterpret_cast<u32x4*>(smem + IO::out_base + (2 * wave) * 1024 + lane * 16) = o[0];
                *reinterpret_cast<u32x4*>(smem + IO::out_base + (2 * wave + 1) * 1024 + lane * 16) = o[1];
                *reinterpret_cast<u32x4*>(sig + L.l * 8 * 1024) = sq;
                if constexpr (SAVE) {
                    sdf2::plane_store16(Hs + L.l * PS + prow + T * 1024, o[0]);
                    sdf2::plane_store16(Hs + L.l * PS + prow + T * 1024 + 512, o[1]);
                }
            }
        } else if constexpr (L.kind == sdf2::LAST) {
            // feature tile T to HBM; v7 tile T = (W8 row 0 / scale) (.) 255 sigma_7 -> the sweep's first input
            const u32x4 sq7 = *reinterpret_cast<const u32x4*>(sig + 7 * 8 * 1024);
            u32x4 of[2], ov[2], vs[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                of[q >> 1][2 * (q & 1)] = pack_bf16x2(acc[4 * q], acc[4 * q + 1]);
                of[q >> 1][2 * (q & 1) + 1] = pack_bf16x2(acc[4 * q + 2], acc[4 * q + 3]);
                const f32x4 w = *reinterpret_cast<const f32x4*>(smem + kW8 + (8 * T + 2 * q + h) * 16);
                const unsigned sw = sq7[q];
                const float v0 = w[0] * inv_scale * sdf2::ubyte_f32(sw, 0), v1 = w[1] * inv_scale * sdf2::ubyte_f32(sw, 1);
                const float v2 = w[2] * inv_scale * sdf2::ubyte_f32(sw, 2), v3 = w[3] * inv_scale * sdf2::ubyte_f32(sw, 3);
                ov[q >> 1][2 * (q & 1)] = pack_bf16x2(v0, v1);
                ov[q >> 1][2 * (q & 1) + 1] = pack_bf16x2(v2, v3);
                vs[q >> 1][2 * (q & 1)] = pack_bf16x2(v0 * kVSave, v1 * kVSave);
                vs[q >> 1][2 * (q & 1) + 1] = pack_bf16x2(v2 * kVSave, v3 * kVSave);
            }
            sdf2::plane_store16(feat + prow + T * 1024, of[0]);
            sdf2::plane_store16(feat + prow + T * 1024 + 512, of[1]);
            *reinterpret_cast<u32x4*>(smem + IO::out_base + (2 * wave) * 1024 + lane * 16) = ov[0];
            *reinterpret_cast<u32x4*>(smem + IO::out_base + (2 * wave + 1) * 1024 + lane * 16) = ov[1];
            if constexpr (SAVE) {
                sdf2::plane_store16(Vs + 7 * PS + prow + T * 1024, vs[0]);
                sdf2::plane_store16(Vs + 7 * PS + prow + T * 1024 + 512, vs[1]);
            }
        } else if constexpr (L.kind == sdf2::SWEEP || L.kind == sdf2::SWEEP_SKIP) {
            // v_l tile = u (.) 255 sigma_l   (the 1/255 is in the next transposed image)
            if (L.kind == sdf2::SWEEP || wave < 7) {
                const u32x4 sq = *reinterpret_cast<const u32x4*>(sig + L.l * 8 * 1024);
                u32x4 ov[2], vs[2];
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) {
                    const float v0 = acc[2 * pr] * sdf2::ubyte_f32(sq[pr >> 1], 2 * (pr & 1));
                    const float v1 = acc[2 * pr + 1] * sdf2::ubyte_f32(sq[pr >> 1], 2 * (pr & 1) + 1);
                    ov[pr >> 2][pr & 3] = pack_bf16x2(v0, v1);
                    vs[pr >> 2][pr & 3] = pack_bf16x2(v0 * kVSave, v1 * kVSave);
                }
                *reinterpret_cast<u32x4*>(smem + IO::out_base + (2 * wave) * 1024 + lane * 16) = ov[0];
                *reinterpret_cast<u32x4*>(smem + IO::out_base + (2 * wave + 1) * 1024 + lane * 16) = ov[1];
                if constexpr (SAVE) {
                    sdf2::plane_store16(Vs + L.l * PS + prow + T * 1024, vs[0]);
                    sdf2::plane_store16(Vs + L.l * PS + prow + T * 1024 + 512, vs[1]);
                }
            } else {
                pe_backward();
            }
        } else {                                    // SWEEP_PE (wave 7)
            pe_backward();
        }
        if constexpr (LI + 1 < PG::NL) lds_barrier();
    };
    step(std::integral_constant<int, 0>{}, WA);
    step(std::integral_constant<int, 1>{}, WB);
    step(std::integral_constant<int, 2>{}, WA);
    step(std::integral_constant<int, 3>{}, WB);
    step(std::integral_constant<int, 4>{}, WA);
    step(std::integral_constant<int, 5>{}, WB);
    step(std::integral_constant<int, 6>{}, WA);
    step(std::integral_constant<int, 7>{}, WB);
    step(std::integral_constant<int, 8>{}, WA);
    step(std::integral_constant<int, 9>{}, WB);
    step(std::integral_constant<int, 10>{}, WA);
    step(std::integral_constant<int, 11>{}, WB);
    step(std::integral_constant<int, 12>{}, WA);
    step(std::integral_constant<int, 13>{}, WB);
    step(std::integral_constant<int, 14>{}, WA);
    step(std::integral_constant<int, 15>{}, WB);
    if (wave == 7) {
        step(std::integral_constant<int, 16>{}, WA);
        if (ok && h == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) a.normals[pd * 3 + d] = n[d] * a.scale;
        }
    } else if (wave == 0) {
        // sdf = W8[0,:] . h8 + b8[0]: MODE 1's f32 chain (k_sdf_fwd2.h, layer 7's epilogue), tile by tile, pair by pair
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
        const float sdf = fmaf(dot, 1.0f / kC1, b0) * inv_scale;
        if (ok && h == 0) a.sdf[sdf_idx] = sdf;
    }
}

template <bool SAVE>
int launch(const VdnSdfArgs* args, hipStream_t stream) {
    static bool once = (allow_big_lds(sdf_fwd1_split_kernel<SAVE>, kLds), true);
    (void)once;
    const int grid = (args->tail_max_rows + 31) / 32;
    hipLaunchKernelGGL((sdf_fwd1_split_kernel<SAVE>), dim3(grid), dim3(kWaves * 64), kLds, stream, *args);
    return (int)hipGetLastError();
}

}  // namespace sdf1s
}  // namespace vdn
