// SDF value (reference dpt_models/fields.py:72-105, SDFNetwork.sdf) on gfx950, layer-wise engine.
//
// k_sdf_fwd2.h streams every weight chunk through LDS and every one of a workgroup's 4 waves reads the whole chunk back as
// its MFMA A operand: 1 KiB of LDS reads per MFMA, i.e. 128 B / clock / CU at the matrix pipe's peak rate - exactly the
// LDS's bandwidth - plus 5 LDS-DMA issues, a counted wait and a barrier per 16 MFMAs. This engine turns the roles around:
//
//  * a workgroup owns 128 points (4 blocks of 32) and walks the network LAYER BY LAYER. For one layer, wave w holds the
//    weights of output tiles 2w and 2w+1 in registers (read once per layer per workgroup with plain coalesced loads - the
//    chunk format of mlp_engine.h is already "16 bytes per lane per k-step" - one whole layer ahead of their use);
//  * the activations of all 128 points live in LDS in B-fragment order ([point block][k-step][lane] x 16 B, conflict-free,
//    the same 16-byte pieces as the PT32 planes), two buffers that alternate per layer. One B fragment read feeds two MFMAs
//    (the wave's two tiles): 0.5 KiB of LDS reads per MFMA;
//  * the unit of work is a STEP = (layer, point block): 2 x NS MFMAs in two independent accumulator chains, with the
//    epilogue (softplus, bf16 packing, LDS write-back) of the PREVIOUS step issued between them, also across layer
//    boundaries. One barrier per step: the tiles a step reads were written at least NPB-1 steps earlier.
//
// Arithmetic is k_sdf_fwd2.h's, operation for operation (same weight stream "sdf2", scaled units, bias as the chain's first
// addend, k-steps in order, f32 last-layer row from the unrounded activations as per-tile partial sums added in tile order).
#pragma once
#include "k_sdf_fwd2.h"

namespace vdn {
namespace sdflw {

constexpr int kWaves = 4;
constexpr int kNPB = 4;                                  // point blocks (of 32) per workgroup
constexpr int kPeb = 0;                                  // encoded input: [pb][4 k-steps] x 1 KiB
constexpr int kBuf0 = kPeb + kNPB * 4096;                // hidden activations: [pb][16 k-steps] x 1 KiB, two buffers
constexpr int kBuf1 = kBuf0 + kNPB * 16384;
constexpr int kW8 = kBuf1 + kNPB * 16384;                // row 0 of the last layer, 256 f32
constexpr int kBias = kW8 + 1024;                        // biases of the 8 hidden layers: [layer][256] f32
constexpr int kLds = kBias + 8 * 1024;
constexpr int kPart = kBuf1;                             // per-tile partial sums of the sdf row, [pb][8 tiles][64 lanes] f32: written by
                                                         // layer 7's epilogue, when nothing reads or writes buffer 1 any more
constexpr int kPre = 6;                                  // B fragments read ahead of their MFMAs
#ifndef VDN_LW_ABL
#define VDN_LW_ABL 0     // timing-only ablations (development): 1 no epilogue math, 2 no MFMA, 4 no step barrier, 8 no B-fragment reads
#endif

using PG = sdf2::Prog<0>;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int L>
struct IO {
    static constexpr int kt = PG::layer(L).kt, nt = PG::layer(L).nt, ns = 2 * kt;
    static constexpr int in_off(int pb, int s) {
        if (L == 0) return kPeb + pb * 4096 + s * 1024;
        if (L == 4 && s >= 14) return kPeb + pb * 4096 + (s - 14) * 1024;
        return ((L & 1) ? kBuf0 : kBuf1) + pb * 16384 + s * 1024;       // layer L-1 wrote buffer (L-1) & 1
    }
    static constexpr int out_off(int pb, int ks) { return ((L & 1) ? kBuf1 : kBuf0) + pb * 16384 + ks * 1024; }
};

struct WSet {
    bf16x8 w[2][18];
};

// the chunks of (layer L, tiles 2 wave, 2 wave + 1) -> registers
template <int L>
VDN_DEV void load_weights(WSet& W, const char* blob, int wave, int lane) {
    constexpr int nt = PG::layer(L).nt, kt = PG::layer(L).kt;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int tile = 2 * wave + j;
        const int t = tile < nt ? tile : nt - 1;        // layer 3 has 7 tiles: the 8th slot recomputes tile 6 and drops it
        const char* ch = blob + (long)(PG::first_chunk(L) + t) * sdf2::kStride;
        const bf16x8* wa = reinterpret_cast<const bf16x8*>(ch) + lane;
        static_for<2 * kt>([&](auto s_c) VDN_INL { W.w[j][decltype(s_c)::value] = wa[decltype(s_c)::value * 64]; });
    }
}

// bias block of tile T (32 f32 behind the chunk's weights) -> this lane's accumulator rows, from the LDS copy
template <int L>
VDN_DEV f32x16 bias_rows(const char* smem, int tile, int h) {
    const int t = tile < PG::layer(L).nt ? tile : PG::layer(L).nt - 1;
    const f32x4* bb = reinterpret_cast<const f32x4*>(smem + kBias + L * 1024 + t * 128);
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = bb[2 * q + h];
        r[4 * q] = v[0]; r[4 * q + 1] = v[1]; r[4 * q + 2] = v[2]; r[4 * q + 3] = v[3];
    }
    return r;
}

VDN_DEV void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int VID = 0>
__global__ __launch_bounds__(kWaves * 64, 1) void sdf_lw0_kernel(SdfArgs a) {
    using P = BF16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const WorkRow wr = work_row(a.active_idx, a.n_active, a.P, kWaves, wave, c);
    if (wr.none) return;
    char* const my = smem + lane * 16;

    WSet WA, WB;                                    // even / odd layers
    load_weights<0>(WA, a.blob, wave, lane);
    __builtin_amdgcn_sched_barrier(0);

    // ---- encoded input of this wave's point block (k_sdf_fwd2.h's prologue) -------------------------------------------
    const long pd = wr.point;
    long sdf_idx = pd;
    {
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
#pragma unroll
        for (int i = 0; i < 39; ++i) pe[i] = pe39[i] * sdf2::kC1;
#pragma unroll
        for (int i = 0; i < 25; ++i) pe[39 + i] = fmaf(pe39[i], sdf2::kC1, -bf16_lo(pack_bf16x2(pe[i], 0.0f)));
        typename P::template Act<2> X;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) X.set(kt, vals_tile<64>(pe, h, kt));
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<bf16x8*>(my + kPeb + wave * 4096 + s * 1024) = X.r[s];
    }
    const float b0 = *reinterpret_cast<const float*>(a.blob + (long)PG::first_chunk(8) * sdf2::kStride + 8 * 2048);
    if (wave == 1) {                                // row 0 of W8 (f32, in every chunk's tail) -> LDS
        const f32x4 v = *(reinterpret_cast<const f32x4*>(a.blob + sdf2::kTail) + lane);
        *reinterpret_cast<f32x4*>(my + kW8) = v;
    }
    static_for<2>([&](auto i_c) VDN_INL {           // the hidden layers' biases -> LDS: wave w copies layers 2w, 2w+1 (lane = 4 f32 of tile lane/8)
        constexpr int i = decltype(i_c)::value;
        const int L = 2 * wave + i;
        int first = 0, kt = 8;
        static_for<8>([&](auto l_c) VDN_INL { if (decltype(l_c)::value == L) { first = PG::first_chunk(decltype(l_c)::value); kt = PG::layer(decltype(l_c)::value).kt; } });
        const int t = lane >> 3;                    // (a tile that layer 3 does not have reads chunk 0 of layer 4: never used)
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.blob + (long)(first + t) * sdf2::kStride + kt * 2048 + (lane & 7) * 16);
        *reinterpret_cast<f32x4*>(my + kBias + L * 1024) = v;
    });
    lds_barrier();

    f32x16 pa, pb_;                                 // accumulators of the previous step (its epilogue runs under this step's MFMAs)
    u32x4 hold;                                     // packed pairs waiting for their 16-byte piece
    float part = 0.0f;                              // layer 7: this tile's partial sum of the sdf row
    f32x4 w8q;
    float sw[32], sl[32];                           // softplus in flight: 1 + 2^t, log2 of it (compile-time indexed: registers)

    // The epilogue of step JP is software-pipelined over the slots (k-steps) of the next step: one wave per SIMD has nobody to
    // hide the latency of exp2 -> add -> log2 -> med3 behind, so slot s runs stage 1 (1 + 2^t) of pair chunk s, stage 2 (log2)
    // of chunk s-1 and stage 3 (median, bf16 pack, LDS store / sdf partial) of chunk s-2: every instruction of a slot is
    // independent of the others, and dependent ones are a whole slot (two MFMAs) apart.
    auto value_of = [&](auto sl_c) VDN_INL -> float {
        constexpr int i = decltype(sl_c)::value;       // 0..31: tile i >> 4, register i & 15
        return (i >> 4) == 0 ? pa[i & 15] : pb_[i & 15];
    };
    auto stage = [&](auto jp_c, auto st_c, auto ch_c, auto pps_c) VDN_INL {
        constexpr int JP = decltype(jp_c)::value, ST = decltype(st_c)::value, CH = decltype(ch_c)::value, PPS = decltype(pps_c)::value;
        constexpr int NC = 16 / PPS;
        if constexpr (JP >= 0 && CH >= 0 && CH < NC && !(VDN_LW_ABL & 1)) {
            constexpr int L = JP / kNPB, PB = JP % kNPB;
            using O = IO<L>;
            static_for<PPS>([&](auto i_c) VDN_INL {
                constexpr int sp = CH * PPS + decltype(i_c)::value, j = sp >> 3, pr = sp & 7;      // pair sp: tile j, pair pr
                constexpr int v0 = 16 * j + 2 * pr, v1 = v0 + 1;
                if constexpr (ST == 1) {
                    sw[v0] = 1.0f + __builtin_amdgcn_exp2f(value_of(std::integral_constant<int, v0>{}));
                    sw[v1] = 1.0f + __builtin_amdgcn_exp2f(value_of(std::integral_constant<int, v1>{}));
                } else if constexpr (ST == 2) {
                    sl[v0] = __builtin_amdgcn_logf(sw[v0]);
                    sl[v1] = __builtin_amdgcn_logf(sw[v1]);
                } else {
                    const float g0 = __builtin_amdgcn_fmed3f(sl[v0], value_of(std::integral_constant<int, v0>{}), 25.0f);
                    const float g1 = __builtin_amdgcn_fmed3f(sl[v1], value_of(std::integral_constant<int, v1>{}), 25.0f);
                    if constexpr (L < 7) {
                        unsigned pk = pack_bf16x2(g0, g1);
                        asm volatile("" : "+v"(pk));
                        hold[pr & 3] = pk;
                        if constexpr ((pr & 3) == 3) {
                            if (O::nt == 8 || 2 * wave + j < O::nt)
                                *reinterpret_cast<u32x4*>(my + O::out_off(PB, 0) + (2 * (2 * wave + j) + (pr >> 2)) * 1024) = hold;
                        }
                    } else {
                        if constexpr ((pr & 1) == 0) w8q = *reinterpret_cast<const f32x4*>(smem + kW8 + (8 * (2 * wave + j) + 2 * (pr >> 1) + h) * 16);
                        if constexpr (pr == 0) part = 0.0f;
                        part = fmaf(g0, w8q[2 * (pr & 1)], fmaf(g1, w8q[2 * (pr & 1) + 1], part));
                        if constexpr (pr == 7) *reinterpret_cast<float*>(smem + kPart + PB * 2048 + (2 * wave + j) * 256 + lane * 4) = part;
                    }
                }
            });
        }
    };
    // slot S of NSLOT of the step after JP (S >= NSLOT: the drain behind the last slot)
    auto epilogue_slot = [&](auto jp_c, auto s_c, auto ns_c) VDN_INL {
        constexpr int JP = decltype(jp_c)::value, S = decltype(s_c)::value, NSL = decltype(ns_c)::value;
        constexpr int PPS = NSL >= 16 ? 1 : (NSL >= 8 ? 2 : (NSL >= 4 ? 4 : 16));
        using PPSC = std::integral_constant<int, PPS>;
        stage(jp_c, std::integral_constant<int, 3>{}, std::integral_constant<int, S - 2>{}, PPSC{});
        stage(jp_c, std::integral_constant<int, 2>{}, std::integral_constant<int, S - 1>{}, PPSC{});
        stage(jp_c, std::integral_constant<int, 1>{}, std::integral_constant<int, S>{}, PPSC{});
    };

    static_for<8 * kNPB>([&](auto j_c) VDN_INL {
        constexpr int J = decltype(j_c)::value, L = J / kNPB, PB = J % kNPB;
        using I = IO<L>;
        constexpr int NS = I::ns;
        WSet& W = (L & 1) ? WB : WA;
        WSet& Wn = (L & 1) ? WA : WB;
        if constexpr (PB == 0 && L < 7) load_weights<L + 1>(Wn, a.blob, wave, lane);     // the other set's layer (L-1) is done
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 x[NS];
        constexpr int PRE = kPre < NS ? kPre : NS;
        static_for<PRE>([&](auto s_c) VDN_INL {
            constexpr int s = decltype(s_c)::value;
            x[s] = *reinterpret_cast<const bf16x8*>(my + I::in_off(PB, s));
        });
        f32x16 ca, cb;
        static_for<NS>([&](auto g_c) VDN_INL {
            constexpr int gi = decltype(g_c)::value;
            if constexpr (gi == 0) {
                ca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.w[0][0], x[0], bias_rows<L>(smem, 2 * wave, h), 0, 0, 0);
                cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.w[1][0], x[0], bias_rows<L>(smem, 2 * wave + 1, h), 0, 0, 0);
            } else if constexpr (VDN_LW_ABL & 2) {
                asm volatile("" ::"v"(W.w[0][gi]), "v"(W.w[1][gi]), "v"(x[gi]));
            } else {
                ca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.w[0][gi], x[gi], ca, 0, 0, 0);
                cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.w[1][gi], x[gi], cb, 0, 0, 0);
            }
            if constexpr (gi + PRE < NS) {
                if constexpr (VDN_LW_ABL & 8) x[gi + PRE] = x[gi];
                else x[gi + PRE] = *reinterpret_cast<const bf16x8*>(my + I::in_off(PB, gi + PRE));
            }
            epilogue_slot(std::integral_constant<int, J - 1>{}, g_c, std::integral_constant<int, NS>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        // drain: the chunks whose later stages fall behind the last slot (none when the step has 18 slots)
        static_for<2>([&](auto d_c) VDN_INL {
            epilogue_slot(std::integral_constant<int, J - 1>{}, std::integral_constant<int, NS + decltype(d_c)::value>{}, std::integral_constant<int, NS>{});
            __builtin_amdgcn_sched_barrier(0);
        });
        pa = ca;
        pb_ = cb;
        if constexpr (VDN_LW_ABL & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else lds_barrier();
    });
    static_for<18>([&](auto s_c) VDN_INL {          // the last step's epilogue has no MFMAs to run under
        epilogue_slot(std::integral_constant<int, 8 * kNPB - 1>{}, s_c, std::integral_constant<int, 16>{});
    });
    lds_barrier();

    // sdf = W8[0,:] . h8 + b8[0] for this wave's point block: the tiles' partial sums in tile order, then the two lane halves
    float tot = *reinterpret_cast<const float*>(smem + kPart + wave * 2048 + lane * 4);
#pragma unroll
    for (int T = 1; T < 8; ++T) tot += *reinterpret_cast<const float*>(smem + kPart + wave * 2048 + T * 256 + lane * 4);
    const float dot = tot + __shfl_xor(tot, 32);
    if (wr.ok && h == 0) a.sdf[sdf_idx] = fmaf(dot, 1.0f / sdf2::kC1, b0) * (1.0f / a.scale);
}

template <int VID = 0>
int launch0(const VdnSdfArgs* args, hipStream_t stream) {
    static bool once = (allow_big_lds(sdf_lw0_kernel<VID>, kLds), true);
    (void)once;
    const int grid = (args->P + kWaves * 32 - 1) / (kWaves * 32);
    hipLaunchKernelGGL((sdf_lw0_kernel<VID>), dim3(grid), dim3(kWaves * 64), kLds, stream, *args);
    return (int)hipGetLastError();
}

}  // namespace sdflw
}  // namespace vdn
