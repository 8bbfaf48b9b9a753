// Backward of the SDF network (rbar chain, fbar chain, weight gradients of the hidden layers) as ONE persistent,
// layer-pipelined launch on gfx950 - bf16 path. Same mathematics as k_sdf_bwd.h + train_dw_bf16.hip (the hand-derived form
// of what autograd builds for reference fields.py:72-108 with create_graph=True, differentiated in dpt_runner.py:253):
//   rbar (ascending l):   vb_l = W_l ub_l;  ub_{l+1} = vb_l * s_l;  ex_l = k * vb_l * v_l * (1 - s_l);   dW_l += v_l ub_l^T
//   fbar (descending l):  hb_l = W_l^T ab_l;  ab_{l-1} = hb_l * s_{l-1} + ex_{l-1};                      dW_l += ab_l x_l^T,  db_l += ab_l
//
// Why: as three kernels the step moved every saved plane through HBM once per chain and once more for the weight-gradient
// GEMM (17 KB per point for the GEMM alone), and all three ran on the HBM roof. A weight gradient can only be fused into the
// chain that produces its operands if ONE workgroup sees MANY rows of ONE layer - a 256 x 256 f32 partial is 256 KiB, half the
// register file of a CU - so here a workgroup belongs to a layer, not to a row block:
//   * stage = one layer of one chain (8 rbar stages, 8 fbar stages, 1 stage for the layer-0 gradient), `lanes` workgroups
//     per stage; a lane is a contiguous range of 32-point blocks that flows through the stages from workgroup to workgroup;
//   * a workgroup (8 waves, 2 per SIMD) keeps its layer's weights in LDS (128 KiB, loaded once) and its weight-gradient
//     partial in registers (128 per wave) for the whole launch; wave w owns output tile w of the chain AND the strip of the
//     weight gradient that belongs to that tile (rbar: row strip, fbar: column strip), so the only plane tiles it loads per
//     block are the two its epilogue needs anyway (H for softplus', V or EX) - one of them doubles as its gradient operand;
//     the block's input tiles are staged through LDS once (LDS-DMA, double buffered) and serve the chain (B operand) and the
//     weight gradient (the other operand);
//   * the weight gradient contracts over POINTS, which sit on the MFMA lanes in every tile: a tile is transposed on the
//     matrix core itself (X^T = X^T I: the tile's registers as the A operand against an identity B operand, 2 MFMAs), whose
//     result converts into A / B fragments of the gradient product with no lane movement and no LDS round trip;
//   * hand-off between stages: write-through (sc1) 16-byte stores, every wave drains its stores (vmcnt(0)), workgroup barrier,
//     one agent-scope counter store per block; the consumer polls the counter (relaxed, agent scope) before it touches the
//     block. Every handed-off line is read exactly once per launch, by a CU that has not touched it before (no stale L1 / L2
//     copy can exist); kernel boundaries do the rest. Logical workgroup ids come from a ticket, so a workgroup only ever
//     waits for workgroups that started before it: no assumption about dispatch order or co-residency. Waits are bounded
//     (status word).
// K splits (= lanes) are summed by vdn_dw_finalize as before, deterministically.
#include <cstdlib>
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {
namespace pipe {

constexpr int kWaves = 8;
constexpr int kLds = 160 * 1024;          // the layer's LDS-resident weights (<= 128 KiB) + two buffers of input tiles (<= 18 KiB each)
#ifndef VDN_PIPE_DW_GROUP
#define VDN_PIPE_DW_GROUP 1
#endif
#ifndef VDN_PIPE_CH_GROUP
#define VDN_PIPE_CH_GROUP 2
#endif
constexpr int kRegKt = 7;                 // stages with register-resident weights keep k-tiles 7 and 8 there (LDS holds 0 .. 6)
constexpr int kChunkStride = 20480;       // BF16::stride(9): chunk stride of the SDF streams
constexpr int kSpinLimit = 1 << 21;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using Stage = ::VdnSdfPipeStage;
using Args = ::VdnSdfPipeArgs;

VDN_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
VDN_DEV T* uni_ptr(T* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (T*)(((unsigned long long)hi << 32) | lo);
}

// LDS-DMA of 16 B per lane that bypasses this CU's L1 (sc1): the source may have been written by another CU in this launch
VDN_DEV void glds16_sc1(const void* gsrc_lane, void* lds_wave_base) {
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(gsrc_lane), "s"(lds) : "memory", "m0");
}
// write-through 16-byte store (reaches memory, not just this XCD's L2); completion is awaited by the caller's vmcnt(0)
VDN_DEV void store16_wt(void* p, const u32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
VDN_DEV int flag_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
VDN_DEV void flag_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wait until *f >= need (every wave polls for itself: wave-uniform address, no LDS broadcast needed); bounded
VDN_DEV void wait_counter(const int* f, int need, int& seen, int* status) {
    if (seen >= need) return;
    int spins = 0;
    int v = flag_load(f);
    while (v < need) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > kSpinLimit) {
            flag_store(status, 1);
            break;
        }
        v = flag_load(f);
    }
    seen = v < need ? need : v;
}

VDN_DEV bf16x8 as_frag(const uint4& u) {
    u32x4 v = {u.x, u.y, u.z, u.w};
    return __builtin_bit_cast(bf16x8, v);
}
VDN_DEV bf16x8 pack8(const f32x16& z, int s) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(z[8 * s + 2 * j], z[8 * s + 2 * j + 1]);
    return __builtin_bit_cast(bf16x8, v);
}

// X^T of a 32 x 32 tile held as two fragments (registers 8s .. 8s+7 of the accumulator layout, packed): the tile's
// registers as the A operand against the identity -> rows (registers) = the tile's lane index (points), lane = its row
// index (features). `valid` < 32 zeroes the rows of points beyond the work list (their planes hold whatever was there).
// Straight-line code (no branch inside: the caller's loops must stay one scheduling region): PARTIAL masks, CS sums the columns
// and adds them to `colsum` scaled by `cs_w` (1 for the tile this wave is responsible for, else 0).
template <bool PARTIAL, bool CS>
VDN_DEV void transpose_tile(const bf16x8& x0, const bf16x8& x1, const bf16x8 (&ident)[2], int h, int valid, bf16x8 (&out)[2], float& colsum, float cs_w) {
    f32x16 z;
#pragma unroll
    for (int t = 0; t < 16; ++t) z[t] = 0.0f;
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x0, ident[0], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x1, ident[1], z, 0, 0, 0);
    if constexpr (PARTIAL) {
#pragma unroll
        for (int t = 0; t < 16; ++t) z[t] = rho(t, h) < valid ? z[t] : 0.0f;
    }
    if constexpr (CS) {
        float s = 0.0f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += z[t];
        colsum = fmaf(s, cs_w, colsum);
    }
    out[0] = pack8(z, 0);
    out[1] = pack8(z, 1);
}

// One instantiation per stage shape, so that the chain and gradient loops are branch-free straight-line code the scheduler can
// software-pipeline (with run-time trip counts every k-tile was a basic block of its own: LDS round trip, two dependent MFMAs,
// next block - 13 K cycles per 32-point block for 1.6 K cycles of matrix work):
//   KIND 0 rbar / 1 fbar / 2 layer-0 gradient; KTW = k-tiles with LDS-resident weights (= input tiles that take part in the
//   weight gradient); HASREG: two more k-tiles (7, 8) with register-resident weights.
template <int KIND, int KTW, bool HASREG>
VDN_DEV void run_stage(const Args& a, char* smem, int stage, int ln) {
    const int wave = uni(threadIdx.x >> 6), lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const Stage& sd = a.stages[stage];
    // kt_lds input tiles are staged through LDS per block (kt_extra of them from the plane reg_out); the weights of the first
    // kt_w = kt_lds - kt_reg k-tiles sit in LDS, those of the last kt_reg (HASREG stages: k-tiles 7, 8) in registers
    constexpr int kind = KIND, kt_w = KTW, kt_reg = HASREG ? 2 : 0, kt_lds = KTW + kt_reg, n_dw = KTW;
    const int nt = uni(sd.nt), kt_extra = HASREG ? uni(sd.kt_extra) : 0;
    const int has_dw = uni(sd.has_dw);
    const int in_ld = uni(sd.in_ld), out_ld = uni(sd.out_ld), out_tile0 = uni(sd.out_tile0), reg_tile0 = uni(sd.reg_tile0), own_ld = uni(sd.own_ld);
    const int reg_ld = uni(sd.reg_ld), copy_in = uni(sd.copy_in);
    const char* blob = uni_ptr(sd.blob);
    const unsigned short* x_in = uni_ptr(reinterpret_cast<const unsigned short*>(sd.x_in));
    unsigned short* x_out = uni_ptr(reinterpret_cast<unsigned short*>(sd.x_out));
    unsigned short* reg_out = uni_ptr(reinterpret_cast<unsigned short*>(sd.reg_out));
    const unsigned short* Splane = uni_ptr(reinterpret_cast<const unsigned short*>(sd.S));
    const unsigned short* aux = uni_ptr(reinterpret_cast<const unsigned short*>(sd.aux));
    unsigned short* ex_out = uni_ptr(reinterpret_cast<unsigned short*>(sd.ex_out));
    const unsigned short* own = uni_ptr(reinterpret_cast<const unsigned short*>(sd.own));
    int* status = a.sync + 1;
    int* counters = a.sync + 2;
    int* my_counter = counters + stage * a.lanes + ln;
    const int in_stage = uni(sd.in_stage), ex_stage = uni(sd.ex_stage);
    const int* in_counter = in_stage >= 0 ? counters + in_stage * a.lanes + ln : nullptr;
    const int* ex_counter = ex_stage >= 0 ? counters + ex_stage * a.lanes + ln : nullptr;

    const long n_rows = a.active_idx != nullptr ? (long)uni(*a.n_active) : (long)a.P;
    const int nb = (int)((n_rows + 31) >> 5);
    const int j0 = (int)((long)ln * nb / a.lanes), j1 = (int)((long)(ln + 1) * nb / a.lanes);
    const bool active = wave < nt;                  // this wave owns an output tile

    // ---- the layer's weights: LDS-resident k-tiles (one DMA pass), register-resident extra k-tiles
    char* W = smem;
    const int w_bytes = kind == 2 ? 0 : nt * kt_w * 2048;
    char* X = smem + w_bytes;
    const int x_buf = kt_lds * 2048;
    {
        const int pieces = w_bytes >> 10;                        // 1-KiB pieces: [tile][k-step] (the layer-0 stage has no chain)
        for (int u = wave; u < pieces; u += kWaves) {
            const int t = u / (kt_w * 2), ks = u - t * (kt_w * 2);
            glds16(blob + (long)(uni(sd.chunk0) + t) * kChunkStride + ks * 1024 + lane * 16, W + u * 1024);
        }
    }
    bf16x8 wreg[HASREG ? 4 : 1];
#pragma unroll
    for (int i = 0; i < (HASREG ? 4 : 1); ++i) wreg[i] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (HASREG && active && kind != 2) {
        const char* ch = blob + (long)(uni(sd.chunk0) + wave) * kChunkStride + (long)kt_w * 2048 + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) wreg[i] = *reinterpret_cast<const bf16x8*>(ch + i * 1024);
    }
    // identity as B fragments: element j of k-step s is k = 16 s + 8 (j >> 2) + 4 h + (j & 3) (the k order of an accumulator tile)
    bf16x8 ident[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) ident[s][j] = (16 * s + 8 * (j >> 2) + 4 * h + (j & 3)) == c ? (short)0x3F80 : (short)0;

    f32x16 dw[n_dw];
#pragma unroll
    for (int i = 0; i < n_dw; ++i)
#pragma unroll
        for (int t = 0; t < 16; ++t) dw[i][t] = 0.0f;
    // fbar: column sums of the input tiles (bias gradient). Every wave transposes every input tile: wave w sums tile w (one register)
    float cs = 0.0f;
    const float ex_k = 0.6931471805599453f;      // V holds 100 log2(e) v: softplus'' = 100 s (1 - s)  (k_sdf_bwd.h)

    const long in_blk = 32L * in_ld, out_blk = 32L * out_ld;
    // DMA of block j's input tiles into buffer (j & 1): piece u = [tile][k-step], 1 KiB, linear copy of the plane's bytes
    auto issue_x = [&](int j) VDN_INL {
        const char* src = reinterpret_cast<const char*>(x_in + (long)j * in_blk);
        char* dst = X + (j & 1) * x_buf;
        const int n_in = (kt_lds - kt_extra) * 2;
        for (int u = wave; u < n_in; u += kWaves) glds16_sc1(src + u * 1024 + lane * 16, dst + u * 1024);
        if (HASREG && kt_extra > 0) {      // the last tiles come from another plane (fbar W8^T: the sdf adjoint, AB(8) tile 8)
            const char* src2 = reinterpret_cast<const char*>(reg_out + (long)j * (32L * reg_ld) + (long)reg_tile0 * 1024);
            for (int u = wave; u < kt_extra * 2; u += kWaves) glds16_sc1(src2 + u * 1024 + lane * 16, dst + (n_in + u) * 1024);
        }
    };
    typedef BF16::raw_tile Raw;
    struct Aux { Raw s, v; };
    auto load_aux = [&](int j) VDN_INL {
        Aux r{};
        const long row = 32L * j + c;
        if (active) {
            if (kind != 2) r.s = BF16::load_raw(Splane, row, 256, wave, h);
            if (kind == 0) r.v = BF16::load_raw(aux, row, 256, wave, h);       // V[l]: epilogue AND this wave's gradient operand
            else if (kind == 1) r.v = BF16::load_raw(aux, row, 256, wave, h);  // EX[l-1]
            else r.s = BF16::load_raw(own, row, own_ld, wave, h);              // layer-0 gradient: PE tile
        }
        return r;
    };
    // counters are polled one iteration ahead of their use (the load's round trip - microseconds when the chip streams - stays off
    // the critical path): `pend` is the value an earlier, unwaited load returned
    int seen_in = 0, seen_ex = 0, pend_in = 0, pend_ex = 0;
    auto poll_ahead = [&]() VDN_INL {
        if (in_counter != nullptr) pend_in = flag_load(in_counter);
        if (ex_counter != nullptr) pend_ex = flag_load(ex_counter);
    };
    auto wait_inputs = [&](int j) VDN_INL {      // block j (lane-relative index j - j0) published by the producer stages
        const int need = j - j0 + 1;
        if (in_counter != nullptr) {
            seen_in = pend_in > seen_in ? pend_in : seen_in;
            wait_counter(in_counter, need, seen_in, status);
        }
        if (ex_counter != nullptr) {
            seen_ex = pend_ex > seen_ex ? pend_ex : seen_ex;
            wait_counter(ex_counter, need, seen_ex, status);
        }
    };
    // write-through stores this wave issues per block: they stay in flight across the next block's barrier (counted vmcnt), and a
    // block is published one iteration after the one that found its stores complete
    int n_st = 0;
    if (active && kind != 2) n_st += (x_out != nullptr ? 2 : 0) + ((kind == 0 && ex_out != nullptr) ? 2 : 0);
    if (HASREG && copy_in && wave < kt_lds - kt_extra) n_st += 2;
    const bool publishes = x_out != nullptr || ex_out != nullptr;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // weights are in LDS
    Aux aux_next{};
    if (j0 < j1) {
        poll_ahead();
        wait_inputs(j0);
        issue_x(j0);
        aux_next = load_aux(j0);
        poll_ahead();
    }
#ifdef VDN_PIPE_STAMP
    long long st_total = __builtin_amdgcn_s_memtime(), st_vm = 0, st_bar = 0, st_wait = 0, st_chain = 0, st_dw = 0;
#define STAMP(var) { const long long t_ = __builtin_amdgcn_s_memtime(); var += t_ - st_last; st_last = t_; }
    long long st_last = st_total;
#else
#define STAMP(var)
#endif
    for (int j = j0; j < j1; ++j) {
        STAMP(st_dw)
        // DMA and plane loads of block j have landed (this wave's; everybody's after the barrier), and so have the stores of
        // block j-2; the stores of block j-1 (the youngest n_st operations of this wave) may still be on their way
        switch (n_st) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        STAMP(st_vm)
        __syncthreads();
        STAMP(st_bar)
        if (threadIdx.x == 0 && publishes && j - 1 > j0) flag_store(my_counter, j - 1 - j0);      // blocks before j-1 are complete in memory
        const Aux aux_cur = aux_next;
        if (j + 1 < j1) {
            wait_inputs(j + 1);
            STAMP(st_wait)
            issue_x(j + 1);
            aux_next = load_aux(j + 1);
            poll_ahead();
        }
        const char* xb = X + (j & 1) * x_buf + h * 512 + c * 16;       // this lane's 16-byte unit of a tile: + tile * 2048 + k-step * 1024
        const int valid = (int)(n_rows - 32L * j < 32 ? n_rows - 32L * j : 32);
        if (valid < 32) {
            // the very last block of the work list: its padding rows hold whatever the producers left there (possibly NaN patterns)
            char* xw = X + (j & 1) * x_buf;
            for (int u = threadIdx.x; u < kt_lds * 128; u += kWaves * 64)      // 16-byte units: [tile][k][h][point]
                if ((u & 31) >= valid) *reinterpret_cast<u32x4*>(xw + u * 16) = u32x4{0u, 0u, 0u, 0u};
            __syncthreads();
        }
        // ---- chain: output tile `wave`
        f32x16 acc;
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
        if (active && kind != 2) {
            const char* wt = W + wave * (kt_w * 2048) + lane * 16;
#pragma unroll
            for (int kt = 0; kt < kt_w; ++kt) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 A = *reinterpret_cast<const bf16x8*>(wt + kt * 2048 + s * 1024);
                    const bf16x8 B = *reinterpret_cast<const bf16x8*>(xb + kt * 2048 + s * 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc, 0, 0, 0);
                }
                if ((kt % VDN_PIPE_CH_GROUP) == VDN_PIPE_CH_GROUP - 1) __builtin_amdgcn_sched_barrier(0);      // k-tiles of fragments in flight
            }
            if constexpr (HASREG) {      // k-tiles 7, 8: weights in registers, inputs from LDS like the others
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 B = *reinterpret_cast<const bf16x8*>(xb + (kRegKt + (i >> 1)) * 2048 + (i & 1) * 1024);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[i], B, acc, 0, 0, 0);
                }
            }
            // ---- epilogue + write-through stores of the hand-off tile (and EX), one 16-byte piece (8 values) at a time: the
            // register budget has no room for whole unpacked tiles beside the 128 gradient accumulators
            unsigned short* po = x_out + (long)j * out_blk + (long)(out_tile0 + wave) * 1024 + h * 256 + c * 8;
            unsigned short* pe = ex_out + (long)j * (32L * 256) + (long)wave * 1024 + h * 256 + c * 8;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const uint4 su = aux_cur.s.k[k], vu = aux_cur.v.k[k];
                const unsigned sw[4] = {su.x, su.y, su.z, su.w}, vw[4] = {vu.x, vu.y, vu.z, vu.w};
                u32x4 ob, eb;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float o2[2], e2[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const float sv = e ? bf16_hi(sw[q]) : bf16_lo(sw[q]), vv = e ? bf16_hi(vw[q]) : bf16_lo(vw[q]);
                        const float ac = acc[8 * k + 2 * q + e];
                        const float sp = sprime(sv, 2);
                        if (kind == 0) {
                            o2[e] = ac * sp;
                            e2[e] = ex_k * ac * vv * (1.0f - sp);
                        } else {
                            o2[e] = ac * sp + vv;
                            e2[e] = 0.0f;
                        }
                    }
                    ob[q] = pack_bf16x2(o2[0], o2[1]);
                    eb[q] = pack_bf16x2(e2[0], e2[1]);
                }
                store16_wt(po + 512 * k, ob);                          // (every chain stage has an x_out; every rbar stage an ex_out)
                if constexpr (KIND == 0) store16_wt(pe + 512 * k, eb);
            }
        }
        if (HASREG && copy_in && wave < kt_lds - kt_extra) {
            // fbar W8^T: vdn_dw_gemm still contracts layer 8 after this launch, from AB(8) = [g_feat | g_sdf / scale]: the input
            // tiles are copied to AB(8) tiles 0 .. 7 (tile 8 is written by sdf_pipe_prep_kernel)
            unsigned short* rp = reg_out + (long)j * (32L * reg_ld) + h * 256 + c * 8;
            const u32x4 u0 = *reinterpret_cast<const u32x4*>(xb + wave * 2048), u1 = *reinterpret_cast<const u32x4*>(xb + wave * 2048 + 1024);
            store16_wt(rp + (long)wave * 1024, u0);
            store16_wt(rp + (long)wave * 1024 + 512, u1);
        }
        STAMP(st_chain)
        // ---- weight gradient: own tile (V / H / PE tile `wave`) x every input tile. Straight-line code per block (the padding rows of
        // a partial last block were zeroed in the LDS buffer above and are zeroed in the own tile here: no masking in the loop - two
        // copies of the loop behind a branch made the register allocator spill the accumulators)
        auto gradient = [&]() VDN_INL {
            constexpr bool PARTIAL = false;
            constexpr bool CS = KIND != 0;          // fbar stages and the layer-0 stage also sum the input tiles' columns (bias gradient)
            float dummy = 0.0f;
            // (a wave without an output tile - 7-tile layers, the layer-0 stage - runs the same loop on a zero own tile: it still owes
            // the column sum of input tile `wave`, and a second code path beside this loop costs the accumulators their registers)
            if (has_dw && (active || (CS && wave < n_dw))) {
                bf16x8 zo[2];
                Raw ow = kind == 0 ? aux_cur.v : aux_cur.s;             // rbar: V tile; fbar: H tile; layer 0: PE tile
                if (c >= valid) ow.k[0] = ow.k[1] = uint4{0u, 0u, 0u, 0u};  // (this lane's point lies beyond the work list)
                transpose_tile<PARTIAL, false>(as_frag(ow.k[0]), as_frag(ow.k[1]), ident, h, valid, zo, dummy, 0.0f);
                const bf16x8 zown0 = zo[0], zown1 = zo[1];
#pragma unroll
                for (int kt = 0; kt < n_dw; ++kt) {
                    const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xb + kt * 2048);
                    const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xb + kt * 2048 + 1024);
                    transpose_tile<PARTIAL, CS>(x0, x1, ident, h, valid, zo, cs, kt == wave ? 1.0f : 0.0f);
                    if (kind == 0) {      // rows = own tile (v), columns = input tile (ub)
                        dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zown0, zo[0], dw[kt], 0, 0, 0);
                        dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zown1, zo[1], dw[kt], 0, 0, 0);
                    } else {              // rows = input tile (ab), columns = own tile (x)
                        dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zo[0], zown0, dw[kt], 0, 0, 0);
                        dw[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zo[1], zown1, dw[kt], 0, 0, 0);
                    }
                    if ((kt % VDN_PIPE_DW_GROUP) == VDN_PIPE_DW_GROUP - 1) __builtin_amdgcn_sched_barrier(0);      // input tiles in flight (register budget)
                }
            }
        };
        gradient();
    }
    // ---- the last block's stores, then its counter
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef VDN_PIPE_STAMP
    if (threadIdx.x == 0) {
        long long* o = reinterpret_cast<long long*>(a.sync + 2 + a.n_stages * a.lanes + 2) + (stage * a.lanes + ln) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - st_total; o[1] = st_vm; o[2] = st_bar; o[3] = st_wait; o[4] = st_chain; o[5] = st_dw; o[6] = j1 - j0;
    }
#endif
    if (threadIdx.x == 0 && j1 > j0 && (x_out != nullptr || ex_out != nullptr)) flag_store(my_counter, j1 - j0);
    // ---- partial sums of this lane (zero when it had no rows)
    if (has_dw) {
        const int M = uni(sd.slab_m), N = uni(sd.slab_n);
        const int split = uni(sd.split) + ln;
        float* slab = uni_ptr(sd.slab) + (long)split * M * N;
        if (active) {
#pragma unroll
            for (int kt = 0; kt < n_dw; ++kt) {
                {
                    // accumulator tile: rows (registers) x columns (lane)
                    const int row0 = kind == 0 ? 32 * wave : 32 * kt, col0 = kind == 0 ? 32 * kt : 32 * wave;
                    float* base = slab + (long)row0 * N + col0 + c;
#pragma unroll
                    for (int t = 0; t < 16; ++t) base[(long)rho(t, h) * N] = dw[kt][t];
                }
            }
        }
        if (KIND != 0 && sd.colsum != nullptr && wave < n_dw) {
            float* csum = uni_ptr(sd.colsum) + (long)split * M;
            const float tot = cs + __shfl_xor(cs, 32);
            if (h == 0) csum[32 * wave + c] = tot;
        }
    }
}

// The per-point inputs of the chains that are not planes yet, as PT32 tiles (one wave per 32 rows of the work list):
//   UB(0) tiles 0, 1 = UB(4) tiles 7, 8 = scale * J_PE(x) g_normals  (39 values: the adjoint of  normal = scale * J_PE^T u)
//   AB(8) tile 8 = [g_sdf / scale, 0 ...]
// so that the pipelined launch reads them like every other input tile (no dependent index -> ray -> depth loads, no trigonometry
// on its critical path: its first stage paces all the others).
__global__ __launch_bounds__(256) void sdf_pipe_prep_kernel(Args a, unsigned short* ub0, unsigned short* ub4, unsigned short* ab8) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const long n_rows = a.active_idx != nullptr ? (long)*a.n_active : (long)a.P;
    const long row = ((long)blockIdx.x * 4 + wave) * 32 + c;
    if (((long)blockIdx.x * 4 + wave) * 32 >= n_rows) return;
    const bool ok = row < n_rows;
    const long rowc = ok ? row : n_rows - 1;
    const long pd = a.active_idx != nullptr ? (long)a.active_idx[rowc] : rowc;
    const long r = pd / a.n_per_ray;
    const float z = a.z[r * a.z_ld + (pd - r * a.n_per_ray)];
    float xin[3], gn[3], ub39[39];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        xin[d] = (a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z) * a.scale;
        gn[d] = a.g_normals[pd * 3 + d] * a.scale;
        ub39[d] = gn[d];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float sn, co;
            sincos_pe<false>(xin[d] * f, sn, co);
            ub39[3 + 6 * k + d] = f * co * gn[d];
            ub39[3 + 6 * k + 3 + d] = -f * sn * gn[d];
        }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const f32x16 t16 = vals_tile<39>(ub39, h, kt);
        BF16::store_tile(ub0, row, 64, kt, h, t16, true);          // (rows beyond the list: inside the block padding)
        BF16::store_tile(ub4, row, 288, 7 + kt, h, t16, true);
    }
    float g1[1] = {a.g_sdf[pd] / a.scale};
    BF16::store_tile(ab8, row, 288, 8, h, vals_tile<1>(g1, h, 0), true);
}

__global__ __launch_bounds__(kWaves * 64, 2) void sdf_bwd_pipe_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- logical id: workgroups are numbered in the order they START, so every counter a workgroup waits on belongs to a
    // workgroup that is already running (or done)
    if (threadIdx.x == 0) *reinterpret_cast<volatile int*>(smem) = atomicAdd(a.sync, 1);
    __syncthreads();
    const int lid = uni(*reinterpret_cast<volatile int*>(smem));
    __syncthreads();
    const int stage = lid / a.lanes, ln = lid - stage * a.lanes;
    if (stage >= a.n_stages) return;
    const int kind = uni(a.stages[stage].kind), kt_reg = uni(a.stages[stage].kt_reg), ktw = uni(a.stages[stage].kt_lds) - kt_reg;
#ifdef VDN_PIPE_ONLY      // (register-usage diagnostics: one variant per build)
    if (VDN_PIPE_ONLY == 0) run_stage<0, 8, false>(a, smem, stage, ln);
    if (VDN_PIPE_ONLY == 1) run_stage<0, 7, true>(a, smem, stage, ln);
    if (VDN_PIPE_ONLY == 2) run_stage<1, 8, false>(a, smem, stage, ln);
    if (VDN_PIPE_ONLY == 3) run_stage<1, 7, true>(a, smem, stage, ln);
    if (VDN_PIPE_ONLY == 4) run_stage<2, 8, false>(a, smem, stage, ln);
#else
    if (kind == 0) {
        if (kt_reg > 0) run_stage<0, 7, true>(a, smem, stage, ln);
        else if (ktw == 8) run_stage<0, 8, false>(a, smem, stage, ln);
        else run_stage<0, 2, false>(a, smem, stage, ln);
    } else if (kind == 1) {
        if (kt_reg > 0) run_stage<1, 7, true>(a, smem, stage, ln);
        else if (ktw == 8) run_stage<1, 8, false>(a, smem, stage, ln);
        else run_stage<1, 7, false>(a, smem, stage, ln);
    } else {
        run_stage<2, 8, false>(a, smem, stage, ln);
    }
#endif
}

}  // namespace pipe
}  // namespace vdn

extern "C" int vdn_sdf_bwd_pipe_bf16(const VdnSdfPipeArgs* args, void* stream_) {
    using namespace vdn;
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || !args->stages || args->n_stages <= 0 || args->lanes <= 0 || !args->sync || args->P <= 0) return -1;
    if (!args->rays_o || !args->rays_d || !args->z || args->n_per_ray <= 0 || args->z_ld < args->n_per_ray || !args->g_normals || !args->g_sdf) return -2;
    static bool once = (allow_big_lds(pipe::sdf_bwd_pipe_kernel, pipe::kLds), true);
    (void)once;
    const int n_wg = args->n_stages * args->lanes;
    hipError_t e = hipMemsetAsync(args->sync, 0, sizeof(int32_t) * (size_t)(2 + n_wg), stream);
    if (e != hipSuccess) return (int)e;
    if (!args->ub0 || !args->ub4 || !args->ab8) return -3;
    hipLaunchKernelGGL(pipe::sdf_pipe_prep_kernel, dim3((args->P + 127) / 128), dim3(256), 0, stream, *args,
                       reinterpret_cast<unsigned short*>(args->ub0), reinterpret_cast<unsigned short*>(args->ub4),
                       reinterpret_cast<unsigned short*>(args->ab8));
    hipLaunchKernelGGL(pipe::sdf_bwd_pipe_kernel, dim3(n_wg), dim3(pipe::kWaves * 64), pipe::kLds, stream, *args);
    return (int)hipGetLastError();
}
