// Flat-stream MLP engine for the bf16 kernels (round 2): the execution structure of k_sdf_fwd2.h behind the interface of
// mlp_engine.h's dense(), for the colour / VDN heads and the background network (forward and backward chains).
//
//   * a kernel's weight stream is a compile-time PROGRAM: for every 32-row chunk its contraction width (k-tiles), whether it
//     carries a bias block, and how many vector-memory instructions its tile's pre() (loads) and epi() (stores) issue;
//   * chunk step c issues chunk c's MFMAs; behind its first two MFMAs it certifies chunk c+1 (counted vmcnt + s_barrier), then
//     issues the DMA pieces of chunk c+DEPTH, runs the epilogue of chunk c-1's tile (also across layer
//     boundaries: before the MFMA group that reads the pending tile), and reads the opening fragments of chunk c+1 during its
//     last MFMAs: the matrix pipe streams across chunk and layer boundaries, no drain steps;
//   * every wait count is a compile-time constant: every lane issues every plane load / store (out-of-range lanes work on a
//     clamped row: loads read it, stores write duplicates of it);
//   * LDS-DMA is inline asm (glds16_saddr): the compiler does not see it, so its own LDS waits stay counted and its waits for
//     ordinary loads count only the loads / stores it knows (which can only wait longer, never shorter).
//
// The chunk index and the pending epilogue travel in the TYPE of the flow object: dense2() takes Flow<C, Pend> and returns
// Flow<C + NT, Pending<Epi, Aux>>; flow_finish() runs the last pending epilogue.
#pragma once
#include <type_traits>
#include "mlp_engine.h"

namespace vdn {
namespace flow {

constexpr int kPre = 4;        // weight fragments read ahead of their MFMA
constexpr int kGroup = 2;      // MFMAs per scheduling group

template <int N>
VDN_DEV void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA of 16 B per lane: scalar base (wave-uniform) + 32-bit per-lane offset; destination = wave-uniform LDS byte address
// through M0 (used by nothing else in these kernels). See vdn_common.h: glds16 for why this is not the builtin.
// Immediate-offset addressing, one M0 write per chunk: vdn_common.h (glds16_imm*).

// A program: struct with
//   static constexpr int total;                      chunks in the stream
//   static constexpr int kt(int c);                  k-tiles of chunk c (0 beyond the stream)
//   static constexpr bool bias(int c);               chunk c initialises its accumulator from its bias block
//   static constexpr int loads(int c);               vector-memory loads pre() issues for chunk c's tile (in step c)
//   static constexpr int stores(int c);              vector-memory stores epi() issues for chunk c's tile (in step c+1)
//   static constexpr bool drained(int c);            the kernel runs chunk c's epilogue right behind step c (flow_drain)

template <int NWAVES, int STRIDE, int NSLOT, int DEPTH>
struct Pipe {
    static_assert(STRIDE % (1024 * NWAVES) == 0, "chunk stride must be a multiple of NWAVES KiB");
    static_assert(NSLOT >= DEPTH + 1, "ring: the chunk being read, the one being opened and DEPTH-1 in flight");
    static constexpr int kG = STRIDE / 1024 / NWAVES;     // DMA instructions per wave per chunk
    static constexpr int kDepth = DEPTH;
    const char* g;      // weight stream (wave-uniform)
    char* lds;          // ring base
    int wave, lane;
    unsigned lane16;
    bf16x8 fr[kPre];    // opening fragments of the next chunk step (already read)
    f32x4 bias[4];      // its bias rows
    unsigned voff0, m0_wave;        // this wave's range of a chunk (+ 4096: the centre of the pieces' immediate offsets): per-lane global offset, LDS address in slot 0
    VDN_DEV void init(const char* blob, char* smem) {
        static_assert(kG <= 8 && (long)STRIDE * 256 < (1L << 31), "one group of immediate offsets; 32-bit chunk offsets");
        g = blob;
        lds = smem;
        wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        lane = threadIdx.x & 63;
        lane16 = lane * 16;
        voff0 = lane16 + wave * (kG * 1024) + 4096;
        m0_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + wave * (kG * 1024) + 4096);
    }
    // (the dump area of the warm-up's LDS-DMA: this wave's own first piece of ring slot 0 - vdn_common.h)
    VDN_DEV char* warm_dump() const { return lds + wave * (kG * 1024); }
    template <int C>
    VDN_DEV char* slot() const { return lds + (C % NSLOT) * STRIDE; }
    template <int C, int I>
    VDN_DEV void issue_piece() {
        const unsigned voff = voff0 + (unsigned)((long)C * STRIDE);
        if constexpr (I == 0) glds16_imm_m0add<(C % NSLOT) * STRIDE, glds_imm(I)>(g, voff, m0_wave);
        else glds16_imm<glds_imm(I)>(g, voff);
    }
    // ring start: chunks 0 .. DEPTH-1 in flight, chunk 0 certified, its opening fragments read. Call it behind every ordinary
    // load / store of the prologue (nothing but counted operations may be younger than a DMA): it drains them first.
    template <class PG>
    VDN_DEV void start() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_for<DEPTH>([&](auto c_c) VDN_INL {
            constexpr int C = decltype(c_c)::value;
            if constexpr (C < PG::total) static_for<kG>([&](auto i_c) VDN_INL { issue_piece<C, decltype(i_c)::value>(); });
        });
        wait_vmcnt<(DEPTH - 1 < PG::total - 1 ? DEPTH - 1 : PG::total - 1) * kG>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr int KT0 = PG::kt(0);
        const char* w = slot<0>();
        const bf16x8* wa = reinterpret_cast<const bf16x8*>(w) + lane;
#pragma unroll
        for (int s = 0; s < (kPre < 2 * KT0 ? kPre : 2 * KT0); ++s) fr[s] = wa[s * 64];
        if constexpr (PG::bias(0)) {
            const f32x4* b = reinterpret_cast<const f32x4*>(w + KT0 * 2048);
#pragma unroll
            for (int q = 0; q < 4; ++q) bias[q] = b[2 * q + (lane >> 5)];
        }
    }
};

// s_waitcnt vmcnt(N) that retires this wave's DMA of chunk c+1 in step c, before its barrier. Step j issues, in order: the
// loads of pre(j); [its first MFMAs; the wait; the barrier]; ALL DMA pieces of chunk j+DEPTH; then the stores of epi(j-1).
// (The DMA goes first on purpose: vmcnt retires in issue order, so a store issued BEFORE the awaited DMA would have to be
// acknowledged by memory before the wait returns - every step would pay an HBM write round trip.) Younger than DMA(c+1),
// issued in step c+1-DEPTH, are therefore: that step's stores, everything of steps c+2-DEPTH .. c-1, and the loads of pre(c).
template <class PG, int KG, int DEPTH>
constexpr int wait_count(int c) {
    int n = PG::loads(c);
    for (int j = c + 1 - DEPTH; j <= c - 1; ++j) {
        if (j < 0) continue;
        // the stores of epi(j-1) run in step j behind its DMA - unless the kernel drained that epilogue at the end of step
        // j-1 (flow_drain): then they are OLDER than step j's DMA and must not be counted (a count too large releases the
        // wait while DMA pieces are still in flight)
        if (!PG::drained(j - 1)) n += PG::stores(j - 1);
        if (j >= c + 2 - DEPTH) {
            n += PG::loads(j);
            if (j + DEPTH < PG::total) n += KG;
        }
    }
    return n < 63 ? n : 63;
}

struct NoPend {
    static constexpr int tile = -1;
    VDN_DEV void run() const {}
};
// the epilogue of a chunk's tile, waiting to run under the next chunk's MFMAs
template <class Epi, class Aux, int TILE = -1>
struct Pending {
    static constexpr int tile = TILE;       // the output tile it belongs to (known once it is a layer's last tile)
    Epi epi;
    f32x16 acc;
    Aux aux;
    int nt;
    VDN_DEV void run() const { epi(nt, acc, aux); }
};

template <int C, class Pend>
struct Flow {
    Pend pend;
};

// One chunk step (see the header). `mid` runs once, in the group chosen for the pending epilogue.
template <class PG, int C, int G_EPI, class PipeT, class ActT, class Mid>
VDN_DEV f32x16 chunk_step(PipeT& pp, const ActT& X, Mid&& mid) {
    constexpr int KT = PG::kt(C);
    constexpr bool BIAS = PG::bias(C);
    constexpr int NS = KT * 2, NG = (NS + kGroup - 1) / kGroup;
    constexpr int KTN = PG::kt(C + 1);
    constexpr bool HAS_NEXT = C + 1 < PG::total;
    constexpr bool HAS_DMA = C + PipeT::kDepth < PG::total;
    constexpr int KG = PipeT::kG;
    const bf16x8* wa = reinterpret_cast<const bf16x8*>(pp.template slot<C>()) + pp.lane;
    const bf16x8* wn = reinterpret_cast<const bf16x8*>(pp.template slot<C + 1>()) + pp.lane;
    bf16x8 fr[NS];
    f32x16 acc;
    constexpr int PF = kPre < NS ? kPre : NS;
    constexpr int PFN = kPre < 2 * KTN ? kPre : 2 * KTN;
    static_for<PF>([&](auto s_c) VDN_INL { fr[decltype(s_c)::value] = pp.fr[decltype(s_c)::value]; });
    if constexpr (BIAS) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[4 * q + 0] = pp.bias[q][0]; acc[4 * q + 1] = pp.bias[q][1]; acc[4 * q + 2] = pp.bias[q][2]; acc[4 * q + 3] = pp.bias[q][3];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<NG>([&](auto g_c) VDN_INL {
        constexpr int gi = decltype(g_c)::value;
        constexpr int s0 = gi * kGroup, s1 = (gi + 1) * kGroup < NS ? (gi + 1) * kGroup : NS;
        static_for<s1 - s0>([&](auto j_c) VDN_INL {
            constexpr int s = s0 + decltype(j_c)::value;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s], X.r[s], acc, 0, 0, 0);
        });
        if constexpr (gi == 0 && HAS_NEXT) {
            __builtin_amdgcn_sched_barrier(0);      // the step's first MFMAs are in the pipe while the wave waits
            wait_vmcnt<wait_count<PG, KG, PipeT::kDepth>(C)>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HAS_DMA && gi == 0)       // all pieces right behind the barrier, ahead of this step's stores (see wait_count)
            static_for<KG>([&](auto i_c) VDN_INL { pp.template issue_piece<C + PipeT::kDepth, decltype(i_c)::value>(); });
        static_for<s1 - s0>([&](auto j_c) VDN_INL {
            constexpr int s = s0 + decltype(j_c)::value;
            if constexpr (s + PF < NS) fr[s + PF] = wa[(s + PF) * 64];
            else if constexpr (HAS_NEXT && s + PF - NS < PFN) pp.fr[s + PF - NS] = wn[(s + PF - NS) * 64];
            if constexpr (HAS_NEXT && s == NS - 1)
                static_for<(PFN > PF ? PFN - PF : 0)>([&](auto e_c) VDN_INL { pp.fr[PF + decltype(e_c)::value] = wn[(PF + decltype(e_c)::value) * 64]; });
        });
        if constexpr (gi == NG - 1 && HAS_NEXT && KTN > 0 && PG::bias(C + 1)) {
            const f32x4* b = reinterpret_cast<const f32x4*>(pp.template slot<C + 1>() + KTN * 2048);
#pragma unroll
            for (int q = 0; q < 4; ++q) pp.bias[q] = b[2 * q + (pp.lane >> 5)];
        }
        if constexpr (gi == (G_EPI < NG ? G_EPI : NG - 1)) mid();
        __builtin_amdgcn_sched_barrier(0);
    });
    return acc;
}

// One dense layer on the wave's 32 points: for every output tile nt, acc = bias + W[nt] . X, then epi(nt, acc, aux) with
// aux = pre(nt) (its loads are issued at the top of the tile's chunk step, one step before epi consumes them).
//   WRITES_INPUT: this layer's epilogue writes the activation tile nt of the NEXT layer's input (so, across the layer boundary,
//   it has to complete before the MFMA group that reads k-steps 2 nt, 2 nt + 1).
// Returns the flow advanced by NT chunks, carrying this layer's last tile as the pending epilogue.
template <class PG, int NT, bool PREV_WRITES_INPUT = true, int C, class PendIn, class PipeT, class ActT, class Pre, class Epi>
VDN_DEV auto dense2(Flow<C, PendIn> f, PipeT& pp, const ActT& X, Pre&& pre, Epi&& epi) {
    using EpiT = std::remove_cv_t<std::remove_reference_t<Epi>>;
    using AuxT = decltype(pre(0));
    Pending<EpiT, AuxT> cur{epi, f32x16{}, AuxT{}, 0};
    static_for<NT>([&](auto t_c) VDN_INL {
        constexpr int T = decltype(t_c)::value;
        constexpr int CC = C + T;
        constexpr int NS = 2 * PG::kt(CC), NG = (NS + kGroup - 1) / kGroup;
        auto aux_cur = pre(T);                                  // loads for this tile's epilogue (runs in step CC + 1)
        f32x16 acc_cur;
        if constexpr (T == 0) {
            // the previous layer's last tile: before the MFMA group that reads it (group 1 at the latest keeps the barrier first)
            constexpr int TP = PendIn::tile;
            constexpr int glimit = (PREV_WRITES_INPUT && TP >= 0) ? (2 * TP) / kGroup : NG;
            if constexpr (glimit == 0) {        // the pending tile is read by this step's very first MFMAs: nothing to hide under
                f.pend.run();
                acc_cur = chunk_step<PG, CC, 1>(pp, X, []() VDN_INL {});
            } else {
                constexpr int G = glimit > 1 ? 1 : 0;
                acc_cur = chunk_step<PG, CC, G>(pp, X, [&]() VDN_INL { f.pend.run(); });
            }
        } else {
            acc_cur = chunk_step<PG, CC, 1>(pp, X, [&]() VDN_INL { cur.run(); });
        }
        cur.acc = acc_cur;
        cur.aux = aux_cur;
        cur.nt = T;
        __builtin_amdgcn_sched_barrier(0);
    });
    using Out = Pending<EpiT, AuxT, NT - 1>;
    return Flow<C + NT, Out>{Out{cur.epi, cur.acc, cur.aux, cur.nt}};
}

template <int C, class Pend>
VDN_DEV void flow_finish(Flow<C, Pend>& f) { f.pend.run(); }
// run the pending epilogue now (code between two layers needs the previous layer's last tile): no overlap at this boundary
template <int C, class Pend>
VDN_DEV Flow<C, NoPend> flow_drain(Flow<C, Pend>& f) {
    f.pend.run();
    __builtin_amdgcn_sched_barrier(0);
    return Flow<C, NoPend>{NoPend{}};
}
VDN_DEV Flow<0, NoPend> flow_begin() { return Flow<0, NoPend>{NoPend{}}; }

struct NoLoad {
    VDN_DEV int operator()(int) const { return 0; }
};

}  // namespace flow
}  // namespace vdn
