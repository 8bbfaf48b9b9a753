// SDF network forward + analytic input-gradient sweep on gfx950, second-generation bf16 kernel.
// Replaces reference dpt_models/fields.py:72-108 (SDFNetwork.forward / .sdf / .gradient) like k_sdf_fwd.h, with a
// different execution structure. What the measurements said about the first kernel (profiles/README.md, round 2):
// one wave's instruction ISSUE, not the matrix pipe, bounds it (per 32 points: 61 K cycles of MFMA against ~130 K cycles
// of VALU, LDS-read and DMA issue), and its softplus' round trip through HBM was 17x its boundary I/O. Hence:
//
//  * softplus'(a_l) of all 8 hidden layers never leaves the chip: 8-bit fixed point (255 sigma; exact at the saturated
//    values 0 and 1; on the bench scene 4-5 % of the hidden activations are that far out, tools/dev/saturation.py), the first tiles in LDS, the rest in registers.
//    That needs the whole register file: 1 wave per SIMD, 4 waves = 128 points per CU.
//  * activations are carried in units of 1/(100 log2 e):  t = 100 log2(e) a,  g = 100 log2(e) softplus(a) =
//    max(t, 0) + log2(1 + 2^-|t|),  sigma(100 a) = 1 - 2^-g.  Hidden-layer weights are unchanged by this
//    (W g = 100 log2(e) W h), biases are scaled by 100 log2(e), the encoded input is scaled at the source, the last
//    layer's weights carry 1/(100 log2 e) and the sweep's transposed weights 1/255: no multiply is left in the epilogues.
//  * the stream of chunk steps is flat: step c issues chunk c's MFMAs with the epilogue of chunk c-1 in their shadow,
//    also across layer boundaries (a boundary only demands that the pending tile is packed before the MFMA group that
//    reads it); the sweep's first operand v7 is formed under the last layer's MFMAs.
//  * the chunk barrier is off the MFMA path: step c certifies chunk c+1 (counted vmcnt + s_barrier behind its first
//    MFMAs), the opening weight fragments of chunk c+1 are read during the last MFMAs of chunk c, and the DMA pieces of
//    chunk c+DEPTH are issued one per MFMA group. All wait counts are compile-time constants (every lane issues every
//    store: out-of-range lanes work on a clamped row and store duplicates of it).
//
// Weight stream ("sdf2" / "full2", vdn_hip/images.py): chunk format of mlp_engine.h (BF16 policy), uniform 20-KiB stride,
// same chunk order as the first kernel's streams.
#pragma once
#include <cstdlib>
#include "k_ray_rows.h"
#include "k_composite_row.h"
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {
namespace sdf2 {

#ifndef VDN_SDF2_PRE
#define VDN_SDF2_PRE 4
#endif
#ifndef VDN_SDF2_GROUP
#define VDN_SDF2_GROUP 2
#endif
#ifndef VDN_SDF2_ABL
#define VDN_SDF2_ABL 0   // timing-only ablations (development harness), bit mask: 1 no epilogue math, 2 no MFMA, 4 no weight DMA, 8 no chunk barrier
#endif
#ifndef VDN_SDF2_S16
#define VDN_SDF2_S16 0   // MFMA shape of the chunk steps: 0 = v_mfma_f32_32x32x16_bf16, 1 = v_mfma_f32_16x16x32_bf16 (same 32-point x 32-feature tile per wave)
#endif
constexpr bool kS16 = VDN_SDF2_S16 != 0;
#ifndef VDN_SDF2_B2
// Barrier cadence of instantiations whose ring has DEPTH + 2 slots (the others re-synchronise every chunk step): 1 = TWO chunks
// certified per workgroup barrier, i.e. one barrier every second chunk step; 2 (default, round 6) = that in front of the sweep only;
// 0 = every step. Same arithmetic, same bits; see chunk_step. Development harness, 65 536-point inference launch, variants
// interleaved in one process (profiles/r06_sdf2_barrier_cadence_ab.log): 130 980 cycles per 128-point pass -> 128 268 (1: hidden step
// 1 268 -> 1 193, but the sweep's 682 -> 730) -> 125 084 (2: 1 193 / 678), at an unchanged clock; launch 146.3 -> 141.8 us. Two builds
// of the library on one box, alternating processes (profiles/r06_sdf2_barrier_cadence_library_ab.log): 148.2 -> 145.9 us (-1.5 %).
#define VDN_SDF2_B2 2
#endif
constexpr int kStride = 20480;          // BF16::stride(9)
constexpr int kTail = 9 * 2048 + 1024;  // row 0 of W8 (f32 x 256) in every chunk's tail (vdn_hip/images.py: SDF_TAIL_OFF)
constexpr int kWaves = 4;
constexpr int kG = kStride / 1024 / kWaves;   // global_load_lds instructions per wave per chunk
constexpr int kPre = VDN_SDF2_PRE;      // weight fragments read ahead of their MFMA
constexpr int kGroup = VDN_SDF2_GROUP;  // MFMAs per scheduling group (the epilogue slices of one group interleave freely)
constexpr float kC1 = 144.26950408889634f;    // 100 log2(e)
constexpr float kVSave = kC1 / 255.0f;        // 255 sigma u -> saved V plane (100 log2(e) v)
constexpr int kSTiles = 63;             // softplus' tiles per point block: 8 + 8 + 8 + 7 + 8 + 8 + 8 + 8

// ---- the two MFMA shapes ------------------------------------------------------------------------------------------
// Shape 0 (32x32x16): lane = c + 32 h owns point c; accumulator register t = feature (t&3) + 8 (t>>2) + 4 h (vdn_common.h).
// Shape 1 (16x16x32, round 6: the shape on which the chip holds the higher clock - MI355X_MICROARCH.md, DVFS give-back item 7):
// the wave's 32 x 32 tile is four 16 x 16 accumulators [ph][fh] (point half, feature half); lane = c16 + 16 q owns points c16 and
// c16 + 16, and LOGICAL accumulator register t = 8 ph + 4 fh + i is MFMA row 4 q + i of feature half fh. The weight image
// ("sdf2x" / "full2x" / "c2x", vdn_hip/images.py) assigns MFMA row 4 q + i of half fh to feature 16 (q>>1) + 8 fh + 4 (q&1) + i, so
// that registers 8 ph .. 8 ph + 7 of lane (c16, q) are - as in shape 0 - one 16-byte unit of the PT32 plane layout
// (mlp_engine.h): unit (k = q>>1, h = q&1) of point 16 ph + c16, and the B fragment of k-step T for point half ph IS tile T's
// packed unit. Planes, bias blocks and tails are the same bytes in both shapes; the A fragments are a permutation of 16-byte units.
// Per-point scalar work (encoding, its adjoint, outputs) stays on the "home" lanes of shape 0 (lane c + 32 h); operands cross
// between the two lane geometries through a wave-private LDS scratch, three times per launch.
#if VDN_SDF2_S16
struct AccT {
    f32x4 v[4];         // [2 ph + fh]
    VDN_DEV float operator[](int t) const { return v[t >> 2][t & 3]; }
    VDN_DEV void fill(const f32x4 (&bias)[4], bool with_bias) {
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = with_bias ? bias[g & 1] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    // fragment s of a chunk = (k-step s >> 1 of 32 inputs, feature half s & 1), against both point halves
    template <int S, class ActT>
    VDN_DEV void mfma(const bf16x8& a, const ActT& X) {
        v[S & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, X.r[S & ~1], v[S & 1], 0, 0, 0);
        v[2 + (S & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, X.r[(S & ~1) + 1], v[2 + (S & 1)], 0, 0, 0);
    }
    VDN_DEV void add(const AccT& o) {
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] += o.v[g];
    }
};
#else
struct AccT {
    f32x16 w;
    VDN_DEV float operator[](int t) const { return w[t]; }
    VDN_DEV void fill(const f32x4 (&bias)[4], bool with_bias) {
#pragma unroll
        for (int t = 0; t < 16; ++t) w[t] = with_bias ? bias[t >> 2][t & 3] : 0.0f;
    }
    template <int S, class ActT>
    VDN_DEV void mfma(const bf16x8& a, const ActT& X) { w = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, X.r[S], w, 0, 0, 0); }
    VDN_DEV void add(const AccT& o) { w += o.w; }
};
#endif
// accumulator registers 4 g .. 4 g + 3 of this lane are features 4 quad_of(g) .. + 3 of the tile's natural order
VDN_DEV int quad_of(int g, int lane) { return kS16 ? 4 * (lane >> 5) + 2 * (g & 1) + ((lane >> 4) & 1) : 2 * g + (lane >> 5); }
// the point half a logical register pair belongs to
constexpr int ph_of_pair(int pr) { return kS16 ? pr >> 2 : 0; }

// ---- compile-time program of the chunk stream -------------------------------------------------------------------
// COL0 / COLH / COLOUT (MODE 2): the colour head behind the sweep (fields.py:148-176): first layer [feature (8 tiles) | points,
// PE4(view), normal x, y (1 tile)], three hidden ReLU layers, the 3-channel sigmoid output
enum { HID = 0, LAST = 1, SWEEP = 2, SWEEP_SKIP = 3, SWEEP_PE = 4, COL0 = 5, COLH = 6, COLOUT = 7 };
struct LayerDesc { int kind, kt, nt, l; };

template <int MODE>
struct Prog {
    static constexpr int NL = MODE == 0 ? 9 : (MODE == 1 ? 17 : 22);
    static constexpr LayerDesc layer(int i) {
        constexpr LayerDesc full[22] = {
            {HID, 2, 8, 0}, {HID, 8, 8, 1}, {HID, 8, 8, 2}, {HID, 8, 7, 3}, {HID, 9, 8, 4}, {HID, 8, 8, 5}, {HID, 8, 8, 6}, {HID, 8, 8, 7},
            {LAST, 8, 9, 8},
            {SWEEP, 8, 8, 6}, {SWEEP, 8, 8, 5}, {SWEEP, 8, 8, 4}, {SWEEP_SKIP, 8, 9, 3}, {SWEEP, 7, 8, 2}, {SWEEP, 8, 8, 1}, {SWEEP, 8, 8, 0},
            {SWEEP_PE, 8, 2, -1},
            {COL0, 9, 8, 0}, {COLH, 8, 8, 1}, {COLH, 8, 8, 2}, {COLH, 8, 8, 3}, {COLOUT, 8, 1, 4}};
        LayerDesc d = full[i];
        if (MODE == 0 && i == 8) d.nt = 1;
        return d;
    }
    static constexpr int first_chunk(int i) {
        int c = 0;
        for (int j = 0; j < i; ++j) c += layer(j).nt;
        return c;
    }
    static constexpr int total = first_chunk(NL);
    static constexpr int sdf_total = first_chunk(MODE == 0 ? 9 : 17);      // chunks of the SDF network's stream; the colour head's follow in a blob of their own
    static constexpr int layer_of(int c) {
        int i = 0;
        while (i + 1 < NL && first_chunk(i + 1) <= c) ++i;
        return i;
    }
    static constexpr int tile_of(int c) { return c - first_chunk(layer_of(c)); }
    static constexpr int kt_of(int c) { return (c >= 0 && c < total) ? layer(layer_of(c)).kt : 0; }
    static constexpr bool bias_of(int c) { return c >= 0 && c < total && (layer(layer_of(c)).kind <= LAST || layer(layer_of(c)).kind >= COL0); }
    // first softplus' tile of hidden layer l
    static constexpr int s_tile0(int l) { return l <= 3 ? 8 * l : 8 * l - 1; }
};

// vector-memory stores issued by the epilogue of chunk c's tile (every lane of every wave issues all of them);
// they are issued during step c+1
template <int MODE, bool SAVE>
constexpr int tile_stores(int c) {
    using PG = Prog<MODE>;
    if (c < 0 || c >= PG::total) return 0;
    const LayerDesc d = PG::layer(PG::layer_of(c));
    const int t = PG::tile_of(c);
    constexpr bool TS = SAVE && (MODE == 1 || MODE == 3);       // the training saves (MODE 2: SAVE = the feature plane is written, for the VDN head)
    switch (d.kind) {      // (two 16-byte stores per plane tile: BF16::kTileOps)
        case HID: return TS ? 2 : 0;                                             // H plane tile
        case LAST: return t < 8 ? ((MODE == 1 || MODE == 3) ? (SAVE ? 4 : 2) : (MODE == 2 && SAVE ? 2 : 0)) : 0;   // feature tile (+ V[7] tile)
        case SWEEP: return TS ? 2 : 0;                                           // V[l]
        case SWEEP_SKIP: return (TS && t < 7) ? 2 : 0;
        case COL0: case COLH: return MODE == 3 ? 2 : 0;                          // MODE 3: the colour head's saved activations
        default: return 0;
    }
}
// steps whose epilogue stores planes issue their DMA pieces in one burst right behind the barrier, AHEAD of those stores:
// vmcnt retires in issue order, so a store issued before the awaited DMA would have to be acknowledged by memory before the
// wait returns. Steps without stores spread the pieces over the MFMA groups (less contention in the CU's memory path).
template <int MODE, bool SAVE>
constexpr bool dma_burst(int c) { return tile_stores<MODE, SAVE>(c - 1) > 0; }
// s_waitcnt vmcnt(N) that retires this wave's DMA of chunk c+1 in step c (before its barrier). Younger than that DMA
// (issued during step c+1-DEPTH) are the DMAs of chunks c+2 .. c+DEPTH-1, the stores of steps c+2-DEPTH .. c-1, and - when
// step c+1-DEPTH issued its DMA as a burst ahead of its stores - that step's stores too.
// Two chunks per barrier (B2; a ring of >= DEPTH + 2 slots): the even step c certifies chunks c+1 AND c+2. This wave's DMA of chunk
// c+2 was issued during step c-1; younger than it are only that step's stores when it issued its DMA as a burst ahead of them
// (a step without stores spreads the pieces and issues nothing behind them), and nothing of step c (the wait sits in front of its
// first epilogue slice).
template <int MODE, bool SAVE>
constexpr int wait_count_b2(int c) {
    const int n = tile_stores<MODE, SAVE>(c - 2);       // step c-1 runs the epilogue of chunk c-2
    return n < 63 ? n : 63;
}
template <int MODE, bool SAVE, int DEPTH>
constexpr int wait_count(int c) {
    using PG = Prog<MODE>;
    int n = 0;
    for (int j = c + 2; j <= c + DEPTH - 1; ++j)
        if (j < PG::total) n += kG;
    for (int j = c + 1 - DEPTH; j <= c - 1; ++j)
        if (j >= c + 2 - DEPTH || dma_burst<MODE, SAVE>(j)) n += tile_stores<MODE, SAVE>(j - 1);     // step j runs the epilogue of chunk j-1
    return n < 63 ? n : 63;
}

template <int N>
VDN_DEV void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA of 16 B per lane, written as inline asm: the builtin makes hipcc's wait-count pass treat every later LDS
// wait as out of order (it marks a pending FLAT access), and it then emits s_waitcnt lgkmcnt(0) in front of every MFMA
// that consumes a fragment - a full LDS round trip per MFMA group instead of a counted wait. Completion is tracked by the
// kernel's own counted vmcnt (the compiler does not see these loads).
// Addressing (round 5): immediate offsets, one M0 write per chunk (vdn_common.h: glds16_imm*). VDN_SDF2_DMA_IMM=0: the A/B arm
// with one scalar base and one M0 write per piece.
#ifndef VDN_SDF2_DMA_IMM
#define VDN_SDF2_DMA_IMM 1
#endif
VDN_DEV void glds16_saddr(const char* base_uniform, unsigned lane_off, char* lds_wave_base) {
    const unsigned lds = (unsigned)(size_t)lds_wave_base;
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" ::"v"(lane_off), "s"(lds), "s"(base_uniform) : "memory", "m0");
}
// ---- pipeline state ---------------------------------------------------------------------------------------------
template <int NSLOT, int SPLIT = (1 << 30)>
struct Pipe {
    const char* g;      // weight stream (wave-uniform)
    const char* g2;     // chunks from SPLIT on: a second stream (MODE 2: the colour head's)
    char* lds;          // ring base
    int wave, lane;
    unsigned lane16;    // lane * 16
    bf16x8 fr[kPre];    // first fragments of the next chunk step (already read)
    f32x4 bias[4];      // its bias rows
    template <int C>
    VDN_DEV char* slot() const { return lds + (C % NSLOT) * kStride; }
    // DMA piece I (of kG) of chunk C: 1 KiB, wave-uniform base + per-lane 32-bit offset (scalar-base addressing)
    unsigned voff0;     // lane * 16 + this wave's first byte in a chunk + 4096 (the centre of the pieces' immediate offsets)
    unsigned m0_wave;   // LDS byte address of the same place in ring slot 0
    VDN_DEV void init_dma() {
        voff0 = lane16 + wave * (kG * 1024) + 4096;
        m0_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + wave * (kG * 1024) + 4096);
    }
    template <int C, int I>
    VDN_DEV void issue_piece() {
#if !(VDN_SDF2_ABL & 4)
#if VDN_SDF2_DMA_IMM
        static_assert(kG <= 8, "one group of immediate offsets");
        constexpr int IMM = glds_imm(I);
        constexpr long COFF = (C >= SPLIT ? (long)(C - SPLIT) : (long)C) * kStride;
        static_assert(COFF + kStride < (1L << 31), "32-bit chunk offsets");
        const char* base = C >= SPLIT ? g2 : g;
        const unsigned voff = voff0 + (unsigned)COFF;
        if constexpr (I == 0) glds16_imm_m0add<(C % NSLOT) * kStride, IMM>(base, voff, m0_wave);
        else glds16_imm<IMM>(base, voff);
#else
        const int piece = wave + I * kWaves;
        if constexpr (C >= SPLIT) glds16_saddr(g2 + ((long)(C - SPLIT) * kStride + piece * 1024), lane16, slot<C>() + piece * 1024);
        else glds16_saddr(g + ((long)C * kStride + piece * 1024), lane16, slot<C>() + piece * 1024);
#endif
#endif
    }
    template <int C>
    VDN_DEV void issue() {
        static_for<kG>([&](auto i_c) VDN_INL { issue_piece<C, decltype(i_c)::value>(); });
    }
    // reads that open chunk step C (its first fragments, and its bias rows)
    template <int C, int KT, bool BIAS>
    VDN_DEV void prefetch() {
        const char* w = slot<C>();
        const bf16x8* wa = reinterpret_cast<const bf16x8*>(w) + lane;
#pragma unroll
        for (int s = 0; s < (kPre < 2 * KT ? kPre : 2 * KT); ++s) fr[s] = wa[s * 64];
        if constexpr (BIAS) load_bias(reinterpret_cast<const f32x4*>(w + KT * 2048));
    }
    VDN_DEV void load_bias(const f32x4* b) {
#pragma unroll
        for (int g = 0; g < (kS16 ? 2 : 4); ++g) bias[g] = b[quad_of(g, lane)];
    }
};

// One chunk step: acc = (bias) + W[chunk C] . X over KT input tiles. group(gi, NG) = the VALU work assigned to MFMA
// group gi of NG (kGroup MFMAs per group). Group 0 certifies chunk C+1; the DMA pieces of chunk C+DEPTH follow one per
// group; the tail reads the opening fragments of chunk C+1.
template <int MODE, bool SAVE, int NSLOT, int DEPTH, int C, class PipeT, class ActT, class Group>
VDN_DEV AccT chunk_step(PipeT& pp, const ActT& X, Group&& group) {
    using PG = Prog<MODE>;
    constexpr int KT = PG::kt_of(C);
    constexpr bool BIAS = PG::bias_of(C);
    constexpr int NS = KT * 2, NG = (NS + kGroup - 1) / kGroup;
    constexpr int KTN = PG::kt_of(C + 1);
    constexpr bool HAS_NEXT = C + 1 < PG::total;
    constexpr bool HAS_DMA = C + DEPTH < PG::total;
    const bf16x8* wa = reinterpret_cast<const bf16x8*>(pp.template slot<C>()) + pp.lane;
    const bf16x8* wn = reinterpret_cast<const bf16x8*>(pp.template slot<C + 1>()) + pp.lane;
    bf16x8 fr[NS];
    AccT acc;
    constexpr int PF = kPre < NS ? kPre : NS;                       // fragments of this chunk read by the previous step
    constexpr int PFN = kPre < 2 * KTN ? kPre : 2 * KTN;            // fragments of the next chunk this step reads
    static_for<PF>([&](auto s_c) VDN_INL { fr[decltype(s_c)::value] = pp.fr[decltype(s_c)::value]; });
    acc.fill(pp.bias, BIAS);
    __builtin_amdgcn_sched_barrier(0);
    static_for<NG>([&](auto g_c) VDN_INL {
        constexpr int gi = decltype(g_c)::value;
        constexpr int s0 = gi * kGroup, s1 = (gi + 1) * kGroup < NS ? (gi + 1) * kGroup : NS;
        static_for<s1 - s0>([&](auto j_c) VDN_INL {
            constexpr int s = s0 + decltype(j_c)::value;
#if VDN_SDF2_ABL & 2
            { const bf16x8 keep = fr[s]; asm volatile("" ::"v"(keep)); }
#else
            acc.template mfma<s>(fr[s], X);
#endif
        });
        // B2: with two spare ring slots the workgroup re-synchronises every SECOND step - the even step certifies chunks C+1 and C+2
        // (the latter's DMA is one step old: an L2-warm piece lands in 250 - 400 cycles of the ~1 300 a step takes); the slot a step's
        // DMA overwrites, that of chunk C-2, was last read in step C-2, which every wave had left at the latest barrier (C or C-1)
        // (VDN_SDF2_B2 = 2: only for the chunks in front of the sweep - behind them one barrier per step again, with the spare slot idle;
        // the change-over needs nothing: a step of the second kind certifies chunk C+1 by the ordinary count, whether or not the
        // even step before it already did)
        constexpr bool B2 = VDN_SDF2_B2 != 0 && NSLOT >= DEPTH + 2 && (VDN_SDF2_B2 != 2 || C < PG::first_chunk(PG::NL < 9 ? PG::NL : 9));
        if constexpr (gi == 0 && HAS_NEXT && (!B2 || C % 2 == 0)) {
            __builtin_amdgcn_sched_barrier(0);      // the step's first MFMAs are in the pipe while the wave waits
            wait_vmcnt<B2 ? wait_count_b2<MODE, SAVE>(C) : wait_count<MODE, SAVE, DEPTH>(C)>();
#if !(VDN_SDF2_ABL & 8)
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // DMA of chunk C+DEPTH (into the slot chunk C-1 was read from: every wave is past the barrier), spread over the groups
        if constexpr (HAS_DMA && dma_burst<MODE, SAVE>(C)) {
            if constexpr (gi == 0) static_for<kG>([&](auto i_c) VDN_INL { pp.template issue_piece<C + DEPTH, decltype(i_c)::value>(); });
        } else if constexpr (HAS_DMA) {
            static_for<kG>([&](auto i_c) VDN_INL {
                constexpr int i = decltype(i_c)::value;
                if constexpr ((NG >= kG ? i * NG / kG : (i < NG ? i : NG - 1)) == gi) pp.template issue_piece<C + DEPTH, i>();
            });
        }
        // fragment reads kPre MFMAs ahead: the rest of this chunk, then the opening fragments of chunk C+1
        static_for<s1 - s0>([&](auto j_c) VDN_INL {
            constexpr int s = s0 + decltype(j_c)::value;
            if constexpr (s + PF < NS) fr[s + PF] = wa[(s + PF) * 64];
            else if constexpr (HAS_NEXT && s + PF - NS < PFN) pp.fr[s + PF - NS] = wn[(s + PF - NS) * 64];
            // a chunk shorter than the next one's opening: its last MFMA slot reads the remainder
            if constexpr (HAS_NEXT && s == NS - 1)
                static_for<(PFN > PF ? PFN - PF : 0)>([&](auto e_c) VDN_INL { pp.fr[PF + decltype(e_c)::value] = wn[(PF + decltype(e_c)::value) * 64]; });
        });
        if constexpr (gi == NG - 1 && HAS_NEXT && KTN > 0 && PG::bias_of(C + 1)) {
            pp.load_bias(reinterpret_cast<const f32x4*>(pp.template slot<C + 1>() + KTN * 2048));
        }
        group(g_c, std::integral_constant<int, NG>{});
        __builtin_amdgcn_sched_barrier(0);
    });
    return acc;
}

#ifndef VDN_SDF2_SP
#define VDN_SDF2_SP 1   // softplus form: 0 = max(t,0) + log2(1 + 2^-|t|), sigma from 2^-g;  1 = med3(log2(1 + 2^t), t, 25), sigma from 1 / (1 + 2^t)
#endif
// softplus in scaled units, t -> g = log2(1 + 2^t), and E = 2^-g = 1 - sigma(100 a) = 1 / (1 + 2^t).
// Form 1 (one VALU instruction fewer per value, and E no longer waits for g): w = 1 + 2^t overflows to +inf from t = 128 on
// (log2 -> +inf) and for t >= 25 the f32 value of log2(w) is t itself, so g = median(log2(w), t, 25) - one v_med3_f32 - is exact
// on both sides: below 25 the logarithm lies between t and 25, from 25 on t lies between 25 and the (possibly infinite)
// logarithm. For t <= -25 w rounds to 1 and g to 0 (true value < 2^-25 units = 2e-10 in h). E = 1/w: 0 at +inf, 1 at w = 1.
struct SpE { float g, e; };
VDN_DEV SpE softplus_sigma(float t) {
    SpE r;
#if VDN_SDF2_ABL & 1
    r.g = t; r.e = t;
    return r;
#endif
#if VDN_SDF2_SP == 1
    const float w = 1.0f + __builtin_amdgcn_exp2f(t);
    r.g = __builtin_amdgcn_fmed3f(__builtin_amdgcn_logf(w), t, 25.0f);
    r.e = __builtin_amdgcn_rcpf(w);
#else
    const float e = __builtin_amdgcn_exp2f(-fabsf(t));
    r.g = relu0(t) + __builtin_amdgcn_logf(1.0f + e);
    r.e = __builtin_amdgcn_exp2f(-r.g);
#endif
    return r;
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef VDN_SDF2_WT_SAVES
#define VDN_SDF2_WT_SAVES 1
#endif
// 16-byte store of a saved-plane piece. VDN_SDF2_WT_SAVES = 1: write-through (sc1), which does not keep the line in this XCD's
// L2 - the planes are next read by other kernels, from HBM anyway, while the L2 is what feeds this kernel's weight stream
// (development harness, 65 536 rows, variants interleaved: training launch 205.6 -> 196.8 us, inference launch 161.9 -> 157.1)
VDN_DEV void plane_store16(unsigned short* p, const u32x4& v) {
#if VDN_SDF2_WT_SAVES
#if VDN_SAVE_NT             // non-temporal as well (vdn_common.h)
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#endif
#else
    *reinterpret_cast<u32x4*>(p) = v;
#endif
}

// four values E = 1 - sigma in [0,1] -> one dword of 255 sigma (8 bits each): unorm16 conversion (round(65535 E), two
// values per instruction), the high bytes gathered by one v_perm, complemented (255 - q). Exact at E = 0 and E = 1.
VDN_DEV unsigned sigma255_pack(float e0, float e1, float e2, float e3) {
    const unsigned d0 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(e0, e1));
    const unsigned d1 = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(e2, e3));
    return ~__builtin_amdgcn_perm(d1, d0, 0x07050301u);
}
VDN_DEV float ubyte_f32(unsigned w, int b) { return (float)((w >> (8 * b)) & 0xffu); }

// S store: tiles [0, NLDS) in LDS (16 B per lane per tile, wave-private), the rest in registers
template <int NLDS>
struct SStore {
    static constexpr int NREG = kSTiles > NLDS ? (kSTiles - NLDS) : 0;
    u32x4 reg[NREG > 0 ? NREG : 1];
    char* lds;      // this lane's 16 bytes of tile 0
    template <int T>
    VDN_DEV void put(const u32x4& v) {
        if constexpr (T < NLDS) *reinterpret_cast<u32x4*>(lds + T * 1024) = v;
        else reg[T - NLDS] = v;
    }
    template <int T>
    VDN_DEV u32x4 get() const {
        if constexpr (T < NLDS) return *reinterpret_cast<const u32x4*>(lds + T * 1024);
        else return reg[T - NLDS];
    }
};

// element pairs [pair_begin(gi), pair_begin(gi+1)) of a 16-register tile are worked on in MFMA group gi of the GA groups
// available to the tile's epilogue
constexpr int pair_begin(int gi, int GA) { return gi >= GA ? 8 : gi * 8 / GA; }

// MODE 0: sdf only (sampler passes, 2 waves / SIMD). MODE 1: sdf + feature + normals, softplus' on chip (1 wave / SIMD);
// SAVE adds the training saves (H in scaled units g = 100 log2(e) h, V, PE).
// (VID only gives the development harness's co-linked tuning variants distinct symbols)
// UPS (MODE 0, ray form with 64 samples per ray): the first up-sampling round (renderer.py:147-191 on the coarse samples)
// behind the pass - a 128-point workgroup is two rays, whose z / sdf rows go through LDS to upsample_row (k_ray_rows.h),
// vdn_upsample_round's work without its launch.
// MODE 3 (round 5, the training step's forward: fields.py:72-108 + 148-176 in one launch): MODE 1 with the training saves AND the colour
// head on the feature vector kept in registers, its hidden activations / small inputs / output saved for the backward (the planes
// rendernet_fwd_kernel writes); no compositing (the step's rows are a compacted work list: a workgroup is not a ray).
// MODE 2 (the north-star kernel of the inference path, reference renderer.py:239-315 in one launch): MODE 1 with the feature
// vector kept in registers, then the colour head on the same 32 points per wave, then - a 128-point workgroup being exactly one
// ray of 128 samples - the ray's NeuS alpha, background blend, transmittance scan and weighted sums (k_composite_row.h) by wave 0
// on the samples handed over in LDS, and the eikonal sums of all rays by the ray that finishes last.
struct ShadeExtra {
    const char* color_blob;     // the colour head's chunk stream ("c2", vdn_hip/images.py), same chunk format and stride
    int* ticket;                // [1] arrival counter, zero before the first launch (the last ray leaves it zero)
    int squeeze_out;            // fields.py:170-171
    int warm_bytes, warm_bytes2;    // all modes: bytes of the weight stream(s) the first round of workgroups pulls into L2 up front (0 = off; vdn_common.h: warm_l2)
    void* col_h;                // MODE 3: [4, rows, 256] the colour head's saved hidden activations (PT32 planes, compact rows)
    void* col_small;            // MODE 3: [rows, 64] its 33 small inputs (points, PE4(view), normal) as the weight-gradient GEMM reads them
    float* col_out;             // MODE 3: [P,3] the sampled colour (dense point id)
    CompositeArgs cm;           // sdf / normals / color are not read (the samples come through LDS)
};

template <int MODE, bool SAVE, int NSLOT, int DEPTH, int VID = 0, bool UPS = false>
__global__ __launch_bounds__(kWaves * 64, MODE == 0 ? 2 : 1) void sdf_fwd2_kernel(SdfArgs a, UpsampleArgs up, ShadeExtra ex) {
    static_assert(!UPS || MODE == 0, "the up-sampling round follows the sdf-only pass");
    using PG = Prog<MODE>;
    static_assert(MODE != 3 || SAVE, "MODE 3 is the training forward");
    constexpr bool TS = SAVE && (MODE == 1 || MODE == 3);          // training saves (H, V, PE planes)
    constexpr bool FEAT = MODE == 1 || MODE == 3 || (MODE == 2 && SAVE);     // the feature plane goes to HBM
    constexpr bool COL = MODE >= 2;                 // the colour head follows the sweep
    using P = BF16;
    using ST = unsigned short;
    static_assert(NSLOT >= DEPTH + 1, "ring: the chunk being read, the one being opened and DEPTH-1 in flight");
    constexpr int kRing = NSLOT * kStride;
    constexpr int kW8 = 0;                             // (W8 row 0 rides in every chunk's tail: kTail)
    constexpr int kLdsTotal = MODE >= 1 ? 160 * 1024 : kRing + kW8;
    constexpr int NLDS = MODE >= 1 ? ((kLdsTotal - kRing - kW8) / (kWaves * 1024) < kSTiles ? (kLdsTotal - kRing - kW8) / (kWaves * 1024) : kSTiles) : 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Pipe<NSLOT, MODE >= 2 ? PG::sdf_total : (1 << 30)> pp;
    pp.g = a.blob;
    pp.g2 = ex.color_blob;
    pp.lds = smem;
    pp.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    pp.lane = threadIdx.x & 63;
    pp.lane16 = pp.lane * 16;
    pp.init_dma();
    const int lane = pp.lane, c = lane & 31, h = lane >> 5;
    const WorkRow wr = work_row(a.active_idx, a.n_active, a.P, kWaves, pp.wave, c);
    if (wr.none) return;
    if constexpr (MODE == 1) {
        // the list's tail is vdn_sdf_fwd_tail_bf16's (k_sdf_fwd1_split.h) when it ends within tail_max_rows behind tail_row0
        if (a.tail_max_rows > 0) {
            const long n_rows = a.active_idx != nullptr ? (long)*a.n_active : (long)a.P;
            if (n_rows > a.tail_row0 && n_rows - a.tail_row0 <= a.tail_max_rows && (long)blockIdx.x * (kWaves * 32) >= a.tail_row0) return;
        }
    }
#ifdef VDN_SDF2_STAMP      // development harness: shader-clock and 100-MHz stamps of the workgroup, into the (otherwise unused) PE buffer
    const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long stamp_p1 = 0, stamp_p2 = 0;     // end of the hidden layers / of the last layer
#endif
    const bool ok = wr.ok;
    const long p = wr.row, pd = wr.point;
    const int c16 = lane & 15, q4 = lane >> 4;      // shape 1: this lane's column of both 16-point halves, and its row quad
    (void)c16; (void)q4;

    float xin[3];
    float px[3] = {0.0f, 0.0f, 0.0f}, dir[3] = {0.0f, 0.0f, 0.0f};     // MODE 2: the point and the view direction (the colour head's inputs)
    long sdf_idx = pd;
    float z_keep = 0.0f, sdf_keep = 0.0f;           // UPS: this point's depth and sdf for the up-sampling round
    if (a.pts != nullptr) {
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = a.pts[pd * 3 + d] * a.scale;
    } else {
        const long r = pd / a.n_per_ray;
        const long sidx = pd - r * a.n_per_ray;
        const float z = a.z[r * a.z_ld + sidx];
        z_keep = z;
        sdf_idx = r * a.sdf_ld + sidx;
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = (a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z) * a.scale;
        if constexpr (COL) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                dir[d] = a.rays_d[r * 3 + d];
                px[d] = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;          // renderer.py:233
            }
        }
    }
    ST* Hs = reinterpret_cast<ST*>(a.H);
    ST* Vs = reinterpret_cast<ST*>(a.V);
    ST* feat = reinterpret_cast<ST*>(a.feat);
    const long PS = P::plane(a.P, 256);
    const long prow = (p >> 5) * (32L * 256) + h * 256 + (p & 31) * 8;       // PT32 offset of this lane's 16-byte pieces (mlp_engine.h)
    // shape 1: the lane's unit (k = q4 >> 1, h = q4 & 1) of its two points, rows clamped as work_row clamps them
    long prow2[2] = {0, 0};
    if constexpr (kS16) {
        const long n_rows = a.active_idx != nullptr ? (long)*a.n_active : (long)a.P;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const long raw = ((long)blockIdx.x * kWaves + pp.wave) * 32 + 16 * ph + c16;
            const long r = raw < n_rows ? raw : (n_rows > 0 ? n_rows - 1 : 0);
            prow2[ph] = (r >> 5) * (32L * 256) + (q4 >> 1) * 512 + (q4 & 1) * 256 + (r & 31) * 8;
        }
    }
    // element offset of 16-byte piece k of tile T in a [*, 256] PT32 plane: shape 0 k = the tile's k-step, shape 1 k = the point half
    auto piece = [&](int T, int k) VDN_INL -> long { return kS16 ? prow2[k] + T * 1024 : prow + T * 1024 + 512 * k; };
    const float inv_scale = 1.0f / a.scale;
    // the input loads above have to be back first: the compiler waits for them with vmcnt(0), which would also wait for younger
    // warm-up loads; behind this wait the stream arrives in L2 while the encoding below is computed from registers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    char* const wdump = smem + pp.wave * (VDN_SDF2_DMA_IMM ? kG * 1024 : 1024);       // (this wave's own first DMA piece of ring slot 0: vdn_common.h)
    warm_l2_issue(a.blob, ex.warm_bytes, wr.n_wg, MODE == 0 ? 512 : 256, wdump);
    warm_l2_issue(COL ? ex.color_blob : nullptr, COL ? ex.warm_bytes2 : 0, wr.n_wg, 256, wdump);
    // (and the kernel's own code: vdn_common.h; MODE 2 is the inference launch and is never cold)
    constexpr int kCode = MODE == 0 ? kWarmCodeSdfFwd2Mode0 : (MODE == 1 ? (SAVE ? kWarmCodeSdfFwd2Save : kWarmCodeSdfFwd2) : (MODE == 3 ? kWarmCodeSdfFwd3 : 0));
    warm_code_issue(ex.warm_bytes > 0 ? kCode : 0, wr.n_wg, MODE == 0 ? 512 : 256, wdump);

    typename P::template Act<9> X, Y;
    typename P::template Act<9> F;          // MODE 2: the feature vector as the colour head's input fragments (tiles 0..7) + its small tile (8)
    float nz = 0.0f;                        // MODE 2: the normal's z component, the 33rd small input (an f32 rank-1 term in COL0's epilogue)
    float col[3] = {0.0f, 0.0f, 0.0f};      // MODE 2: the sampled colour of this lane's point
    SStore<NLDS> SS;
    SS.lds = smem + kRing + kW8 + pp.wave * (NLDS * 1024) + lane * 16;
    // shape 1: wave-private scratch for the crossings between home lanes and tile lanes. MODE >= 1: the wave's S region (not yet /
    // no longer holding softplus' tiles when used); MODE 0: a piece of the ring's last slot, which no DMA touches before step 0
    static_assert(!kS16 || MODE == 0 || NLDS * 1024 >= 32 * 272, "exchange scratch");
    static_assert(!kS16 || NSLOT - 1 >= DEPTH, "the last ring slot is free at the start");
    char* const xch = MODE >= 1 ? smem + kRing + kW8 + pp.wave * (NLDS * 1024) : smem + (NSLOT - 1) * kStride + pp.wave * 4096;
    // NF fragments in home-lane form (unit (s, h, c) = lane c + 32 h of fragment s) -> the same units in tile-lane form:
    // fragment 2 T + ph of lane (c16, q4) = unit (s = 2 T + (q4 >> 1), h = q4 & 1, c = 16 ph + c16)
    auto to_tile_lanes = [&](bf16x8* fr, auto nf_c) VDN_INL {
        constexpr int NF = decltype(nf_c)::value;
#pragma unroll
        for (int s = 0; s < NF; ++s) *reinterpret_cast<bf16x8*>(xch + s * 1024 + lane * 16) = fr[s];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < NF; ++s)
            fr[s] = *reinterpret_cast<const bf16x8*>(xch + ((s & ~1) + (q4 >> 1)) * 1024 + (q4 & 1) * 512 + (16 * (s & 1) + c16) * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // positional encoding in scaled units: X tiles 0,1 (layer 0) and a copy for the skip input of layer 4 (tiles 7,8):
    // tile 8 of X is not touched by anything else, tile 7's copy waits in registers
    bf16x8 pe7[2];
    {
        float pe39[39], pe[64];
        posenc<3, 6, false>(xin, pe39);
        // slots 39..63 of the 64-wide input (zero padding in the first kernel) carry the bf16 rounding residue of the first
        // 25 encoded values - the raw coordinates and the low octaves - against the same weight columns (kmap of the scaled
        // streams, vdn_hip/images.py): the network sees those inputs to ~16 bits. A bf16 coordinate alone is an SDF error of
        // up to 2e-3 at |x| ~ 1, against an alpha that multiplies the SDF by inv_s ~ 1e2 .. 1e4.
#pragma unroll
        for (int i = 0; i < 39; ++i) pe[i] = pe39[i] * kC1;
#pragma unroll
        for (int i = 0; i < 25; ++i) pe[39 + i] = fmaf(pe39[i], kC1, -bf16_lo(pack_bf16x2(pe[i], 0.0f)));      // (the residue of the exact product)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) X.set(kt, vals_tile<64>(pe, h, kt));
        if constexpr (TS) {         // saved in the same scaled units as H and V (include/vdn_render.h: VdnSdfArgs); the residue
            if (a.PE != nullptr) {  // slots are saved as zeros (the weight-gradient GEMM contracts over the 39 encoded values)
#pragma unroll
                for (int i = 39; i < 64; ++i) pe[i] = 0.0f;
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) P::store_tile(reinterpret_cast<ST*>(a.PE), p, 64, kt, h, vals_tile<64>(pe, h, kt), true);
            }
        }
        if constexpr (kS16) to_tile_lanes(&X.r[0], std::integral_constant<int, 4>{});
        pe7[0] = X.r[0]; pe7[1] = X.r[1];
        X.r[16] = X.r[2]; X.r[17] = X.r[3];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the ordinary loads and stores above: nothing but DMA and counted stores from here on
    // ring start (behind the PE stores, so that nothing but counted operations is younger than a DMA): W8 row 0 into its
    // fixed place (wave 0), chunks 0 .. DEPTH-1 in flight, chunk 0 certified, its opening fragments read
    static_for<DEPTH>([&](auto i_c) VDN_INL { pp.template issue<decltype(i_c)::value>(); });
    wait_vmcnt<(DEPTH - 1) * kG>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    pp.template prefetch<0, 2, true>();

    AccT acc_prev;              // accumulator of the previous chunk's tile (its epilogue runs under this chunk's MFMAs)
    u32x4 sq_prev;              // 255 sigma of the previous chunk's tile: read by the sweep's epilogue, built by a hidden layer's
    u32x4 sq_v7;                // 255 sigma_7 tile while v7 is formed
    unsigned hold0 = 0, hold1 = 0;  // a pair's values waiting for their partners (one 4-value pack / one 16-byte store)
    unsigned fhold0 = 0, fhold1 = 0, vhold0 = 0, vhold1 = 0;    // first halves of 16-byte plane pieces (feature / V)
    f32x4 w8hold;               // W8 row 0 at the features of the pair being worked on
    float sdf_dot[2] = {0.0f, 0.0f};    // this lane's part of  W8[0,:] . g8  (f32), per point half (shape 0: [0] only)
    float nz2[2] = {0.0f, 0.0f};        // shape 1: nz of the lane's two points
    AccT UPE[2];                // d sdf / d(PE) tiles (W4^T rows 7,8 and W0^T)
    float n[3] = {0.0f, 0.0f, 0.0f};
    auto pe_backward = [&](bool first) VDN_INL {      // n += J_PE^T u  (transpose Jacobian of the encoding)
        float u[39];
#if VDN_SDF2_S16
        {   // tile lanes -> home lanes through the scratch: f32 [32 points][68] (272-byte rows: the b128 reads of 32 rows spread over the banks)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(xch + (16 * (g >> 1) + c16) * 272 + (32 * t + 4 * quad_of(g, lane)) * 4) = UPE[t].v[g];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const f32x4 r = *reinterpret_cast<const f32x4*>(xch + c * 272 + i * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (4 * i + j < 39) u[4 * i + j < 39 ? 4 * i + j : 0] = r[j];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#else
        {
            const f32x16 upe[2] = {UPE[0].w, UPE[1].w};
            tiles_vals<39, 2>(upe, h, u);
        }
#endif
        // u = u_4[PE part] + u_0 for the ray adjoint (VdnSdfArgs.U_pe); vector-memory operations the wait counts do not
        // know of are harmless: they only make a counted wait return later
        if (a.U_pe != nullptr && ok && h == 0) {
#pragma unroll
            for (int i = 0; i < 39; ++i) a.U_pe[p * 39 + i] = first ? u[i] : a.U_pe[p * 39 + i] + u[i];
        }
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

    // epilogue of chunk CP's tile, element pairs [PB, PE): runs in step CP+1
    auto epilogue = [&](auto cp_c, auto pb_c, auto pe_c) VDN_INL {
        constexpr int CP = decltype(cp_c)::value, PB = decltype(pb_c)::value, PE_ = decltype(pe_c)::value;
        if constexpr (CP >= 0 && PE_ > PB) {
            constexpr int LI = PG::layer_of(CP), T = PG::tile_of(CP);
            constexpr LayerDesc L = PG::layer(LI);
            auto& D = (LI & 1) ? X : Y;                 // layer LI reads (LI & 1 ? Y : X) and writes the other
            if constexpr (L.kind == HID) {
                static_for<PE_ - PB>([&](auto i_c) VDN_INL {
                    constexpr int pr = PB + decltype(i_c)::value;       // elements 2pr, 2pr+1
                    const SpE sp0 = softplus_sigma(acc_prev[2 * pr]), sp1 = softplus_sigma(acc_prev[2 * pr + 1]);
                    const float g0 = sp0.g, g1 = sp1.g;
                    if constexpr (L.l == 7) {
                        // the sdf row of the last layer in f32 on the VALU, from the unrounded activations: 2 FMAs per pair on
                        // one layer's epilogue, and the output that the alpha multiplies by inv_s loses no bits to bf16
                        if constexpr ((pr & 1) == 0) w8hold = *(reinterpret_cast<const f32x4*>(pp.template slot<CP + 1>() + kTail) + (8 * T + quad_of(pr >> 1, lane)));
                        sdf_dot[ph_of_pair(pr)] = fmaf(g0, w8hold[2 * (pr & 1)], fmaf(g1, w8hold[2 * (pr & 1) + 1], sdf_dot[ph_of_pair(pr)]));
                    }
                    unsigned pk = pack_bf16x2(g0, g1);
                    asm volatile("" : "+v"(pk));
                    u32x4 cur = __builtin_bit_cast(u32x4, D.r[T * 2 + (pr >> 2)]);
                    cur[pr & 3] = pk;
                    D.r[T * 2 + (pr >> 2)] = __builtin_bit_cast(bf16x8, cur);
                    if constexpr (MODE >= 1) {
                        if constexpr ((pr & 1) == 0) {
                            hold0 = __builtin_bit_cast(unsigned, sp0.e);
                            hold1 = __builtin_bit_cast(unsigned, sp1.e);
                            asm volatile("" : "+v"(hold0), "+v"(hold1));
                        } else {
                            unsigned w = sigma255_pack(__builtin_bit_cast(float, hold0), __builtin_bit_cast(float, hold1), sp0.e, sp1.e);
                            asm volatile("" : "+v"(w));      // materialise here: otherwise the chain sinks to the tile's end
                            sq_prev[pr >> 1] = w;
                            if constexpr (TS && (pr & 3) == 3)          // H plane piece k = pr >> 2 = this k-step's whole B fragment
                                plane_store16(Hs + L.l * PS + piece(T, pr >> 2), cur);
                            if constexpr (pr == 7) SS.template put<PG::s_tile0(L.l) + T>(sq_prev);
                        }
                    }
                });
            } else if constexpr (L.kind == LAST) {
                if constexpr (MODE >= 1 && T < 8) {
                    // feature tile T to HBM (MODE 2: into F, the colour head's input fragments; to HBM only for a VDN head); v7 tile T = (W8 row 0 / scale) (.) 255 sigma_7 -> Y (free while layer 8 reads X)
                    static_for<PE_ - PB>([&](auto i_c) VDN_INL {
                        constexpr int pr = PB + decltype(i_c)::value;
                        if constexpr ((pr & 1) == 1) {
                            constexpr int q = pr >> 1;
                            if constexpr ((q & 1) == 0) {       // first half of the 16-byte piece k = q >> 1 waits for the second
                                fhold0 = pack_bf16x2(acc_prev[4 * q], acc_prev[4 * q + 1]);
                                fhold1 = pack_bf16x2(acc_prev[4 * q + 2], acc_prev[4 * q + 3]);
                                asm volatile("" : "+v"(fhold0), "+v"(fhold1));
                            } else {
                                u32x4 o;
                                o[0] = fhold0;
                                o[1] = fhold1;
                                o[2] = pack_bf16x2(acc_prev[4 * q], acc_prev[4 * q + 1]);
                                o[3] = pack_bf16x2(acc_prev[4 * q + 2], acc_prev[4 * q + 3]);
                                if constexpr (FEAT) plane_store16(feat + piece(T, q >> 1), o);
                                if constexpr (COL) F.r[T * 2 + (q >> 1)] = __builtin_bit_cast(bf16x8, o);      // (the piece IS the B fragment)
                            }
                            const f32x4 w = *(reinterpret_cast<const f32x4*>(pp.template slot<CP + 1>() + kTail) + (8 * T + quad_of(q, lane)));
                            if constexpr (q == 0) sq_v7 = SS.template get<PG::s_tile0(7) + T>();
                            const unsigned sw = sq_v7[q];
                            const float v0 = w[0] * inv_scale * ubyte_f32(sw, 0), v1 = w[1] * inv_scale * ubyte_f32(sw, 1);
                            const float v2 = w[2] * inv_scale * ubyte_f32(sw, 2), v3 = w[3] * inv_scale * ubyte_f32(sw, 3);
                            u32x4 cur = __builtin_bit_cast(u32x4, Y.r[T * 2 + (q >> 1)]);
                            unsigned k0 = pack_bf16x2(v0, v1), k1 = pack_bf16x2(v2, v3);
                            asm volatile("" : "+v"(k0), "+v"(k1));      // materialise here (the sweep is the first reader)
                            cur[2 * (q & 1)] = k0;
                            cur[2 * (q & 1) + 1] = k1;
                            Y.r[T * 2 + (q >> 1)] = __builtin_bit_cast(bf16x8, cur);
                            if constexpr (TS) {
                                if constexpr ((q & 1) == 0) {
                                    vhold0 = pack_bf16x2(v0 * kVSave, v1 * kVSave);
                                    vhold1 = pack_bf16x2(v2 * kVSave, v3 * kVSave);
                                    asm volatile("" : "+v"(vhold0), "+v"(vhold1));
                                } else {
                                    u32x4 ov;
                                    ov[0] = vhold0;
                                    ov[1] = vhold1;
                                    ov[2] = pack_bf16x2(v0 * kVSave, v1 * kVSave);
                                    ov[3] = pack_bf16x2(v2 * kVSave, v3 * kVSave);
                                    plane_store16(Vs + 7 * PS + piece(T, q >> 1), ov);
                                }
                            }
                        }
                    });
                }
            } else if constexpr (L.kind == SWEEP || (L.kind == SWEEP_SKIP && T < 7)) {
                // v_l tile = u (.) 255 sigma_l   (the 1/255 is in the next transposed image)
                static_for<PE_ - PB>([&](auto i_c) VDN_INL {
                    constexpr int pr = PB + decltype(i_c)::value;
                    const float v0 = acc_prev[2 * pr] * ubyte_f32(sq_prev[pr >> 1], 2 * (pr & 1));
                    const float v1 = acc_prev[2 * pr + 1] * ubyte_f32(sq_prev[pr >> 1], 2 * (pr & 1) + 1);
                    unsigned pk = pack_bf16x2(v0, v1);
                    asm volatile("" : "+v"(pk));
                    u32x4 cur = __builtin_bit_cast(u32x4, D.r[T * 2 + (pr >> 2)]);
                    cur[pr & 3] = pk;
                    D.r[T * 2 + (pr >> 2)] = __builtin_bit_cast(bf16x8, cur);
                    if constexpr (TS) {
                        // the V plane holds 100 log2(e) v, v = u (.) sigma: the units of H and PE
                        unsigned pv = pack_bf16x2(v0 * kVSave, v1 * kVSave);
                        if constexpr ((pr & 3) == 0) {
                            vhold0 = pv;
                            asm volatile("" : "+v"(vhold0));
                        } else if constexpr ((pr & 3) == 1) {
                            vhold1 = pv;
                            asm volatile("" : "+v"(vhold1));
                        } else if constexpr ((pr & 3) == 2) {
                            hold0 = pv;
                            asm volatile("" : "+v"(hold0));
                        } else {        // 16-byte piece k = pr >> 2
                            u32x4 o;
                            o[0] = vhold0;
                            o[1] = vhold1;
                            o[2] = hold0;
                            o[3] = pv;
                            plane_store16(Vs + L.l * PS + piece(T, pr >> 2), o);
                        }
                    }
                });
            } else if constexpr (L.kind == COL0 || L.kind == COLH) {
                // colour head, hidden layers: ReLU, packed as the next layer's input fragments. COL0 adds the 33rd small input, the
                // normal's z component, as an f32 rank-1 term: its weight column rides in every colour chunk's tail (f32 x 256)
                static_for<PE_ - PB>([&](auto i_c) VDN_INL {
                    constexpr int pr = PB + decltype(i_c)::value;
                    float a0 = acc_prev[2 * pr], a1 = acc_prev[2 * pr + 1];
                    if constexpr (L.kind == COL0) {
                        if constexpr ((pr & 1) == 0) w8hold = *(reinterpret_cast<const f32x4*>(pp.template slot<CP + 1>() + kTail) + (8 * T + quad_of(pr >> 1, lane)));
                        const float nzp = kS16 ? nz2[ph_of_pair(pr)] : nz;
                        a0 = fmaf(w8hold[2 * (pr & 1)], nzp, a0);
                        a1 = fmaf(w8hold[2 * (pr & 1) + 1], nzp, a1);
                    }
                    unsigned pk = pack_bf16x2(relu0(a0), relu0(a1));
                    asm volatile("" : "+v"(pk));
                    u32x4 cur = __builtin_bit_cast(u32x4, D.r[T * 2 + (pr >> 2)]);
                    cur[pr & 3] = pk;
                    D.r[T * 2 + (pr >> 2)] = __builtin_bit_cast(bf16x8, cur);
                    if constexpr (MODE == 3 && (pr & 3) == 3)          // the saved plane piece k = pr >> 2 = this k-step's whole B fragment (as the H planes)
                        plane_store16(reinterpret_cast<ST*>(ex.col_h) + L.l * PS + piece(T, pr >> 2), cur);
                });
            } else if constexpr (L.kind == COLOUT) {
                // rows 0..2 of the output tile = registers 0..2 of the h = 0 lanes (fields.py:166-171)
                if constexpr (PB == 0) {
                    if constexpr (kS16) {
                        // rows 0..2 = registers 0..2 of feature half 0 of the q4 = 0 lanes, per point half: home lane c takes its point's from
                        // lane c & 15 (itself for c < 16)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const float lo = acc_prev[j], hi = __shfl(acc_prev[8 + j], lane & 15);
                            const float x = (lane & 16) ? hi : lo;
                            col[j] = ex.squeeze_out ? sigmoidf_(x) : relu0(x);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 3; ++j) col[j] = ex.squeeze_out ? sigmoidf_(acc_prev[j]) : relu0(acc_prev[j]);
                    }
                }
            }
        }
    };

    // ---- the flat stream of chunk steps ---------------------------------------------------------------------------
    static_for<PG::total + 1>([&](auto c_c) VDN_INL {
        constexpr int C = decltype(c_c)::value;
        constexpr int CP = C - 1;
        constexpr int CPs = CP >= 0 ? CP : 0;
        if constexpr (C < PG::total) {
            constexpr int LI = PG::layer_of(C), T = PG::tile_of(C);
            constexpr LayerDesc L = PG::layer(LI);
            constexpr int NS = 2 * L.kt, NG = (NS + kGroup - 1) / kGroup;
#ifdef VDN_SDF2_STAMP
            if constexpr (C == PG::first_chunk(8)) stamp_p1 = __builtin_amdgcn_s_memtime();
            if constexpr (MODE == 1 && C == PG::first_chunk(MODE == 1 ? 9 : 8)) stamp_p2 = __builtin_amdgcn_s_memtime();
#endif
            // groups available to the pending epilogue: inside a layer all of them; across a layer boundary those before
            // the MFMA group that reads the pending tile (k-steps 2 T', 2 T' + 1 of this layer's input)
            constexpr bool boundary = CP >= 0 && PG::layer_of(CPs) != LI;
            constexpr int KP = PG::layer(PG::layer_of(CPs)).kind;
            constexpr bool prev_writes_input = boundary && (KP == HID || KP == SWEEP || KP == COL0 || KP == COLH);
            constexpr int GA = prev_writes_input ? ((2 * PG::tile_of(CPs)) / kGroup < NG ? (2 * PG::tile_of(CPs)) / kGroup : NG) : NG;
            static_assert(GA >= 1, "a pending tile needs at least one MFMA group before its reader");
            if constexpr (L.kind == HID && L.l == 4 && T == 0) { X.r[14] = pe7[0]; X.r[15] = pe7[1]; }
            // sweep: fetch this tile's 255 sigma one step before its epilogue runs
            constexpr bool sweep_tile = L.kind == SWEEP || (L.kind == SWEEP_SKIP && T < 7);
            u32x4 sq_next;
            if constexpr (sweep_tile) sq_next = SS.template get<PG::s_tile0(L.l) + T>();
            const auto& Xin = L.kind == COL0 ? F : ((LI & 1) ? Y : X);
            const AccT acc_cur = chunk_step<MODE, SAVE, NSLOT, DEPTH, C>(pp, Xin, [&](auto g_c, auto) VDN_INL {
                constexpr int gi = decltype(g_c)::value;
                epilogue(std::integral_constant<int, CP>{}, std::integral_constant<int, pair_begin(gi, GA)>{},
                         std::integral_constant<int, pair_begin(gi + 1, GA)>{});
            });
            if constexpr (L.kind == SWEEP_SKIP && T >= 7) UPE[T - 7] = acc_cur;
            if constexpr (L.kind == SWEEP_PE) {
                // shape 1: ONE crossing to the home lanes, on u_4[PE part] + u_0 (the encoding's adjoint is linear)
                if constexpr (kS16) UPE[T].add(acc_cur);
                else UPE[T] = acc_cur;
            }
            if constexpr (L.kind == LAST && T == L.nt - 1) {
                // sdf = W8[0,:] . h8 + b8[0]: the f32 dot of layer 7's epilogue (g8 = 100 log2(e) h8), the bias from this
                // chunk's bias block (row 0); the MFMA's own bf16 value of the row is not used
                const float b0 = *reinterpret_cast<const float*>(pp.template slot<C>() + L.kt * 2048);
                float dot;
                if constexpr (kS16) {       // the four row quads of a point; home lane c + 32 h is a tile lane of point half (c >> 4)
                    float d0 = sdf_dot[0], d1 = sdf_dot[1];
                    d0 += __shfl_xor(d0, 16); d1 += __shfl_xor(d1, 16);
                    d0 += __shfl_xor(d0, 32); d1 += __shfl_xor(d1, 32);
                    dot = (lane & 16) ? d1 : d0;
                } else {
                    dot = sdf_dot[0] + __shfl_xor(sdf_dot[0], 32);
                }
                sdf_keep = fmaf(dot, 1.0f / kC1, b0) * inv_scale;
                if (ok && h == 0) a.sdf[sdf_idx] = sdf_keep;
            }
            acc_prev = acc_cur;
            if constexpr (sweep_tile) sq_prev = sq_next;
            if constexpr ((!kS16 && L.kind == SWEEP_SKIP && T == 8) || (L.kind == SWEEP_PE && T == 1)) pe_backward(kS16 || L.kind == SWEEP_SKIP);
            if constexpr (COL && L.kind == SWEEP_PE && T == 1) {
                // the normal is complete: the colour head's small input tile [points (3), PE4(view) (27), normal x, y] (fields.py:154;
                // k order of the "c2" stream), z component kept in f32
                float small[32], pe[27];
                posenc<3, 4, P::kAccurateTrig>(dir, pe);
#pragma unroll
                for (int d = 0; d < 3; ++d) small[d] = px[d];
#pragma unroll
                for (int i = 0; i < 27; ++i) small[3 + i] = pe[i];
                small[30] = n[0] * a.scale;
                small[31] = n[1] * a.scale;
                nz = n[2] * a.scale;
                F.set(8, vals_tile<32>(small, h, 0));
                if constexpr (kS16) {
                    to_tile_lanes(&F.r[16], std::integral_constant<int, 2>{});
                    reinterpret_cast<float*>(xch)[c] = nz;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    nz2[0] = reinterpret_cast<const float*>(xch)[c16];
                    nz2[1] = reinterpret_cast<const float*>(xch)[16 + c16];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if constexpr (MODE == 3) {
                    // the colour head's 33 small inputs as rendernet_fwd_kernel saves them ([rows, 64]: the weight-gradient GEMM's operand).
                    // Uncounted vector-memory operations only make a counted wait return later (as the U_pe stores above)
                    float s33[33];
#pragma unroll
                    for (int i = 0; i < 32; ++i) s33[i] = small[i];
                    s33[32] = nz;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt) P::store_tile(reinterpret_cast<ST*>(ex.col_small), p, 64, kt, h, vals_tile<33>(s33, h, kt), true);
                }
            }
        } else {
            // drain: the last chunk's tile (MODE 0: nothing is pending, the sdf row was stored above)
            epilogue(std::integral_constant<int, CP>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
        }
        __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (MODE >= 1) {
        if (ok && h == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) a.normals[pd * 3 + d] = n[d] * a.scale;
            if constexpr (MODE == 3) {
#pragma unroll
                for (int d = 0; d < 3; ++d) ex.col_out[pd * 3 + d] = col[d];
            }
        }
    }
    if constexpr (MODE == 2) {
        // this workgroup's 128 points are the 128 samples of ray blockIdx.x: hand them to wave 0 through LDS (the weight ring is
        // free behind the barrier) and composite there; rows = [7][128] floats: sdf | normal x, y, z | colour r, g, b
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* rows = reinterpret_cast<float*>(smem);
        if (h == 0) {
            const int i = pp.wave * 32 + c;
            rows[i] = sdf_keep;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                rows[(1 + d) * 128 + i] = n[d] * a.scale;
                rows[(4 + d) * 128 + i] = col[d];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (pp.wave != 0) return;
        const int ray = blockIdx.x;
        CompositeArgs cm = ex.cm;
        float* eik_partial = cm.eik_partial;
        cm.eik_partial = nullptr;                   // (stored below, write-through, ahead of the arrival count)
        const EikPair ep = composite_row(cm, ray, lane, CompositeLdsSrc{rows, 128}, rows + 7 * 128, rows + 7 * 128 + kMaxT);
        // gradient_error = sum(num) / (sum(den) + 1e-5) over ALL rays (renderer.py:313-315) without a launch of its own: every ray
        // publishes its partial sums (one 8-byte write-through store, drained) and then counts itself in; the ray whose count comes
        // back last reads all partials (sc1 loads: served by L2 / memory, never by this CU's L1) and reduces them exactly as
        // eikonal_reduce_kernel does (lane-strided double sums, then the wave sum): the same bits as the separate launch.
        int old = 0;
        if (lane == 0) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 pv = {ep.num, ep.den};
            asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(eik_partial + 2 * ray), "v"(pv) : "memory");
            old = __hip_atomic_fetch_add(ex.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == cm.B - 1) {
            double sn = 0.0, sd = 0.0;
            for (int i = lane; i < cm.B; i += 64) {
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 v;
                asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(eik_partial + 2 * i) : "memory");
                sn += (double)v[0];
                sd += (double)v[1];
            }
            sn = wave_sum(sn);
            sd = wave_sum(sd);
            if (lane == 0) {
                cm.eik_out[0] = (float)sn / ((float)sd + 1e-5f);
                cm.eik_out[1] = (float)sn;
                cm.eik_out[2] = (float)sd;
                __hip_atomic_store(ex.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if constexpr (UPS) {
        // wave w holds samples 32 (w & 1) .. +31 of ray w >> 1 of this workgroup; the weight ring is free behind the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float* rows = reinterpret_cast<float*>(smem) + (pp.wave >> 1) * 3 * kMaxT;
        if (h == 0) {
            rows[(pp.wave & 1) * 32 + c] = z_keep;
            rows[kMaxT + (pp.wave & 1) * 32 + c] = sdf_keep;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int r = blockIdx.x * 2 + (pp.wave >> 1);
        if ((pp.wave & 1) == 0 && r < up.B) upsample_row(up, r, lane, 64, rows, rows + kMaxT, rows + 2 * kMaxT);
    }
#ifdef VDN_SDF2_STAMP
    if (threadIdx.x == 0 && a.PE != nullptr) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(a.PE) + 8 * blockIdx.x;
        o[0] = stamp_c0; o[1] = __builtin_amdgcn_s_memtime(); o[2] = stamp_r0; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = stamp_p1; o[5] = stamp_p2;
    }
#endif
}

template <int MODE, bool SAVE, int NSLOT, int DEPTH, int VID = 0, bool UPS = false>
int launch(const VdnSdfArgs* args, hipStream_t stream, const VdnUpsampleArgs* up = nullptr, const ShadeExtra* ex = nullptr) {
    constexpr size_t lds_min = MODE >= 1 ? 160 * 1024 : NSLOT * kStride;
    static bool once = (allow_big_lds(sdf_fwd2_kernel<MODE, SAVE, NSLOT, DEPTH, VID, UPS>, lds_min), true);
    (void)once;
    const int grid = (args->P + kWaves * 32 - 1) / (kWaves * 32);
    const size_t lds = lds_min;
    // VDN_SDF2_WARM=0: no L2 warm-up (A/B); by default every launch that fills the chip at least once warms the stream it walks
    static const bool warm = [] { const char* e = getenv("VDN_SDF2_WARM"); return e == nullptr || e[0] != '0'; }();
    ShadeExtra exv = ex != nullptr ? *ex : ShadeExtra{};
    // (a training step's launches find the stream cold: the saving launch always, the others when the caller says so; a render()
    // loop keeps its few MB of weights in L2 / MALL, where the warm-up would only cost its 3-5 us)
    const bool cold = (SAVE && (MODE == 1 || MODE == 3)) || args->cold_start != 0;
    exv.warm_bytes = (warm && cold) ? Prog<MODE>::sdf_total * kStride : 0;
    exv.warm_bytes2 = (warm && cold && MODE >= 2) ? (Prog<MODE>::total - Prog<MODE>::sdf_total) * kStride : 0;
    hipLaunchKernelGGL((sdf_fwd2_kernel<MODE, SAVE, NSLOT, DEPTH, VID, UPS>), dim3(grid), dim3(kWaves * 64), lds, stream, *args,
                       up != nullptr ? *up : VdnUpsampleArgs{}, exv);
    return (int)hipGetLastError();
}

}  // namespace sdf2
}  // namespace vdn
