// MLP engine for gfx950: weights streamed through LDS as the MFMA A operand, activations chained
// in registers as the B operand (see vdn_common.h for the activation-tile layout). Two precision
// policies share every kernel body:
//
//   F32  : v_mfma_f32_32x32x2_f32, exact fp32 (parity path).       4 waves / workgroup, 1 wave / SIMD.
//          chunk = [KT*4 groups][64 lanes][4 f32] + 32 f32 bias (+pad) = KT*4096 + 1024 bytes
//          lane (i,h) of group g holds W[nt*32+i][8g+4h .. +3]; HBM activations stored as f32.
//   BF16 : v_mfma_f32_32x32x16_bf16, bf16 operands / fp32 accumulate (throughput path).
//          4 waves / workgroup, 2 workgroups / CU = 2 waves / SIMD.
//          chunk = [KT*2 k-steps][64 lanes][8 bf16] + 32 f32 bias (+pad) = KT*2048 + 1024 bytes
//          lane (i,h) element j of k-step s holds W[nt*32+i][16s + 8(j>>2) + 4h + (j&3)], which is the
//          k order in which a 32x32 accumulator tile converts to B fragments with no lane movement;
//          HBM activations stored as bf16.
//
// Chunks of one kernel lie in global memory in the exact order the kernel consumes them, at a uniform stride, so
// weight streaming is one linear walk: each chunk is fetched with global_load_lds (1 KiB per wave-instruction)
// into a ring of LDS slots, two chunk steps ahead of its use (WStream).
#pragma once
#include <type_traits>
#include "vdn_common.h"

namespace vdn {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

VDN_DEV unsigned pack_bf16x2(float a, float b) {
    bf16x2_t v = {(__bf16)a, (__bf16)b};      // v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}
VDN_DEV float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
VDN_DEV float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }


// Weight stream: a ring of NSLOT LDS slots, each chunk fetched DEPTH = NSLOT-1 chunk steps before it is used
// (measured: with a depth of 1 every step waited ~0.9 us for its chunk - the L2 -> LDS latency - which put a
// 58 us floor under a 63-step forward chain whose MFMAs take 31 us). All chunks of a stream sit at one uniform
// stride STRIDE (a multiple of NWAVES KiB, >= the largest chunk), so every wave issues exactly G = STRIDE/1024/NWAVES
// global_load_lds instructions per chunk and the number of younger in-flight loads is known without tables.
template <int NWAVES, int STRIDE, int NSLOT = 3>
struct WStream {
    static_assert(STRIDE % (1024 * NWAVES) == 0, "chunk stride must be a multiple of NWAVES KiB");
    static constexpr int DEPTH = NSLOT - 1;
    static constexpr int G = STRIDE / 1024 / NWAVES;
    const char* g;   // global cursor: first byte of the next chunk to fetch
    char* lds;       // base of the ring
    int cur;         // index of the chunk being consumed
    int issued;      // chunks issued so far
    int total;       // chunks in the stream
    int wave, lane;
    bool all_issue;  // set by the kernel: this wave has at least one in-range lane (so it issues every store)

    unsigned voff0, m0_wave;    // this wave's G KiB of a chunk (+ 4096: the centre of the pieces' immediate offsets, vdn_common.h): lane offset / LDS address in slot 0
    VDN_DEV void init(const char* blob, char* smem, int total_chunks) {
        g = blob;
        lds = smem;
        cur = -1;
        issued = 0;
        total = total_chunks;
        wave = threadIdx.x >> 6;
        lane = threadIdx.x & 63;
        all_issue = false;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        voff0 = lane * 16 + wv * (G * 1024) + 4096;
        m0_wave = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem + wv * (G * 1024) + 4096);
    }
    // L2 warm-up of the whole stream by the launch's first round of workgroups (warm_l2 above); before start()
    // (the dump area of the warm-up's LDS-DMA: this wave's own first piece of ring slot 0 - vdn_common.h)
    VDN_DEV char* warm_dump() const { return lds + (threadIdx.x >> 6) * (G * 1024); }
    VDN_DEV void warm(long n_wg, int resident) const { warm_l2(g, total * STRIDE, n_wg, resident, warm_dump()); }
    VDN_DEV void warm_issue(long n_wg, int resident) const { warm_l2_issue(g, total * STRIDE, n_wg, resident, warm_dump()); }
    VDN_DEV void issue_next() {
        // (vdn_common.h: glds16_imm*) one M0 write per group of 8 pieces, the pieces by immediate offset from the chunk's scalar base
        const unsigned m0v = m0_wave + (unsigned)((issued % NSLOT) * STRIDE);
        static_for<G>([&](auto i_c) VDN_INL {
            constexpr int I = decltype(i_c)::value;
            if constexpr (I % 8 == 0) glds16_imm_m0<glds_imm(I)>(g, voff0 + glds_group_off(I), m0v + glds_group_off(I));
            else glds16_imm<glds_imm(I)>(g, voff0 + glds_group_off(I));
        });
        g += STRIDE;
        ++issued;
    }
    VDN_DEV void start() {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i)
            if (issued < total) issue_next();
    }
    // Make the next chunk current (all waves), then start fetching the chunk DEPTH steps ahead.
    // `younger` = vector-memory instructions this wave issued during the previous step that may stay in flight
    // (epilogue stores + prefetched epilogue loads); the glds of the chunks still ahead are added here. vmcnt
    // counts loads, stores and LDS-DMA together in issue order, so vmcnt(N) retires this chunk without draining
    // them. `all_issue` must be false for a wave whose lanes are all out of range (it skips its stores).
    VDN_DEV const char* acquire(int younger = 0) {
        ++cur;
        if (!all_issue) younger = 0;
        const int ahead = min(issued - cur - 1, DEPTH - 1);      // chunks after `cur` already in flight
        wait_vm(younger + ahead * G);
#if !defined(VDN_ABLATE) || VDN_ABLATE != 3
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
        if (issued < total) issue_next();
        return lds + (cur % NSLOT) * STRIDE;
    }
    static VDN_DEV void wait_vm(int n) {   // wave-uniform n; s_waitcnt takes an immediate
        switch (n) {
#define VDN_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
            VDN_W(1) VDN_W(2) VDN_W(3) VDN_W(4) VDN_W(5) VDN_W(6) VDN_W(7) VDN_W(8) VDN_W(9) VDN_W(10) VDN_W(11) VDN_W(12)
            VDN_W(13) VDN_W(14) VDN_W(15) VDN_W(16) VDN_W(17) VDN_W(18) VDN_W(19) VDN_W(20) VDN_W(21) VDN_W(22) VDN_W(23) VDN_W(24)
            VDN_W(25) VDN_W(26) VDN_W(27) VDN_W(28) VDN_W(29) VDN_W(30) VDN_W(31) VDN_W(32)
#undef VDN_W
            default: if (n > 32) { asm volatile("s_waitcnt vmcnt(32)" ::: "memory"); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } break;
        }
    }
};

// softplus' from a loaded plane value. from_h 0: the plane holds softplus' itself; 1: the plane holds h = softplus(a):
// sigma(100 a) = 1 - exp(-100 h); 2: the plane holds g = 100 log2(e) h (the units the second-generation SDF kernel
// saves in, k_sdf_fwd2.h): sigma = 1 - 2^-g.
VDN_DEV float sprime(float raw, int from_h) {
    if (from_h == 0) return raw;
    return 1.0f - __builtin_amdgcn_exp2f(from_h == 2 ? -raw : -144.26950408889634f * raw);
}

// Row bookkeeping of the MLP kernels: lane c of wave w of workgroup b works on row b*waves*32 + w*32 + c of the work
// list. Without a list the row is the point; with one (vdn_background_active / vdn_foreground_active) `row` addresses
// the training saves and `point` = active_idx[row] the per-point inputs and outputs. `none` = the whole workgroup lies
// beyond the list (it must return before its first barrier).
struct WorkRow {
    long row, point;
    bool ok, none;
    long n_wg;          // workgroups of the launch that have rows
};
VDN_DEV WorkRow work_row(const int32_t* active_idx, const int32_t* n_active, long P, int waves, int wave, int c) {
    WorkRow w;
    const long n_rows = active_idx != nullptr ? (long)*n_active : P;
    w.none = (long)blockIdx.x * waves * 32 >= n_rows;
    const long raw = ((long)blockIdx.x * waves + wave) * 32 + c;
    w.ok = raw < n_rows;
    w.row = w.ok ? raw : (n_rows > 0 ? n_rows - 1 : 0);
    w.point = active_idx != nullptr ? (long)active_idx[w.row] : w.row;
    w.n_wg = (n_rows + waves * 32 - 1) / (waves * 32);
    return w;
}

struct NoPre {
    VDN_DEV int operator()(int) const { return 0; }
};

// ---------------------------------------------------------------------------------------------
struct F32 {
    static constexpr int kWaves = 4;
    static constexpr int kMinWavesPerEU = 1;
    static constexpr bool kAccurateTrig = true;     // ocml sincosf in the positional encoding
    static constexpr bool kOverlapEpilogue = false; // an f32 tile's MFMAs take 4x longer than its epilogue: nothing to hide
    static constexpr bool kDeriveS = false;         // the parity path keeps softplus' exactly as the forward computed it
    static constexpr int chunk_bytes(int KT) { return KT * 4096 + 1024; }
    static constexpr int stride(int KTMAX) { return (chunk_bytes(KTMAX) + 4095) / 4096 * 4096; }   // uniform chunk stride
    static constexpr int kTileOps = 4;      // vector-memory instructions per store_tile / load_tile (the kernels' vmcnt accounting)
    using store_t = float;

    template <int NT>
    struct Act {
        float r[NT * 16];
        VDN_DEV void set(int tile, const f32x16& v) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                float x = v[t];
                // materialise here: otherwise LLVM sinks a register-only epilogue down to its use in the
                // next layer and keeps every tile's accumulator alive (volatile asm is not reordered
                // across the volatile wait in WStream::acquire)
                asm volatile("" : "+v"(x));
                r[tile * 16 + t] = x;
            }
        }
        template <class A2>
        VDN_DEV void copy_tile(int dt, const A2& src, int st) {
#pragma unroll
            for (int t = 0; t < 16; ++t) r[dt * 16 + t] = src.r[st * 16 + t];
        }
    };

    // acc = (bias) + W[chunk] . X over KT input tiles starting at X tile `x0`
    template <int KT, bool BIAS, class ActT>
    static VDN_DEV f32x16 mma(const char* w, const ActT& X, int x0, int lane) {
        const int h = lane >> 5;
        const f32x4* wa = reinterpret_cast<const f32x4*>(w) + lane;
        f32x16 acc;
        if constexpr (BIAS) {
            const f32x4* bias = reinterpret_cast<const f32x4*>(w + KT * 4096);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = bias[2 * q + h];   // features 8q+4h .. +3
                acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
        }
        static_for<KT * 4>([&](auto g_c) VDN_INL {
            constexpr int g = decltype(g_c)::value;
            const f32x4 a = wa[g * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], X.r[x0 * 16 + 4 * g + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], X.r[x0 * 16 + 4 * g + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], X.r[x0 * 16 + 4 * g + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], X.r[x0 * 16 + 4 * g + 3], acc, 0, 0, 0);
        });
        return acc;
    }

    // elements of one [P, ld] activation plane / padded point count (row-major: no padding)
    static VDN_DEV long plane(long P, int ld) { return P * ld; }
    static VDN_DEV long rows(long P) { return P; }
    // row-major [P, ld] <-> activation tile: lane (c,h) owns 16-byte pieces at column 32*tile + 8q + 4h
    static VDN_DEV void store_tile(float* base, long row, int ld, int tile, int h, const f32x16& v, bool ok) {
        if (!ok) return;
        float* p = base + row * ld + tile * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            *reinterpret_cast<f32x4*>(p + 8 * q) = o;
        }
    }
    // a tile as loaded, before conversion (what the chains keep in flight across chunk steps): f32 planes need none
    using raw_tile = f32x16;
    static VDN_DEV raw_tile load_raw(const float* base, long row, int ld, int tile, int h) { return load_tile(base, row, ld, tile, h); }
    static VDN_DEV f32x16 unpack(const raw_tile& r) { return r; }
    static VDN_DEV f32x16 load_tile(const float* base, long row, int ld, int tile, int h) {
        const float* p = base + row * ld + tile * 32 + 4 * h;
        f32x16 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(p + 8 * q);
            r[4 * q] = o[0]; r[4 * q + 1] = o[1]; r[4 * q + 2] = o[2]; r[4 * q + 3] = o[3];
        }
        return r;
    }
};

// (VDN_PLANE_ST_MODE, vdn_common.h: cache policy of the chain kernels' plane stores - store_tile below. 0 plain; 1 write-through:
// measured neutral on the step in round 3; 2 non-temporal: neutral; 3 both: the default since round 4)
struct BF16 {
    // 4 waves per workgroup and two workgroups resident per CU (2 waves / SIMD): the workgroups run their
    // chunk barriers independently, so one computes while the other waits on memory
    static constexpr int kWaves = 4;
    static constexpr int kMinWavesPerEU = 2;
    static constexpr bool kAccurateTrig = false;    // hardware v_sin / v_cos (see vdn_common.h: sincos_pe)
    // A softplus tile epilogue is ~110 VALU instructions against 16 MFMAs: elementwise epilogues (ElemEpi) are issued in
    // the shadow of the NEXT tile's MFMAs (mma_slots), one accumulator register per MFMA.
    static constexpr bool kOverlapEpilogue = true;
    // training: softplus' is not stored; consumers re-derive it from the saved activation h = softplus(a):
    // sigma(100 a) = 1 - exp(-100 h). From a bf16 h this is at least as accurate as a bf16-rounded sigma (the relative
    // error of h carries over for small h and is damped by exp(-100 h) for large h) and saves writing 8 planes per step.
    static constexpr bool kDeriveS = true;
#ifndef VDN_SLOT_GROUP
#define VDN_SLOT_GROUP 2
#endif
    static constexpr int kSlotGroup = VDN_SLOT_GROUP;
    static constexpr int chunk_bytes(int KT) { return KT * 2048 + 1024; }
    static constexpr int stride(int KTMAX) { return (chunk_bytes(KTMAX) + 4095) / 4096 * 4096; }   // uniform chunk stride
    using store_t = unsigned short;   // bf16 bits

    template <int NT>
    struct Act {
        bf16x8 r[NT * 2];
        VDN_DEV void set(int tile, const f32x16& v) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                u32x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned x = pack_bf16x2(v[8 * s + 2 * j], v[8 * s + 2 * j + 1]);
                    asm volatile("" : "+v"(x));      // materialise here (see F32::Act::set)
                    pk[j] = x;
                }
                r[tile * 2 + s] = __builtin_bit_cast(bf16x8, pk);
            }
        }
        template <class A2>
        VDN_DEV void copy_tile(int dt, const A2& src, int st) {
            r[dt * 2] = src.r[st * 2];
            r[dt * 2 + 1] = src.r[st * 2 + 1];
        }
    };

    template <int KT, bool BIAS, class ActT>
    static VDN_DEV f32x16 mma(const char* w, const ActT& X, int x0, int lane) {
        const int h = lane >> 5;
        const bf16x8* wa = reinterpret_cast<const bf16x8*>(w) + lane;
        f32x16 acc;
        if constexpr (BIAS) {
            const f32x4* bias = reinterpret_cast<const f32x4*>(w + KT * 2048);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = bias[2 * q + h];
                acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
        }
#if defined(VDN_ABLATE) && VDN_ABLATE == 2
        (void)wa;
        return acc;
#endif
        // Fragment reads are software-pipelined PRE deep under the MFMAs. Left to itself hipcc emits
        // {2 x ds_read_b128, s_waitcnt lgkmcnt(0), 2 x MFMA} x 8 per tile - a full LDS round trip exposed eight
        // times (measured ~1700 cycles per tile for 512 cycles of MFMA) - so the order is pinned with
        // sched_group_barrier: PRE reads up front, then one read per MFMA, then the last PRE MFMAs.
        constexpr int NS = KT * 2;
        constexpr int PRE = NS < 6 ? NS : 6;
        bf16x8 fr[NS];
        static_for<NS>([&](auto s_c) VDN_INL {
            constexpr int s = decltype(s_c)::value;
            fr[s] = wa[s * 64];
        });
        static_for<NS>([&](auto s_c) VDN_INL {
            constexpr int s = decltype(s_c)::value;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s], X.r[x0 * 2 + s], acc, 0, 0, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x100, PRE + (BIAS ? 2 : 0), 0);      // DS reads (bias rows + first fragments)
        static_for<NS - PRE>([&](auto) VDN_INL {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                      // 1 DS read
        });
        __builtin_amdgcn_sched_group_barrier(0x008, PRE, 0);
        return acc;
    }

    // mma() with the issue order written out by hand and fenced: PRE fragment reads ahead, then per MFMA one more read and
    // slot(k) - a slice of independent VALU work that runs while the (dependent) MFMA chain occupies the matrix pipe.
    // hipcc's scheduler does not produce this interleaving from sched_group_barrier hints (it keeps MFMAs and the epilogue
    // in two blocks in two steps out of three), hence the sched_barrier(0) fences.
    template <int KT, bool BIAS, class ActT, class Slot>
    static VDN_DEV f32x16 mma_slots(const char* w, const ActT& X, int x0, int lane, Slot&& slot) {
        const int h = lane >> 5;
        const bf16x8* wa = reinterpret_cast<const bf16x8*>(w) + lane;
        constexpr int NS = KT * 2;
        constexpr int PRE = NS < 6 ? NS : 6;
        f32x16 acc;
        bf16x8 fr[NS];
        static_for<PRE>([&](auto s_c) VDN_INL {
            constexpr int s = decltype(s_c)::value;
            fr[s] = wa[s * 64];
        });
        if constexpr (BIAS) {
            const f32x4* bias = reinterpret_cast<const f32x4*>(w + KT * 2048);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = bias[2 * q + h];
                acc[4 * q + 0] = b[0]; acc[4 * q + 1] = b[1]; acc[4 * q + 2] = b[2]; acc[4 * q + 3] = b[3];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<NS>([&](auto s_c) VDN_INL {
            constexpr int s = decltype(s_c)::value;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[s], X.r[x0 * 2 + s], acc, 0, 0, 0);
            if constexpr (s + PRE < NS) fr[s + PRE] = wa[(s + PRE) * 64];
            slot(s_c, std::integral_constant<int, NS>{});
            // fence every kSlotGroup MFMAs: inside a group the scheduler interleaves that many independent epilogue
            // chains (a lone softplus chain is 7 dependent instructions and would not fill its MFMA's shadow)
            if constexpr ((s + 1) % kSlotGroup == 0 || s + 1 == NS) __builtin_amdgcn_sched_barrier(0);
        });
        return acc;
    }

    // bf16 activation planes use the tile-blocked "PT32" layout: points in blocks of 32 (= one wave's tile); inside a
    // block, per 32-feature tile (2 KiB):  [k (2)][h (2)][point (32)][8 features = 16 bytes], where the 16-byte unit (k, h, point)
    // holds accumulator registers 8k .. 8k+7 of lane (point, h) - features 16k + 4h + {0..3} and 16k + 8 + 4h + {0..3}, i.e. exactly
    // the MFMA B-operand fragment of k-step k:
    //   element (p, f) at  (p>>5)*(32*ld) + (f>>5)*1024 + ((f>>4)&1)*512 + ((f>>2)&1)*256 + (p&31)*8 + ((f>>3)&1)*4 + (f&3)
    // A wave's store / load of (tile, k) is ONE contiguous 1-KiB run of 16-byte pieces (two vector-memory instructions per tile;
    // the first layout of this path used four 8-byte ones - the training-mode kernels are bound by the ISSUE of those
    // instructions, ~50 cycles each beside the MFMA stream), and the weight-gradient GEMM reads 1-KiB runs. P is padded to 32.
    static constexpr int kTileOps = 2;      // vector-memory instructions per store_tile / load_tile (the kernels' vmcnt accounting)
    static VDN_DEV long rows(long P) { return (P + 31) & ~31L; }
    static VDN_DEV long plane(long P, int ld) { return rows(P) * ld; }
    // MODE: cache policy of the two 16-byte stores. 0 plain; 1 write-through (sc1: the line is not kept in this XCD's L2 - saved
    // planes and deltas are next read by other kernels, while the L2 is what feeds every MLP kernel's weight stream); 2 non-temporal;
    // 3 both (vdn_common.h: VDN_PLANE_ST_MODE, the default)
    template <int MODE = VDN_PLANE_ST_MODE>
    static VDN_DEV void store_tile(unsigned short* base, long row, int ld, int tile, int h, const f32x16& v, bool ok) {
        if (!ok) return;
        unsigned short* p = base + (row >> 5) * (32L * ld) + tile * 1024 + h * 256 + (row & 31) * 8;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            uint4 o;
            o.x = pack_bf16x2(v[8 * k], v[8 * k + 1]);
            o.y = pack_bf16x2(v[8 * k + 2], v[8 * k + 3]);
            o.z = pack_bf16x2(v[8 * k + 4], v[8 * k + 5]);
            o.w = pack_bf16x2(v[8 * k + 6], v[8 * k + 7]);
            typedef unsigned wt_u32x4 __attribute__((ext_vector_type(4)));
            const wt_u32x4 ov = {o.x, o.y, o.z, o.w};
            if constexpr (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p + 512 * k), "v"(ov) : "memory");
            else if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p + 512 * k), "v"(ov) : "memory");
            else if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p + 512 * k), "v"(ov) : "memory");
            else if constexpr (MODE == 4) { asm volatile("" ::"v"(p + 512 * k), "v"(ov)); }      // (timing ablation: no store)
            else *reinterpret_cast<uint4*>(p + 512 * k) = o;
        }
    }
    // a tile as loaded: 8 registers of packed bf16 (the chains keep these, not the 16 converted values, in flight across steps)
    struct raw_tile { uint4 k[2]; };
    static VDN_DEV raw_tile load_raw(const unsigned short* base, long row, int ld, int tile, int h) {
        const unsigned short* p = base + (row >> 5) * (32L * ld) + tile * 1024 + h * 256 + (row & 31) * 8;
        raw_tile r;
#ifdef VDN_PLANE_LD_NT          // (development A/B: the chain kernels read a saved plane once)
        typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
        const nt_u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(p));
        const nt_u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(p + 512));
        r.k[0] = make_uint4(a[0], a[1], a[2], a[3]);
        r.k[1] = make_uint4(b[0], b[1], b[2], b[3]);
#else
        r.k[0] = *reinterpret_cast<const uint4*>(p);
        r.k[1] = *reinterpret_cast<const uint4*>(p + 512);
#endif
        return r;
    }
    static VDN_DEV f32x16 unpack(const raw_tile& t) {
        f32x16 r;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint4 o = t.k[k];
            r[8 * k] = bf16_lo(o.x); r[8 * k + 1] = bf16_hi(o.x); r[8 * k + 2] = bf16_lo(o.y); r[8 * k + 3] = bf16_hi(o.y);
            r[8 * k + 4] = bf16_lo(o.z); r[8 * k + 5] = bf16_hi(o.z); r[8 * k + 6] = bf16_lo(o.w); r[8 * k + 7] = bf16_hi(o.w);
        }
        return r;
    }
    static VDN_DEV f32x16 load_tile(const unsigned short* base, long row, int ld, int tile, int h) {
        return unpack(load_raw(base, row, ld, tile, h));
    }
};

// One dense layer on the wave's 32 points: for every output tile nt, acc = bias + W[nt] . X[x0 ..],
// then epi(nt, acc, aux) with aux = pre(nt) evaluated right after the chunk is acquired (so its
// loads overlap the MFMA loop).
// epi_stores = vector-memory store instructions every in-range wave issues in epi() per tile, pre_loads = load
// instructions in pre() per tile (P::kTileOps per store_tile / load_tile; a lower bound is safe, 0 drains the queue
// at every step). Both are younger than the glds of the chunk being acquired, so they may stay in flight across
// the barrier: the prefetched loads then have two chunk steps to land.
// An epilogue split into a per-register part and a per-tile part:
//   elem(nt, t, acc_t, scratch)  pure VALU work on accumulator register t of tile nt (results into `scratch`)
//   finish(nt, scratch, aux)     the rest (packing into the next layer's operand, stores)
// Policies with kOverlapEpilogue run elem() of tile nt-1 in the MFMA shadow of tile nt; others call both in place.
template <class ScratchT, class Elem, class Finish>
struct ElemEpi {
    using Scratch = ScratchT;
    Elem elem;
    Finish finish;
    template <class Aux>
    VDN_DEV void operator()(int nt, const f32x16& acc, const Aux& aux) const {
        Scratch sc;
#pragma unroll
        for (int t = 0; t < 16; ++t) elem(nt, t, acc[t], sc);
        finish(nt, sc, aux);
    }
};
template <class ScratchT, class Elem, class Finish>
VDN_DEV ElemEpi<ScratchT, Elem, Finish> elem_epi(Elem e, Finish f) { return {e, f}; }
template <class T> struct is_elem_epi : std::false_type {};
template <class S, class E, class F> struct is_elem_epi<ElemEpi<S, E, F>> : std::true_type {};

// PF = how many chunk steps ahead of its use pre(nt) is issued (plain epilogues only): the HBM-bound backward chains run at
// half their wave slots on the training step's work lists, so each wave keeps two steps of plane loads in flight.
template <class P, int KT, int NT, bool BIAS, int PF = 1, class WS, class ActT, class Pre, class Epi>
VDN_DEV void dense(WS& ws, const ActT& X, int x0, Pre&& pre, Epi&& epi, int epi_stores = 0, int pre_loads = 0) {
    const int lane = ws.lane;
    using EpiT = std::remove_cv_t<std::remove_reference_t<Epi>>;
    if constexpr (P::kOverlapEpilogue && is_elem_epi<EpiT>::value) {
        // Software pipeline over the output tiles: step s acquires chunk s and issues tile s's MFMAs with elem() of tile
        // s-1 in their shadow (epi never writes X, so the order is free), then finish(s-1); step NT drains the last tile.
        f32x16 acc_prev;
        decltype(pre(0)) aux_prev{};
        static_for<NT + 1>([&](auto s_c) VDN_INL {
            constexpr int s = decltype(s_c)::value;
            typename EpiT::Scratch sc;
            if constexpr (s < NT) {
                // younger than chunk s's DMA (issued DEPTH steps ago) are the loads of pre(j) and the stores of finish(j-1)
                // of the steps j = s-DEPTH .. s-1 of THIS layer (earlier layers' counts are unknown: not counted)
                constexpr int D = WS::DEPTH;
                constexpr int n_pre = s < D ? s : D;
                constexpr int n_epi = s <= D ? (s > 0 ? s - 1 : 0) : D;
                const int yg = n_pre * pre_loads + n_epi * epi_stores;
                const char* w = ws.acquire(yg);
                auto aux_cur = pre(s);
                const f32x16 acc_cur = P::template mma_slots<KT, BIAS>(w, X, x0, lane, [&](auto k_c, auto ns_c) VDN_INL {
                    if constexpr (s > 0) {
                        constexpr int k = decltype(k_c)::value, NS = decltype(ns_c)::value;
                        static_for<(k + 1) * 16 / NS - k * 16 / NS>([&](auto j_c) VDN_INL {
                            constexpr int t = k * 16 / NS + decltype(j_c)::value;
                            epi.elem(s - 1, t, acc_prev[t], sc);
                        });
                    }
                });
                if constexpr (s > 0) epi.finish(s - 1, sc, aux_prev);
                acc_prev = acc_cur;
                aux_prev = aux_cur;
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) epi.elem(s - 1, t, acc_prev[t], sc);
                epi.finish(s - 1, sc, aux_prev);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        return;
    }
    // pre(nt) (HBM loads feeding tile nt's epilogue) is issued PF chunk steps ahead of its use, so PF full steps of
    // MFMA + epilogue work hide its latency; the first PF tiles' are issued before the layer's first barrier.
    using Aux = decltype(pre(0));
    Aux ring[PF];
    static_for<PF>([&](auto i_c) VDN_INL {
        constexpr int i = decltype(i_c)::value;
        if constexpr (i < NT) ring[i] = pre(i);
    });
    static_for<NT>([&](auto nt_c) VDN_INL {
        constexpr int nt = decltype(nt_c)::value;
        // tile 0 follows another layer's epilogue (unknown store count): only the chunks ahead stay in flight
        constexpr int n_steps = nt < WS::DEPTH ? nt : WS::DEPTH;     // steps of this layer since chunk nt's DMA was issued
        // younger than that DMA: the stores of those steps' epilogues and the loads they issued (step j issues pre(j + PF),
        // if that tile exists) - counted exactly: a count that is too large would release the wait early
        constexpr int n_pre = [] { int n = 0; for (int j = nt - n_steps; j < nt; ++j) n += (j + PF < NT) ? 1 : 0; return n; }();
        const int yg = n_steps * epi_stores + n_pre * pre_loads;
        const char* w = ws.acquire(yg);
        Aux nxt{};
        if constexpr (nt + PF < NT) nxt = pre(nt + PF);
        const f32x16 acc = P::template mma<KT, BIAS>(w, X, x0, lane);
        epi(nt, acc, ring[nt % PF]);
        if constexpr (nt + PF < NT) ring[nt % PF] = nxt;
        // keep each tile's epilogue inside its own chunk step: without this the scheduler sinks the
        // register-only epilogues of several tiles past the barriers and runs out of registers
        __builtin_amdgcn_sched_barrier(0);
    });
}

// plane-load prefetch distance of the colour / background backward chains (the SDF chains choose at launch: k_sdf_bwd.h)
#ifndef VDN_BWD_PF
#define VDN_BWD_PF 2
#endif
constexpr int kBwdPrefetch = VDN_BWD_PF;

// per-point vector -> one activation tile (f32x16), zero beyond NF
template <int NF>
VDN_DEV f32x16 vals_tile(const float (&vals)[NF], int h, int tile) {
    f32x16 o;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int f0 = 32 * tile + (t & 3) + 8 * (t >> 2);
        const int f1 = f0 + 4;
        const float a = f0 < NF ? vals[f0 < NF ? f0 : 0] : 0.0f;
        const float b = f1 < NF ? vals[f1 < NF ? f1 : 0] : 0.0f;
        o[t] = h ? b : a;
    }
    return o;
}
// NT tiles (f32x16 each) -> full per-point vector on every lane
template <int NF, int NT>
VDN_DEV void tiles_vals(const f32x16 (&T)[NT], int h, float (&vals)[NF]) {
#pragma unroll
    for (int tile = 0; tile < NT; ++tile) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int f0 = 32 * tile + (t & 3) + 8 * (t >> 2);
            const int f1 = f0 + 4;
            const float mine = T[tile][t];
            const float other = __shfl_xor(mine, 32);
            if (f0 < NF) vals[f0 < NF ? f0 : 0] = h ? other : mine;
            if (f1 < NF) vals[f1 < NF ? f1 : 0] = h ? mine : other;
        }
    }
}

// Adjoint of the positional encoding, one 32-slot tile at a time: `tile` holds d loss / d(encoded input) for slots
// [32 T0, 32 T0 + 32) of [x (D), sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] (embedder.py:27-36); adds J^T of those slots to dx.
template <int D, int L, int T0, bool ACCURATE>
VDN_DEV void pe_adjoint_tile(const f32x16& tile, int h, const float (&x)[D], float (&dx)[D]) {
    float g[32];
    const f32x16 t1[1] = {tile};
    tiles_vals<32, 1>(t1, h, g);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int slot = 32 * T0 + j;
        if (slot >= D * (1 + 2 * L)) continue;
        if (slot < D) {
            dx[slot] += g[j];
        } else {
            const int k = (slot - D) / (2 * D), w = (slot - D) % (2 * D), d = w % D;
            const float f = (float)(1 << k);
            float sn, co;
            sincos_pe<ACCURATE>(x[d] * f, sn, co);
            dx[d] += w >= D ? -f * sn * g[j] : f * co * g[j];
        }
    }
}

}  // namespace vdn
