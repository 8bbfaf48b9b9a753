// fp32 MLP engine: exact-f32 MFMA (v_mfma_f32_32x32x2_f32) with weights streamed through LDS.
//
// Weight chunk (one 32-row output tile `nt` of one layer), built by vdn_build_images():
//   [KT*4 groups][64 lanes][4 floats]   lane (i,h) of group g holds W[nt*32+i][8g+4h .. 8g+4h+3]
//   [32 floats bias][pad to 1 KiB]
// i.e. chunk bytes = KT*4096 + 1024, KT = padded input width / 32. Chunks of a kernel are laid out
// in global memory in the exact order the kernel consumes them, so streaming is one linear walk:
// every chunk is fetched with global_load_lds (1 KiB per wave-instruction) into one of two LDS
// slots while the previous chunk is being multiplied.
#pragma once
#include "vdn_common.h"

namespace vdn {

constexpr int chunk_bytes_f32(int KT) { return KT * 4096 + 1024; }

template <int NWAVES, int SLOT_BYTES>
struct WStream {
    const char* g;   // global cursor: first byte of the next chunk to fetch
    char* lds;       // base of the two slots
    int slot;        // slot holding the current chunk
    int wave, lane;

    VDN_DEV void init(const char* blob, char* smem) {
        g = blob;
        lds = smem;
        slot = 1;
        wave = threadIdx.x >> 6;
        lane = threadIdx.x & 63;
    }
    template <int BYTES>
    VDN_DEV void issue(int s) {
        static_assert(BYTES % 1024 == 0 && BYTES <= SLOT_BYTES, "chunk size");
        constexpr int pieces = BYTES / 1024;
#pragma unroll
        for (int i = 0; i < (pieces + NWAVES - 1) / NWAVES; ++i) {
            const int piece = wave + i * NWAVES;
            if (piece < pieces) glds16(g + piece * 1024 + lane * 16, lds + s * SLOT_BYTES + piece * 1024);
        }
        g += BYTES;
    }
    template <int FIRST_BYTES>
    VDN_DEV void start() { issue<FIRST_BYTES>(0); }
    // Make the chunk issued last current (all waves), then start fetching the next one.
    template <int NEXT_BYTES>
    VDN_DEV const char* acquire() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        slot ^= 1;
        if constexpr (NEXT_BYTES > 0) issue<NEXT_BYTES>(slot ^ 1);
        return lds + slot * SLOT_BYTES;
    }
};

// One dense layer on the wave's 32 points: for every output tile nt, acc = bias + W[nt] . X, then
// epi(nt, acc). X holds KT input tiles (KT*16 registers). NEXT_BYTES = size of the chunk that
// follows this layer's last chunk in the stream (0 at the end of the stream).
struct NoPre {
    VDN_DEV int operator()(int) const { return 0; }
};

// `pre(nt)` runs right after the chunk is acquired (its loads overlap the MFMA loop) and its
// result is handed to `epi(nt, acc, aux)`.
template <int KT, int NT, int NEXT_BYTES, bool BIAS, class WS, class Pre, class Epi>
VDN_DEV void dense_f32(WS& ws, const float* X, Pre&& pre, Epi&& epi) {
    const int lane = ws.lane;
    const int h = lane >> 5;
    static_for<NT>([&](auto nt_c) {
        constexpr int nt = decltype(nt_c)::value;
        const char* w = (nt + 1 < NT) ? ws.template acquire<chunk_bytes_f32(KT)>()
                                      : ws.template acquire<NEXT_BYTES>();
        const f32x4* wa = reinterpret_cast<const f32x4*>(w) + lane;
        auto aux = pre(nt);
        f32x16 acc;
        if constexpr (BIAS) {
            const f32x4* bias = reinterpret_cast<const f32x4*>(w + KT * 4096);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = bias[2 * q + h];   // features 8q+4h .. +3
                acc[4 * q + 0] = b[0];
                acc[4 * q + 1] = b[1];
                acc[4 * q + 2] = b[2];
                acc[4 * q + 3] = b[3];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
        }
        static_for<KT * 4>([&](auto g_c) {
            constexpr int g = decltype(g_c)::value;
            const f32x4 a = wa[g * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], X[4 * g + 0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], X[4 * g + 1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], X[4 * g + 2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], X[4 * g + 3], acc, 0, 0, 0);
        });
        epi(nt, acc, aux);
    });
}

// Row-major [P, ld] <-> activation tile helpers (lane (c,h): 16-byte pieces at col 32*tile+8q+4h).
VDN_DEV void store_tile_rowmajor(float* base, long row, int ld, int tile, int h, const f32x16& v, bool ok) {
    if (!ok) return;
    float* p = base + row * ld + tile * 32 + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        *reinterpret_cast<f32x4*>(p + 8 * q) = o;
    }
}
VDN_DEV f32x16 load_tile_rowmajor_v(const float* base, long row, int ld, int tile, int h) {
    const float* p = base + row * ld + tile * 32 + 4 * h;
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(p + 8 * q);
        r[4 * q] = o[0];
        r[4 * q + 1] = o[1];
        r[4 * q + 2] = o[2];
        r[4 * q + 3] = o[3];
    }
    return r;
}
VDN_DEV void load_tile_rowmajor(const float* base, long row, int ld, int tile, int h, float* X16) {
    const float* p = base + row * ld + tile * 32 + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(p + 8 * q);
        X16[4 * q] = o[0];
        X16[4 * q + 1] = o[1];
        X16[4 * q + 2] = o[2];
        X16[4 * q + 3] = o[3];
    }
}

}  // namespace vdn
