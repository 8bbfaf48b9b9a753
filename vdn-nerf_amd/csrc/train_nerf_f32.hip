// Backward of the background NeRF MLP on gfx950, fp32: delta chain through the transposed
// layers with ReLU masks from the saved activations. Adjoint of fields.py:324-353.
#include "mlp_engine_f32.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kNbWaves = 4;
constexpr int kNbSlot = chunk_bytes_f32(9);
using NbStream = WStream<kNbWaves, kNbSlot>;

struct MaskStoreN {
    float* Y;
    float* dst;
    int ld;
    long row;
    bool ok;
    int h;
    VDN_DEV void operator()(int nt, const f32x16& acc, const f32x16& hv) const {
        f32x16 o;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            o[t] = hv[t] > 0.0f ? acc[t] : 0.0f;
            Y[nt * 16 + t] = o[t];
        }
        store_tile_rowmajor(dst, row, ld, nt, h, o, ok);
    }
};

template <bool DPT>
__global__ __launch_bounds__(kNbWaves * 64, 1) void nerf_bwd_f32_kernel(NerfBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NbStream ws;
    ws.init(a.blob, smem);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const long p_raw = ((long)blockIdx.x * kNbWaves + ws.wave) * 32 + c;
    const bool ok = p_raw < a.P;
    const long p = ok ? p_raw : (long)a.P - 1;
    const long PS = (long)a.P * 256;
    constexpr int KO = DPT ? 4 : 1;
    constexpr int LDO = DPT ? 128 : 32;

    float X[144], Y[128];
    {   // delta of [rgb (tile 0, rows 0..2) | dpt (tiles 1..3)]: no activation on these heads
        float g3[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) g3[d] = a.g_rgb[p * 3 + d];
        vals_to_tiles<3, 1>(g3, h, X);
        f32x16 t16;
#pragma unroll
        for (int t = 0; t < 16; ++t) t16[t] = X[t];
        store_tile_rowmajor(a.delta_o, p, LDO, 0, h, t16, ok);
        if constexpr (DPT) {
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const f32x16 g = load_tile_rowmajor_v(a.g_feat, p, 96, kt, h);
#pragma unroll
                for (int t = 0; t < 16; ++t) X[(kt + 1) * 16 + t] = g[t];
                store_tile_rowmajor(a.delta_o, p, LDO, kt + 1, h, g, ok);
            }
        }
    }
    constexpr int C4 = chunk_bytes_f32(4), C8 = chunk_bytes_f32(8), C9 = chunk_bytes_f32(9), CO = chunk_bytes_f32(KO);
    auto ldH = [&](int l) { return [=](int nt) { return load_tile_rowmajor_v(a.save_h + l * PS, p, 256, nt, h); }; };
    ws.start<CO>();
    // Wout^T: -> d hv (128), masked by the views layer's ReLU
    dense_f32<KO, 4, C4, false>(ws, X, [&](int nt) { return load_tile_rowmajor_v(a.save_hv, p, 128, nt, h); },
                                MaskStoreN{Y, a.delta_v, 128, p, ok, h});
    // Wviews^T: -> d [feature (8 tiles) | PE(view) (dropped)]; feature_linear has no activation
    dense_f32<4, 9, C9, false>(ws, Y, NoPre{}, [&](int nt, const f32x16& acc, int) {
        if (nt < 8) {
#pragma unroll
            for (int t = 0; t < 16; ++t) X[nt * 16 + t] = acc[t];
            store_tile_rowmajor(a.delta_head, p, 288, nt, h, acc, ok);
        }
    });
    {   // head delta = [d feature (256) | d density at row 256]
        float g1[1] = {a.g_density[p]};
        vals_to_tiles<1, 1>(g1, h, X + 128);
        f32x16 t16;
#pragma unroll
        for (int t = 0; t < 16; ++t) t16[t] = X[128 + t];
        store_tile_rowmajor(a.delta_head, p, 288, 8, h, t16, ok);
    }
    dense_f32<9, 8, C8, false>(ws, X, ldH(7), MaskStoreN{Y, a.delta_h + 7 * PS, 256, p, ok, h});     // Whead^T
    dense_f32<8, 8, C8, false>(ws, Y, ldH(6), MaskStoreN{X, a.delta_h + 6 * PS, 256, p, ok, h});     // W7^T
    dense_f32<8, 8, C8, false>(ws, X, ldH(5), MaskStoreN{Y, a.delta_h + 5 * PS, 256, p, ok, h});     // W6^T
    // W5^T: 11 output tiles = [PE (3, dropped) | h4 (8)]
    dense_f32<8, 11, C8, false>(ws, Y,
        [&](int nt) { return nt >= 3 ? load_tile_rowmajor_v(a.save_h + 4 * PS, p, 256, nt - 3, h) : f32x16{}; },
        [&](int nt, const f32x16& acc, const f32x16& hv) {
            if (nt >= 3) {
                f32x16 o;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    o[t] = hv[t] > 0.0f ? acc[t] : 0.0f;
                    X[(nt - 3) * 16 + t] = o[t];
                }
                store_tile_rowmajor(a.delta_h + 4 * PS, p, 256, nt - 3, h, o, ok);
            }
        });
    dense_f32<8, 8, C8, false>(ws, X, ldH(3), MaskStoreN{Y, a.delta_h + 3 * PS, 256, p, ok, h});     // W4^T
    dense_f32<8, 8, C8, false>(ws, Y, ldH(2), MaskStoreN{X, a.delta_h + 2 * PS, 256, p, ok, h});     // W3^T
    dense_f32<8, 8, C8, false>(ws, X, ldH(1), MaskStoreN{Y, a.delta_h + 1 * PS, 256, p, ok, h});     // W2^T
    dense_f32<8, 8, 0, false>(ws, Y, ldH(0), MaskStoreN{X, a.delta_h + 0 * PS, 256, p, ok, h});      // W1^T
}

}  // namespace vdn

extern "C" int vdn_nerf_mlp_bwd_f32(const VdnNerfBwdArgs* args, void* stream_) {
    using namespace vdn;
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->P <= 0 || !args->blob || !args->g_density || !args->g_rgb || !args->save_h || !args->save_hv ||
        !args->delta_o || !args->delta_v || !args->delta_head || !args->delta_h) return -1;
    const int grid = (args->P + kNbWaves * 32 - 1) / (kNbWaves * 32);
    const size_t lds = 2 * kNbSlot;
    static bool once = (allow_big_lds(nerf_bwd_f32_kernel<false>, 2 * kNbSlot), allow_big_lds(nerf_bwd_f32_kernel<true>, 2 * kNbSlot), true);
    (void)once;
    if (args->g_feat != nullptr)
        hipLaunchKernelGGL(nerf_bwd_f32_kernel<true>, dim3(grid), dim3(kNbWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL(nerf_bwd_f32_kernel<false>, dim3(grid), dim3(kNbWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}
