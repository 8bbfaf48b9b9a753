// Backward of the background NeRF MLP on gfx950, shared body for both policies: delta chain through
// the transposed layers with ReLU masks from the saved activations. Adjoint of fields.py:324-353.
#pragma once
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

template <class P, bool DPT>
__global__ __launch_bounds__(P::kWaves * 64, P::kMinWavesPerEU) void nerf_bwd_kernel(NerfBwdArgs a) {
    using ST = typename P::store_t;
    constexpr int kSlot = P::stride(9);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WStream<P::kWaves, kSlot> ws;
    ws.init(a.blob, smem, 80);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    // p = row in the (possibly compacted) work list = row of the saves and deltas; pd = dense point id of the upstream grads
    const long n_rows = a.active_idx != nullptr ? (long)*a.n_active : (long)a.P;
    if ((long)blockIdx.x * P::kWaves * 32 >= n_rows) return;
    const long p_raw = ((long)blockIdx.x * P::kWaves + ws.wave) * 32 + c;
    const bool ok = p_raw < n_rows;
    const long p = ok ? p_raw : n_rows - 1;
    const long pd = a.active_idx != nullptr ? (long)a.active_idx[p] : p;
    const long PS = P::plane(a.P, 256);
    constexpr int KO = DPT ? 4 : 1;
    constexpr int LDO = DPT ? 128 : 32;
    const ST* save_h = reinterpret_cast<const ST*>(a.save_h);
    const ST* save_hv = reinterpret_cast<const ST*>(a.save_hv);
    ST* delta_o = reinterpret_cast<ST*>(a.delta_o);
    ST* delta_v = reinterpret_cast<ST*>(a.delta_v);
    ST* delta_head = reinterpret_cast<ST*>(a.delta_head);
    ST* delta_h = reinterpret_cast<ST*>(a.delta_h);

    typename P::template Act<9> X;
    typename P::template Act<8> Y;
    {   // delta of [rgb (tile 0, rows 0..2) | dpt (tiles 1..3)]: no activation on these heads
        float g3[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) g3[d] = a.g_rgb[pd * 3 + d];
        const f32x16 t16 = vals_tile<3>(g3, h, 0);
        X.set(0, t16);
        P::store_tile(delta_o, p, LDO, 0, h, t16, ok);
        if constexpr (DPT) {
#pragma unroll
            for (int kt = 0; kt < 3; ++kt) {
                const f32x16 g = F32::load_tile(a.g_feat, pd, 96, kt, h);
                X.set(kt + 1, g);
                P::store_tile(delta_o, p, LDO, kt + 1, h, g, ok);
            }
        }
    }
    auto ldH = [&](int l) VDN_INL { return [=](int nt) VDN_INL { return P::load_tile(save_h + l * PS, p, 256, nt, h); }; };
    auto mask_store = [&](auto& D, ST* dst, int ld) VDN_INL {
        return [&D, dst, ld, p, ok, h](int nt, const f32x16& acc, const f32x16& hv) VDN_INL {
            f32x16 o;
#pragma unroll
            for (int t = 0; t < 16; ++t) o[t] = hv[t] > 0.0f ? acc[t] : 0.0f;
            D.set(nt, o);
            P::store_tile(dst, p, ld, nt, h, o, ok);
        };
    };
    ws.all_issue = __any(ok);
    ws.start();
    // Wout^T: -> d hv (128), masked by the views layer's ReLU
    dense<P, KO, 4, false>(ws, X, 0, [&](int nt) VDN_INL { return P::load_tile(save_hv, p, 128, nt, h); },
                               mask_store(Y, delta_v, 128), 4);
    // Wviews^T: -> d [feature (8 tiles) | PE(view) (dropped)]; feature_linear has no activation
    dense<P, 4, 9, false>(ws, Y, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL {
        if (nt < 8) {
            X.set(nt, acc);
            P::store_tile(delta_head, p, 288, nt, h, acc, ok);
        }
    });
    {   // head delta = [d feature (256) | d density at row 256]
        float g1[1] = {a.g_density[pd]};
        const f32x16 t16 = vals_tile<1>(g1, h, 0);
        X.set(8, t16);
        P::store_tile(delta_head, p, 288, 8, h, t16, ok);
    }
    dense<P, 9, 8, false>(ws, X, 0, ldH(7), mask_store(Y, delta_h + 7 * PS, 256), 4, 4);     // Whead^T
    dense<P, 8, 8, false>(ws, Y, 0, ldH(6), mask_store(X, delta_h + 6 * PS, 256), 4, 4);     // W7^T
    dense<P, 8, 8, false>(ws, X, 0, ldH(5), mask_store(Y, delta_h + 5 * PS, 256), 4, 4);     // W6^T
    // W5^T: 11 output tiles = [PE (3, dropped) | h4 (8)]
    dense<P, 8, 11, false>(ws, Y, 0,
        [&](int nt) VDN_INL { return nt >= 3 ? P::load_tile(save_h + 4 * PS, p, 256, nt - 3, h) : f32x16{}; },
        [&](int nt, const f32x16& acc, const f32x16& hv) VDN_INL {
            if (nt >= 3) {
                f32x16 o;
#pragma unroll
                for (int t = 0; t < 16; ++t) o[t] = hv[t] > 0.0f ? acc[t] : 0.0f;
                X.set(nt - 3, o);
                P::store_tile(delta_h + 4 * PS, p, 256, nt - 3, h, o, ok);
            }
        });
    dense<P, 8, 8, false>(ws, X, 0, ldH(3), mask_store(Y, delta_h + 3 * PS, 256), 4, 4);     // W4^T
    dense<P, 8, 8, false>(ws, Y, 0, ldH(2), mask_store(X, delta_h + 2 * PS, 256), 4, 4);     // W3^T
    dense<P, 8, 8, false>(ws, X, 0, ldH(1), mask_store(Y, delta_h + 1 * PS, 256), 4, 4);     // W2^T
    dense<P, 8, 8, false>(ws, Y, 0, ldH(0), mask_store(X, delta_h + 0 * PS, 256), 4, 4);      // W1^T
}

template <class P>
int launch_nerf_bwd(const VdnNerfBwdArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->P <= 0 || !args->blob || !args->g_density || !args->g_rgb || !args->save_h || !args->save_hv ||
        !args->delta_o || !args->delta_v || !args->delta_head || !args->delta_h) return -1;
    const int ppw = P::kWaves * 32;
    const int grid = (args->P + ppw - 1) / ppw;
    const size_t lds = 3 * P::stride(9);
    static bool once = (allow_big_lds(nerf_bwd_kernel<P, false>, lds), allow_big_lds(nerf_bwd_kernel<P, true>, lds), true);
    (void)once;
    if (args->g_feat != nullptr)
        hipLaunchKernelGGL((nerf_bwd_kernel<P, true>), dim3(grid), dim3(P::kWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL((nerf_bwd_kernel<P, false>), dim3(grid), dim3(P::kWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}

}  // namespace vdn
