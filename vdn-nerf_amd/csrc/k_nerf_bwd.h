// Backward of the background NeRF MLP on gfx950, shared body for both policies: delta chain through
// the transposed layers with ReLU masks from the saved activations. Adjoint of fields.py:324-353.
#pragma once
#include <type_traits>
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

template <class P, bool DPT>
__global__ __launch_bounds__(P::kWaves * 64, P::kMinWavesPerEU) void nerf_bwd_kernel(NerfBwdArgs a) {
    using ST = typename P::store_t;
    constexpr int kSlot = P::stride(9);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WStream<P::kWaves, kSlot> ws;
    const bool want_pts = a.d_pts != nullptr;      // differentiable rays: three more chunks (W0^T) and the encodings' adjoints
    ws.init(a.blob, smem, want_pts ? 83 : 80);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    // p = row in the (possibly compacted) work list = row of the saves and deltas; pd = dense point id of the upstream grads
    const long n_rows = a.active_idx != nullptr ? (long)*a.n_active : (long)a.P;
    if ((long)blockIdx.x * P::kWaves * 32 >= n_rows) return;
    ws.warm_issue((n_rows + P::kWaves * 32 - 1) / (P::kWaves * 32), 256 * P::kMinWavesPerEU);      // (mlp_engine.h)
    warm_code_issue((std::is_same<P, BF16>::value) ? kWarmCodeNerfBwd : 0, (n_rows + P::kWaves * 32 - 1) / (P::kWaves * 32), 256 * P::kMinWavesPerEU, ws.warm_dump());      // (the kernel's own code: vdn_common.h)
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

    // inverted-sphere point and view direction of this lane's sample, as the forward builds them (renderer.py:112-115)
    float x4[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dir[3] = {0.0f, 0.0f, 0.0f}, pw[3] = {0.0f, 0.0f, 0.0f}, nrm = 1.0f;
    float dx4[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ddir[3] = {0.0f, 0.0f, 0.0f};
    if (want_pts) {
        const long r = pd / a.n_per_ray;
        const float z = a.z[pd];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            dir[d] = a.rays_d[r * 3 + d];
            pw[d] = a.rays_o[r * 3 + d] + dir[d] * z;
        }
        nrm = sqrtf(pw[0] * pw[0] + pw[1] * pw[1] + pw[2] * pw[2]);
        const float rr = fminf(fmaxf(nrm, 1.0f), 1e10f);
#pragma unroll
        for (int d = 0; d < 3; ++d) x4[d] = pw[d] / rr;
        x4[3] = 1.0f / rr;
    }
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
    warm_l2_wait();
    ws.start();
    // Wout^T: -> d hv (128), masked by the views layer's ReLU
    dense<P, KO, 4, false>(ws, X, 0, [&](int nt) VDN_INL { return P::load_tile(save_hv, p, 128, nt, h); },
                               mask_store(Y, delta_v, 128), 4);
    // Wviews^T: -> d [feature (8 tiles) | PE(view) (only wanted for differentiable rays)]; feature_linear has no activation
    dense<P, 4, 9, false>(ws, Y, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL {
        if (nt < 8) {
            X.set(nt, acc);
            P::store_tile(delta_head, p, 288, nt, h, acc, ok);
        } else if (want_pts) {
            pe_adjoint_tile<3, 4, 0, P::kAccurateTrig>(acc, h, dir, ddir);
        }
    });
    {   // head delta = [d feature (256) | d density at row 256]
        float g1[1] = {a.g_density[pd]};
        const f32x16 t16 = vals_tile<1>(g1, h, 0);
        X.set(8, t16);
        P::store_tile(delta_head, p, 288, 8, h, t16, ok);
    }
    dense<P, 9, 8, false, kBwdPrefetch>(ws, X, 0, ldH(7), mask_store(Y, delta_h + 7 * PS, 256), P::kTileOps, P::kTileOps);     // Whead^T
    dense<P, 8, 8, false, kBwdPrefetch>(ws, Y, 0, ldH(6), mask_store(X, delta_h + 6 * PS, 256), P::kTileOps, P::kTileOps);     // W7^T
    dense<P, 8, 8, false, kBwdPrefetch>(ws, X, 0, ldH(5), mask_store(Y, delta_h + 5 * PS, 256), P::kTileOps, P::kTileOps);     // W6^T
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
            } else if (want_pts) {          // the skip input's adjoint: slots 32 nt .. of the 10-octave encoding of pts4
                if (nt == 0) pe_adjoint_tile<4, 10, 0, P::kAccurateTrig>(acc, h, x4, dx4);
                else if (nt == 1) pe_adjoint_tile<4, 10, 1, P::kAccurateTrig>(acc, h, x4, dx4);
                else pe_adjoint_tile<4, 10, 2, P::kAccurateTrig>(acc, h, x4, dx4);
            }
        });
    dense<P, 8, 8, false, kBwdPrefetch>(ws, X, 0, ldH(3), mask_store(Y, delta_h + 3 * PS, 256), P::kTileOps, P::kTileOps);     // W4^T
    dense<P, 8, 8, false, kBwdPrefetch>(ws, Y, 0, ldH(2), mask_store(X, delta_h + 2 * PS, 256), P::kTileOps, P::kTileOps);     // W3^T
    dense<P, 8, 8, false, kBwdPrefetch>(ws, X, 0, ldH(1), mask_store(Y, delta_h + 1 * PS, 256), P::kTileOps, P::kTileOps);     // W2^T
    dense<P, 8, 8, false, kBwdPrefetch>(ws, Y, 0, ldH(0), mask_store(X, delta_h + 0 * PS, 256), P::kTileOps, P::kTileOps);      // W1^T
    if (want_pts) {
        dense<P, 8, 3, false>(ws, X, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL {       // W0^T
            if (nt == 0) pe_adjoint_tile<4, 10, 0, P::kAccurateTrig>(acc, h, x4, dx4);
            else if (nt == 1) pe_adjoint_tile<4, 10, 1, P::kAccurateTrig>(acc, h, x4, dx4);
            else pe_adjoint_tile<4, 10, 2, P::kAccurateTrig>(acc, h, x4, dx4);
        });
        // pts4 = [p / r, 1 / r], r = clip(|p|, 1, 1e10): inside the clip r = |p| (d r / d p = p / r), outside it is a constant
        if (ok && h == 0) {
            const bool free_r = nrm >= 1.0f && nrm <= 1e10f;
            const float rr = fminf(fmaxf(nrm, 1.0f), 1e10f);
            const float pu = pw[0] * dx4[0] + pw[1] * dx4[1] + pw[2] * dx4[2];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float v = dx4[d] / rr;
                if (free_r) v -= pw[d] * (pu / (rr * rr * rr) + dx4[3] / (rr * rr * rr));
                a.d_pts[pd * 3 + d] = v;
                a.d_dirs[pd * 3 + d] = ddir[d];
            }
        }
    }
}

template <class P>
int launch_nerf_bwd(const VdnNerfBwdArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->P <= 0 || !args->blob || !args->g_density || !args->g_rgb || !args->save_h || !args->save_hv ||
        !args->delta_o || !args->delta_v || !args->delta_head || !args->delta_h) return -1;
    if (args->d_pts && (!args->d_dirs || !args->rays_o || !args->rays_d || !args->z || args->n_per_ray <= 0)) return -2;
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
