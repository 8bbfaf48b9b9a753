// Backward of RenderingNetwork (colour head / VDN head) on gfx950, shared body for both policies.
// delta chain: delta_4 = g_out * act'(out); delta_{l-1} = (W_l^T delta_l) * [h_l > 0]; the last
// transposed layer yields d loss / d [feature | points, PE(view), normals]. Per-layer deltas go to
// HBM row-major for the weight-gradient GEMM. Adjoint of fields.py:148-176.
#pragma once
#include <type_traits>
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

template <class P, int NT_OUT, int EX = 0>      // EX: extra input tiles of the d_feature = 352 network (k_render_fwd.h)
__global__ __launch_bounds__(P::kWaves * 64, P::kMinWavesPerEU) void rendernet_bwd_kernel(RenderNetBwdArgs a) {
    using ST = typename P::store_t;
    constexpr int kSlot = P::stride(8);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WStream<P::kWaves, kSlot> ws;
    ws.init(a.blob, smem, 42 + EX);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const WorkRow wr = work_row(a.active_idx, a.n_active, a.P, P::kWaves, ws.wave, c);
    if (wr.none) return;
    ws.warm_issue(wr.n_wg, 256 * P::kMinWavesPerEU);      // (mlp_engine.h; ends before the ring starts)
    warm_code_issue((std::is_same<P, BF16>::value) ? kWarmCodeRenderBwd : 0, wr.n_wg, 256 * P::kMinWavesPerEU, ws.warm_dump());      // (the kernel's own code: vdn_common.h)
    const bool ok = wr.ok;
    const long p = wr.row, pd = wr.point;          // p: row of saves / deltas / d_feat; pd: dense point id
    const long PS = P::plane(a.P, 256);
    const ST* save_h = reinterpret_cast<const ST*>(a.save_h);
    ST* delta_h = reinterpret_cast<ST*>(a.delta_h);
    ST* delta_out = reinterpret_cast<ST*>(a.delta_out);
    ST* d_feat = reinterpret_cast<ST*>(a.d_feat);

    typename P::template Act<8> X, Y;
    if constexpr (NT_OUT == 1) {
        float dl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dl[j] = 0.0f;
            if (j < a.d_out) {
                const float o = a.out[pd * a.d_out + j], g = a.g_out[pd * a.d_out + j];
                dl[j] = a.squeeze_out ? g * o * (1.0f - o) : (o > 0.0f ? g : 0.0f);
            }
        }
        const f32x16 t16 = vals_tile<4>(dl, h, 0);
        X.set(0, t16);
        P::store_tile(delta_out, p, 32, 0, h, t16, ok);
    } else {
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const f32x16 o = F32::load_tile(a.out, pd, 96, kt, h);
            const f32x16 g = F32::load_tile(a.g_out, pd, 96, kt, h);
            f32x16 dl;
#pragma unroll
            for (int t = 0; t < 16; ++t) dl[t] = a.squeeze_out ? g[t] * o[t] * (1.0f - o[t]) : (o[t] > 0.0f ? g[t] : 0.0f);
            X.set(kt, dl);
            P::store_tile(delta_out, p, 96, kt, h, dl, ok);
        }
    }
    auto ldH = [&](int l) VDN_INL { return [=](int nt) VDN_INL { return P::load_tile(save_h + l * PS, p, 256, nt, h); }; };
    // D = acc * [saved activation > 0]; kept in registers and stored row-major
    auto mask_store = [&](auto& D, int l) VDN_INL {
        return [&D, l, delta_h, PS, p, ok, h](int nt, const f32x16& acc, const f32x16& hv) VDN_INL {
            f32x16 o;
#pragma unroll
            for (int t = 0; t < 16; ++t) o[t] = hv[t] > 0.0f ? acc[t] : 0.0f;
            D.set(nt, o);
            P::store_tile(delta_h + l * PS, p, 256, nt, h, o, ok);
        };
    };
    ws.all_issue = __any(ok);
    warm_l2_wait();
    ws.start();
    dense<P, NT_OUT, 8, false, kBwdPrefetch>(ws, X, 0, ldH(3), mask_store(Y, 3), P::kTileOps, P::kTileOps);   // W4^T
    dense<P, 8, 8, false, kBwdPrefetch>(ws, Y, 0, ldH(2), mask_store(X, 2), P::kTileOps, P::kTileOps);        // W3^T
    dense<P, 8, 8, false, kBwdPrefetch>(ws, X, 0, ldH(1), mask_store(Y, 1), P::kTileOps, P::kTileOps);        // W2^T
    dense<P, 8, 8, false, kBwdPrefetch>(ws, Y, 0, ldH(0), mask_store(X, 0), P::kTileOps, P::kTileOps);        // W1^T
    f32x16 SM[2];
    dense<P, 8, 10 + EX, false>(ws, X, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL {   // W0^T
        if (nt >= 10) {
            if constexpr (EX > 0) {         // adjoint of the appended VDN channels: joins the VDN head's output adjoint
                f32x16 o = F32::load_tile(a.d_extra, pd, 32 * EX, nt - 10, h);
#pragma unroll
                for (int t = 0; t < 16; ++t) o[t] += acc[t];
                F32::store_tile(a.d_extra, pd, 32 * EX, nt - 10, h, o, ok);
            }
        } else if (nt < 8) {
            f32x16 o = acc;
            if (a.acc_feat) {
                const f32x16 prev = P::load_tile(d_feat, p, 256, nt, h);
#pragma unroll
                for (int t = 0; t < 16; ++t) o[t] += prev[t];
            }
            P::store_tile(d_feat, p, 256, nt, h, o, ok);
        } else {
            SM[nt - 8] = acc;
        }
    });
    float small[33];
    tiles_vals<33, 2>(SM, h, small);        // [points(3), PE(view)(27), normals(3)]
    if (ok && h == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float prev = a.acc_normals ? a.d_normals[pd * 3 + d] : 0.0f;
            a.d_normals[pd * 3 + d] = prev + small[30 + d];
        }
        if (a.d_pts != nullptr) {
            // differentiable rays: d loss / d points (slots 0..2) and, through the transpose Jacobian of the 4-octave
            // encoding (slots 3..29 = [dir, sin(2^k dir), cos(2^k dir)]), d loss / d view_dirs
            const long r = pd / a.n_per_ray;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float dir = a.dirs ? a.dirs[pd * 3 + d] : a.rays_d[r * 3 + d];
                float g = small[3 + d];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float f = (float)(1 << k);
                    float sn, co;
                    sincos_pe<P::kAccurateTrig>(dir * f, sn, co);
                    g += f * (co * small[3 + 3 + 6 * k + d] - sn * small[3 + 3 + 6 * k + 3 + d]);
                }
                const float pp = a.acc_pts ? a.d_pts[pd * 3 + d] : 0.0f;
                const float pg = a.acc_pts ? a.d_dirs[pd * 3 + d] : 0.0f;
                a.d_pts[pd * 3 + d] = pp + small[d];
                a.d_dirs[pd * 3 + d] = pg + g;
            }
        }
    }
}

template <class P>
int launch_rendernet_bwd(const VdnRenderNetBwdArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->P <= 0 || !args->blob || !args->g_out || !args->out || !args->save_h || !args->delta_out ||
        !args->delta_h || !args->d_feat || !args->d_normals) return -1;
    if (!(args->d_out == 96 || (args->d_out >= 1 && args->d_out <= 4))) return -2;
    if (args->d_pts && (!args->d_dirs || (!args->dirs && (!args->rays_d || args->n_per_ray <= 0)))) return -3;
    const int ppw = P::kWaves * 32;
    const int grid = (args->P + ppw - 1) / ppw;
    const size_t lds = 3 * P::stride(8);
    static bool once = (allow_big_lds(rendernet_bwd_kernel<P, 1>, lds), allow_big_lds(rendernet_bwd_kernel<P, 3>, lds),
                        allow_big_lds(rendernet_bwd_kernel<P, 1, 3>, lds), true);
    (void)once;
    if (args->d_extra != nullptr) {
        if (args->d_out == 96) return -4;
        hipLaunchKernelGGL((rendernet_bwd_kernel<P, 1, 3>), dim3(grid), dim3(P::kWaves * 64), lds, stream, *args);
        return (int)hipGetLastError();
    }
    if (args->d_out == 96)
        hipLaunchKernelGGL((rendernet_bwd_kernel<P, 3>), dim3(grid), dim3(P::kWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL((rendernet_bwd_kernel<P, 1>), dim3(grid), dim3(P::kWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}

}  // namespace vdn
