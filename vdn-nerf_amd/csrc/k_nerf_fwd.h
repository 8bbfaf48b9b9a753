// Background NeRF++ MLP forward on gfx950, shared body for both precision policies.
// Fuses the inverted-sphere parameterisation of renderer.py:112-115 (pts4 = [pts/r, 1/r],
// r = max(|pts|,1)), PE10(pts4) / PE4(view), the 8x256 ReLU trunk with its skip after layer 4,
// and the alpha / feature / views / rgb (/ 96-ch dpt) heads. Replaces reference
// dpt_models/fields.py:324-353 as called from renderer.py:100-123.
#pragma once
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

template <class P, bool DPT>
__global__ __launch_bounds__(P::kWaves * 64, P::kMinWavesPerEU) void nerf_fwd_kernel(NerfArgs a) {
    using ST = typename P::store_t;
    constexpr int kSlot = P::stride(11);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WStream<P::kWaves, kSlot> ws;
    ws.init(a.blob, smem, 64 + 9 + 4 + (DPT ? 4 : 1));
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    // q = row of this lane in the (possibly compacted) work list = row of its training saves; p = its dense point id
    const long n_rows = a.active_idx != nullptr ? (long)*a.n_active : (long)a.P;
    if ((long)blockIdx.x * P::kWaves * 32 >= n_rows) return;          // whole workgroup beyond the active list
    if (a.save_h != nullptr) ws.warm((n_rows + P::kWaves * 32 - 1) / (P::kWaves * 32), 256 * P::kMinWavesPerEU);
    const long q_raw = ((long)blockIdx.x * P::kWaves + ws.wave) * 32 + c;
    const bool ok = q_raw < n_rows;
    const long q = ok ? q_raw : n_rows - 1;
    const long p = a.active_idx != nullptr ? (long)a.active_idx[q] : q;
    const long r = p / a.n_per_ray;
    ST* save_h = reinterpret_cast<ST*>(a.save_h);
    const long PS = P::plane(a.P, 256);

    float dir[3], p4[4];
#pragma unroll
    for (int d = 0; d < 3; ++d) dir[d] = a.dirs ? a.dirs[p * 3 + d] : a.rays_d[r * 3 + d];
    if (a.pts4 != nullptr) {
#pragma unroll
        for (int d = 0; d < 4; ++d) p4[d] = a.pts4[p * 4 + d];
    } else {
        const float z = a.z[p];
        float q[3];
        float n2 = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            q[d] = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;
            n2 += q[d] * q[d];
        }
        const float rr = fminf(fmaxf(sqrtf(n2), 1.0f), 1e10f);   // renderer.py:114
#pragma unroll
        for (int d = 0; d < 3; ++d) p4[d] = q[d] / rr;
        p4[3] = 1.0f / rr;
    }
    typename P::template Act<11> X;
    typename P::template Act<9> Y;
    auto put_pe = [&](bool save) VDN_INL {
        float pe[84];
        posenc<4, 10, P::kAccurateTrig>(p4, pe);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const f32x16 t16 = vals_tile<84>(pe, h, kt);
            X.set(kt, t16);
            if (save && a.save_pe != nullptr) P::store_tile(reinterpret_cast<ST*>(a.save_pe), q, 96, kt, h, t16, ok);
        }
    };
    // D tile (t0 + nt) <- relu(acc); optionally kept row-major for the backward
    auto relu_into = [&](auto& D, int t0, ST* save, int ld) VDN_INL {
        return [&D, t0, save, ld, q, ok, h](int nt, const f32x16& acc, int) VDN_INL {
            f32x16 o;
#pragma unroll
            for (int t = 0; t < 16; ++t) o[t] = relu0(acc[t]);
            D.set(t0 + nt, o);
            if (save != nullptr) P::store_tile(save, q, ld, nt, h, o, ok);
        };
    };
    auto sv = [&](int l) VDN_INL { return save_h ? save_h + l * PS : (ST*)nullptr; };
    const int est = save_h != nullptr ? P::kTileOps : 0;
    ws.all_issue = __any(ok);
    put_pe(true);
    ws.start();
    dense<P, 3, 8, true>(ws, X, 0, NoPre{}, relu_into(Y, 0, sv(0), 256), est);          // pts_linears.0
    dense<P, 8, 8, true>(ws, Y, 0, NoPre{}, relu_into(X, 0, sv(1), 256), est);          // 1
    dense<P, 8, 8, true>(ws, X, 0, NoPre{}, relu_into(Y, 0, sv(2), 256), est);          // 2
    dense<P, 8, 8, true>(ws, Y, 0, NoPre{}, relu_into(X, 0, sv(3), 256), est);          // 3
    dense<P, 8, 8, true>(ws, X, 0, NoPre{}, relu_into(Y, 0, sv(4), 256), est);         // 4
    // skip (fields.py:334-335): h = cat([input_pts, h]) -> X = [PE (3 tiles) | h (8 tiles)]
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) X.copy_tile(3 + kt, Y, kt);
    put_pe(false);
    dense<P, 11, 8, true>(ws, X, 0, NoPre{}, relu_into(Y, 0, sv(5), 256), est);         // 5
    dense<P, 8, 8, true>(ws, Y, 0, NoPre{}, relu_into(X, 0, sv(6), 256), est);          // 6
    dense<P, 8, 8, true>(ws, X, 0, NoPre{}, relu_into(Y, 0, sv(7), 256), est);          // 7
    // heads on h: image rows 0..255 feature_linear, row 256 alpha_linear
    dense<P, 8, 9, true>(ws, Y, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL {
        if (nt < 8) {
            X.set(nt, acc);
            if (a.save_feature != nullptr) P::store_tile(reinterpret_cast<ST*>(a.save_feature), q, 256, nt, h, acc, ok);
        } else {
            if (ok && h == 0) a.density[p] = acc[0];
        }
    });
    {   // views_linears.0 on cat([feature, PE4(view)])  (fields.py:340-344)
        float pe[27];
        posenc<3, 4, P::kAccurateTrig>(dir, pe);
        const f32x16 t16 = vals_tile<27>(pe, h, 0);
        X.set(8, t16);
        if (a.save_vpe != nullptr) P::store_tile(reinterpret_cast<ST*>(a.save_vpe), q, 32, 0, h, t16, ok);
    }
    dense<P, 9, 4, true>(ws, X, 0, NoPre{}, relu_into(Y, 0, reinterpret_cast<ST*>(a.save_hv), 128));
    // rgb_linear (image tile 0, rows 0..2) and dpt_linear (image tiles 1..3)
    dense<P, 4, DPT ? 4 : 1, true>(ws, Y, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL {
        if (nt == 0) {
            if (ok && h == 0) {
                a.rgb[p * 3 + 0] = acc[0];
                a.rgb[p * 3 + 1] = acc[1];
                a.rgb[p * 3 + 2] = acc[2];
            }
        } else {
            F32::store_tile(a.feat, p, 96, nt - 1, h, acc, ok);      // network outputs: always f32
        }
    });
}

template <class P>
int launch_nerf_fwd(const VdnNerfArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || args->P <= 0 || !args->blob || !args->density || !args->rgb || args->n_per_ray <= 0) return -1;
    if (!args->pts4 && (!args->rays_o || !args->rays_d || !args->z)) return -1;
    if (!args->dirs && !args->rays_d) return -1;
    const int ppw = P::kWaves * 32;
    const int grid = (args->P + ppw - 1) / ppw;
    const size_t lds = 3 * P::stride(11);
    static bool once = (allow_big_lds(nerf_fwd_kernel<P, false>, lds), allow_big_lds(nerf_fwd_kernel<P, true>, lds), true);
    (void)once;
    if (args->feat != nullptr)
        hipLaunchKernelGGL((nerf_fwd_kernel<P, true>), dim3(grid), dim3(P::kWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL((nerf_fwd_kernel<P, false>), dim3(grid), dim3(P::kWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}

}  // namespace vdn
