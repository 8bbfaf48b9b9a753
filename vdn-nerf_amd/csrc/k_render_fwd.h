// RenderingNetwork forward (colour head d_out<=4, VDN feature head d_out=96) on gfx950, shared body
// for both precision policies. Input assembly [points(3), PE4(view_dirs)(27), normals(3),
// feature(256)] (mode 'idr'), 4 hidden ReLU layers of 256, sigmoid output. Replaces reference
// dpt_models/fields.py:148-176. K order inside the kernel is [feature(256) | points, PE(view),
// normals (33 -> 64)]; the weight image builder permutes the first layer's columns accordingly.
#pragma once
#include <type_traits>
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

// EX: extra feature tiles behind the 10 standard input tiles (3 = the 96 VDN channels of depth_before_color, renderer.py:247-248)
template <class P, int NT_OUT, int EX = 0>   // NT_OUT 1: d_out <= 4 (colour); 3: d_out = 96 (VDN head)
VDN_DEV void rendernet_fwd_body(const RenderNetArgs& a, char* smem) {
    using ST = typename P::store_t;
    constexpr int kSlot = P::stride(10 + EX);
    constexpr int kNSlot = 3 * kSlot > 160 * 1024 ? 2 : 3;      // the 13-k-tile f32 chunks are 56 KiB: two ring slots
    WStream<P::kWaves, kSlot, kNSlot> ws;
    ws.init(a.blob, smem, 32 + NT_OUT);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const WorkRow wr = work_row(a.active_idx, a.n_active, a.P, P::kWaves, ws.wave, c);
    if (wr.none) return;
    // training step: the stream is cold (vdn_common.h: warm_l2); it arrives in L2 underneath the input assembly below
    if (a.save_h != nullptr) ws.warm_issue(wr.n_wg, 256 * P::kMinWavesPerEU);
    warm_code_issue((a.save_h != nullptr && std::is_same<P, BF16>::value) ? kWarmCodeRenderFwd : 0, wr.n_wg, 256 * P::kMinWavesPerEU, ws.warm_dump());      // (the kernel's own code: vdn_common.h)
    const bool ok = wr.ok;
    const long p = wr.row, pd = wr.point;          // p: row of feat and of the saves; pd: dense point id
    const long r = pd / a.n_per_ray;
    const ST* feat = reinterpret_cast<const ST*>(a.feat);
    ST* save_h = reinterpret_cast<ST*>(a.save_h);
    ST* save_small = reinterpret_cast<ST*>(a.save_small);
    const long PS = P::plane(a.P, 256);

    typename P::template Act<10 + EX> X;
    typename P::template Act<8> Y;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) X.set(kt, P::load_tile(feat, p, 256, kt, h));
    if constexpr (EX > 0) {
#pragma unroll
        for (int kt = 0; kt < EX; ++kt) {
            const f32x16 t16 = F32::load_tile(a.extra, pd, 32 * EX, kt, h);
            X.set(10 + kt, t16);
            if (a.save_extra != nullptr) P::store_tile(reinterpret_cast<ST*>(a.save_extra), p, 32 * EX, kt, h, t16, ok);
        }
    }
    {
        float small[33];
        float dir[3];
        const float z = a.pts ? 0.0f : a.z[pd];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            dir[d] = a.dirs ? a.dirs[pd * 3 + d] : a.rays_d[r * 3 + d];
            small[d] = a.pts ? a.pts[pd * 3 + d] : a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;   // renderer.py:233
            small[30 + d] = a.normals[pd * 3 + d];
        }
        float pe[27];
        posenc<3, 4, P::kAccurateTrig>(dir, pe);
#pragma unroll
        for (int i = 0; i < 27; ++i) small[3 + i] = pe[i];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const f32x16 t16 = vals_tile<33>(small, h, kt);
            X.set(8 + kt, t16);
            if (save_small != nullptr) P::store_tile(save_small, p, 64, kt, h, t16, ok);
        }
    }
    auto relu_into = [&](auto& D, int l) VDN_INL {
        return [&D, l, save_h, PS, p, ok, h](int nt, const f32x16& acc, int) VDN_INL {
            f32x16 o;
#pragma unroll
            for (int t = 0; t < 16; ++t) o[t] = relu0(acc[t]);
            D.set(nt, o);
            if (save_h != nullptr) P::store_tile(save_h + l * PS, p, 256, nt, h, o, ok);
        };
    };
    const int est = save_h != nullptr ? P::kTileOps : 0;
    ws.all_issue = __any(ok);
    warm_l2_wait();
    ws.start();
    dense<P, 10 + EX, 8, true>(ws, X, 0, NoPre{}, relu_into(Y, 0), est);
    dense<P, 8, 8, true>(ws, Y, 0, NoPre{}, relu_into(X, 1), est);
    dense<P, 8, 8, true>(ws, X, 0, NoPre{}, relu_into(Y, 2), est);
    dense<P, 8, 8, true>(ws, Y, 0, NoPre{}, relu_into(X, 3), est);
    dense<P, 8, NT_OUT, true>(ws, X, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL {
        f32x16 o;
#pragma unroll
        for (int t = 0; t < 16; ++t) o[t] = a.squeeze_out ? sigmoidf_(acc[t]) : relu0(acc[t]);
        if constexpr (NT_OUT == 1) {
            if (ok && h == 0) {
                for (int j = 0; j < a.d_out && j < 4; ++j) a.out[pd * a.d_out + j] = o[j];
            }
        } else {
            F32::store_tile(a.out, pd, 96, nt, h, o, ok);      // network outputs feed the per-ray kernels: always f32
        }
    });
}

template <class P, int NT_OUT, int EX = 0>
__global__ __launch_bounds__(P::kWaves * 64, P::kMinWavesPerEU) void rendernet_fwd_kernel(RenderNetArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    rendernet_fwd_body<P, NT_OUT, EX>(a, smem);
}

template <class P>
int launch_rendernet_fwd(const VdnRenderNetArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || args->P <= 0 || !args->blob || !args->normals || !args->feat || !args->out || args->n_per_ray <= 0) return -1;
    if (!args->pts && (!args->rays_o || !args->rays_d || !args->z)) return -1;
    if (!args->dirs && !args->rays_d) return -1;
    if (!(args->d_out == 96 || (args->d_out >= 1 && args->d_out <= 4))) return -2;
    const int ppw = P::kWaves * 32;
    const int grid = (args->P + ppw - 1) / ppw;
    if (args->extra != nullptr) {
        if (args->d_out == 96) return -3;          // only the colour head takes the VDN channels
        const size_t lds3 = (3 * P::stride(13) > 160 * 1024 ? 2 : 3) * P::stride(13);
        static bool once3 = (allow_big_lds(rendernet_fwd_kernel<P, 1, 3>, lds3), true);
        (void)once3;
        hipLaunchKernelGGL((rendernet_fwd_kernel<P, 1, 3>), dim3(grid), dim3(P::kWaves * 64), lds3, stream, *args);
        return (int)hipGetLastError();
    }
    const size_t lds = 3 * P::stride(10);
    static bool once = (allow_big_lds(rendernet_fwd_kernel<P, 1>, lds), allow_big_lds(rendernet_fwd_kernel<P, 3>, lds), true);
    (void)once;
    if (args->d_out == 96)
        hipLaunchKernelGGL((rendernet_fwd_kernel<P, 3>), dim3(grid), dim3(P::kWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL((rendernet_fwd_kernel<P, 1>), dim3(grid), dim3(P::kWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}

}  // namespace vdn
