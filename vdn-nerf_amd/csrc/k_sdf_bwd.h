// Backward of the SDF network including the double backward through d sdf/d x, on gfx950; shared
// body for both precision policies.
//
// Forward quantities (k_sdf_fwd.h):  a_l = W_l x_l + b_l,  h_{l+1} = softplus(a_l),  s_l = softplus'(a_l),
// sweep  v_l = u_{l+1} * s_l,  u_l = W_l^T v_l,  normal = scale * J_PE^T (u_0 + u_4[PE part]).
// Adjoint, given (g_sdf, g_feat, g_normal):
//   rbar (ascending l):  ub_0 = ub_4[PE] = scale * J_PE g_normal;   vb_l = W_l ub_l;
//                        ub_{l+1} = vb_l * s_l;   ex_l = 100 * vb_l * v_l * (1 - s_l)      [softplus'' = 100 s (1-s)]
//                        weight grads:  dW_l += v_l ub_l^T  (dw gemm), dW_8[sdf row] += colsum(ub_8) / scale
//   fbar (descending l): ab_8 = [g_feat | g_sdf/scale];  hb_l = W_l^T ab_l;  ab_{l-1} = hb_l * s_{l-1} + ex_{l-1}
//                        weight grads:  dW_l += ab_l x_l^T,  db_l = colsum(ab_l)
// This is the hand-derived form of what autograd builds for reference fields.py:97-108 with
// create_graph=True and then differentiates in dpt_runner.py:253.
#pragma once
#include <cstdlib>
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

// PF: chunk steps between the issue of a tile's plane loads and their use (mlp_engine.h: dense)
template <class P, int PF>
__global__ __launch_bounds__(P::kWaves * 64, P::kMinWavesPerEU) void sdf_rbar_kernel(SdfRbarArgs a) {
    using ST = typename P::store_t;
    constexpr int kSlot = P::stride(9);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WStream<P::kWaves, kSlot> ws;
    ws.init(a.blob, smem, 63);        // hidden layers 0..7 of the forward stream
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const WorkRow wr = work_row(a.active_idx, a.n_active, a.P, P::kWaves, ws.wave, c);
    if (wr.none) return;
    ws.warm(wr.n_wg, 256 * P::kMinWavesPerEU);
    const bool ok = wr.ok;
    const long p = wr.row, pd = wr.point;          // p: row of the saves / adjoint planes; pd: dense point id
    const long Pn = P::rows(a.P), PS = Pn * 256;
    const ST* S = reinterpret_cast<const ST*>(a.S);
    const int from_h = a.s_from_h;
    // V holds v (s_from_h 0/1) or 100 log2(e) v (s_from_h 2): softplus'' = 100 s (1 - s) either way
    const float ex_k = from_h == 2 ? 0.6931471805599453f : 100.0f;
    const ST* V = reinterpret_cast<const ST*>(a.V);
    ST* EX = reinterpret_cast<ST*>(a.EX);

    float xin[3];
    if (a.pts != nullptr) {
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = a.pts[pd * 3 + d] * a.scale;
    } else {
        const long r = pd / a.n_per_ray;
        const float z = a.z[r * a.z_ld + (pd - r * a.n_per_ray)];
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = (a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z) * a.scale;
    }
    // ub_total = scale * J_PE g_n  (39 values): adjoint of n = scale * J^T u
    float ub39[39];
    {
        float gn[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            gn[d] = a.g_normals[pd * 3 + d] * a.scale;
            ub39[d] = gn[d];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float f = (float)(1 << k);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float sn, co;
                sincos_pe<P::kAccurateTrig>(xin[d] * f, sn, co);
                ub39[3 + 6 * k + d] = f * co * gn[d];
                ub39[3 + 6 * k + 3 + d] = -f * sn * gn[d];
            }
        }
    }
    ST* ub0 = reinterpret_cast<ST*>(a.UB);
    ST* ub1 = ub0 + Pn * 64;
    ST* ub2 = ub1 + PS;
    ST* ub3 = ub2 + PS;
    ST* ub4 = ub3 + PS;
    ST* ub5 = ub4 + Pn * 288;
    ST* ub6 = ub5 + PS;
    ST* ub7 = ub6 + PS;
    ST* ub8 = ub7 + PS;

    typename P::template Act<9> X;
    typename P::template Act<8> Y;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const f32x16 t16 = vals_tile<39>(ub39, h, kt);
        X.set(kt, t16);
        P::store_tile(ub0, p, 64, kt, h, t16, ok);
    }
    struct SV { typename P::raw_tile s, v; };
    auto ldSV = [&](int l) VDN_INL {
        return [=](int nt) VDN_INL {
            SV r;
            r.s = P::load_raw(S + l * PS, p, 256, nt, h);
            r.v = P::load_raw(V + l * PS, p, 256, nt, h);
            return r;
        };
    };
    // epilogue of layer l: ub_{l+1} = vb * s_l -> D (registers) and dst (HBM); ex_l -> EX[l]
    auto epi = [&](auto& D, ST* dst, int ld, int l) VDN_INL {
        return [&D, dst, ld, l, EX, PS, p, ok, h, from_h, ex_k](int nt, const f32x16& acc, const SV& sv) VDN_INL {
            f32x16 ub, ex;
            const f32x16 sr = P::unpack(sv.s), vr = P::unpack(sv.v);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float s = sprime(sr[t], from_h);
                ub[t] = acc[t] * s;
                ex[t] = ex_k * acc[t] * vr[t] * (1.0f - s);
            }
            D.set(nt, ub);
            P::store_tile(dst, p, ld, nt, h, ub, ok);
            P::store_tile(EX + l * PS, p, 256, nt, h, ex, ok);
        };
    };
    ws.all_issue = __any(ok);
    ws.start();
    dense<P, 2, 8, false, PF>(ws, X, 0, ldSV(0), epi(Y, ub1, 256, 0), 2 * P::kTileOps, 2 * P::kTileOps);
    dense<P, 8, 8, false, PF>(ws, Y, 0, ldSV(1), epi(X, ub2, 256, 1), 2 * P::kTileOps, 2 * P::kTileOps);
    dense<P, 8, 8, false, PF>(ws, X, 0, ldSV(2), epi(Y, ub3, 256, 2), 2 * P::kTileOps, 2 * P::kTileOps);
    dense<P, 8, 7, false, PF>(ws, Y, 0, ldSV(3), epi(X, ub4, 288, 3), 2 * P::kTileOps, 2 * P::kTileOps);      // ub_4[h part]: 7 tiles
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {                                        // ub_4[PE part] = ub_total
        const f32x16 t16 = vals_tile<39>(ub39, h, kt);
        X.set(7 + kt, t16);
        P::store_tile(ub4, p, 288, 7 + kt, h, t16, ok);
    }
    dense<P, 9, 8, false, PF>(ws, X, 0, ldSV(4), epi(Y, ub5, 256, 4), 2 * P::kTileOps, 2 * P::kTileOps);
    dense<P, 8, 8, false, PF>(ws, Y, 0, ldSV(5), epi(X, ub6, 256, 5), 2 * P::kTileOps, 2 * P::kTileOps);
    dense<P, 8, 8, false, PF>(ws, X, 0, ldSV(6), epi(Y, ub7, 256, 6), 2 * P::kTileOps, 2 * P::kTileOps);
    dense<P, 8, 8, false, PF>(ws, Y, 0, ldSV(7), epi(X, ub8, 256, 7), 2 * P::kTileOps, 2 * P::kTileOps);
}

template <class P, int PF>
__global__ __launch_bounds__(P::kWaves * 64, P::kMinWavesPerEU) void sdf_fbar_kernel(SdfFbarArgs a) {
    using ST = typename P::store_t;
    constexpr int kSlot = P::stride(9);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    WStream<P::kWaves, kSlot> ws;
    const bool want_pts = a.d_pts != nullptr;      // differentiable rays: two more chunks (W0^T) and the encoding's adjoint
    ws.init(a.blob, smem, want_pts ? 67 : 65);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const WorkRow wr = work_row(a.active_idx, a.n_active, a.P, P::kWaves, ws.wave, c);
    if (wr.none) return;
    ws.warm(wr.n_wg, 256 * P::kMinWavesPerEU);
    const bool ok = wr.ok;
    const long p = wr.row, pd = wr.point;          // p: row of the saves / adjoint planes; pd: dense point id
    const long Pn = P::rows(a.P), PS = Pn * 256;
    const ST* S = reinterpret_cast<const ST*>(a.S);
    const int from_h = a.s_from_h;
    const ST* EX = reinterpret_cast<const ST*>(a.EX);
    float xin[3] = {0.0f, 0.0f, 0.0f};
    if (want_pts) {
        if (a.pts != nullptr) {
#pragma unroll
            for (int d = 0; d < 3; ++d) xin[d] = a.pts[pd * 3 + d] * a.scale;
        } else {
            const long r = pd / a.n_per_ray;
            const float z = a.z[r * a.z_ld + (pd - r * a.n_per_ray)];
#pragma unroll
            for (int d = 0; d < 3; ++d) xin[d] = (a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z) * a.scale;
        }
    }
    // d loss / d xin accumulated from the adjoints of the encoded input (two tiles: the skip layer's PE part, then W0^T ab_0):
    // J_PE(xin)^T applied at once, so that only three values stay alive across the remaining layers
    float dx[3] = {0.0f, 0.0f, 0.0f};
    auto pe_adjoint = [&](const f32x16 (&T2)[2]) VDN_INL {
        float g[39];
        tiles_vals<39, 2>(T2, h, g);
#pragma unroll
        for (int d = 0; d < 3; ++d) dx[d] += g[d];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float f = (float)(1 << k);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float sn, co;
                sincos_pe<P::kAccurateTrig>(xin[d] * f, sn, co);
                dx[d] += f * (co * g[3 + 6 * k + d] - sn * g[3 + 6 * k + 3 + d]);
            }
        }
    };
    const ST* g_feat = reinterpret_cast<const ST*>(a.g_feat);
    ST* ab8 = reinterpret_cast<ST*>(a.AB);
    auto ab = [&](int l) VDN_INL { return ab8 + Pn * 288 + (long)(7 - l) * PS; };   // l = 7..0

    typename P::template Act<9> X;
    typename P::template Act<8> Y;
#pragma unroll
    for (int kt = 0; kt < 9; ++kt) {
        f32x16 t16;
        if (kt < 8) {
            t16 = P::load_tile(g_feat, p, 256, kt, h);
        } else {
            float g1[1] = {a.g_sdf[pd] / a.scale};
            t16 = vals_tile<1>(g1, h, 0);
        }
        X.set(kt, t16);
        P::store_tile(ab8, p, 288, kt, h, t16, ok);
    }
    struct SE { typename P::raw_tile s, e; };
    auto ldSE = [&](int l) VDN_INL {
        return [=](int nt) VDN_INL {
            SE r;
            r.s = P::load_raw(S + l * PS, p, 256, nt, h);
            r.e = P::load_raw(EX + l * PS, p, 256, nt, h);
            return r;
        };
    };
    auto epi = [&](auto& D, int l) VDN_INL {       // ab_l = hb_{l+1} * s_l + ex_l
        ST* dst = ab(l);
        return [&D, dst, p, ok, h, from_h](int nt, const f32x16& acc, const SE& se) VDN_INL {
            f32x16 o;
            const f32x16 sr = P::unpack(se.s), er = P::unpack(se.e);
#pragma unroll
            for (int t = 0; t < 16; ++t) o[t] = acc[t] * sprime(sr[t], from_h) + er[t];
            D.set(nt, o);
            P::store_tile(dst, p, 256, nt, h, o, ok);
        };
    };
    ws.all_issue = __any(ok);
    ws.start();
    dense<P, 9, 8, false, PF>(ws, X, 0, ldSE(7), epi(Y, 7), P::kTileOps, 2 * P::kTileOps);     // W8^T
    dense<P, 8, 8, false, PF>(ws, Y, 0, ldSE(6), epi(X, 6), P::kTileOps, 2 * P::kTileOps);     // W7^T
    dense<P, 8, 8, false, PF>(ws, X, 0, ldSE(5), epi(Y, 5), P::kTileOps, 2 * P::kTileOps);     // W6^T
    dense<P, 8, 8, false, PF>(ws, Y, 0, ldSE(4), epi(X, 4), P::kTileOps, 2 * P::kTileOps);     // W5^T
    {   // W4^T: 9 output tiles = [h4 part (7) | PE part (2: the skip input's adjoint, only wanted for differentiable rays)]
        ST* dst = ab(3);
        f32x16 PE4[2];
        dense<P, 8, 9, false, PF>(ws, X, 0,
            [&](int nt) VDN_INL {
                SE r{};
                if (nt < 7) {
                    r.s = P::load_raw(S + 3 * PS, p, 256, nt, h);
                    r.e = P::load_raw(EX + 3 * PS, p, 256, nt, h);
                }
                return r;
            },
            [&](int nt, const f32x16& acc, const SE& se) VDN_INL {
                if (nt < 7) {
                    f32x16 o;
                    const f32x16 sr = P::unpack(se.s), er = P::unpack(se.e);
#pragma unroll
                    for (int t = 0; t < 16; ++t) o[t] = acc[t] * sprime(sr[t], from_h) + er[t];
                    Y.set(nt, o);
                    P::store_tile(dst, p, 256, nt, h, o, ok);
                } else {
                    PE4[nt - 7] = acc;
                }
            });
        if (want_pts) pe_adjoint(PE4);
    }
    dense<P, 7, 8, false, PF>(ws, Y, 0, ldSE(2), epi(X, 2), P::kTileOps, 2 * P::kTileOps);     // W3^T
    dense<P, 8, 8, false, PF>(ws, X, 0, ldSE(1), epi(Y, 1), P::kTileOps, 2 * P::kTileOps);     // W2^T
    dense<P, 8, 8, false, PF>(ws, Y, 0, ldSE(0), epi(X, 0), P::kTileOps, 2 * P::kTileOps);      // W1^T
    if (want_pts) {
        // d loss / d point = scale * d loss / d xin:  J_PE^T (W0^T ab_0 + [W4^T ab_4]_PE)  through the activations, plus the
        // explicit dependence of normal = scale * J_PE(xin)^T u on xin at fixed u (the u-dependence went through rbar):
        //   d/d xin_d [ f (cos(f xin_d) u_sin - sin(f xin_d) u_cos) ] = -f^2 (sin u_sin + cos u_cos)
        f32x16 U0[2];
        dense<P, 8, 2, false>(ws, X, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL { U0[nt] = acc; });   // W0^T
        pe_adjoint(U0);
        if (ok && h == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float gn = a.g_normals[pd * 3 + d] * a.scale;
                float e = 0.0f;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const float f = (float)(1 << k);
                    float sn, co;
                    sincos_pe<P::kAccurateTrig>(xin[d] * f, sn, co);
                    e -= f * f * (sn * a.U_pe[p * 39 + 3 + 6 * k + d] + co * a.U_pe[p * 39 + 3 + 6 * k + 3 + d]);
                }
                const float v = (dx[d] + gn * e) * a.scale;
                a.d_pts[pd * 3 + d] = a.acc_pts ? a.d_pts[pd * 3 + d] + v : v;
            }
        }
    }
}

// VDN_PLANE_PREFETCH = 1 | 2 (default 2): read once per process
inline int plane_prefetch_depth() {
    static const int d = [] { const char* e = getenv("VDN_PLANE_PREFETCH"); return (e != nullptr && e[0] == '1') ? 1 : 2; }();
    return d;
}

template <class P>
int launch_sdf_rbar(const VdnSdfRbarArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->P <= 0 || !args->blob || !args->g_normals || !args->S || !args->V || !args->UB || !args->EX) return -1;
    if (!args->pts && (!args->rays_o || !args->rays_d || !args->z || args->n_per_ray <= 0 || args->z_ld < args->n_per_ray)) return -2;
    const int ppw = P::kWaves * 32;
    const size_t lds = 3 * P::stride(9);
    static bool once = (allow_big_lds(sdf_rbar_kernel<P, 1>, lds), allow_big_lds(sdf_rbar_kernel<P, 2>, lds), true);
    (void)once;
    if (plane_prefetch_depth() == 2)
        hipLaunchKernelGGL((sdf_rbar_kernel<P, 2>), dim3((args->P + ppw - 1) / ppw), dim3(P::kWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL((sdf_rbar_kernel<P, 1>), dim3((args->P + ppw - 1) / ppw), dim3(P::kWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}

template <class P>
int launch_sdf_fbar(const VdnSdfFbarArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->P <= 0 || !args->blob || !args->g_sdf || !args->g_feat || !args->S || !args->EX || !args->AB) return -1;
    if (args->d_pts && (!args->g_normals || !args->U_pe ||
                        (!args->pts && (!args->rays_o || !args->rays_d || !args->z || args->n_per_ray <= 0 || args->z_ld < args->n_per_ray)))) return -2;
    const int ppw = P::kWaves * 32;
    const size_t lds = 3 * P::stride(9);
    static bool once = (allow_big_lds(sdf_fbar_kernel<P, 1>, lds), allow_big_lds(sdf_fbar_kernel<P, 2>, lds), true);
    (void)once;
    if (plane_prefetch_depth() == 2)
        hipLaunchKernelGGL((sdf_fbar_kernel<P, 2>), dim3((args->P + ppw - 1) / ppw), dim3(P::kWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL((sdf_fbar_kernel<P, 1>), dim3((args->P + ppw - 1) / ppw), dim3(P::kWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}

}  // namespace vdn
