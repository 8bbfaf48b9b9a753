// SDF network forward on gfx950 (shared body for the F32 and BF16 policies of mlp_engine.h).
// Fuses: point generation (o + d*z) -> positional encoding -> 9 weight-normed layers with
// Softplus(beta=100) and the skip at layer 4 -> [sdf | 256-d feature], and (FULL mode) the analytic
// reverse sweep that yields d sdf / d x, all with activations resident in registers.
// Replaces reference dpt_models/fields.py:72-108 (SDFNetwork.forward / .sdf / .gradient).
#pragma once
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {

// MODE 0: sdf only; 1: sdf + feature + normals (+ training saves). NW = waves per workgroup (32 points each):
// small launches use fewer waves per workgroup so that the grid still covers the 256 CUs.
// DERIVE (bf16 training launch only): softplus' is not stored but re-derived from the saved activations.
// (the body is a device function so that the fp32 one-launch shading kernel - k_shade_f32.h - can run it in front of the colour head)
template <class P, int MODE, int NW, bool DERIVE = false>
VDN_DEV void sdf_fwd_body(const SdfArgs& a, char* smem) {
    using ST = typename P::store_t;
    constexpr int kSlot = P::stride(9);
#ifndef VDN_NSLOT
#define VDN_NSLOT 3
#endif
    WStream<NW, kSlot, VDN_NSLOT> ws;
    ws.init(a.blob, smem, MODE == 0 ? 64 : 131);       // chunks in the 'sdf' / 'full' stream
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const WorkRow wr = work_row(a.active_idx, a.n_active, a.P, NW, ws.wave, c);
    if (wr.none) return;
    if (a.H != nullptr || a.cold_start) ws.warm(wr.n_wg, 256 * P::kMinWavesPerEU);
    const bool ok = wr.ok;
    const long p = wr.row, pd = wr.point;          // p: row of the saves (and of feat); pd: dense point id
    ws.all_issue = __any(ok);

    float xin[3];
    long sdf_idx = pd;
    if (a.pts != nullptr) {
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = a.pts[pd * 3 + d] * a.scale;
    } else {
        const long r = pd / a.n_per_ray;
        const long sidx = pd - r * a.n_per_ray;
        const float z = a.z[r * a.z_ld + sidx];
        sdf_idx = r * a.sdf_ld + sidx;
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = (a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z) * a.scale;
    }
    constexpr bool SV = (MODE == 1);
    ST* S = reinterpret_cast<ST*>(a.S);
    ST* Hs = reinterpret_cast<ST*>(a.H);
    // bf16 training launch: softplus' is re-derived from the saved activations instead of being stored (mlp_engine.h)
    constexpr bool derive = DERIVE;
    const ST* Ssrc = derive ? Hs : S;
    ST* Vs = reinterpret_cast<ST*>(a.V);
    const long PS = P::plane(a.P, 256);

    typename P::template Act<9> X, Y;
    auto put_pe = [&](int tile0) VDN_INL {
        float pe[39];
        posenc<3, 6, P::kAccurateTrig>(xin, pe);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const f32x16 t16 = vals_tile<39>(pe, h, kt);
            X.set(tile0 + kt, t16);
            if constexpr (SV) {
                if (tile0 == 0 && a.PE != nullptr) P::store_tile(reinterpret_cast<ST*>(a.PE), p, 64, kt, h, t16, ok);
            }
        }
    };
    // hidden layer epilogue: D <- softplus100(acc); S <- softplus'(acc) (and H <- the activation). The softplus is the
    // per-register part (runs under the next tile's MFMAs on the bf16 path), packing / stores the per-tile part.
    struct HS { f32x16 hv, sv; };
    auto hidden = [&](auto& D, int l) VDN_INL {
        return elem_epi<HS>(
            [](int, int t, float acc, HS& o) VDN_INL {
                if constexpr (SV) {
                    float a_, b_;
                    softplus100_both(acc, a_, b_);
                    o.hv[t] = a_;
                    o.sv[t] = b_;
                } else {
#if defined(VDN_ABLATE) && VDN_ABLATE == 1
                    o.hv[t] = acc;
#else
                    o.hv[t] = softplus100_fast(acc);
#endif
                }
            },
            [&D, l, S, Hs, PS, p, ok, h, derive](int nt, const HS& o, int) VDN_INL {
                D.set(nt, o.hv);
                if constexpr (SV) {
                    if (!derive) P::store_tile(S + l * PS, p, 256, nt, h, o.sv, ok);
                    if (Hs != nullptr) P::store_tile(Hs + l * PS, p, 256, nt, h, o.hv, ok);
                }
            });
    };
    const int est_h = SV ? (Hs != nullptr && !derive ? 2 * P::kTileOps : P::kTileOps) : 0;     // stores per hidden-layer tile (S and / or H)
    const int est_v = Vs != nullptr ? P::kTileOps : 0;                // stores per sweep tile (V when training)
    put_pe(0);
    ws.start();
    dense<P, 2, 8, true>(ws, X, 0, NoPre{}, hidden(Y, 0), est_h);
    dense<P, 8, 8, true>(ws, Y, 0, NoPre{}, hidden(X, 1), est_h);
    dense<P, 8, 8, true>(ws, X, 0, NoPre{}, hidden(Y, 2), est_h);
    dense<P, 8, 7, true>(ws, Y, 0, NoPre{}, hidden(X, 3), est_h);
    put_pe(7);   // skip: layer-4 input = [h4 (217 -> 7 tiles) | PE (39 -> 2 tiles)] / sqrt2 (1/sqrt2 is in the image)
    dense<P, 9, 8, true>(ws, X, 0, NoPre{}, hidden(Y, 4), est_h);
    dense<P, 8, 8, true>(ws, Y, 0, NoPre{}, hidden(X, 5), est_h);
    dense<P, 8, 8, true>(ws, X, 0, NoPre{}, hidden(Y, 6), est_h);
    dense<P, 8, 8, true>(ws, Y, 0, NoPre{}, hidden(X, 7), est_h);

    const float inv_scale = 1.0f / a.scale;
    if constexpr (MODE == 0) {
        // last layer reduced to its sdf row (image row 0 = W8 row 0)
        dense<P, 8, 1, true>(ws, X, 0, NoPre{}, [&](int, const f32x16& acc, int) VDN_INL {
            if (ok && h == 0) a.sdf[sdf_idx] = acc[0] * inv_scale;
        });
    } else {
        // image rows: 0..255 = feature rows (W8 rows 1..256), 256 = sdf row (W8 row 0)
        ST* feat = reinterpret_cast<ST*>(a.feat);
        dense<P, 8, 9, true>(ws, X, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL {
            if (nt < 8) {
                P::store_tile(feat, p, 256, nt, h, acc, ok);
            } else {
                if (ok && h == 0) a.sdf[sdf_idx] = acc[0] * inv_scale;
            }
        });
        // ---- reverse sweep: u = d sdf / d(layer input), v = u (.) softplus'(a_l) ------------------
        // Every sweep layer's epilogue multiplies its output tile by the S tile of the layer below
        // (loaded right after the chunk acquire), so X/Y always hold v, ready to be the B operand.
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            const f32x16 w8 = F32::load_tile(a.w8row, 0, 0, kt, h);
            const f32x16 s7 = P::load_tile(Ssrc + 7 * PS, p, 256, kt, h);
            f32x16 v7;
#pragma unroll
            for (int t = 0; t < 16; ++t) v7[t] = w8[t] * inv_scale * sprime(s7[t], derive);
            Y.set(kt, v7);
            if (Vs != nullptr) P::store_tile(Vs + 7 * PS, p, 256, kt, h, v7, ok);
        }
        auto loadS = [&](int layer) VDN_INL {
            return [=](int nt) VDN_INL { return P::load_tile(Ssrc + layer * PS, p, 256, nt, h); };
        };
        auto mulInto = [&](auto& D, int layer) VDN_INL {     // D <- v_layer = u (.) s_layer; optionally kept for the backward
            return [&D, layer, Vs, PS, p, ok, h, derive](int nt, const f32x16& acc, const f32x16& sv) VDN_INL {
                f32x16 v;
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = acc[t] * sprime(sv[t], derive);
                D.set(nt, v);
                if (Vs != nullptr) P::store_tile(Vs + layer * PS, p, 256, nt, h, v, ok);
            };
        };
        // d/dx through the positional encoding (transpose Jacobian), accumulated into n[]
        float n[3] = {0.0f, 0.0f, 0.0f};
        auto pe_backward = [&](const f32x16 (&U2)[2], bool first) VDN_INL {
            float u[39];
            tiles_vals<39, 2>(U2, h, u);
            if (a.U_pe != nullptr && ok && h == 0) {      // u = u_4[PE part] + u_0 for the ray adjoint (VdnSdfArgs.U_pe)
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
                    sincos_pe<P::kAccurateTrig>(xin[d] * f, sn, co);
                    n[d] += f * (co * u[3 + 6 * k + d] - sn * u[3 + 6 * k + 3 + d]);
                }
            }
        };
        dense<P, 8, 8, false>(ws, Y, 0, loadS(6), mulInto(X, 6), est_v, P::kTileOps);   // through W7^T
        dense<P, 8, 8, false>(ws, X, 0, loadS(5), mulInto(Y, 5), est_v, P::kTileOps);   // W6^T
        dense<P, 8, 8, false>(ws, Y, 0, loadS(4), mulInto(X, 4), est_v, P::kTileOps);   // W5^T
        {   // W4^T: 9 output tiles = [h4 part (7 tiles) | PE part (2 tiles)]
            f32x16 UPE[2];
            dense<P, 8, 9, false>(ws, X, 0,
                [&](int nt) VDN_INL { return nt < 7 ? P::load_tile(Ssrc + 3 * PS, p, 256, nt, h) : f32x16{}; },
                [&](int nt, const f32x16& acc, const f32x16& sv) VDN_INL {
                    if (nt < 7) {
                        f32x16 v;
#pragma unroll
                        for (int t = 0; t < 16; ++t) v[t] = acc[t] * sprime(sv[t], derive);
                        Y.set(nt, v);
                        if (Vs != nullptr) P::store_tile(Vs + 3 * PS, p, 256, nt, h, v, ok);
                    } else {
                        UPE[nt - 7] = acc;
                    }
                });
            pe_backward(UPE, true);
        }
        dense<P, 7, 8, false>(ws, Y, 0, loadS(2), mulInto(X, 2), est_v, P::kTileOps);   // W3^T
        dense<P, 8, 8, false>(ws, X, 0, loadS(1), mulInto(Y, 1), est_v, P::kTileOps);   // W2^T
        dense<P, 8, 8, false>(ws, Y, 0, loadS(0), mulInto(X, 0), est_v, P::kTileOps);   // W1^T
        f32x16 U0[2];
        dense<P, 8, 2, false>(ws, X, 0, NoPre{}, [&](int nt, const f32x16& acc, int) VDN_INL { U0[nt] = acc; });   // W0^T
        pe_backward(U0, false);
        if (ok && h == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) a.normals[pd * 3 + d] = n[d] * a.scale;
        }
    }
}

template <class P, int MODE, int NW, bool DERIVE = false>
__global__ __launch_bounds__(NW * 64, P::kMinWavesPerEU) void sdf_fwd_kernel(SdfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    sdf_fwd_body<P, MODE, NW, DERIVE>(a, smem);
}

template <class P, int MODE, int NW, bool DERIVE = false>
void launch_sdf_nw(const VdnSdfArgs* args, hipStream_t stream) {
    const size_t lds = VDN_NSLOT * P::stride(9);
    static bool once = (allow_big_lds(sdf_fwd_kernel<P, MODE, NW, DERIVE>, lds), true);
    (void)once;
    const int grid = (args->P + NW * 32 - 1) / (NW * 32);
    hipLaunchKernelGGL((sdf_fwd_kernel<P, MODE, NW, DERIVE>), dim3(grid), dim3(NW * 64), lds, stream, *args);
}

template <class P>
int launch_sdf_fwd(int mode, const VdnSdfArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || args->P <= 0 || args->blob == nullptr) return -1;
    if (args->pts == nullptr && (args->rays_o == nullptr || args->rays_d == nullptr || args->z == nullptr || args->n_per_ray <= 0 ||
                                 args->z_ld < args->n_per_ray || args->sdf_ld < args->n_per_ray)) return -2;
    if (mode == 0) {
        if (args->sdf == nullptr) return -3;
        // Measured (profiles/README.md): 4 waves per workgroup share each weight chunk's DMA and beat
        // 1- or 2-wave workgroups even when that leaves the grid below one workgroup per CU.
        launch_sdf_nw<P, 0, 4>(args, stream);
    } else if (mode == 1) {
        if (!args->sdf || !args->feat || !args->normals || !args->S || !args->w8row) return -3;
        if constexpr (P::kDeriveS) {
            if (args->H != nullptr) launch_sdf_nw<P, 1, P::kWaves, true>(args, stream);
            else launch_sdf_nw<P, 1, P::kWaves, false>(args, stream);
        } else {
            launch_sdf_nw<P, 1, P::kWaves, false>(args, stream);
        }
    } else {
        return -4;
    }
    return (int)hipGetLastError();
}

}  // namespace vdn
