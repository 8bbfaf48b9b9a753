// SDF network forward on gfx950, exact fp32 (parity path).
// Fuses: point generation (o + d*z) -> positional encoding -> 9 weight-normed layers with
// Softplus(beta=100) and the skip at layer 4 -> [sdf | 256-d feature], and (FULL mode) the analytic
// reverse sweep that yields d sdf / d x, all with activations resident in registers.
// Replaces reference dpt_models/fields.py:72-108 (SDFNetwork.forward / .sdf / .gradient).
#include "mlp_engine_f32.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kSdfWaves = 4;
constexpr int kSdfSlot = chunk_bytes_f32(9);
using SdfStream = WStream<kSdfWaves, kSdfSlot>;

// hidden layer epilogue: Y = softplus100(acc); optionally save s = softplus' row-major for the sweep
template <bool SAVE>
struct HiddenEpi {
    float* Y;
    float* S;       // [P,256] slice of this layer
    long row;
    bool ok;
    int h;
    float* Hs;      // [P,256] slice to keep the activation itself (training) or nullptr
    VDN_DEV void operator()(int nt, const f32x16& acc, int) const {
        f32x16 s;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if constexpr (SAVE) {
                float hv, sv;
                softplus100_both(acc[t], hv, sv);
                Y[nt * 16 + t] = hv;
                s[t] = sv;
            } else {
                Y[nt * 16 + t] = softplus100_fast(acc[t]);
            }
        }
        if constexpr (SAVE) {
            store_tile_rowmajor(S, row, 256, nt, h, s, ok);
            if (Hs != nullptr) {
                f32x16 hv;
#pragma unroll
                for (int t = 0; t < 16; ++t) hv[t] = Y[nt * 16 + t];
                store_tile_rowmajor(Hs, row, 256, nt, h, hv, ok);
            }
        }
    }
};

template <int MODE>   // 0: sdf only; 1: sdf + feature + normals
__global__ __launch_bounds__(kSdfWaves * 64, 1) void sdf_f32_kernel(SdfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SdfStream ws;
    ws.init(a.blob, smem);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const long p_raw = ((long)blockIdx.x * kSdfWaves + ws.wave) * 32 + c;
    const bool ok = p_raw < a.P;
    const long p = ok ? p_raw : (long)a.P - 1;

    float xin[3];
    long sdf_idx = p;
    if (a.pts != nullptr) {
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = a.pts[p * 3 + d] * a.scale;
    } else {
        const long r = p / a.n_per_ray;
        const long sidx = p - r * a.n_per_ray;
        const float z = a.z[r * a.z_ld + sidx];
        sdf_idx = r * a.sdf_ld + sidx;
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = (a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z) * a.scale;
    }

    float X[144], Y[144];
    {
        float pe[39];
        posenc<3, 6>(xin, pe);
        vals_to_tiles<39, 2>(pe, h, X);
        if constexpr (MODE == 1) {
            if (a.PE != nullptr) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    f32x16 t16;
#pragma unroll
                    for (int t = 0; t < 16; ++t) t16[t] = X[kt * 16 + t];
                    store_tile_rowmajor(a.PE, p, 64, kt, h, t16, ok);
                }
            }
        }
    }
    constexpr int C2 = chunk_bytes_f32(2), C7 = chunk_bytes_f32(7), C8 = chunk_bytes_f32(8), C9 = chunk_bytes_f32(9);
    constexpr bool SV = (MODE == 1);
    float* S = a.S;
    const long PS = (long)a.P * 256;
    ws.start<C2>();
    dense_f32<2, 8, C8, true>(ws, X, NoPre{}, HiddenEpi<SV>{Y, S + 0 * PS, p, ok, h, (SV && a.H) ? a.H + 0 * PS : nullptr});
    dense_f32<8, 8, C8, true>(ws, Y, NoPre{}, HiddenEpi<SV>{X, S + 1 * PS, p, ok, h, (SV && a.H) ? a.H + 1 * PS : nullptr});
    dense_f32<8, 8, C8, true>(ws, X, NoPre{}, HiddenEpi<SV>{Y, S + 2 * PS, p, ok, h, (SV && a.H) ? a.H + 2 * PS : nullptr});
    dense_f32<8, 7, C9, true>(ws, Y, NoPre{}, HiddenEpi<SV>{X, S + 3 * PS, p, ok, h, (SV && a.H) ? a.H + 3 * PS : nullptr});
    {   // skip: layer-4 input = [h4 (217, padded to 7 tiles) | PE (39, 2 tiles)] / sqrt2 (1/sqrt2 is in the image)
        float pe[39];
        posenc<3, 6>(xin, pe);
        vals_to_tiles<39, 2>(pe, h, X + 112);
    }
    dense_f32<9, 8, C8, true>(ws, X, NoPre{}, HiddenEpi<SV>{Y, S + 4 * PS, p, ok, h, (SV && a.H) ? a.H + 4 * PS : nullptr});
    dense_f32<8, 8, C8, true>(ws, Y, NoPre{}, HiddenEpi<SV>{X, S + 5 * PS, p, ok, h, (SV && a.H) ? a.H + 5 * PS : nullptr});
    dense_f32<8, 8, C8, true>(ws, X, NoPre{}, HiddenEpi<SV>{Y, S + 6 * PS, p, ok, h, (SV && a.H) ? a.H + 6 * PS : nullptr});
    dense_f32<8, 8, C8, true>(ws, Y, NoPre{}, HiddenEpi<SV>{X, S + 7 * PS, p, ok, h, (SV && a.H) ? a.H + 7 * PS : nullptr});

    const float inv_scale = 1.0f / a.scale;
    if constexpr (MODE == 0) {
        // last layer reduced to its sdf row (image row 0 = W8 row 0)
        dense_f32<8, 1, 0, true>(ws, X, NoPre{}, [&](int, const f32x16& acc, int) {
            if (ok && h == 0) a.sdf[sdf_idx] = acc[0] * inv_scale;
        });
        return;
    } else {
        // image rows: 0..255 = feature rows (W8 rows 1..256), 256 = sdf row (W8 row 0)
        dense_f32<8, 9, C8, true>(ws, X, NoPre{}, [&](int nt, const f32x16& acc, int) {
            if (nt < 8) {
                store_tile_rowmajor(a.feat, p, 256, nt, h, acc, ok);
            } else {
                if (ok && h == 0) a.sdf[sdf_idx] = acc[0] * inv_scale;
            }
        });

        // ---- reverse sweep: u = d sdf / d(layer input), v = u (.) softplus'(a_l) ------------------
        // Every sweep layer's epilogue multiplies its output tile by the S tile of the layer below
        // (loaded right after the chunk acquire), so X/Y always hold v, ready to be the B operand.
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            const f32x16 w8 = load_tile_rowmajor_v(a.w8row, 0, 0, kt, h);
            const f32x16 s7 = load_tile_rowmajor_v(S + 7 * PS, p, 256, kt, h);
            f32x16 v7;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                v7[t] = w8[t] * inv_scale * s7[t];
                Y[kt * 16 + t] = v7[t];
            }
            if (a.V != nullptr) store_tile_rowmajor(a.V + 7 * PS, p, 256, kt, h, v7, ok);
        }
        auto loadS = [&](int layer) {
            return [=](int nt) { return load_tile_rowmajor_v(S + layer * PS, p, 256, nt, h); };
        };
        auto mulInto = [&](float* D, int layer) {     // D <- v_layer = u (.) s_layer; optionally kept for the backward
            return [=](int nt, const f32x16& acc, const f32x16& sv) {
                f32x16 v;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    v[t] = acc[t] * sv[t];
                    D[nt * 16 + t] = v[t];
                }
                if (a.V != nullptr) store_tile_rowmajor(a.V + layer * PS, p, 256, nt, h, v, ok);
            };
        };
        // d/dx through the positional encoding (transpose Jacobian), accumulated into n[]
        float n[3] = {0.0f, 0.0f, 0.0f};
        auto pe_backward = [&](const float* U2) {
            float u[39];
            tiles_to_vals<39, 2>(U2, h, u);
#pragma unroll
            for (int d = 0; d < 3; ++d) n[d] += u[d];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float f = (float)(1 << k);
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    float sn, co;
                    sincosf(xin[d] * f, &sn, &co);
                    n[d] += f * (co * u[3 + 6 * k + d] - sn * u[3 + 6 * k + 3 + d]);
                }
            }
        };
        dense_f32<8, 8, C8, false>(ws, Y, loadS(6), mulInto(X, 6));   // through W7^T
        dense_f32<8, 8, C8, false>(ws, X, loadS(5), mulInto(Y, 5));   // W6^T
        dense_f32<8, 8, C8, false>(ws, Y, loadS(4), mulInto(X, 4));   // W5^T
        {   // W4^T: 9 output tiles = [h4 part (7 tiles) | PE part (2 tiles)]
            float UPE[32];
            dense_f32<8, 9, C7, false>(ws, X,
                [&](int nt) { return nt < 7 ? load_tile_rowmajor_v(S + 3 * PS, p, 256, nt, h) : f32x16{}; },
                [&](int nt, const f32x16& acc, const f32x16& sv) {
                    f32x16 v;
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        v[t] = acc[t] * sv[t];
                        if (nt < 7) Y[nt * 16 + t] = v[t];
                        else UPE[(nt - 7) * 16 + t] = acc[t];
                    }
                    if (nt < 7 && a.V != nullptr) store_tile_rowmajor(a.V + 3 * PS, p, 256, nt, h, v, ok);
                });
            pe_backward(UPE);
        }
        dense_f32<7, 8, C8, false>(ws, Y, loadS(2), mulInto(X, 2));   // W3^T
        dense_f32<8, 8, C8, false>(ws, X, loadS(1), mulInto(Y, 1));   // W2^T
        dense_f32<8, 8, C8, false>(ws, Y, loadS(0), mulInto(X, 0));   // W1^T
        dense_f32<8, 2, 0, false>(ws, X, NoPre{}, [&](int nt, const f32x16& acc, int) {   // W0^T
#pragma unroll
            for (int t = 0; t < 16; ++t) Y[nt * 16 + t] = acc[t];
        });
        pe_backward(Y);
        if (ok && h == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) a.normals[p * 3 + d] = n[d] * a.scale;
        }
    }
}

}  // namespace vdn

extern "C" int vdn_sdf_mlp_fwd_f32(int mode, const VdnSdfArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    using namespace vdn;
    if (args == nullptr || args->P <= 0 || args->blob == nullptr) return -1;
    if (args->pts == nullptr && (args->rays_o == nullptr || args->rays_d == nullptr || args->z == nullptr || args->n_per_ray <= 0 ||
                                 args->z_ld < args->n_per_ray || args->sdf_ld < args->n_per_ray)) return -2;
    const int grid = (args->P + kSdfWaves * 32 - 1) / (kSdfWaves * 32);
    const size_t lds = 2 * kSdfSlot;
    static bool once = (allow_big_lds(sdf_f32_kernel<0>, 2 * kSdfSlot), allow_big_lds(sdf_f32_kernel<1>, 2 * kSdfSlot), true);
    (void)once;
    if (mode == 0) {
        if (args->sdf == nullptr) return -3;
        hipLaunchKernelGGL(sdf_f32_kernel<0>, dim3(grid), dim3(kSdfWaves * 64), lds, stream, *args);
    } else if (mode == 1) {
        if (!args->sdf || !args->feat || !args->normals || !args->S || !args->w8row) return -3;
        hipLaunchKernelGGL(sdf_f32_kernel<1>, dim3(grid), dim3(kSdfWaves * 64), lds, stream, *args);
    } else {
        return -4;
    }
    return (int)hipGetLastError();
}
