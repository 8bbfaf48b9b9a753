// Backward of the SDF network including the double backward through d sdf/d x, on gfx950, fp32.
//
// Forward quantities (sdf_f32.hip):  a_l = W_l x_l + b_l,  h_{l+1} = softplus(a_l),  s_l = softplus'(a_l),
// sweep  v_l = u_{l+1} * s_l,  u_l = W_l^T v_l,  normal = scale * J_PE^T (u_0 + u_4[PE part]).
// Adjoint, given (g_sdf, g_feat, g_normal):
//   rbar (ascending l):  ub_0 = ub_4[PE] = scale * J_PE g_normal;   vb_l = W_l ub_l;
//                        ub_{l+1} = vb_l * s_l;   ex_l = 100 * vb_l * v_l * (1 - s_l)      [softplus'' = 100 s (1-s)]
//                        weight grads:  dW_l += v_l ub_l^T  (dw_gemm), dW_8[sdf row] += colsum(ub_8) / scale
//   fbar (descending l): ab_8 = [g_feat | g_sdf/scale];  hb_l = W_l^T ab_l;  ab_{l-1} = hb_l * s_{l-1} + ex_{l-1}
//                        weight grads:  dW_l += ab_l x_l^T,  db_l = colsum(ab_l)
// This is the hand-derived form of what autograd builds for reference fields.py:97-108 with
// create_graph=True and then differentiates in dpt_runner.py:253.
#include "mlp_engine_f32.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kSbWaves = 4;
constexpr int kSbSlot = chunk_bytes_f32(9);
using SbStream = WStream<kSbWaves, kSbSlot>;

__global__ __launch_bounds__(kSbWaves * 64, 1) void sdf_rbar_f32_kernel(SdfRbarArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SbStream ws;
    ws.init(a.blob, smem);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const long p_raw = ((long)blockIdx.x * kSbWaves + ws.wave) * 32 + c;
    const bool ok = p_raw < a.P;
    const long p = ok ? p_raw : (long)a.P - 1;
    const long P = a.P, PS = P * 256;

    float xin[3];
    if (a.pts != nullptr) {
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = a.pts[p * 3 + d] * a.scale;
    } else {
        const long r = p / a.n_per_ray;
        const float z = a.z[r * a.z_ld + (p - r * a.n_per_ray)];
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = (a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z) * a.scale;
    }
    // ub_total = scale * J_PE g_n  (39 values): adjoint of n = scale * J^T u
    float ub39[39];
    {
        float gn[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            gn[d] = a.g_normals[p * 3 + d] * a.scale;
            ub39[d] = gn[d];
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float f = (float)(1 << k);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                float sn, co;
                sincosf(xin[d] * f, &sn, &co);
                ub39[3 + 6 * k + d] = f * co * gn[d];
                ub39[3 + 6 * k + 3 + d] = -f * sn * gn[d];
            }
        }
    }
    float* ub0 = a.UB;
    float* ub1 = ub0 + P * 64;
    float* ub2 = ub1 + PS;
    float* ub3 = ub2 + PS;
    float* ub4 = ub3 + PS;
    float* ub5 = ub4 + P * 288;
    float* ub6 = ub5 + PS;
    float* ub7 = ub6 + PS;
    float* ub8 = ub7 + PS;

    float X[144], Y[128];
    vals_to_tiles<39, 2>(ub39, h, X);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        f32x16 t16;
#pragma unroll
        for (int t = 0; t < 16; ++t) t16[t] = X[kt * 16 + t];
        store_tile_rowmajor(ub0, p, 64, kt, h, t16, ok);
    }
    struct SV { f32x16 s, v; };
    auto ldSV = [&](int l) {
        return [=](int nt) {
            SV r;
            r.s = load_tile_rowmajor_v(a.S + l * PS, p, 256, nt, h);
            r.v = load_tile_rowmajor_v(a.V + l * PS, p, 256, nt, h);
            return r;
        };
    };
    // epilogue of layer l: ub_{l+1} = vb * s_l -> D (registers) and dst (HBM); ex_l -> EX[l]
    auto epi = [&](float* D, float* dst, int ld, int l) {
        return [=](int nt, const f32x16& acc, const SV& sv) {
            f32x16 ub, ex;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                ub[t] = acc[t] * sv.s[t];
                ex[t] = 100.0f * acc[t] * sv.v[t] * (1.0f - sv.s[t]);
                D[nt * 16 + t] = ub[t];
            }
            store_tile_rowmajor(dst, p, ld, nt, h, ub, ok);
            store_tile_rowmajor(a.EX + l * PS, p, 256, nt, h, ex, ok);
        };
    };
    constexpr int C2 = chunk_bytes_f32(2), C8 = chunk_bytes_f32(8), C9 = chunk_bytes_f32(9);
    ws.start<C2>();
    dense_f32<2, 8, C8, false>(ws, X, ldSV(0), epi(Y, ub1, 256, 0));
    dense_f32<8, 8, C8, false>(ws, Y, ldSV(1), epi(X, ub2, 256, 1));
    dense_f32<8, 8, C8, false>(ws, X, ldSV(2), epi(Y, ub3, 256, 2));
    dense_f32<8, 7, C9, false>(ws, Y, ldSV(3), epi(X, ub4, 288, 3));      // ub_4[h part]: 7 tiles
    vals_to_tiles<39, 2>(ub39, h, X + 112);                               // ub_4[PE part] = ub_total
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        f32x16 t16;
#pragma unroll
        for (int t = 0; t < 16; ++t) t16[t] = X[112 + kt * 16 + t];
        store_tile_rowmajor(ub4, p, 288, 7 + kt, h, t16, ok);
    }
    dense_f32<9, 8, C8, false>(ws, X, ldSV(4), epi(Y, ub5, 256, 4));
    dense_f32<8, 8, C8, false>(ws, Y, ldSV(5), epi(X, ub6, 256, 5));
    dense_f32<8, 8, C8, false>(ws, X, ldSV(6), epi(Y, ub7, 256, 6));
    dense_f32<8, 8, 0, false>(ws, Y, ldSV(7), epi(X, ub8, 256, 7));
}

__global__ __launch_bounds__(kSbWaves * 64, 1) void sdf_fbar_f32_kernel(SdfFbarArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    SbStream ws;
    ws.init(a.blob, smem);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const long p_raw = ((long)blockIdx.x * kSbWaves + ws.wave) * 32 + c;
    const bool ok = p_raw < a.P;
    const long p = ok ? p_raw : (long)a.P - 1;
    const long P = a.P, PS = P * 256;
    float* ab8 = a.AB;
    auto ab = [&](int l) { return ab8 + P * 288 + (long)(7 - l) * PS; };   // l = 7..0

    float X[144], Y[144];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) load_tile_rowmajor(a.g_feat, p, 256, kt, h, X + kt * 16);
    {
        float g1[1] = {a.g_sdf[p] / a.scale};
        vals_to_tiles<1, 1>(g1, h, X + 128);
    }
#pragma unroll
    for (int kt = 0; kt < 9; ++kt) {
        f32x16 t16;
#pragma unroll
        for (int t = 0; t < 16; ++t) t16[t] = X[kt * 16 + t];
        store_tile_rowmajor(ab8, p, 288, kt, h, t16, ok);
    }
    struct SE { f32x16 s, e; };
    auto ldSE = [&](int l) {
        return [=](int nt) {
            SE r;
            r.s = load_tile_rowmajor_v(a.S + l * PS, p, 256, nt, h);
            r.e = load_tile_rowmajor_v(a.EX + l * PS, p, 256, nt, h);
            return r;
        };
    };
    auto epi = [&](float* D, int l) {       // ab_l = hb_{l+1} * s_l + ex_l
        float* dst = ab(l);
        return [=](int nt, const f32x16& acc, const SE& se) {
            f32x16 o;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                o[t] = acc[t] * se.s[t] + se.e[t];
                D[nt * 16 + t] = o[t];
            }
            store_tile_rowmajor(dst, p, 256, nt, h, o, ok);
        };
    };
    constexpr int C7 = chunk_bytes_f32(7), C8 = chunk_bytes_f32(8), C9 = chunk_bytes_f32(9);
    ws.start<C9>();
    dense_f32<9, 8, C8, false>(ws, X, ldSE(7), epi(Y, 7));     // W8^T
    dense_f32<8, 8, C8, false>(ws, Y, ldSE(6), epi(X, 6));     // W7^T
    dense_f32<8, 8, C8, false>(ws, X, ldSE(5), epi(Y, 5));     // W6^T
    dense_f32<8, 8, C8, false>(ws, Y, ldSE(4), epi(X, 4));     // W5^T
    {   // W4^T: 9 output tiles = [h4 part (7) | PE part (2, no gradient wanted)]
        float* dst = ab(3);
        dense_f32<8, 9, C7, false>(ws, X,
            [&](int nt) {
                SE r;
                if (nt < 7) {
                    r.s = load_tile_rowmajor_v(a.S + 3 * PS, p, 256, nt, h);
                    r.e = load_tile_rowmajor_v(a.EX + 3 * PS, p, 256, nt, h);
                } else {
                    r.s = f32x16{};
                    r.e = f32x16{};
                }
                return r;
            },
            [&](int nt, const f32x16& acc, const SE& se) {
                if (nt < 7) {
                    f32x16 o;
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        o[t] = acc[t] * se.s[t] + se.e[t];
                        Y[nt * 16 + t] = o[t];
                    }
                    store_tile_rowmajor(dst, p, 256, nt, h, o, ok);
                }
            });
    }
    dense_f32<7, 8, C8, false>(ws, Y, ldSE(2), epi(X, 2));     // W3^T
    dense_f32<8, 8, C8, false>(ws, X, ldSE(1), epi(Y, 1));     // W2^T
    dense_f32<8, 8, 0, false>(ws, Y, ldSE(0), epi(X, 0));      // W1^T
}

}  // namespace vdn

extern "C" int vdn_sdf_bwd_rbar_f32(const VdnSdfRbarArgs* args, void* stream_) {
    using namespace vdn;
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->P <= 0 || !args->blob || !args->g_normals || !args->S || !args->V || !args->UB || !args->EX) return -1;
    if (!args->pts && (!args->rays_o || !args->rays_d || !args->z || args->n_per_ray <= 0 || args->z_ld < args->n_per_ray)) return -2;
    const int grid = (args->P + kSbWaves * 32 - 1) / (kSbWaves * 32);
    static bool once = (allow_big_lds(sdf_rbar_f32_kernel, 2 * kSbSlot), allow_big_lds(sdf_fbar_f32_kernel, 2 * kSbSlot), true);
    (void)once;
    hipLaunchKernelGGL(sdf_rbar_f32_kernel, dim3(grid), dim3(kSbWaves * 64), 2 * kSbSlot, stream, *args);
    return (int)hipGetLastError();
}

extern "C" int vdn_sdf_bwd_fbar_f32(const VdnSdfFbarArgs* args, void* stream_) {
    using namespace vdn;
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->P <= 0 || !args->blob || !args->g_sdf || !args->g_feat || !args->S || !args->EX || !args->AB) return -1;
    const int grid = (args->P + kSbWaves * 32 - 1) / (kSbWaves * 32);
    static bool once = (allow_big_lds(sdf_rbar_f32_kernel, 2 * kSbSlot), allow_big_lds(sdf_fbar_f32_kernel, 2 * kSbSlot), true);
    (void)once;
    hipLaunchKernelGGL(sdf_fbar_f32_kernel, dim3(grid), dim3(kSbWaves * 64), 2 * kSbSlot, stream, *args);
    return (int)hipGetLastError();
}
