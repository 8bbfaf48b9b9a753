// Background NeRF++ MLP forward on gfx950, exact fp32.
// Fuses the inverted-sphere parameterisation of renderer.py:112-115 (pts4 = [pts/r, 1/r],
// r = max(|pts|,1)), PE10(pts4) / PE4(view), the 8x256 ReLU trunk with its skip after layer 4,
// and the alpha / feature / views / rgb (/ 96-ch dpt) heads. Replaces reference
// dpt_models/fields.py:324-353 as called from renderer.py:100-123.
#include "mlp_engine_f32.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kNfWaves = 4;
constexpr int kNfSlot = chunk_bytes_f32(11);
using NfStream = WStream<kNfWaves, kNfSlot>;

struct ReluIntoN {
    float* Y;
    float* save;    // row-major slice [P,ld] or nullptr
    int ld;
    long row;
    bool ok;
    int h;
    VDN_DEV void operator()(int nt, const f32x16& acc, int) const {
        f32x16 o;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            o[t] = fmaxf(acc[t], 0.0f);
            Y[nt * 16 + t] = o[t];
        }
        if (save != nullptr) store_tile_rowmajor(save, row, ld, nt, h, o, ok);
    }
};
template <int NTILES>
VDN_DEV void save_tiles(float* dst, int ld, const float* X, long row, int h, bool ok) {
    if (dst == nullptr) return;
#pragma unroll
    for (int kt = 0; kt < NTILES; ++kt) {
        f32x16 t16;
#pragma unroll
        for (int t = 0; t < 16; ++t) t16[t] = X[kt * 16 + t];
        store_tile_rowmajor(dst, row, ld, kt, h, t16, ok);
    }
}

template <bool DPT>
__global__ __launch_bounds__(kNfWaves * 64, 1) void nerf_f32_kernel(NerfArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    NfStream ws;
    ws.init(a.blob, smem);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const long p_raw = ((long)blockIdx.x * kNfWaves + ws.wave) * 32 + c;
    const bool ok = p_raw < a.P;
    const long p = ok ? p_raw : (long)a.P - 1;
    const long r = p / a.n_per_ray;

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
    float X[176], Y[144];
    auto put_pe = [&](float* dst) {
        float pe[84];
        posenc<4, 10>(p4, pe);
        vals_to_tiles<84, 3>(pe, h, dst);
    };
    constexpr int C3 = chunk_bytes_f32(3), C8 = chunk_bytes_f32(8), C9 = chunk_bytes_f32(9), C11 = chunk_bytes_f32(11),
                  C4 = chunk_bytes_f32(4);
    put_pe(X);
    save_tiles<3>(a.save_pe, 96, X, p, h, ok);
    const long PS = (long)a.P * 256;
    auto sv = [&](int l) { return a.save_h ? a.save_h + l * PS : nullptr; };
    ws.start<C3>();
    dense_f32<3, 8, C8, true>(ws, X, NoPre{}, ReluIntoN{Y, sv(0), 256, p, ok, h});          // pts_linears.0
    dense_f32<8, 8, C8, true>(ws, Y, NoPre{}, ReluIntoN{X, sv(1), 256, p, ok, h});          // 1
    dense_f32<8, 8, C8, true>(ws, X, NoPre{}, ReluIntoN{Y, sv(2), 256, p, ok, h});          // 2
    dense_f32<8, 8, C8, true>(ws, Y, NoPre{}, ReluIntoN{X, sv(3), 256, p, ok, h});          // 3
    dense_f32<8, 8, C11, true>(ws, X, NoPre{}, ReluIntoN{Y, sv(4), 256, p, ok, h});         // 4
    // skip (fields.py:334-335): h = cat([input_pts, h]) -> X = [PE (3 tiles) | h (8 tiles)]
#pragma unroll
    for (int i = 0; i < 128; ++i) X[48 + i] = Y[i];
    put_pe(X);
    dense_f32<11, 8, C8, true>(ws, X, NoPre{}, ReluIntoN{Y, sv(5), 256, p, ok, h});         // 5
    dense_f32<8, 8, C8, true>(ws, Y, NoPre{}, ReluIntoN{X, sv(6), 256, p, ok, h});          // 6
    dense_f32<8, 8, C8, true>(ws, X, NoPre{}, ReluIntoN{Y, sv(7), 256, p, ok, h});          // 7
    // heads on h: image rows 0..255 feature_linear, row 256 alpha_linear
    dense_f32<8, 9, C9, true>(ws, Y, NoPre{}, [&](int nt, const f32x16& acc, int) {
        if (nt < 8) {
#pragma unroll
            for (int t = 0; t < 16; ++t) X[nt * 16 + t] = acc[t];
            if (a.save_feature != nullptr) store_tile_rowmajor(a.save_feature, p, 256, nt, h, acc, ok);
        } else {
            if (ok && h == 0) a.density[p] = acc[0];
        }
    });
    {   // views_linears.0 on cat([feature, PE4(view)])  (fields.py:340-344)
        float pe[27];
        posenc<3, 4>(dir, pe);
        vals_to_tiles<27, 1>(pe, h, X + 128);
        save_tiles<1>(a.save_vpe, 32, X + 128, p, h, ok);
    }
    dense_f32<9, 4, C4, true>(ws, X, NoPre{}, ReluIntoN{Y, a.save_hv, 128, p, ok, h});
    // rgb_linear (image tile 0, rows 0..2) and dpt_linear (image tiles 1..3)
    dense_f32<4, DPT ? 4 : 1, 0, true>(ws, Y, NoPre{}, [&](int nt, const f32x16& acc, int) {
        if (nt == 0) {
            if (ok && h == 0) {
                a.rgb[p * 3 + 0] = acc[0];
                a.rgb[p * 3 + 1] = acc[1];
                a.rgb[p * 3 + 2] = acc[2];
            }
        } else {
            store_tile_rowmajor(a.feat, p, 96, nt - 1, h, acc, ok);
        }
    });
}

}  // namespace vdn

extern "C" int vdn_nerf_mlp_fwd_f32(const VdnNerfArgs* args, void* stream_) {
    using namespace vdn;
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || args->P <= 0 || !args->blob || !args->density || !args->rgb || args->n_per_ray <= 0) return -1;
    if (!args->pts4 && (!args->rays_o || !args->rays_d || !args->z)) return -1;
    if (!args->dirs && !args->rays_d) return -1;
    const int grid = (args->P + kNfWaves * 32 - 1) / (kNfWaves * 32);
    const size_t lds = 2 * kNfSlot;
    static bool once = (allow_big_lds(nerf_f32_kernel<false>, 2 * kNfSlot), allow_big_lds(nerf_f32_kernel<true>, 2 * kNfSlot), true);
    (void)once;
    if (args->feat != nullptr)
        hipLaunchKernelGGL(nerf_f32_kernel<true>, dim3(grid), dim3(kNfWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL(nerf_f32_kernel<false>, dim3(grid), dim3(kNfWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}
