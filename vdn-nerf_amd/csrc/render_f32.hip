// RenderingNetwork forward (colour head d_out=3, VDN feature head d_out=96) on gfx950, exact fp32.
// Input assembly [points(3), PE4(view_dirs)(27), normals(3), feature(256)] (mode 'idr'), 4 hidden
// ReLU layers of 256, sigmoid output. Replaces reference dpt_models/fields.py:148-176.
// K order inside the kernel is [feature(256) | points, PE(view), normals (33 -> 64)]; the weight
// image builder permutes the first layer's columns accordingly.
#include "mlp_engine_f32.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kRnWaves = 4;
constexpr int kRnSlot = chunk_bytes_f32(10);
using RnStream = WStream<kRnWaves, kRnSlot>;

struct ReluInto {
    float* Y;
    float* save;    // [P,256] slice or nullptr
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
        if (save != nullptr) store_tile_rowmajor(save, row, 256, nt, h, o, ok);
    }
};

template <int NT_OUT>   // 1: d_out <= 32 (colour); 3: d_out = 96 (VDN head)
__global__ __launch_bounds__(kRnWaves * 64, 1) void rendernet_f32_kernel(RenderNetArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RnStream ws;
    ws.init(a.blob, smem);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const long p_raw = ((long)blockIdx.x * kRnWaves + ws.wave) * 32 + c;
    const bool ok = p_raw < a.P;
    const long p = ok ? p_raw : (long)a.P - 1;
    const long r = p / a.n_per_ray;

    float X[160], Y[128];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) load_tile_rowmajor(a.feat, p, 256, kt, h, X + kt * 16);
    {
        float small[33];
        float dir[3];
        const float z = a.pts ? 0.0f : a.z[p];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            dir[d] = a.dirs ? a.dirs[p * 3 + d] : a.rays_d[r * 3 + d];
            small[d] = a.pts ? a.pts[p * 3 + d] : a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;   // renderer.py:233
            small[30 + d] = a.normals[p * 3 + d];
        }
        float pe[27];
        posenc<3, 4>(dir, pe);
#pragma unroll
        for (int i = 0; i < 27; ++i) small[3 + i] = pe[i];
        vals_to_tiles<33, 2>(small, h, X + 128);
        if (a.save_small != nullptr) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 t16;
#pragma unroll
                for (int t = 0; t < 16; ++t) t16[t] = X[128 + kt * 16 + t];
                store_tile_rowmajor(a.save_small, p, 64, kt, h, t16, ok);
            }
        }
    }
    const long PS = (long)a.P * 256;
    auto sv = [&](int l) { return a.save_h ? a.save_h + l * PS : nullptr; };
    constexpr int C10 = chunk_bytes_f32(10), C8 = chunk_bytes_f32(8);
    ws.start<C10>();
    dense_f32<10, 8, C8, true>(ws, X, NoPre{}, ReluInto{Y, sv(0), p, ok, h});
    dense_f32<8, 8, C8, true>(ws, Y, NoPre{}, ReluInto{X, sv(1), p, ok, h});
    dense_f32<8, 8, C8, true>(ws, X, NoPre{}, ReluInto{Y, sv(2), p, ok, h});
    dense_f32<8, 8, C8, true>(ws, Y, NoPre{}, ReluInto{X, sv(3), p, ok, h});
    dense_f32<8, NT_OUT, 0, true>(ws, X, NoPre{}, [&](int nt, const f32x16& acc, int) {
        f32x16 o;
#pragma unroll
        for (int t = 0; t < 16; ++t) o[t] = a.squeeze_out ? sigmoidf_(acc[t]) : fmaxf(acc[t], 0.0f);
        if constexpr (NT_OUT == 1) {
            if (ok && h == 0) {
                for (int j = 0; j < a.d_out && j < 4; ++j) a.out[p * a.d_out + j] = o[j];
            }
        } else {
            store_tile_rowmajor(a.out, p, 96, nt, h, o, ok);
        }
    });
}

}  // namespace vdn

extern "C" int vdn_rendernet_fwd_f32(const VdnRenderNetArgs* args, void* stream_) {
    using namespace vdn;
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || args->P <= 0 || !args->blob || !args->normals || !args->feat || !args->out || args->n_per_ray <= 0) return -1;
    if (!args->pts && (!args->rays_o || !args->rays_d || !args->z)) return -1;
    if (!args->dirs && !args->rays_d) return -1;
    if (!(args->d_out == 96 || (args->d_out >= 1 && args->d_out <= 4))) return -2;
    const int grid = (args->P + kRnWaves * 32 - 1) / (kRnWaves * 32);
    const size_t lds = 2 * kRnSlot;
    static bool once = (allow_big_lds(rendernet_f32_kernel<1>, 2 * kRnSlot), allow_big_lds(rendernet_f32_kernel<3>, 2 * kRnSlot), true);
    (void)once;
    if (args->d_out == 96)
        hipLaunchKernelGGL(rendernet_f32_kernel<3>, dim3(grid), dim3(kRnWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL(rendernet_f32_kernel<1>, dim3(grid), dim3(kRnWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}
