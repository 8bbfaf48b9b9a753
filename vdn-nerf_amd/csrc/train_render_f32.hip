// Backward of RenderingNetwork (colour head / VDN head) on gfx950, fp32.
// delta chain: delta_4 = g_out * act'(out); delta_{l-1} = (W_l^T delta_l) * [h_l > 0]; the last
// transposed layer yields d loss / d [feature | points, PE(view), normals]. Per-layer deltas go to
// HBM row-major for the weight-gradient GEMM (dw_gemm_f32.hip). Adjoint of fields.py:148-176.
#include "mlp_engine_f32.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kRbWaves = 4;
constexpr int kRbSlot = chunk_bytes_f32(8);
using RbStream = WStream<kRbWaves, kRbSlot>;

// epilogue: D = acc * [saved activation > 0]; keep in registers and store row-major
struct MaskStore {
    float* Y;
    float* dst;     // [P,ld]
    int ld;
    long row;
    bool ok;
    int h;
    VDN_DEV void operator()(int nt, const f32x16& acc, const f32x16& hv) const {
        f32x16 o;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            o[t] = hv[t] > 0.0f ? acc[t] : 0.0f;
            Y[nt * 16 + t] = o[t];
        }
        store_tile_rowmajor(dst, row, ld, nt, h, o, ok);
    }
};

template <int NT_OUT>
__global__ __launch_bounds__(kRbWaves * 64, 1) void rendernet_bwd_f32_kernel(RenderNetBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RbStream ws;
    ws.init(a.blob, smem);
    const int lane = ws.lane, c = lane & 31, h = lane >> 5;
    const long p_raw = ((long)blockIdx.x * kRbWaves + ws.wave) * 32 + c;
    const bool ok = p_raw < a.P;
    const long p = ok ? p_raw : (long)a.P - 1;
    const long PS = (long)a.P * 256;

    float X[128], Y[128];
    if constexpr (NT_OUT == 1) {
        float dl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dl[j] = 0.0f;
            if (j < a.d_out) {
                const float o = a.out[p * a.d_out + j], g = a.g_out[p * a.d_out + j];
                dl[j] = a.squeeze_out ? g * o * (1.0f - o) : (o > 0.0f ? g : 0.0f);
            }
        }
        vals_to_tiles<4, 1>(dl, h, X);
        f32x16 t16;
#pragma unroll
        for (int t = 0; t < 16; ++t) t16[t] = X[t];
        store_tile_rowmajor(a.delta_out, p, 32, 0, h, t16, ok);
    } else {
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const f32x16 o = load_tile_rowmajor_v(a.out, p, 96, kt, h);
            const f32x16 g = load_tile_rowmajor_v(a.g_out, p, 96, kt, h);
            f32x16 dl;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                dl[t] = a.squeeze_out ? g[t] * o[t] * (1.0f - o[t]) : (o[t] > 0.0f ? g[t] : 0.0f);
                X[kt * 16 + t] = dl[t];
            }
            store_tile_rowmajor(a.delta_out, p, 96, kt, h, dl, ok);
        }
    }
    constexpr int C8 = chunk_bytes_f32(8), CO = chunk_bytes_f32(NT_OUT);
    auto ldH = [&](int l) { return [=](int nt) { return load_tile_rowmajor_v(a.save_h + l * PS, p, 256, nt, h); }; };
    ws.start<CO>();
    dense_f32<NT_OUT, 8, C8, false>(ws, X, ldH(3), MaskStore{Y, a.delta_h + 3 * PS, 256, p, ok, h});   // W4^T
    dense_f32<8, 8, C8, false>(ws, Y, ldH(2), MaskStore{X, a.delta_h + 2 * PS, 256, p, ok, h});         // W3^T
    dense_f32<8, 8, C8, false>(ws, X, ldH(1), MaskStore{Y, a.delta_h + 1 * PS, 256, p, ok, h});         // W2^T
    dense_f32<8, 8, C8, false>(ws, Y, ldH(0), MaskStore{X, a.delta_h + 0 * PS, 256, p, ok, h});         // W1^T
    float SM[32];
    dense_f32<8, 10, 0, false>(ws, X, NoPre{}, [&](int nt, const f32x16& acc, int) {                     // W0^T
        if (nt < 8) {
            f32x16 o = acc;
            if (a.acc_feat) {
                const f32x16 prev = load_tile_rowmajor_v(a.d_feat, p, 256, nt, h);
#pragma unroll
                for (int t = 0; t < 16; ++t) o[t] += prev[t];
            }
            store_tile_rowmajor(a.d_feat, p, 256, nt, h, o, ok);
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) SM[(nt - 8) * 16 + t] = acc[t];
        }
    });
    float small[33];
    tiles_to_vals<33, 2>(SM, h, small);        // [points(3), PE(view)(27), normals(3)]
    if (ok && h == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float prev = a.acc_normals ? a.d_normals[p * 3 + d] : 0.0f;
            a.d_normals[p * 3 + d] = prev + small[30 + d];
        }
    }
}

}  // namespace vdn

extern "C" int vdn_rendernet_bwd_f32(const VdnRenderNetBwdArgs* args, void* stream_) {
    using namespace vdn;
    hipStream_t stream = (hipStream_t)stream_;
    if (!args || args->P <= 0 || !args->blob || !args->g_out || !args->out || !args->save_h || !args->delta_out ||
        !args->delta_h || !args->d_feat || !args->d_normals) return -1;
    if (!(args->d_out == 96 || (args->d_out >= 1 && args->d_out <= 4))) return -2;
    const int grid = (args->P + kRbWaves * 32 - 1) / (kRbWaves * 32);
    const size_t lds = 2 * kRbSlot;
    static bool once = (allow_big_lds(rendernet_bwd_f32_kernel<1>, 2 * kRbSlot), allow_big_lds(rendernet_bwd_f32_kernel<3>, 2 * kRbSlot), true);
    (void)once;
    if (args->d_out == 96)
        hipLaunchKernelGGL(rendernet_bwd_f32_kernel<3>, dim3(grid), dim3(kRbWaves * 64), lds, stream, *args);
    else
        hipLaunchKernelGGL(rendernet_bwd_f32_kernel<1>, dim3(grid), dim3(kRbWaves * 64), lds, stream, *args);
    return (int)hipGetLastError();
}
