// Background NeRF++ MLP forward on gfx950, bf16 path on the flat-stream engine (mlp_flow.h). Same fusion and the same
// results as k_nerf_fwd.h (which the fp32 path keeps): inverted-sphere parameterisation of renderer.py:112-115, PE10(pts4) /
// PE4(view), the 8x256 ReLU trunk with its skip after layer 4, and the alpha / feature / views / rgb (/ 96-ch dpt) heads.
// Replaces reference dpt_models/fields.py:324-353 as called from renderer.py:100-123.
#pragma once
#include "mlp_flow.h"
#include "vdn_kernels.h"

namespace vdn {

// chunk program of the 'fwd' stream (vdn_hip/images.py: nerf_streams): pts_linears.0 (3 k-tiles), .1-.4, .5 (11: skip), .6, .7,
// heads (feature 8 + alpha 1), views_linears.0 (9 k-tiles, 4 chunks), rgb (+ dpt) (4 k-tiles)
template <bool DPT, bool SAVE>
struct NerfFwdProg {
    static constexpr int total = 64 + 9 + 4 + (DPT ? 4 : 1);
    static constexpr int kt(int c) {
        if (c < 0 || c >= total) return 0;
        if (c < 8) return 3;
        if (c < 40) return 8;
        if (c < 48) return 11;
        if (c < 73) return 8;
        if (c < 77) return 9;
        return 4;
    }
    static constexpr bool bias(int c) { return c >= 0 && c < total; }
    static constexpr int loads(int) { return 0; }
    static constexpr bool drained(int c) { return c == 39; }      // layer 4's last tile: the skip copy needs it (flow_drain below)
    // plane stores of the tile's epilogue (every lane issues them); the per-point outputs (density, rgb, dpt features) are
    // conditional and uncounted, which can only make a wait longer
    static constexpr int stores(int c) {
        if (!SAVE || c < 0 || c >= total) return 0;
        if (c < 64) return BF16::kTileOps;      // save_h tile
        if (c < 72) return BF16::kTileOps;      // save_feature tile
        if (c < 73) return 0;                   // alpha row
        if (c < 77) return BF16::kTileOps;      // save_hv tile
        return 0;
    }
};

#ifndef VDN_NERF_FWD_ST_MODE
#define VDN_NERF_FWD_ST_MODE VDN_PLANE_ST_MODE      // cache policy of this kernel's saves (development A/B; mlp_engine.h: BF16::store_tile)
#endif
template <bool DPT, bool SAVE>
__global__ __launch_bounds__(256, 2) void nerf_fwd2_kernel(NerfArgs a) {
    constexpr int kSt = VDN_NERF_FWD_ST_MODE;
    using P = BF16;
    using ST = unsigned short;
    using PG = NerfFwdProg<DPT, SAVE>;
    constexpr int kSlot = P::stride(11);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    flow::Pipe<4, kSlot, 3, 2> pp;
    pp.init(a.blob, smem);
    const int lane = pp.lane, c = lane & 31, h = lane >> 5;
    // q = row of this lane in the (possibly compacted) work list = row of its training saves; p = its dense point id
    const long n_rows = a.active_idx != nullptr ? (long)*a.n_active : (long)a.P;
    if ((long)blockIdx.x * 4 * 32 >= n_rows) return;          // whole workgroup beyond the active list
    // (mlp_engine.h: cold weight stream inside a training step; it arrives underneath the encodings below)
    char* const wdump = pp.warm_dump();              // (this wave's own first DMA piece of ring slot 0: vdn_common.h)
    if constexpr (SAVE) warm_l2_issue(a.blob, PG::total * kSlot, (n_rows + 127) / 128, 512, wdump);
    warm_code_issue(SAVE ? kWarmCodeNerfFwd2 : 0, (n_rows + 127) / 128, 512, wdump);
    const long q_raw = ((long)blockIdx.x * 4 + pp.wave) * 32 + c;
    const bool ok = q_raw < n_rows;
    const long q = ok ? q_raw : n_rows - 1;                   // out-of-range lanes repeat the last row (their plane stores are duplicates)
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
        float qq[3];
        float n2 = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            qq[d] = a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z;
            n2 += qq[d] * qq[d];
        }
        const float rr = fminf(fmaxf(sqrtf(n2), 1.0f), 1e10f);   // renderer.py:114
#pragma unroll
        for (int d = 0; d < 3; ++d) p4[d] = qq[d] / rr;
        p4[3] = 1.0f / rr;
    }
    typename P::template Act<11> X;
    typename P::template Act<9> Y;
    bf16x8 pe_keep[6];              // the encoded input, kept for the skip after layer 4
    {
        float pe[84];
        posenc<4, 10, false>(p4, pe);
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
            const f32x16 t16 = vals_tile<84>(pe, h, kt);
            X.set(kt, t16);
            if constexpr (SAVE) {
                if (a.save_pe != nullptr) P::template store_tile<kSt>(reinterpret_cast<ST*>(a.save_pe), q, 96, kt, h, t16, true);
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) pe_keep[i] = X.r[i];
    }
    // D tile (t0 + nt) <- relu(acc); in training kept for the backward
    auto relu_into = [&](auto& D, int t0, ST* save, int ld) VDN_INL {
        return [&D, t0, save, ld, q, h](int nt, const f32x16& acc, int) VDN_INL {
            f32x16 o;
#pragma unroll
            for (int t = 0; t < 16; ++t) o[t] = relu0(acc[t]);
            D.set(t0 + nt, o);
            if constexpr (SAVE) P::template store_tile<kSt>(save, q, ld, nt, h, o, true);
        };
    };
    auto sv = [&](int l) VDN_INL { return save_h + l * PS; };
    warm_l2_wait();
    pp.template start<PG>();
    auto f0 = flow::flow_begin();
    auto f1 = flow::dense2<PG, 8>(f0, pp, X, flow::NoLoad{}, relu_into(Y, 0, sv(0), 256));          // pts_linears.0
    auto f2 = flow::dense2<PG, 8>(f1, pp, Y, flow::NoLoad{}, relu_into(X, 0, sv(1), 256));          // 1
    auto f3 = flow::dense2<PG, 8>(f2, pp, X, flow::NoLoad{}, relu_into(Y, 0, sv(2), 256));          // 2
    auto f4 = flow::dense2<PG, 8>(f3, pp, Y, flow::NoLoad{}, relu_into(X, 0, sv(3), 256));          // 3
    auto f5 = flow::dense2<PG, 8>(f4, pp, X, flow::NoLoad{}, relu_into(Y, 0, sv(4), 256));          // 4
    // skip (fields.py:334-335): h = cat([input_pts, h]) -> X = [PE (3 tiles) | h (8 tiles)]. The copy needs layer 4's last
    // tile: its pending epilogue runs first (the one layer boundary of this kernel without overlap)
    auto f5d = flow::flow_drain(f5);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) X.copy_tile(3 + kt, Y, kt);
#pragma unroll
    for (int i = 0; i < 6; ++i) X.r[i] = pe_keep[i];
    auto f6 = flow::dense2<PG, 8>(f5d, pp, X, flow::NoLoad{}, relu_into(Y, 0, sv(5), 256));         // 5
    auto f7 = flow::dense2<PG, 8>(f6, pp, Y, flow::NoLoad{}, relu_into(X, 0, sv(6), 256));          // 6
    auto f8 = flow::dense2<PG, 8>(f7, pp, X, flow::NoLoad{}, relu_into(Y, 0, sv(7), 256));          // 7
    // heads on h: image rows 0..255 feature_linear, row 256 alpha_linear
    auto f9 = flow::dense2<PG, 9>(f8, pp, Y, flow::NoLoad{}, [&](int nt, const f32x16& acc, int) VDN_INL {
        if (nt < 8) {
            X.set(nt, acc);
            if constexpr (SAVE) {
                if (a.save_feature != nullptr) P::template store_tile<kSt>(reinterpret_cast<ST*>(a.save_feature), q, 256, nt, h, acc, true);
            }
        } else {
            if (ok && h == 0) a.density[p] = acc[0];
        }
    });
    {   // views_linears.0 on cat([feature, PE4(view)])  (fields.py:340-344): tile 8 of X (the pending head tile is the alpha row)
        float pe[27];
        posenc<3, 4, false>(dir, pe);
        const f32x16 t16 = vals_tile<27>(pe, h, 0);
        X.set(8, t16);
        if constexpr (SAVE) {
            if (a.save_vpe != nullptr) P::template store_tile<kSt>(reinterpret_cast<ST*>(a.save_vpe), q, 32, 0, h, t16, true);
        }
    }
    auto f10 = flow::dense2<PG, 4, false>(f9, pp, X, flow::NoLoad{}, relu_into(Y, 0, reinterpret_cast<ST*>(a.save_hv), 128));
    // rgb_linear (image tile 0, rows 0..2) and dpt_linear (image tiles 1..3)
    auto f11 = flow::dense2<PG, DPT ? 4 : 1>(f10, pp, Y, flow::NoLoad{}, [&](int nt, const f32x16& acc, int) VDN_INL {
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
    flow::flow_finish(f11);
}

inline int launch_nerf_fwd2(const VdnNerfArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || args->P <= 0 || !args->blob || !args->density || !args->rgb || args->n_per_ray <= 0) return -1;
    if (!args->pts4 && (!args->rays_o || !args->rays_d || !args->z)) return -1;
    if (!args->dirs && !args->rays_d) return -1;
    const bool save = args->save_h != nullptr;
    if (save && (!args->save_hv || !args->save_feature)) return -1;
    const int grid = (args->P + 127) / 128;
    const size_t lds = 3 * BF16::stride(11);
    static bool once = (allow_big_lds(nerf_fwd2_kernel<false, false>, lds), allow_big_lds(nerf_fwd2_kernel<true, false>, lds),
                        allow_big_lds(nerf_fwd2_kernel<false, true>, lds), allow_big_lds(nerf_fwd2_kernel<true, true>, lds), true);
    (void)once;
    const bool dpt = args->feat != nullptr;
#define VDN_L(D, S) hipLaunchKernelGGL((nerf_fwd2_kernel<D, S>), dim3(grid), dim3(256), lds, stream, *args)
    if (dpt) { if (save) VDN_L(true, true); else VDN_L(true, false); }
    else { if (save) VDN_L(false, true); else VDN_L(false, false); }
#undef VDN_L
    return (int)hipGetLastError();
}

}  // namespace vdn
