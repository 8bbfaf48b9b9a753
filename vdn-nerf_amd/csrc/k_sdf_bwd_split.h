// Backward of the SDF network (both chains of k_sdf_bwd.h: rbar, the adjoint of the reverse sweep, and fbar, the adjoint of
// the forward pass) as ONE launch on gfx950, bf16 policy, training path (no ray gradients).
//
// k_sdf_bwd.h gives a wave 32 points and every output tile of every layer; rbar hands the second-order term ex_l (8 planes,
// 4 KB per point) to fbar through HBM because the two chains walk the layers in opposite directions: ex_7, which fbar needs
// first, is the last thing rbar produces, so all eight planes of a wave's 32 points (128 KB) would have to stay on chip.
// With the FEATURES split over the waves (k_sdf_fwd0_split.h: one workgroup = 32 points, 8 waves, wave w = output tile w
// of every layer, activations through LDS in B-fragment order, weights read straight into registers two steps ahead) the
// same wave owns tile w of ex_l in rbar and consumes exactly that tile in fbar: ex never leaves the wave - seven layers of it
// in a wave-private LDS strip, the last one (the first fbar needs) in registers. What still crosses HBM is what the
// weight-gradient GEMM reads: UB (out), AB (out), and the forward's H and V planes (in).
//
// Arithmetic is k_sdf_bwd.h's, operation for operation (zero-initialised accumulator, k-steps in order, the same epilogue
// expressions, ex rounded to bf16 as the EX planes were): UB and AB are bit-identical to the two-kernel path.
#pragma once
#include "mlp_engine.h"
#include "vdn_kernels.h"

namespace vdn {
namespace sdfbs {

constexpr int kWaves = 8;
constexpr int kStride = 20480;                  // BF16::stride(9)
constexpr int kPe = 0;                          // ub_0 = ub_4[PE part]: 4 k-steps x 1 KiB
constexpr int kT8 = kPe + 4 * 1024;             // tile 8 of ab_8 (g_sdf / scale): 2 k-steps
constexpr int kBuf0 = kT8 + 2 * 1024;           // activations: 16 k-steps x 1 KiB, two buffers
constexpr int kBuf1 = kBuf0 + 16 * 1024;
constexpr int kEx = kBuf1 + 16 * 1024;          // ex_0 .. ex_6: [layer][wave][2][lane] x 16 B (wave-private)
constexpr int kNExLds = 7;
constexpr int kLds = kEx + kNExLds * 16 * 1024;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The 16 steps: rbar layers 0..7 (forward weight stream), then fbar W8^T .. W1^T (transposed stream).
struct StepDesc { int fbar, l, kt, nt, chunk0; };
constexpr StepDesc step_desc(int i) {
    constexpr StepDesc t[16] = {
        {0, 0, 2, 8, 0}, {0, 1, 8, 8, 8}, {0, 2, 8, 8, 16}, {0, 3, 8, 7, 24}, {0, 4, 9, 8, 31}, {0, 5, 8, 8, 39}, {0, 6, 8, 8, 47}, {0, 7, 8, 8, 55},
        // fbar step producing ab_l: W_{l+1}^T, K = width of ab_{l+1}
        {1, 7, 9, 8, 0}, {1, 6, 8, 8, 8}, {1, 5, 8, 8, 16}, {1, 4, 8, 8, 24}, {1, 3, 8, 7, 32}, {1, 2, 7, 8, 41}, {1, 1, 8, 8, 49}, {1, 0, 8, 8, 57}};
    return t[i];
}
// LDS offset of k-step s of step I's input; steps alternate between the two buffers (step I writes buffer I & 1)
template <int I>
constexpr int in_off(int s) {
    constexpr StepDesc d = step_desc(I);
    if (!d.fbar && d.l == 0) return kPe + s * 1024;
    if (!d.fbar && d.l == 4 && s >= 14) return kPe + (s - 14) * 1024;
    if (d.fbar && d.l == 7 && s >= 16) return kT8 + (s - 16) * 1024;
    return ((I & 1) ? kBuf0 : kBuf1) + s * 1024;
}
template <int I>
constexpr int out_base() { return (I & 1) ? kBuf1 : kBuf0; }

struct WSet { bf16x8 w[18]; };

// Every global access of the kernel goes through a buffer descriptor (wave-uniform base in SGPRs, 32-bit per-lane byte offset,
// wave-uniform soffset): plain 64-bit addressing cost ~60 VGPRs of precomputed plane addresses here, i.e. spills.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
VDN_DEV rsrc_t make_rsrc(const void* base, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)(bytes > 0xffffffffL ? 0xffffffffL : bytes), 0x00020000);
}
#ifndef VDN_BS_LD_AUX
#define VDN_BS_LD_AUX 0         // cache-policy bits of the plane loads (bit 1 = non-temporal; the stores': VDN_BS_ST_AUX, vdn_common.h).
#endif                          // 2 costs the step 140 us: the kernel reads every H plane twice
// (VDN_BS_LDV_AUX, vdn_common.h: ... of the V planes' loads (read once) and of the H planes' SECOND reading (fbar))
VDN_DEV u32x4 bload(rsrc_t r, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, VDN_BS_LD_AUX); }
VDN_DEV u32x4 bload_once(rsrc_t r, unsigned voff, unsigned soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, VDN_BS_LDV_AUX); }
VDN_DEV void bstore(rsrc_t r, unsigned voff, unsigned soff, const u32x4& v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, VDN_BS_ST_AUX); }

template <int I>
VDN_DEV void load_weights(WSet& W, rsrc_t blob_r, rsrc_t blob_f, int wave, unsigned lane16) {
    constexpr StepDesc d = step_desc(I);
    const int t = wave < d.nt ? wave : d.nt - 1;        // a layer with 7 tiles: wave 7 recomputes tile 6 and drops it
    const unsigned ch = (unsigned)(d.chunk0 + t) * kStride;
    static_for<2 * d.kt>([&](auto s_c) VDN_INL {
        constexpr int s = decltype(s_c)::value;
        W.w[s] = __builtin_bit_cast(bf16x8, bload(d.fbar ? blob_f : blob_r, lane16, ch + s * 1024));
    });
}

VDN_DEV void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int I>
VDN_DEV f32x16 step_mma(const WSet& W, const char* my) {
    constexpr StepDesc d = step_desc(I);
    constexpr int NS = 2 * d.kt;
    constexpr int PRE = NS < 4 ? NS : 4;            // B fragments read ahead of their MFMAs (6: no difference, 8 more registers)
    f32x16 acc;
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.0f;
    bf16x8 x[NS];
    static_for<NS>([&](auto s_c) VDN_INL {
        constexpr int s = decltype(s_c)::value;
        x[s] = *reinterpret_cast<const bf16x8*>(my + in_off<I>(s));
    });
    static_for<NS>([&](auto s_c) VDN_INL {
        constexpr int s = decltype(s_c)::value;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W.w[s], x[s], acc, 0, 0, 0);
    });
    __builtin_amdgcn_sched_group_barrier(0x100, PRE, 0);
    static_for<NS - PRE>([&](auto) VDN_INL {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    });
    __builtin_amdgcn_sched_group_barrier(0x008, PRE, 0);
    return acc;
}

VDN_DEV u32x4 pack8(const f32x16& v, int k) {
    u32x4 o;
    o[0] = pack_bf16x2(v[8 * k], v[8 * k + 1]);
    o[1] = pack_bf16x2(v[8 * k + 2], v[8 * k + 3]);
    o[2] = pack_bf16x2(v[8 * k + 4], v[8 * k + 5]);
    o[3] = pack_bf16x2(v[8 * k + 6], v[8 * k + 7]);
    return o;
}
VDN_DEV u32x4 pack8v(const float (&v)[8]) {
    u32x4 o;
    o[0] = pack_bf16x2(v[0], v[1]); o[1] = pack_bf16x2(v[2], v[3]); o[2] = pack_bf16x2(v[4], v[5]); o[3] = pack_bf16x2(v[6], v[7]);
    return o;
}
VDN_DEV void unpack8u(const u32x4& a, float (&r)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[2 * j] = bf16_lo(a[j]); r[2 * j + 1] = bf16_hi(a[j]); }
}
VDN_DEV void unpack8(const uint4& a, float (&r)[8]) {
    r[0] = bf16_lo(a.x); r[1] = bf16_hi(a.x); r[2] = bf16_lo(a.y); r[3] = bf16_hi(a.y);
    r[4] = bf16_lo(a.z); r[5] = bf16_hi(a.z); r[6] = bf16_lo(a.w); r[7] = bf16_hi(a.w);
}
VDN_DEV f32x16 unpack16(const u32x4& a, const u32x4& b) {
    f32x16 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        r[2 * j] = bf16_lo(a[j]); r[2 * j + 1] = bf16_hi(a[j]);
        r[8 + 2 * j] = bf16_lo(b[j]); r[8 + 2 * j + 1] = bf16_hi(b[j]);
    }
    return r;
}
// per-lane byte offset of the lane's 16-byte pieces inside a PT32 plane of row length ld (piece k at + 1024 k, tile t at + 2048 t)
VDN_DEV unsigned plane_voff(long row, int ld, int h) { return (unsigned)(((row >> 5) * (32L * ld) + h * 256 + (row & 31) * 8) * 2); }

__global__ __launch_bounds__(kWaves * 64, 1) void sdf_bwd_split_kernel(SdfRbarArgs ra, SdfFbarArgs fa) {
    using P = BF16;
    using ST = unsigned short;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const WorkRow wr = work_row(ra.active_idx, ra.n_active, ra.P, 1, 0, c);
    if (wr.none) return;
    // (vdn_common.h: both weight streams and the code are cold inside a training step; the LDS of this kernel holds activations
    // that any wave writes: the barrier behind the wait keeps them out of a slower wave's dump area)
    warm_l2_issue(ra.blob, 63 * kStride, wr.n_wg, 256, smem + wave * 1024);
    warm_l2_issue(fa.blob, 67 * kStride, wr.n_wg, 256, smem + wave * 1024);
    warm_code_issue(kWarmCodeSdfBwdSplit, wr.n_wg, 256, smem + wave * 1024);
    warm_l2_sync();
    const bool ok = wr.ok;
    const long p = wr.row, pd = wr.point;
    const long Pn = P::rows(ra.P), PS = Pn * 256;
    char* const my = smem + lane * 16;
    constexpr int from_h = 2;                       // the bf16 forward's units (H and V in 1 / (100 log2 e)): the entry point checks
    constexpr float ex_k = 0.6931471805599453f;

    const unsigned lane16 = lane * 16;
    const rsrc_t RW = make_rsrc(ra.blob, 63L * kStride), FW = make_rsrc(fa.blob, 67L * kStride);
    WSet WA, WB;                                    // even / odd steps
    load_weights<0>(WA, RW, FW, wave, lane16);
    load_weights<1>(WB, RW, FW, wave, lane16);
    __builtin_amdgcn_sched_barrier(0);

    // planes: descriptors + this lane's offsets for the three row lengths; byte offsets of the blocks inside UB and AB
    const rsrc_t RS = make_rsrc(ra.S, 8 * PS * 2), RV = make_rsrc(ra.V, 8 * PS * 2);
    const rsrc_t RUB = make_rsrc(ra.UB, Pn * 2144 * 2), RAB = make_rsrc(fa.AB, Pn * 2336 * 2), RG = make_rsrc(fa.g_feat, PS * 2);
    const unsigned v256 = plane_voff(p, 256, h), v288 = plane_voff(p, 288, h), v64 = plane_voff(p, 64, h);
    const unsigned PSB = (unsigned)(PS * 2);
    const unsigned ub4_off = (unsigned)((Pn * 64 + 3 * PS) * 2);
    auto ub_off = [&](int l) VDN_INL -> unsigned {  // ub_l, l = 1..8 (ub_4: row length 288)
        return l <= 4 ? (unsigned)(Pn * 64 * 2) + (unsigned)(l - 1) * PSB : ub4_off + (unsigned)(Pn * 288 * 2) + (unsigned)(l - 5) * PSB;
    };
    auto ab_off = [&](int l) VDN_INL -> unsigned { return (unsigned)(Pn * 288 * 2) + (unsigned)(7 - l) * PSB; };   // l = 7..0
    const unsigned tile_off = (unsigned)wave * 2048;
    auto put = [&](rsrc_t r, unsigned voff, unsigned soff, const u32x4& p0, const u32x4& p1) VDN_INL {
        if (ok) {
            bstore(r, voff, soff, p0);
            bstore(r, voff, soff + 1024, p1);
        }
    };

    // ---- rbar's input: ub_0 = ub_4[PE part] = scale * J_PE g_normal (waves 0, 1: one tile each) ---------------------------
    if (wave < 2) {
        float xin[3];
        if (ra.pts != nullptr) {
#pragma unroll
            for (int d = 0; d < 3; ++d) xin[d] = ra.pts[pd * 3 + d] * ra.scale;
        } else {
            const long r = pd / ra.n_per_ray;
            const float z = ra.z[r * ra.z_ld + (pd - r * ra.n_per_ray)];
#pragma unroll
            for (int d = 0; d < 3; ++d) xin[d] = (ra.rays_o[r * 3 + d] + ra.rays_d[r * 3 + d] * z) * ra.scale;
        }
        float ub39[39];
        float gn[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            gn[d] = ra.g_normals[pd * 3 + d] * ra.scale;
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
        const f32x16 t16 = wave == 0 ? vals_tile<39>(ub39, h, 0) : vals_tile<39>(ub39, h, 1);
        const u32x4 p0 = pack8(t16, 0), p1 = pack8(t16, 1);
        *reinterpret_cast<u32x4*>(my + kPe + (2 * wave) * 1024) = p0;
        *reinterpret_cast<u32x4*>(my + kPe + (2 * wave + 1) * 1024) = p1;
        put(RUB, v64, tile_off, p0, p1);
        put(RUB, v288, ub4_off + 7 * 2048 + tile_off, p0, p1);
    }
    lds_barrier();

    u32x4 exr[8 - kNExLds][2];                      // the last layers' ex of this wave's tile (bf16, as the EX planes held them)
    u32x4 gf[2];                                    // this wave's tile of g_feat (loaded during step 6)
    struct Raw { u32x4 k[2]; };
    Raw rs, rv;                                     // plane tiles of the current step's epilogue (loaded one step ahead)
    rs.k[0] = bload(RS, v256, tile_off); rs.k[1] = bload(RS, v256, tile_off + 1024);
    rv.k[0] = bload_once(RV, v256, tile_off); rv.k[1] = bload_once(RV, v256, tile_off + 1024);

    static_for<16>([&](auto i_c) VDN_INL {
        constexpr int I = decltype(i_c)::value;
        constexpr StepDesc d = step_desc(I);
        WSet& W = (I & 1) ? WB : WA;
        if constexpr (I == 6) {                     // fbar's input tile, two steps ahead of its use
            gf[0] = bload(RG, v256, tile_off);
            gf[1] = bload(RG, v256, tile_off + 1024);
        }
        if constexpr (I == 8) {
            // fbar's input: ab_8 = [g_feat | g_sdf / scale] -> LDS (buffer 1: step 7 wrote nothing there) and the AB plane
            const u32x4 p0 = gf[0], p1 = gf[1];
            *reinterpret_cast<u32x4*>(my + kBuf1 + (2 * wave) * 1024) = p0;
            *reinterpret_cast<u32x4*>(my + kBuf1 + (2 * wave + 1) * 1024) = p1;
            put(RAB, v288, tile_off, p0, p1);
            if (wave == 0) {
                float g1[1] = {fa.g_sdf[pd] / fa.scale};
                const f32x16 t16 = vals_tile<1>(g1, h, 0);
                const u32x4 q0 = pack8(t16, 0), q1 = pack8(t16, 1);
                *reinterpret_cast<u32x4*>(my + kT8) = q0;
                *reinterpret_cast<u32x4*>(my + kT8 + 1024) = q1;
                put(RAB, v288, 8 * 2048, q0, q1);
            }
            lds_barrier();
        }
        const f32x16 acc = step_mma<I>(W, my);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I + 2 < 16) load_weights<I + 2>(W, RW, FW, wave, lane16);     // this set is free again
        // plane tiles of the NEXT step's epilogue
        Raw ns{}, nv{};
        if constexpr (I + 1 < 16) {
            constexpr StepDesc n = step_desc(I + 1);
            const unsigned so = (unsigned)n.l * PSB + (unsigned)(wave < n.nt ? wave : n.nt - 1) * 2048;
            if constexpr (n.fbar) { ns.k[0] = bload_once(RS, v256, so); ns.k[1] = bload_once(RS, v256, so + 1024); }
            else { ns.k[0] = bload(RS, v256, so); ns.k[1] = bload(RS, v256, so + 1024); }
            if constexpr (!n.fbar) { nv.k[0] = bload_once(RV, v256, so); nv.k[1] = bload_once(RV, v256, so + 1024); }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the epilogue in two halves of 8 values (one 16-byte piece each): half the temporaries of a 16-value pass
        u32x4 pp[2];
        static_for<2>([&](auto k_c) VDN_INL {
            constexpr int k = decltype(k_c)::value;
            float sv[8], av[8];
            unpack8u(rs.k[k], sv);
#pragma unroll
            for (int t = 0; t < 8; ++t) av[t] = acc[8 * k + t];
            float ov[8];
            if constexpr (!d.fbar) {
                // ub_{l+1} = vb * s_l ;  ex_l = ex_k * vb * v_l * (1 - s_l)
                float vv[8], ev[8];
                unpack8u(rv.k[k], vv);
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const float sp = sprime(sv[t], from_h);
                    ov[t] = av[t] * sp;
                    ev[t] = ex_k * av[t] * vv[t] * (1.0f - sp);
                }
                const u32x4 e = pack8v(ev);
                if constexpr (d.l < kNExLds) *reinterpret_cast<u32x4*>(my + kEx + (d.l * 16 + 2 * wave + k) * 1024) = e;
                else exr[d.l - kNExLds][k] = e;
            } else {
                // ab_l = hb * s_l + ex_l
                u32x4 e;
                if constexpr (d.l < kNExLds) e = *reinterpret_cast<const u32x4*>(my + kEx + (d.l * 16 + 2 * wave + k) * 1024);
                else e = exr[d.l - kNExLds][k];
                float ev[8];
                unpack8u(e, ev);
#pragma unroll
                for (int t = 0; t < 8; ++t) ov[t] = av[t] * sprime(sv[t], from_h) + ev[t];
            }
            pp[k] = pack8v(ov);
        });
        const u32x4 p0 = pp[0], p1 = pp[1];
        if (wave < d.nt) {
            if constexpr (!(d.fbar == 0 && d.l == 7) && !(d.fbar == 1 && d.l == 0)) {      // ub_8 and ab_0 feed no further step
                *reinterpret_cast<u32x4*>(my + out_base<I>() + (2 * wave) * 1024) = p0;
                *reinterpret_cast<u32x4*>(my + out_base<I>() + (2 * wave + 1) * 1024) = p1;
            }
            if constexpr (!d.fbar) put(RUB, d.l + 1 == 4 ? v288 : v256, ub_off(d.l + 1) + tile_off, p0, p1);
            else put(RAB, v256, ab_off(d.l) + tile_off, p0, p1);
        }
        rs = ns;
        rv = nv;
        if constexpr (I != 7) lds_barrier();        // (step 8's prologue has its own barrier behind the ab_8 tiles)
    });
}

inline int launch(const VdnSdfRbarArgs* ra, const VdnSdfFbarArgs* fa, hipStream_t stream) {
    static bool once = (allow_big_lds(sdf_bwd_split_kernel, kLds), true);
    (void)once;
    const int grid = (ra->P + 31) / 32;
    hipLaunchKernelGGL(sdf_bwd_split_kernel, dim3(grid), dim3(kWaves * 64), kLds, stream, *ra, *fa);
    return (int)hipGetLastError();
}

}  // namespace sdfbs
}  // namespace vdn
