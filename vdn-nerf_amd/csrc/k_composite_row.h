// The per-ray compositor (reference dpt_models/renderer.py:262-315): NeuS alpha from (sdf, normal), inside / outside blend with
// the background pass, exclusive transmittance scan, weighted colour sums, eikonal partial sums - one 64-lane wavefront per ray,
// T = N + n_outside <= 256 samples, lane l owns samples 4l .. 4l+3.
//
// One body, two callers: composite_kernel (rays.hip; sdf / normals / colours of the ray read from HBM) and the fused shading kernel
// (k_sdf_fwd2.h MODE 2: the workgroup that evaluated the ray's 128 points hands them over in LDS). The body carries
// `#pragma clang fp contract(off)` like k_ray_rows.h, so both callers round like rays.hip's -ffp-contract=off build.
#pragma once
#include "k_ray_rows.h"

namespace vdn {

// per-sample inputs of the compositor: q = r * N + i (global sample id), i = sample of the ray
struct CompositeGlobalSrc {
    const float* sdf_;
    const float* normals_;
    const float* color_;
    VDN_DEV float sdf(long q, int) const { return sdf_[q]; }
    VDN_DEV float normal(long q, int, int k) const { return normals_[q * 3 + k]; }
    VDN_DEV float color(long q, int, int k) const { return color_[q * 3 + k]; }
};
// the ray's samples in LDS: [7][N] floats = sdf | normal x, y, z | colour r, g, b
struct CompositeLdsSrc {
    const float* rows;
    int n;
    VDN_DEV float sdf(long, int i) const { return rows[i]; }
    VDN_DEV float normal(long, int i, int k) const { return rows[(1 + k) * n + i]; }
    VDN_DEV float color(long, int i, int k) const { return rows[(4 + k) * n + i]; }
};

// what a caller may need in registers: the ray's eikonal partial sums (what eik_partial[2r], [2r+1] receive) and its composited
// colour incl. the background term (what color_out[3r ..] receives); uniform over the wave
struct RowOut { float num, den, c[3]; };
typedef RowOut EikPair;

// s_w / s_in: kMaxT floats of LDS scratch each (this wave's): the ray's weights and inside flags for the feature channels.
// eik_partial may be NULL (the fused kernel publishes the returned pair itself).
template <class Src>
VDN_DEV RowOut composite_row(const CompositeArgs& a, int r, int lane, const Src& src, float* s_w, float* s_in) {
#pragma clang fp contract(off)
    const int N = a.N, T = a.T;
    const bool has_bg = a.bg_density != nullptr;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = a.rays_o[r * 3 + k];
        d[k] = a.rays_d[r * 3 + k];
    }
    float inv_s = expf(a.variance[0] * 10.0f);                     // fields.py:364
    inv_s = fminf(fmaxf(inv_s, 1e-6f), 1e6f);                      // renderer.py:262
    const float car = a.cos_anneal_dev != nullptr ? a.cos_anneal_dev[0] : a.cos_anneal_ratio;      // (device scalar: graph-captured launches)

    float alpha[kEPL], f[kEPL], Tr[kEPL], wgt[kEPL], col[kEPL][3], ins[kEPL];
    double eik_num = 0.0, eik_den = 0.0;
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        const int i = kEPL * lane + e;
        alpha[e] = 0.0f;
        f[e] = 1.0f;
        ins[e] = 0.0f;
        col[e][0] = col[e][1] = col[e][2] = 0.0f;
        if (i < T) {
            float bg_a = 0.0f;
            if (has_bg) {
                const long q = (long)r * T + i;
                bg_a = 1.0f - expf(-softplus1(a.bg_density[q]) * a.bg_dists[q]);   // renderer.py:124
            }
            if (i < N) {
                const long q = (long)r * N + i;
                const float sdf = src.sdf(q, i), dist = a.dists[q];
                const float g0 = src.normal(q, i, 0), g1 = src.normal(q, i, 1), g2 = src.normal(q, i, 2);
                const float true_cos = d[0] * g0 + d[1] * g1 + d[2] * g2;
                const float iter_cos = -(fmaxf(-true_cos * 0.5f + 0.5f, 0.0f) * (1.0f - car) + fmaxf(-true_cos, 0.0f) * car);
                const float est_next = sdf + iter_cos * dist * 0.5f;
                const float est_prev = sdf - iter_cos * dist * 0.5f;
                const float prev_cdf = sigmoidf_(est_prev * inv_s);
                const float next_cdf = sigmoidf_(est_next * inv_s);
                float al = ((prev_cdf - next_cdf) + 1e-5f) / (prev_cdf + 1e-5f);
                al = fminf(fmaxf(al, 0.0f), 1.0f);
                const float mz = a.mid_z[q];
                const float x = o[0] + d[0] * mz, y = o[1] + d[1] * mz, w = o[2] + d[2] * mz;
                const float pn = sqrtf(x * x + y * y + w * w);
                const float inside = pn < 1.0f ? 1.0f : 0.0f;
                const float relax = pn < 1.2f ? 1.0f : 0.0f;
                const float gn = sqrtf(g0 * g0 + g1 * g1 + g2 * g2) - 1.0f;
                eik_num += (double)(relax * (gn * gn));
                eik_den += (double)relax;
                a.cdf[q] = prev_cdf;
                a.inside_sphere[q] = inside;
                ins[e] = inside;
                float c0 = src.color(q, i, 0), c1 = src.color(q, i, 1), c2 = src.color(q, i, 2);
                if (has_bg) {
                    const long qb = (long)r * T + i;
                    al = al * inside + bg_a * (1.0f - inside);                     // renderer.py:290
                    c0 = c0 * inside + a.bg_rgb[qb * 3] * (1.0f - inside);
                    c1 = c1 * inside + a.bg_rgb[qb * 3 + 1] * (1.0f - inside);
                    c2 = c2 * inside + a.bg_rgb[qb * 3 + 2] * (1.0f - inside);
                }
                alpha[e] = al;
                col[e][0] = c0; col[e][1] = c1; col[e][2] = c2;
            } else {
                const long qb = (long)r * T + i;
                alpha[e] = bg_a;
                col[e][0] = a.bg_rgb[qb * 3]; col[e][1] = a.bg_rgb[qb * 3 + 1]; col[e][2] = a.bg_rgb[qb * 3 + 2];
            }
            f[e] = 1.0f - alpha[e] + 1e-7f;
        }
    }
    ray_excl_cumprod(f, Tr, lane);
    double ws = 0.0, c0 = 0.0, c1 = 0.0, c2 = 0.0;
    float wmax = 0.0f;
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        const int i = kEPL * lane + e;
        wgt[e] = 0.0f;
        if (i < T) {
            wgt[e] = alpha[e] * Tr[e];
            a.weights[(long)r * T + i] = wgt[e];
            if (a.alpha_out != nullptr) a.alpha_out[(long)r * T + i] = alpha[e];
            s_w[i] = wgt[e];
            s_in[i] = ins[e];
            ws += (double)wgt[e];
            c0 += (double)(col[e][0] * wgt[e]);
            c1 += (double)(col[e][1] * wgt[e]);
            c2 += (double)(col[e][2] * wgt[e]);
            wmax = fmaxf(wmax, wgt[e]);
        }
    }
    const float wsum = (float)wave_sum(ws);
    float cr = (float)wave_sum(c0), cg = (float)wave_sum(c1), cb = (float)wave_sum(c2);
    wmax = wave_max(wmax);
    eik_num = wave_sum(eik_num);
    eik_den = wave_sum(eik_den);
    if (a.background_rgb != nullptr) {                                              // renderer.py:309-310
        cr = cr + a.background_rgb[0] * (1.0f - wsum);
        cg = cg + a.background_rgb[1] * (1.0f - wsum);
        cb = cb + a.background_rgb[2] * (1.0f - wsum);
    }
    if (lane == 0) {
        a.color_out[r * 3] = cr; a.color_out[r * 3 + 1] = cg; a.color_out[r * 3 + 2] = cb;
        a.weight_sum[r] = wsum;
        a.weight_max[r] = wmax;
        if (a.s_val != nullptr) a.s_val[r] = 1.0f / inv_s;
        if (a.eik_partial != nullptr) {
            a.eik_partial[r * 2] = (float)eik_num;
            a.eik_partial[r * 2 + 1] = (float)eik_den;
        }
    }
    // 96-channel VDN features: lanes over channels, samples streamed (weights / inside from LDS)
    if (a.feat_out != nullptr) {
        __builtin_amdgcn_wave_barrier();
        const int C = a.feat_ch;
        // lanes over channels (this lane: ch0 = lane, ch1 = lane + 64; up to 128 channels per pass), samples in groups of 8 with
        // all 32 loads of a group issued before the first use - the plain loop waited out one memory round trip per sample and
        // channel pass (160 us for 160 samples x 96 channels). Each channel's sum keeps the samples' order.
        for (int c0 = 0; c0 < C; c0 += 128) {
            const int ch0 = c0 + lane, ch1 = c0 + lane + 64;
            const bool v0 = ch0 < C, v1 = ch1 < C;
            double acc0 = 0.0, acc1 = 0.0;
            for (int i0 = 0; i0 < T; i0 += 8) {
                float fa[8][2], fb[8][2];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k;
                    const bool fg = i < N, bg = i < T && (has_bg || i >= N) && a.bg_feat != nullptr;
                    const long qf = ((long)r * N + i) * C, qb = ((long)r * T + i) * C;
                    fa[k][0] = (fg && v0) ? a.feat[qf + ch0] : 0.0f;
                    fa[k][1] = (fg && v1) ? a.feat[qf + ch1] : 0.0f;
                    fb[k][0] = (bg && v0) ? a.bg_feat[qb + ch0] : 0.0f;
                    fb[k][1] = (bg && v1) ? a.bg_feat[qb + ch1] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k;
                    if (i >= T) break;
                    float f0, f1;
                    if (i < N) {
                        f0 = fa[k][0]; f1 = fa[k][1];
                        if (has_bg) {
                            const float inside = s_in[i];
                            f0 = f0 * inside + fb[k][0] * (1.0f - inside);   // renderer.py:297-298
                            f1 = f1 * inside + fb[k][1] * (1.0f - inside);
                        }
                    } else {
                        f0 = fb[k][0]; f1 = fb[k][1];
                    }
                    const float wi = s_w[i];
                    acc0 += (double)(f0 * wi);
                    acc1 += (double)(f1 * wi);
                }
            }
            if (v0) a.feat_out[(long)r * C + ch0] = (float)acc0;
            if (v1) a.feat_out[(long)r * C + ch1] = (float)acc1;
        }
    }
    return RowOut{(float)eik_num, (float)eik_den, {cr, cg, cb}};
}

}  // namespace vdn
