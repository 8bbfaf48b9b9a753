// Per-ray device functions of the hierarchical sampler, shared by the per-ray kernels (rays.hip) and the fused
// "SDF pass + merge + up-sample" kernel (k_sdf_fwd0_split.h): one 64-lane wavefront per ray, rows in LDS.
// Reference: dpt_models/renderer.py 147-191 (up_sample) with 44-74 (sample_pdf), 193-207 (cat_z_vals).
//
// Numerics follow the reference's CPU path: ATen's CPU cumprod / cumsum accumulate float32 inputs in double and round every
// prefix to float, so the scans here run in double too, and mul/add pairs round like the reference's separate aten ops:
// every function body carries `#pragma clang fp contract(off)` (rays.hip is also built with -ffp-contract=off), so the
// results do not depend on the flags of the translation unit that includes this header.
#pragma once
#include "vdn_common.h"
#include "vdn_kernels.h"

namespace vdn {

constexpr int kRayWaves = 4;      // rays per workgroup
constexpr int kMaxT = 256;        // max samples per ray handled (4 per lane)
constexpr int kEPL = 4;

VDN_DEV double wave_incl_scan_mul(double v, int lane) {
#pragma clang fp contract(off)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(v, off);
        if (lane >= off) v *= t;
    }
    return v;
}
VDN_DEV double wave_incl_scan_add(double v, int lane) {
#pragma clang fp contract(off)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}
VDN_DEV double wave_sum(double v) {
#pragma clang fp contract(off)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
VDN_DEV float wave_max(float v) {
#pragma clang fp contract(off)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// exclusive product scan over a ray: in f[e] (lane owns elements kEPL*lane+e), out T[e] = prod_{k<i} f_k
// rounded to float per element (what alpha * cumprod([1, f...])[:-1] multiplies with).
VDN_DEV void ray_excl_cumprod(const float (&f)[kEPL], float (&T)[kEPL], int lane) {
#pragma clang fp contract(off)
    double loc = 1.0;
#pragma unroll
    for (int e = 0; e < kEPL; ++e) loc *= (double)f[e];
    const double incl = wave_incl_scan_mul(loc, lane);
    double run = __shfl_up(incl, 1);
    if (lane == 0) run = 1.0;
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        T[e] = (float)run;
        run *= (double)f[e];
    }
}

// ------------------------------------------------------------------------------------------
// one up-sampling round: z[M], sdf[M] -> n_imp new z per ray  (renderer.py:147-191, 44-74)
// ------------------------------------------------------------------------------------------
// the round itself, on a ray whose z / sdf rows (M entries) sit in LDS; cdf: M floats of scratch
VDN_DEV void upsample_row(const UpsampleArgs& a, int r, int lane, int M, const float* z, const float* sd, float* cdf) {
#pragma clang fp contract(off)
    const bool given_w = a.weights != nullptr;
    float o[3] = {0.0f, 0.0f, 0.0f}, d[3] = {0.0f, 0.0f, 0.0f};
    if (!given_w) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            o[k] = a.rays_o[r * 3 + k];
            d[k] = a.rays_d[r * 3 + k];
        }
    }
    auto radius = [&](float zz) {
#pragma clang fp contract(off)
        const float x = o[0] + d[0] * zz, y = o[1] + d[1] * zz, w = o[2] + d[2] * zz;
        return sqrtf(x * x + y * y + w * w);
    };
    float alpha[kEPL], f[kEPL], T[kEPL];
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        const int i = kEPL * lane + e;
        alpha[e] = 0.0f;
        f[e] = 1.0f;
        if (i < M - 1) {
            const float z0 = z[i], z1 = z[i + 1], s0 = sd[i], s1 = sd[i + 1];
            const bool inside = (radius(z0) < 1.0f) | (radius(z1) < 1.0f);
            const float mid_sdf = (s0 + s1) * 0.5f;
            float cosv = (s1 - s0) / (z1 - z0 + 1e-5f);
            const float prev_cos = (i == 0) ? 0.0f : (s0 - sd[i - 1]) / (z0 - z[i - 1] + 1e-5f);
            cosv = fminf(prev_cos, cosv);
            cosv = fminf(fmaxf(cosv, -1e3f), 0.0f) * (inside ? 1.0f : 0.0f);
            const float dist = z1 - z0;
            const float prev_esti = mid_sdf - cosv * dist * 0.5f;
            const float next_esti = mid_sdf + cosv * dist * 0.5f;
            const float prev_cdf = sigmoidf_(prev_esti * a.inv_s);
            const float next_cdf = sigmoidf_(next_esti * a.inv_s);
            alpha[e] = (prev_cdf - next_cdf + 1e-5f) / (prev_cdf + 1e-5f);
            f[e] = 1.0f - alpha[e] + 1e-7f;
        }
    }
    ray_excl_cumprod(f, T, lane);
    // sample_pdf (det=True)
    float w[kEPL];
    double wsum = 0.0;
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        const int i = kEPL * lane + e;
        w[e] = (i < M - 1) ? ((given_w ? a.weights[(long)r * a.w_ld + i] : alpha[e] * T[e]) + 1e-5f) : 0.0f;
        wsum += (double)w[e];
    }
    const float tot = (float)wave_sum(wsum);
    double loc = 0.0;
    float pdf[kEPL];
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        const int i = kEPL * lane + e;
        pdf[e] = (i < M - 1) ? w[e] / tot : 0.0f;
        loc += (double)pdf[e];
    }
    const double incl = wave_incl_scan_add(loc, lane);
    double run = __shfl_up(incl, 1);
    if (lane == 0) run = 0.0;
    if (lane == 0) cdf[0] = 0.0f;
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        const int i = kEPL * lane + e;
        run += (double)pdf[e];
        if (i < M - 1) cdf[i + 1] = (float)run;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < a.n_imp) {
        const float u = a.u[lane];
        int lo = 0, hi = M;                       // searchsorted(cdf, u, right=True): first idx with cdf[idx] > u
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, M - 1);
        float denom = cdf[above] - cdf[below];
        denom = denom < 1e-5f ? 1.0f : denom;
        const float t = (u - cdf[below]) / denom;
        a.new_z[(long)r * a.n_imp + lane] = z[below] + t * (z[above] - z[below]);
    }
}

// ------------------------------------------------------------------------------------------
// merge the new samples into the sorted ray (cat + sort + permuted sdf, renderer.py:197-205)
// also used for z_feed = sort(cat(z_vals, z_vals_outside)) (renderer.py:390-391), sdf pointers NULL
// ------------------------------------------------------------------------------------------
// za / zb: this wave's LDS scratch for the old and the new row; lz / ls (optional): LDS copies of the merged z / sdf rows
// old_src (optional): the old row already in LDS (then a.z is not read); new_sdf_src (optional): this ray's K new sdf values on
// chip (then a.new_sdf is not read)
VDN_DEV void merge_row(const MergeArgs& a, int r, int lane, float* za, float* zb, float* lz, float* ls, const float* old_src = nullptr,
                       const float* new_sdf_src = nullptr) {
#pragma clang fp contract(off)
    const int M = a.M, K = a.K;
    for (int i = lane; i < M; i += 64) za[i] = old_src != nullptr ? old_src[i] : a.z[(long)r * a.ld + i];
    for (int j = lane; j < K; j += 64) zb[j] = a.new_z[(long)r * a.K + j];
    __builtin_amdgcn_wave_barrier();
    const bool has_sdf = a.sdf != nullptr;
    // The old row is sorted in every call render() makes (coarse z, earlier merges), so an old element's rank among the
    // old ones is its index and a new element's is a binary search; an unsorted row (possible through z_vals_inject)
    // takes the counting path. Both give the stable ranks of cat + sort: old elements precede equal new ones.
    bool sorted_here = true;
    for (int i = lane; i + 1 < M; i += 64) sorted_here &= za[i] <= za[i + 1];
    const bool old_sorted = __all(sorted_here);
    float oz[kEPL], os[kEPL];
    int opos[kEPL];
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        const int i = lane + 64 * e;
        opos[e] = -1;
        if (i < M) {
            const float v = za[i];
            int cnt = i;
            if (!old_sorted) {
                cnt = 0;
                for (int k = 0; k < M; ++k) cnt += (za[k] < v) | ((za[k] == v) & (k < i));
            }
            for (int j = 0; j < K; ++j) cnt += zb[j] < v;
            opos[e] = cnt;
            oz[e] = v;
            os[e] = has_sdf ? a.sdf[(long)r * a.ld + i] : 0.0f;
        }
    }
    float nz = 0.0f, ns = 0.0f;
    int npos = -1;
    if (lane < K) {
        const float v = zb[lane];
        int cnt = 0;
        if (old_sorted) {
            int lo = 0, hi = M;                       // first index with za[idx] > v  ==  #(za <= v)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (za[mid] <= v) lo = mid + 1; else hi = mid;
            }
            cnt = lo;
        } else {
            for (int k = 0; k < M; ++k) cnt += za[k] <= v;
        }
        for (int j = 0; j < K; ++j) cnt += (zb[j] < v) | ((zb[j] == v) & (j < lane));
        npos = cnt;
        nz = v;
        ns = has_sdf ? (new_sdf_src != nullptr ? new_sdf_src[lane] : a.new_sdf[(long)r * a.K + lane]) : 0.0f;
    }
    __builtin_amdgcn_wave_barrier();   // all reads of the old row are done before it is overwritten in place
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        if (opos[e] >= 0) {
            a.z_out[(long)r * a.ld_out + opos[e]] = oz[e];
            if (has_sdf) a.sdf_out[(long)r * a.ld_out + opos[e]] = os[e];
            if (lz != nullptr) { lz[opos[e]] = oz[e]; ls[opos[e]] = os[e]; }
        }
    }
    if (npos >= 0) {
        a.z_out[(long)r * a.ld_out + npos] = nz;
        if (has_sdf) a.sdf_out[(long)r * a.ld_out + npos] = ns;
        if (lz != nullptr) { lz[npos] = nz; ls[npos] = ns; }
    }
}

}  // namespace vdn
