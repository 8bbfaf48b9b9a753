// Per-ray kernels of the NeuS renderer on gfx950: one 64-lane wavefront per ray, wave-level scans
// (shuffle based) for the exclusive transmittance product and the CDF, LDS only as a per-wave
// scratch line. Replaces the elementwise / cumprod / cumsum / searchsorted / sort / gather op chains of
// reference dpt_models/renderer.py: render 334-359 (coarse + outside z), up_sample 147-191 with
// sample_pdf 44-74, cat_z_vals 193-207, the section set-up of render_core 228-230 /
// render_core_outside 107-109, and render_core's alpha + compositing 262-315.
//
// Numerics follow the reference's CPU path: ATen's CPU cumprod / cumsum accumulate float32 inputs in
// double and round every prefix to float, so the scans here run in double too. Built with
// -ffp-contract=off so that mul/add pairs round like the reference's separate aten ops.
#include "k_ray_rows.h"
#include "k_composite_row.h"

namespace vdn {

// ------------------------------------------------------------------------------------------
// coarse z + outside z  (renderer.py:334-359)
// ------------------------------------------------------------------------------------------
__global__ void coarse_z_kernel(CoarseArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = a.n_samples + a.n_outside;
    if (idx >= a.B * per) return;
    const int r = idx / per, j = idx % per;
    const float nearv = a.near[r], farv = a.far[r];
    if (j < a.n_samples) {
        float z = nearv + (farv - nearv) * a.lin_samples[j];
        if (a.t_rand != nullptr) z = z + (a.t_rand[r] - 0.5f) * 2.0f / (float)a.n_samples;
        a.z[(long)r * a.z_ld + j] = z;
    } else {
        // z_out[k] = far / flip(zo)[k] + 1/n_samples, zo = lower + (upper-lower)*t (stratified) or the linspace itself
        const int k = j - a.n_samples;
        const int kk = a.n_outside - 1 - k;
        float zo;
        if (a.t_rand_out != nullptr)
            zo = a.out_lower[kk] + (a.out_upper[kk] - a.out_lower[kk]) * a.t_rand_out[(long)r * a.n_outside + kk];
        else
            zo = a.lin_outside[kk];
        a.z_out[(long)r * a.n_outside + k] = farv / zo + 1.0f / (float)a.n_samples;
    }
}

__global__ __launch_bounds__(kRayWaves * 64) void upsample_kernel(UpsampleArgs a) {
    __shared__ float s_z[kRayWaves][kMaxT], s_sdf[kRayWaves][kMaxT], s_cdf[kRayWaves][kMaxT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRayWaves + wave;
    if (r >= a.B) return;            // whole wave exits together; no block-level barrier is used
    const int M = a.M;
    float* z = s_z[wave];
    float* sd = s_sdf[wave];
    const bool given_w = a.weights != nullptr;
    for (int i = lane; i < M; i += 64) {
        z[i] = a.z[(long)r * a.ld + i];
        sd[i] = given_w ? 0.0f : a.sdf[(long)r * a.ld + i];
    }
    __builtin_amdgcn_wave_barrier();
    upsample_row(a, r, lane, M, z, sd, s_cdf[wave]);
}

__global__ __launch_bounds__(kRayWaves * 64) void merge_kernel(MergeArgs a) {
    __shared__ float s_a[kRayWaves][kMaxT], s_b[kRayWaves][kMaxT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRayWaves + wave;
    if (r >= a.B) return;
    merge_row(a, r, lane, s_a[wave], s_b[wave], nullptr, nullptr);
}

// merge of round i followed by the up-sampling of round i+1 on the merged row, which never leaves LDS in between
// (renderer.py:372-386: cat_z_vals, then up_sample of the next iteration): one launch instead of two
__global__ __launch_bounds__(kRayWaves * 64) void merge_upsample_kernel(MergeArgs m, UpsampleArgs u) {
    __shared__ float s_a[kRayWaves][kMaxT], s_b[kRayWaves][kMaxT], s_z[kRayWaves][kMaxT], s_sdf[kRayWaves][kMaxT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRayWaves + wave;
    if (r >= m.B) return;
    merge_row(m, r, lane, s_a[wave], s_b[wave], s_z[wave], s_sdf[wave]);
    __builtin_amdgcn_wave_barrier();
    upsample_row(u, r, lane, m.M + m.K, s_z[wave], s_sdf[wave], s_a[wave]);     // the old row's scratch serves as the cdf row
}

// ------------------------------------------------------------------------------------------
// section lengths and mid-points (renderer.py:228-230 and 107-109): dists, mid_z for a sorted row
// ------------------------------------------------------------------------------------------
__global__ void sections_kernel(SectionArgs a) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.B * a.n) return;
    const int r = idx / a.n, i = idx % a.n;
    const float z0 = a.z[(long)r * a.ld + i];
    const float dist = (i + 1 < a.n) ? a.z[(long)r * a.ld + i + 1] - z0 : a.sample_dist;
    a.dists[(long)r * a.n + i] = dist;
    a.mid_z[(long)r * a.n + i] = z0 + dist * 0.5f;
}

// ------------------------------------------------------------------------------------------
// NeuS alpha from (sdf, normal), inside/outside blend with the background pass, transmittance
// scan and weighted sums (renderer.py:262-315). One wave per ray, T = N + n_outside <= 256.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRayWaves * 64) void composite_kernel(CompositeArgs a) {
    __shared__ float s_w[kRayWaves][kMaxT], s_in[kRayWaves][kMaxT];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRayWaves + wave;
    if (r >= a.B) return;
    composite_row(a, r, lane, CompositeGlobalSrc{a.sdf, a.normals, a.color}, s_w[wave], s_in[wave]);
}

// d_feats = sum_i w_i * feature_i (renderer.py:306-308) as its own launch, four waves per ray (the per-ray compositor has two
// waves per CU: 44 us for these 56 MB inside it): wave w sums its quarter of the samples in order, lanes over channels, the
// four partial sums are added in wave order.
// fl (the training step with the VDN head, vdn_composite_fwd_train): also d loss / d render_feats of the depth-feature term
// (dpt_runner.py:239-243 with mask = 1: loss_kernel's expressions, train_opt.hip) from the sums this block has just made
struct FeatLoss {
    const float* gt_feats;     // [B,C] or NULL
    float* g_feats;            // [B,C]
    float depth_weight, grad_scale;
};
VDN_DEV void feat_loss_grad(const CompositeArgs& a, const FeatLoss& fl, long idx, float f) {
    const float mask_sum = (float)a.B + 1e-5f;                     // dpt_runner.py:213 with mask = ones
    const float m = 1.0f;
    const float e = (f - fl.gt_feats[idx]) * m;
    const float sgn = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f);
    fl.g_feats[idx] = sgn * m / mask_sum * fl.depth_weight * fl.grad_scale;
}

__global__ __launch_bounds__(256) void feat_composite_kernel(CompositeArgs a, FeatLoss fl) {
    __shared__ double s_part[4][128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x;
    const int N = a.N, T = a.T, C = a.feat_ch;
    const bool has_bg = a.bg_density != nullptr;
    const int per = (T + 3) / 4, i_begin = wave * per, i_end = min(T, i_begin + per);
    for (int c0 = 0; c0 < C; c0 += 128) {
        const int ch0 = c0 + lane, ch1 = c0 + lane + 64;
        const bool v0 = ch0 < C, v1 = ch1 < C;
        double acc0 = 0.0, acc1 = 0.0;
        for (int i0 = i_begin; i0 < i_end; i0 += 8) {
            float fa[8][2], fb[8][2], wi[8], ins[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k;
                const bool in = i < i_end, fg = in && i < N, bg = in && (has_bg || i >= N) && a.bg_feat != nullptr;
                const long qf = ((long)r * N + i) * C, qb = ((long)r * T + i) * C;
                fa[k][0] = (fg && v0) ? a.feat[qf + ch0] : 0.0f;
                fa[k][1] = (fg && v1) ? a.feat[qf + ch1] : 0.0f;
                fb[k][0] = (bg && v0) ? a.bg_feat[qb + ch0] : 0.0f;
                fb[k][1] = (bg && v1) ? a.bg_feat[qb + ch1] : 0.0f;
                wi[k] = in ? a.weights[(long)r * T + i] : 0.0f;
                ins[k] = fg ? a.inside_sphere[(long)r * N + i] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k;
                if (i >= i_end) break;
                float f0, f1;
                if (i < N) {
                    f0 = fa[k][0]; f1 = fa[k][1];
                    if (has_bg) {
                        f0 = f0 * ins[k] + fb[k][0] * (1.0f - ins[k]);   // renderer.py:297-298
                        f1 = f1 * ins[k] + fb[k][1] * (1.0f - ins[k]);
                    }
                } else {
                    f0 = fb[k][0]; f1 = fb[k][1];
                }
                acc0 += (double)(f0 * wi[k]);
                acc1 += (double)(f1 * wi[k]);
            }
        }
        s_part[wave][lane] = acc0;
        s_part[wave][lane + 64] = acc1;
        __syncthreads();
        if (wave == 0) {
            if (v0) {
                const float f = (float)(((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane]);
                a.feat_out[(long)r * C + ch0] = f;
                if (fl.gt_feats != nullptr) feat_loss_grad(a, fl, (long)r * C + ch0, f);
            }
            if (v1) {
                const float f = (float)(((s_part[0][lane + 64] + s_part[1][lane + 64]) + s_part[2][lane + 64]) + s_part[3][lane + 64]);
                a.feat_out[(long)r * C + ch1] = f;
                if (fl.gt_feats != nullptr) feat_loss_grad(a, fl, (long)r * C + ch1, f);
            }
        }
        __syncthreads();
    }
}

// gradient_error = sum(num) / (sum(den) + 1e-5)  (renderer.py:313-315); also exports (num, den) for
// the data-parallel all-reduce of the two scalars.
__global__ void eikonal_reduce_kernel(const float* partial, int B, float* out3) {
    double n = 0.0, dn = 0.0;
    for (int i = threadIdx.x; i < B; i += 64) {
        n += (double)partial[2 * i];
        dn += (double)partial[2 * i + 1];
    }
    n = wave_sum(n);
    dn = wave_sum(dn);
    if (threadIdx.x == 0) {
        out3[0] = (float)n / ((float)dn + 1e-5f);
        out3[1] = (float)n;
        out3[2] = (float)dn;
    }
}

// The eikonal sums of composite_kernel on their own (same element -> lane mapping, same expressions, same double-precision
// wave sums: bit-identical partials), for the data-parallel step, which reduces them over the ranks early (vdn_eikonal_terms).
__global__ __launch_bounds__(kRayWaves * 64) void eikonal_terms_kernel(VdnEikonalArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRayWaves + wave;
    if (r >= a.B) return;
    const int N = a.N;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = a.rays_o[r * 3 + k];
        d[k] = a.rays_d[r * 3 + k];
    }
    double eik_num = 0.0, eik_den = 0.0;
#pragma unroll
    for (int e = 0; e < kEPL; ++e) {
        const int i = kEPL * lane + e;
        if (i < N) {
            const long q = (long)r * N + i;
            const float g0 = a.normals[q * 3], g1 = a.normals[q * 3 + 1], g2 = a.normals[q * 3 + 2];
            const float mz = a.mid_z[q];
            const float x = o[0] + d[0] * mz, y = o[1] + d[1] * mz, w = o[2] + d[2] * mz;
            const float pn = sqrtf(x * x + y * y + w * w);
            const float relax = pn < 1.2f ? 1.0f : 0.0f;
            const float gn = sqrtf(g0 * g0 + g1 * g1 + g2 * g2) - 1.0f;
            eik_num += (double)(relax * (gn * gn));
            eik_den += (double)relax;
        }
    }
    eik_num = wave_sum(eik_num);
    eik_den = wave_sum(eik_den);
    if (lane == 0) {
        a.eik_partial[r * 2] = (float)eik_num;
        a.eik_partial[r * 2 + 1] = (float)eik_den;
    }
}

// ------------------------------------------------------------------------------------------
// ray generation (poses.py:168-212, dataset.py:111-118): one thread per ray, images resident in HBM
// ------------------------------------------------------------------------------------------
__global__ void gen_rays_kernel(GenRaysArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.B) return;
    const float x = a.pixels_x[i], y = a.pixels_y[i];
    const float* K = a.intrinsic_inv;
    float p[3], v[3], d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) p[r] = K[r * 3 + 0] * x + K[r * 3 + 1] * y + K[r * 3 + 2];
    const float nrm = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = p[r] / nrm;
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = a.pose[r * 4 + 0] * v[0] + a.pose[r * 4 + 1] * v[1] + a.pose[r * 4 + 2] * v[2];
    float* o = a.out + (long)i * a.out_ld;
    float org[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        org[r] = a.pose[r * 4 + 3];
        o[r] = org[r];
        o[3 + r] = d[r];
    }
    const int xi = min(max((int)x, 0), a.W - 1), yi = min(max((int)y, 0), a.H - 1);
    const long pix = (long)yi * a.W + xi;
    if (a.out_ld > 6) o[6] = a.mask ? a.mask[pix * a.mask_ch] : 1.0f;
    if (a.image != nullptr && a.out_ld >= 10) {
        o[7] = a.image[pix * 3];
        o[8] = a.image[pix * 3 + 1];
        o[9] = a.image[pix * 3 + 2];
    }
    if (a.feats != nullptr)
        for (int ch = 0; ch < a.C; ++ch) o[10 + ch] = a.feats[pix * a.C + ch];
    if (a.near != nullptr && a.far != nullptr) {          // dataset.py:111-118
        const float aa = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        const float bb = 2.0f * (org[0] * d[0] + org[1] * d[1] + org[2] * d[2]);
        const float mid = 0.5f * (-bb) / aa;
        a.near[i] = mid - 1.0f;
        a.far[i] = mid + 1.0f;
    }
}

}  // namespace vdn

using namespace vdn;

extern "C" int vdn_gen_rays(const VdnGenRaysArgs* a, void* stream) {
    if (!a || a->B <= 0 || !a->pixels_x || !a->pixels_y || !a->intrinsic_inv || !a->pose || !a->out) return -1;
    if (a->out_ld < 6 || a->H <= 0 || a->W <= 0) return -2;
    if (a->feats && (a->C <= 0 || a->out_ld < 10 + a->C)) return -3;
    if (a->mask && a->mask_ch <= 0) return -4;
    hipLaunchKernelGGL(gen_rays_kernel, dim3((a->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int vdn_coarse_z(const VdnCoarseArgs* a, void* stream) {
    if (!a || a->B <= 0 || !a->near || !a->far || !a->z || !a->lin_samples) return -1;
    if (a->n_outside > 0 && (!a->z_out || !a->lin_outside)) return -2;
    if (a->t_rand_out && (!a->out_lower || !a->out_upper)) return -3;
    const int n = a->B * (a->n_samples + a->n_outside);
    hipLaunchKernelGGL(coarse_z_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int vdn_upsample_round(const VdnUpsampleArgs* a, void* stream) {
    if (!a || a->B <= 0 || !a->z || !a->new_z || !a->u) return -1;
    if (a->weights == nullptr ? (!a->sdf || !a->rays_o || !a->rays_d) : a->w_ld < a->M - 1) return -1;
    if (a->M < 2 || a->M > kMaxT || a->n_imp < 1 || a->n_imp > 64 || a->ld < a->M) return -2;
    hipLaunchKernelGGL(upsample_kernel, dim3((a->B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * 64), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int vdn_merge_sorted(const VdnMergeArgs* a, void* stream) {
    if (!a || a->B <= 0 || !a->z || !a->new_z || !a->z_out) return -1;
    if (a->M < 1 || a->K < 1 || a->K > 64 || a->M + a->K > kMaxT || a->ld < a->M || a->ld_out < a->M + a->K) return -2;
    if (a->sdf && (!a->new_sdf || !a->sdf_out)) return -3;
    hipLaunchKernelGGL(merge_kernel, dim3((a->B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * 64), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// work lists (see include/vdn_render.h): background samples whose NeRF++ output the compositor does not multiply by
// zero, and - for the training step - foreground samples inside the relaxed sphere. The norm test is the compositor's
// own (composite_kernel above: same expression, same -ffp-contract=off file).
// ------------------------------------------------------------------------------------------
struct ActiveJob {
    const float* rays_o;
    const float* rays_d;
    const float* mid_z;     // [B,N]
    int B, N, T;            // T samples listed per ray; samples s >= N are always active (background mode)
    float radius;           // background: active iff !(norm < 1); foreground (T == N): active iff norm < radius
    bool background;
    bool complement;        // foreground only: list what the test rejects
    int32_t* active_idx;
    int32_t* n_active;
    int32_t* ray_counts;
};

VDN_DEV bool sample_active(const ActiveJob& a, int r, int s, const float (&o)[3], const float (&d)[3]) {
    if (s >= a.N) return true;
    const float mz = a.mid_z[(long)r * a.N + s];
    const float x = o[0] + d[0] * mz, y = o[1] + d[1] * mz, w = o[2] + d[2] * mz;
    const float pn = sqrtf(x * x + y * y + w * w);
    return a.background ? !(pn < 1.0f) : (pn < a.radius) != a.complement;
}

template <bool FILL>
__global__ __launch_bounds__(kRayWaves * 64) void active_list_kernel(ActiveJob a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRayWaves + wave;
    if (r >= a.B) return;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = a.rays_o[r * 3 + k];
        d[k] = a.rays_d[r * 3 + k];
    }
    int base = 0;
    if (FILL) {      // offset of this ray = sum of the counts of the rays before it (ascending dense order, deterministic)
        int acc = 0;
        for (int i = lane; i < r; i += 64) acc += a.ray_counts[i];
        base = (int)wave_sum((double)acc);
    }
    int n = 0;
    for (int s0 = 0; s0 < a.T; s0 += 64) {
        const int s = s0 + lane;
        const bool act = s < a.T && sample_active(a, r, s, o, d);
        const unsigned long long m = __ballot(act);
        if (FILL && act) a.active_idx[base + n + __popcll(m & ((1ull << lane) - 1ull))] = r * a.T + s;
        n += __popcll(m);
    }
    if (lane == 0) {
        if (!FILL) a.ray_counts[r] = n;
        else if (r == a.B - 1) a.n_active[0] = base + n;
    }
}

static int launch_active(const ActiveJob& j, void* stream) {
    const dim3 grid((j.B + kRayWaves - 1) / kRayWaves), block(kRayWaves * 64);
    hipLaunchKernelGGL(active_list_kernel<false>, grid, block, 0, (hipStream_t)stream, j);
    hipLaunchKernelGGL(active_list_kernel<true>, grid, block, 0, (hipStream_t)stream, j);
    return (int)hipGetLastError();
}

extern "C" int vdn_background_active(const VdnBackgroundActiveArgs* a, void* stream) {
    if (!a || a->B <= 0 || a->N < 0 || a->T < a->N || !a->rays_o || !a->rays_d || (a->N > 0 && !a->mid_z) ||
        !a->active_idx || !a->n_active || !a->ray_counts) return -1;
    const ActiveJob j = {a->rays_o, a->rays_d, a->mid_z, a->B, a->N, a->T, 1.0f, true, false, a->active_idx, a->n_active, a->ray_counts};
    return launch_active(j, stream);
}

extern "C" int vdn_foreground_active(const VdnForegroundActiveArgs* a, void* stream) {
    if (!a || a->B <= 0 || a->N <= 0 || !a->rays_o || !a->rays_d || !a->mid_z || !(a->radius > 0.0f) ||
        !a->active_idx || !a->n_active || !a->ray_counts) return -1;
    const ActiveJob j = {a->rays_o, a->rays_d, a->mid_z, a->B, a->N, a->N, a->radius, false, a->complement != 0, a->active_idx, a->n_active, a->ray_counts};
    return launch_active(j, stream);
}

// The training step's per-ray preparation fused (include/vdn_render.h: VdnTrainPrepArgs): every one of the launches it
// replaces costs ~4.5 us of fixed time for a few hundred bytes per ray. Pass 0: both section sets + both per-ray counts;
// pass 1 (blocks [0, nb): foreground, [nb, 2 nb): background): the fill passes. Same expressions as sections_kernel /
// active_list_kernel (this file is built with -ffp-contract=off: the norm tests are the compositor's own).
template <int PASS>
__global__ __launch_bounds__(kRayWaves * 64) void train_prep_kernel(TrainPrepArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nb = (a.B + kRayWaves - 1) / kRayWaves;
    const bool second = PASS == 1 && (int)blockIdx.x >= nb;
    const int r = ((int)blockIdx.x - (second ? nb : 0)) * kRayWaves + wave;
    if (r >= a.B) return;
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = a.rays_o[r * 3 + k];
        d[k] = a.rays_d[r * 3 + k];
    }
    if (PASS == 0) {
        // z_feed = stable merge of the inside and the outside depths; both rows and the merged one stay in LDS for the sections
        __shared__ float s_a[kRayWaves][kMaxT], s_b[kRayWaves][kMaxT], s_f[kRayWaves][kMaxT], s_x[kRayWaves][kMaxT], s_z[kRayWaves][kMaxT];
        const float* final_z = nullptr;
        if (a.new_z != nullptr) {       // the last round's merge (no sdf), in place; the completed row stays in LDS
            MergeArgs m1 = {};
            m1.z = a.z; m1.new_z = a.new_z; m1.z_out = a.z;
            m1.B = a.B; m1.M = a.M_old; m1.K = a.N - a.M_old; m1.ld = a.z_ld; m1.ld_out = a.z_ld;
            merge_row(m1, r, lane, s_a[wave], s_b[wave], s_z[wave], s_x[wave]);
            __builtin_amdgcn_wave_barrier();
            final_z = s_z[wave];
        }
        MergeArgs m = {};
        m.z = a.z; m.new_z = a.z_out; m.z_out = a.z_feed;
        m.B = a.B; m.M = a.N; m.K = a.T - a.N; m.ld = a.z_ld; m.ld_out = a.T;
        merge_row(m, r, lane, s_a[wave], s_b[wave], s_f[wave], s_x[wave], final_z);
        __builtin_amdgcn_wave_barrier();
        const float* zi = s_a[wave];
        const float* zf = s_f[wave];
        for (int i = lane; i < a.N; i += 64) {
            const float z0 = zi[i];
            const float dist = (i + 1 < a.N) ? zi[i + 1] - z0 : a.sample_dist;
            a.dists[(long)r * a.N + i] = dist;
            a.mid_z[(long)r * a.N + i] = z0 + dist * 0.5f;
        }
        for (int i = lane; i < a.T; i += 64) {
            const float z0 = zf[i];
            const float dist = (i + 1 < a.T) ? zf[i + 1] - z0 : a.sample_dist;
            a.bg_dists[(long)r * a.T + i] = dist;
            a.bg_mid[(long)r * a.T + i] = z0 + dist * 0.5f;
        }
        __builtin_amdgcn_s_waitcnt(0);          // this wave reads back its own mid_z row below
        __builtin_amdgcn_wave_barrier();
    }
    // the two lists: foreground (inside samples with |p| < radius) and background (not inside the unit sphere, plus every
    // outside sample)
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        if (PASS == 1 && (which == 1) != second) continue;
        if (which == 0 && a.fg_active_idx == nullptr) continue;
        const ActiveJob j = which == 0
            ? ActiveJob{a.rays_o, a.rays_d, a.mid_z, a.B, a.N, a.N, a.fg_radius, false, false, a.fg_active_idx, a.fg_n_active, a.fg_ray_counts}
            : ActiveJob{a.rays_o, a.rays_d, a.mid_z, a.B, a.N, a.T, 1.0f, true, false, a.bg_active_idx, a.bg_n_active, a.bg_ray_counts};
        int base = 0;
        if (PASS == 1) {
            int acc = 0;
            for (int i = lane; i < r; i += 64) acc += j.ray_counts[i];
            base = (int)wave_sum((double)acc);
        }
        int n = 0;
        for (int s0 = 0; s0 < j.T; s0 += 64) {
            const int s = s0 + lane;
            const bool act = s < j.T && sample_active(j, r, s, o, d);
            const unsigned long long m = __ballot(act);
            if (PASS == 1 && act) j.active_idx[base + n + __popcll(m & ((1ull << lane) - 1ull))] = r * j.T + s;
            n += __popcll(m);
        }
        if (lane == 0) {
            if (PASS == 0) j.ray_counts[r] = n;
            else if (r == j.B - 1) j.n_active[0] = base + n;
        }
    }
}

extern "C" int vdn_train_prep(const VdnTrainPrepArgs* a, void* stream) {
    if (!a || a->B <= 0 || a->N <= 0 || a->T <= a->N || a->T > kMaxT || a->z_ld < a->N) return -1;
    if (!a->rays_o || !a->rays_d || !a->z || !a->z_out || !a->z_feed || !a->dists || !a->mid_z || !a->bg_dists || !a->bg_mid) return -2;
    if (a->T - a->N > 64) return -2;
    if (a->new_z && (a->M_old < 1 || a->M_old >= a->N || a->N - a->M_old > 64)) return -2;
    if (!a->bg_active_idx || !a->bg_n_active || !a->bg_ray_counts) return -3;
    if (a->fg_active_idx && (!a->fg_n_active || !a->fg_ray_counts || !(a->fg_radius > 0.0f))) return -4;
    const int nb = (a->B + kRayWaves - 1) / kRayWaves;
    hipLaunchKernelGGL(train_prep_kernel<0>, dim3(nb), dim3(kRayWaves * 64), 0, (hipStream_t)stream, *a);
    hipLaunchKernelGGL(train_prep_kernel<1>, dim3(2 * nb), dim3(kRayWaves * 64), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int vdn_merge_upsample(const VdnMergeArgs* m, const VdnUpsampleArgs* u, void* stream) {
    if (!m || !u || m->B <= 0 || !m->z || !m->new_z || !m->z_out || !m->sdf || !m->new_sdf || !m->sdf_out) return -1;
    if (m->M < 1 || m->K < 1 || m->K > 64 || m->M + m->K > kMaxT || m->ld < m->M || m->ld_out < m->M + m->K) return -2;
    if (u->B != m->B || u->M != m->M + m->K || u->weights || !u->rays_o || !u->rays_d || !u->u || !u->new_z ||
        u->n_imp < 1 || u->n_imp > 64) return -3;
    hipLaunchKernelGGL(merge_upsample_kernel, dim3((m->B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * 64), 0, (hipStream_t)stream, *m, *u);
    return (int)hipGetLastError();
}

extern "C" int vdn_sections(const VdnSectionArgs* a, void* stream) {
    if (!a || a->B <= 0 || a->n <= 0 || !a->z || !a->dists || !a->mid_z || a->ld < a->n) return -1;
    const int n = a->B * a->n;
    hipLaunchKernelGGL(sections_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int vdn_eikonal_terms(const VdnEikonalArgs* a, void* stream) {
    if (!a || a->B <= 0 || a->N <= 0 || a->N > kMaxT) return -1;
    if (!a->rays_o || !a->rays_d || !a->mid_z || !a->normals || !a->eik_partial || !a->eik_out) return -2;
    hipLaunchKernelGGL(eikonal_terms_kernel, dim3((a->B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * 64), 0, (hipStream_t)stream, *a);
    hipLaunchKernelGGL(eikonal_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->eik_partial, a->B, a->eik_out);
    return (int)hipGetLastError();
}

extern "C" int vdn_eikonal_reduce(const float* eik_partial, int32_t B, float* eik_out, void* stream) {
    if (!eik_partial || !eik_out || B <= 0) return -1;
    hipLaunchKernelGGL(eikonal_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, eik_partial, B, eik_out);
    return (int)hipGetLastError();
}

extern "C" int vdn_feat_composite(const VdnCompositeArgs* a, void* stream) {
    if (!a || a->B <= 0 || a->N <= 0 || a->T < a->N || a->T > kMaxT) return -1;
    if (!a->feat || !a->feat_out || a->feat_ch <= 0 || !a->weights || !a->inside_sphere) return -2;
    if (a->T > a->N && (!a->bg_density || !a->bg_feat)) return -3;
    hipLaunchKernelGGL(feat_composite_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, *a, FeatLoss{});
    return (int)hipGetLastError();
}

// The training step's forward compositor with the VDN head in the plain one-rank configuration (no mask): the per-ray kernel and the
// feature channels' weighted sums, which also make d loss / d render_feats on the spot. No eikonal reduce: the adjoint
// (vdn_composite_bwd_train) takes the denominator from the foreground work list, the scalars are reduced off the critical path.
extern "C" int vdn_composite_fwd_train(const VdnCompositeArgs* a, const float* gt_feats, float* g_feats, float depth_weight, float grad_scale,
                                       void* stream) {
    if (!a || a->B <= 0 || a->N <= 0 || a->T < a->N || a->T > kMaxT || !gt_feats || !g_feats) return -1;
    if (!a->rays_o || !a->rays_d || !a->sdf || !a->normals || !a->dists || !a->mid_z || !a->color || !a->variance) return -2;
    if (!a->weights || !a->cdf || !a->inside_sphere || !a->color_out || !a->weight_sum || !a->weight_max || !a->eik_partial) return -3;
    if (a->T > a->N && (!a->bg_density || !a->bg_rgb || !a->bg_dists)) return -4;
    if (!a->feat_out || !a->feat || a->feat_ch <= 0 || (a->T > a->N && !a->bg_feat)) return -5;
    VdnCompositeArgs per_ray = *a;
    per_ray.feat_out = nullptr;
    hipLaunchKernelGGL(composite_kernel, dim3((a->B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * 64), 0, (hipStream_t)stream, per_ray);
    hipLaunchKernelGGL(feat_composite_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, *a, FeatLoss{gt_feats, g_feats, depth_weight, grad_scale});
    return (int)hipGetLastError();
}

extern "C" int vdn_alpha_composite_fwd(const VdnCompositeArgs* a, void* stream) {
    if (!a || a->B <= 0 || a->N <= 0 || a->T < a->N || a->T > kMaxT) return -1;
    if (!a->rays_o || !a->rays_d || !a->sdf || !a->normals || !a->dists || !a->mid_z || !a->color || !a->variance) return -2;
    if (!a->weights || !a->cdf || !a->inside_sphere || !a->color_out || !a->weight_sum || !a->weight_max ||
        !a->eik_partial || !a->eik_out) return -3;
    if (a->T > a->N && (!a->bg_density || !a->bg_rgb || !a->bg_dists)) return -4;
    if (a->feat_out && (!a->feat || a->feat_ch <= 0 || (a->T > a->N && !a->bg_feat))) return -5;
    VdnCompositeArgs per_ray = *a;
    per_ray.feat_out = nullptr;                   // the feature channels have their own launch (feat_composite_kernel)
    hipLaunchKernelGGL(composite_kernel, dim3((a->B + kRayWaves - 1) / kRayWaves), dim3(kRayWaves * 64), 0, (hipStream_t)stream, per_ray);
    if (a->feat_out != nullptr)
        hipLaunchKernelGGL(feat_composite_kernel, dim3(a->B), dim3(256), 0, (hipStream_t)stream, *a, FeatLoss{});
    hipLaunchKernelGGL(eikonal_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->eik_partial, a->B, a->eik_out);
    return (int)hipGetLastError();
}
