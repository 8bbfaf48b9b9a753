// Adjoint of the per-ray compositing kernel (rays.hip: composite_kernel), one wavefront per ray.
// weights w_i = alpha_i T_i, T_i = prod_{k<i} (1 - alpha_k + 1e-7):
//   dL/dalpha_i = Wb_i T_i - (sum_{j>i} Wb_j w_j) / (1 - alpha_i + 1e-7),   Wb_i = dL/dw_i
// then through the inside/outside blend, the NeuS alpha (renderer.py:262-282), the background
// alpha (renderer.py:124) and the eikonal term (renderer.py:313-315). Built with -ffp-contract=off.
#include "vdn_common.h"
#include "vdn_kernels.h"
#include "k_composite_row.h"

namespace vdn {

constexpr int kRW = 4;
constexpr int kMaxTB = 256;
constexpr int kE = 4;

VDN_DEV double wsum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// The 96 feature channels outside the per-ray kernel (CompositeBwdArgs.feat_scratch = [fd (B*T) | cf (B*N) | cb (B*T)]):
// one wave per sample row, lanes over channels, many waves per CU - plain streaming kernels.
//   feat_dot:   fd[r,i] = sum_ch g_feat[r,ch] * blended feature(r,i,ch)       (the features' share of dL/dw_i)
//   feat_outer: d_feat[r,i,:] = cf[r,i] * g_feat[r,:],  d_bg_feat[r,i,:] = cb[r,i] * g_feat[r,:]
__global__ __launch_bounds__(256) void feat_dot_kernel(CompositeBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int N = a.N, T = a.T, C = a.feat_ch;
    if (row >= (long)a.B * T) return;
    const int r = (int)(row / T), i = (int)(row - (long)r * T);
    const bool has_bg = a.bg_density != nullptr;
    float v0 = 0.0f, v1 = 0.0f;
    float b0 = 0.0f, b1 = 0.0f;
    if (has_bg && a.bg_feat != nullptr) {
        if (lane < C) b0 = a.bg_feat[row * C + lane];
        if (lane + 64 < C) b1 = a.bg_feat[row * C + lane + 64];
    }
    if (i < N) {
        const long q = (long)r * N + i;
        if (lane < C) v0 = a.feat[q * C + lane];
        if (lane + 64 < C) v1 = a.feat[q * C + lane + 64];
        if (has_bg) {
            const float mz = a.mid_z[q];
            const float x = a.rays_o[r * 3] + a.rays_d[r * 3] * mz, y = a.rays_o[r * 3 + 1] + a.rays_d[r * 3 + 1] * mz,
                        zz = a.rays_o[r * 3 + 2] + a.rays_d[r * 3 + 2] * mz;
            const float inside = sqrtf(x * x + y * y + zz * zz) < 1.0f ? 1.0f : 0.0f;
            v0 = v0 * inside + b0 * (1.0f - inside);
            v1 = v1 * inside + b1 * (1.0f - inside);
        }
    } else {
        v0 = b0; v1 = b1;
    }
    float part = 0.0f;
    if (lane < C) part += a.g_feat[(long)r * C + lane] * v0;
    if (lane + 64 < C) part += a.g_feat[(long)r * C + lane + 64] * v1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) a.feat_scratch[row] = part;
}

__global__ __launch_bounds__(256) void feat_outer_kernel(CompositeBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int N = a.N, T = a.T, C = a.feat_ch;
    if (row >= (long)a.B * T) return;
    const int r = (int)(row / T), i = (int)(row - (long)r * T);
    const float* cf = a.feat_scratch + (long)a.B * T;
    const float* cb = cf + (long)a.B * N;
    const float g0 = lane < C ? a.g_feat[(long)r * C + lane] : 0.0f, g1 = lane + 64 < C ? a.g_feat[(long)r * C + lane + 64] : 0.0f;
    if (i < N) {
        const long q = (long)r * N + i;
        const float c = cf[q];
        if (lane < C) a.d_feat[q * C + lane] = c * g0;
        if (lane + 64 < C) a.d_feat[q * C + lane + 64] = c * g1;
    }
    if (a.bg_density != nullptr && a.d_bg_feat != nullptr) {
        const float c = cb[row];
        if (lane < C) a.d_bg_feat[row * C + lane] = c * g0;
        if (lane + 64 < C) a.d_bg_feat[row * C + lane + 64] = c * g1;
    }
}

// upstream gradients the caller already holds in registers (the fused forward + loss + backward launch below): the colour
// term's gradient of this ray, d loss / d gradient_error and the eikonal term's GLOBAL denominator
struct CompositeBwdOvr {
    bool on;
    float gc[3], g_eik, eik_den;
};

// the adjoint of one ray (one wavefront); `wave` = this ray's slot in the workgroup's LDS arrays
VDN_DEV void composite_bwd_row(const CompositeBwdArgs& a, int r, int wave, int lane, const CompositeBwdOvr& ov) {
    const int N = a.N, T = a.T, C = a.feat_ch;
    const bool has_bg = a.bg_density != nullptr;
    const bool has_feat = a.d_feat != nullptr && a.g_feat != nullptr;
    float o[3], d[3], gc[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = a.rays_o[r * 3 + k];
        d[k] = a.rays_d[r * 3 + k];
        if (ov.on) gc[k] = ov.gc[k];
        else if (a.g_color != nullptr) gc[k] = a.g_color[r * 3 + k];
    }
    const float var = a.variance[0];
    const float inv_s_raw = expf(var * 10.0f);
    const float inv_s = fminf(fmaxf(inv_s_raw, 1e-6f), 1e6f);
    const bool s_unclipped = inv_s_raw >= 1e-6f && inv_s_raw <= 1e6f;
    const float car = a.cos_anneal_dev != nullptr ? a.cos_anneal_dev[0] : a.cos_anneal_ratio;      // (device scalar: graph-captured launches)
    const float g_eik = ov.on ? ov.g_eik : (a.g_eik != nullptr ? a.g_eik[0] : 0.0f);
    const float eik_den = ov.on ? ov.eik_den : a.eik[2] + 1e-5f;
    float bgc[3] = {0.0f, 0.0f, 0.0f};
    if (a.background_rgb != nullptr) {
        bgc[0] = a.background_rgb[0]; bgc[1] = a.background_rgb[1]; bgc[2] = a.background_rgb[2];
    }

    // The 96-channel VDN features are handled by the whole wave per sample (lanes over channels: rows of 384 contiguous bytes)
    // instead of by the sample's owner lane walking its own row (64 lanes on 64 different rows: 268 us per launch).
    // s_fd[i] = sum_ch g_feat[ch] * blended feature(i, ch): the features' share of dL/dw_i; s_cf / s_cb: the coefficients of
    // g_feat in d_feat[i, :] / d_bg_feat[i, :], filled by the owner lanes below.
    __shared__ float s_fd[kRW][kMaxTB], s_cf[kRW][kMaxTB], s_cb[kRW][kMaxTB];
    float gfa = 0.0f, gfb = 0.0f;               // this lane's channels of g_feat: lane, lane + 64
    const bool ext_feat = has_feat && a.feat_scratch != nullptr;       // feat_dot_kernel ran before, feat_outer_kernel runs after
    if (ext_feat) {
        for (int i = lane; i < T; i += 64) s_fd[wave][i] = a.feat_scratch[(long)r * T + i];
        __builtin_amdgcn_wave_barrier();
    } else if (has_feat) {
        if (lane < C) gfa = a.g_feat[(long)r * C + lane];
        if (lane + 64 < C) gfb = a.g_feat[(long)r * C + lane + 64];
        for (int i0 = 0; i0 < T; i0 += 8) {
            float fa[8][2], fb[8][2], insd[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k;
                fa[k][0] = fa[k][1] = fb[k][0] = fb[k][1] = 0.0f;
                insd[k] = 0.0f;
                if (i >= T) continue;
                const long qt = (long)r * T + i;
                if (i < N) {
                    const long q = (long)r * N + i;
                    const float mz = a.mid_z[q];
                    const float x = o[0] + d[0] * mz, y = o[1] + d[1] * mz, zz = o[2] + d[2] * mz;
                    insd[k] = sqrtf(x * x + y * y + zz * zz) < 1.0f ? 1.0f : 0.0f;
                    if (lane < C) fa[k][0] = a.feat[q * C + lane];
                    if (lane + 64 < C) fa[k][1] = a.feat[q * C + lane + 64];
                }
                if (has_bg && a.bg_feat != nullptr) {
                    if (lane < C) fb[k][0] = a.bg_feat[qt * C + lane];
                    if (lane + 64 < C) fb[k][1] = a.bg_feat[qt * C + lane + 64];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = i0 + k;
                if (i >= T) break;
                float v0, v1;
                if (i < N) {
                    v0 = fa[k][0]; v1 = fa[k][1];
                    if (has_bg) {
                        v0 = v0 * insd[k] + fb[k][0] * (1.0f - insd[k]);
                        v1 = v1 * insd[k] + fb[k][1] * (1.0f - insd[k]);
                    }
                } else {
                    v0 = fb[k][0]; v1 = fb[k][1];
                }
                float part = gfa * v0 + gfb * v1;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
                if (lane == 0) s_fd[wave][i] = part;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    float alpha[kE], w[kE], f[kE], Tr[kE], Wb[kE], ins[kE];
    // pass 1: dL/dw_i
#pragma unroll
    for (int e = 0; e < kE; ++e) {
        const int i = kE * lane + e;
        alpha[e] = 0.0f; w[e] = 0.0f; f[e] = 1.0f; Wb[e] = 0.0f; ins[e] = 0.0f;
        if (i < T) {
            const long qt = (long)r * T + i;
            alpha[e] = a.alpha[qt];
            w[e] = a.weights[qt];
            f[e] = 1.0f - alpha[e] + 1e-7f;
            float c0, c1, c2;
            float inside = 0.0f;
            if (i < N) {
                const long q = (long)r * N + i;
                const float mz = a.mid_z[q];
                const float x = o[0] + d[0] * mz, y = o[1] + d[1] * mz, zz = o[2] + d[2] * mz;
                inside = sqrtf(x * x + y * y + zz * zz) < 1.0f ? 1.0f : 0.0f;
                c0 = a.color[q * 3]; c1 = a.color[q * 3 + 1]; c2 = a.color[q * 3 + 2];
                if (has_bg) {
                    c0 = c0 * inside + a.bg_rgb[qt * 3] * (1.0f - inside);
                    c1 = c1 * inside + a.bg_rgb[qt * 3 + 1] * (1.0f - inside);
                    c2 = c2 * inside + a.bg_rgb[qt * 3 + 2] * (1.0f - inside);
                }
            } else {
                c0 = a.bg_rgb[qt * 3]; c1 = a.bg_rgb[qt * 3 + 1]; c2 = a.bg_rgb[qt * 3 + 2];
            }
            ins[e] = inside;
            float wb = gc[0] * (c0 - bgc[0]) + gc[1] * (c1 - bgc[1]) + gc[2] * (c2 - bgc[2]);
            if (a.g_weights != nullptr) wb += a.g_weights[qt];
            if (has_feat) wb += s_fd[wave][i];
            Wb[e] = wb;
        }
    }
    // transmittance (exclusive product) and suffix sums S_i = sum_{j>i} Wb_j w_j
    {
        double loc = 1.0;
#pragma unroll
        for (int e = 0; e < kE; ++e) loc *= (double)f[e];
        double incl = loc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double t = __shfl_up(incl, off);
            if (lane >= off) incl *= t;
        }
        double run = __shfl_up(incl, 1);
        if (lane == 0) run = 1.0;
#pragma unroll
        for (int e = 0; e < kE; ++e) {
            Tr[e] = (float)run;
            run *= (double)f[e];
        }
    }
    double suf[kE];
    {
        double loc = 0.0;
#pragma unroll
        for (int e = 0; e < kE; ++e) loc += (double)Wb[e] * (double)w[e];
        double incl = loc;                       // inclusive suffix over lanes
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const double t = __shfl_down(incl, off);
            if (lane + off < 64) incl += t;
        }
        double run = __shfl_down(incl, 1);       // sum over higher lanes
        if (lane == 63) run = 0.0;
#pragma unroll
        for (int e = kE - 1; e >= 0; --e) {
            suf[e] = run;
            run += (double)Wb[e] * (double)w[e];
        }
    }
    double dvar = 0.0;
    float ddir[3] = {0.0f, 0.0f, 0.0f};        // sum_i d true_cos_i * normal_i  (true_cos = rays_d . normal, renderer.py:265)
#pragma unroll
    for (int e = 0; e < kE; ++e) {
        const int i = kE * lane + e;
        if (i >= T) continue;
        const long qt = (long)r * T + i;
        const float dalpha = Wb[e] * Tr[e] - (float)(suf[e] / (double)f[e]);
        const float wi = w[e];
        float da_in = 0.0f, da_bg = dalpha;
        if (i < N) {
            const long q = (long)r * N + i;
            const float inside = ins[e];
            if (has_bg) {
                da_in = dalpha * inside;
                da_bg = dalpha * (1.0f - inside);
            } else {
                da_in = dalpha;
                da_bg = 0.0f;
            }
            const float cs = has_bg ? inside : 1.0f;
            a.d_color[q * 3] = wi * gc[0] * cs;
            a.d_color[q * 3 + 1] = wi * gc[1] * cs;
            a.d_color[q * 3 + 2] = wi * gc[2] * cs;
            if (has_feat) s_cf[wave][i] = wi * cs;
            // NeuS alpha backward
            const float sdf = a.sdf[q], dist = a.dists[q];
            const float g0 = a.normals[q * 3], g1 = a.normals[q * 3 + 1], g2 = a.normals[q * 3 + 2];
            const float tc = d[0] * g0 + d[1] * g1 + d[2] * g2;
            const float ra = -tc * 0.5f + 0.5f, rb = -tc;
            const float ic = -(fmaxf(ra, 0.0f) * (1.0f - car) + fmaxf(rb, 0.0f) * car);
            const float en = sdf + ic * dist * 0.5f, ep = sdf - ic * dist * 0.5f;
            const float pc = sigmoidf_(ep * inv_s), nc = sigmoidf_(en * inv_s);
            const float raw = ((pc - nc) + 1e-5f) / (pc + 1e-5f);
            const float graw = (raw >= 0.0f && raw <= 1.0f) ? da_in : 0.0f;
            float dpc = graw * (nc / ((pc + 1e-5f) * (pc + 1e-5f)));
            if (a.g_cdf != nullptr) dpc += a.g_cdf[q];           // cdf_fine is prev_cdf itself (renderer.py:276, 322)
            const float dnc = -graw / (pc + 1e-5f);
            const float dzp = dpc * pc * (1.0f - pc), dzn = dnc * nc * (1.0f - nc);    // wrt ep*s, en*s
            const float dep = dzp * inv_s, den = dzn * inv_s;
            dvar += (double)(dzp * ep + dzn * en);
            const float dic = (den - dep) * dist * 0.5f;
            const float dtc = dic * ((ra > 0.0f ? 0.5f * (1.0f - car) : 0.0f) + (rb > 0.0f ? car : 0.0f));
            a.d_sdf[q] = dep + den;
            if (a.d_dists != nullptr) {
                a.d_dists[q] = (den - dep) * ic * 0.5f;          // en/ep = sdf +- iter_cos * dist / 2
                ddir[0] += dtc * g0; ddir[1] += dtc * g1; ddir[2] += dtc * g2;
            }
            // eikonal: d/dn of relax*(|n|-1)^2 / (den+1e-5)
            const float mz = a.mid_z[q];
            const float x = o[0] + d[0] * mz, y = o[1] + d[1] * mz, zz = o[2] + d[2] * mz;
            const float relax = sqrtf(x * x + y * y + zz * zz) < 1.2f ? 1.0f : 0.0f;
            const float gn = sqrtf(g0 * g0 + g1 * g1 + g2 * g2);
            const float ke = gn > 0.0f ? g_eik * relax * 2.0f * (gn - 1.0f) / (gn * eik_den) : 0.0f;
            a.d_normals[q * 3] = dtc * d[0] + ke * g0;
            a.d_normals[q * 3 + 1] = dtc * d[1] + ke * g1;
            a.d_normals[q * 3 + 2] = dtc * d[2] + ke * g2;
        }
        if (has_bg) {
            const float cs = (i < N) ? (1.0f - ins[e]) : 1.0f;
            a.d_bg_rgb[qt * 3] = wi * gc[0] * cs;
            a.d_bg_rgb[qt * 3 + 1] = wi * gc[1] * cs;
            a.d_bg_rgb[qt * 3 + 2] = wi * gc[2] * cs;
            if (has_feat && a.d_bg_feat != nullptr) s_cb[wave][i] = wi * cs;
            // alpha_bg = 1 - exp(-softplus(rho) * dist)
            const float rho_ = a.bg_density[qt], dist = a.bg_dists[qt];
            const float sp = softplus1(rho_);
            const float dsp = rho_ > 20.0f ? 1.0f : sigmoidf_(rho_);
            a.d_bg_density[qt] = da_bg * expf(-sp * dist) * dist * dsp;
            if (a.d_bg_dists != nullptr) a.d_bg_dists[qt] = da_bg * expf(-sp * dist) * sp;
        }
    }
    if (ext_feat) {
        __builtin_amdgcn_wave_barrier();
        float* cf = a.feat_scratch + (long)a.B * T;
        float* cb = cf + (long)a.B * N;
        for (int i = lane; i < T; i += 64) {
            if (i < N) cf[(long)r * N + i] = s_cf[wave][i];
            if (has_bg && a.d_bg_feat != nullptr) cb[(long)r * T + i] = s_cb[wave][i];
        }
    } else if (has_feat) {
        __builtin_amdgcn_wave_barrier();
        for (int i = 0; i < T; ++i) {
            if (i < N) {
                const long q = (long)r * N + i;
                const float cf = s_cf[wave][i];
                if (lane < C) a.d_feat[q * C + lane] = cf * gfa;
                if (lane + 64 < C) a.d_feat[q * C + lane + 64] = cf * gfb;
            }
            if (has_bg && a.d_bg_feat != nullptr) {
                const long qt = (long)r * T + i;
                const float cb = s_cb[wave][i];
                if (lane < C) a.d_bg_feat[qt * C + lane] = cb * gfa;
                if (lane + 64 < C) a.d_bg_feat[qt * C + lane + 64] = cb * gfb;
            }
        }
    }
    dvar = wsum_d(dvar);
    if (lane == 0) a.d_var_partial[r] = s_unclipped ? (float)(dvar * 10.0 * (double)inv_s) : 0.0f;
    if (a.d_dir_cos != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float s = (float)wsum_d((double)ddir[k]);
            if (lane == 0) a.d_dir_cos[r * 3 + k] = s;
        }
    }
}

__global__ __launch_bounds__(kRW * 64) void composite_bwd_kernel(CompositeBwdArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRW + wave;
    if (r >= a.B) return;
    CompositeBwdOvr ov;
    ov.on = false;
    composite_bwd_row(a, r, wave, lane, ov);
}

// renderer.py:262-315 forward, the colour term's gradient (dpt_runner.py:228-229, mask = 1) and the adjoint of the compositor
// for one ray in ONE launch: the training step's plain configuration (no mask loss, no depth-feature loss, one rank) needs no
// global quantity between the three but the eikonal term's denominator - the number of inside samples within the relaxed
// sphere, which is exactly the length of the foreground work list (vdn_foreground_active uses the compositor's own norm
// expression). The loss SCALARS (logging) are reduced afterwards by the usual kernels, off the critical path. Same device
// functions as composite_kernel / composite_bwd_kernel and loss_kernel's expressions: bit-identical adjoints.
__global__ __launch_bounds__(kRW * 64) void composite_train_kernel(CompositeArgs fa, CompositeBwdArgs ba, const float* true_rgb, float* g_color,
                                                                   const int32_t* fg_count, float igr_weight, float grad_scale) {
    __shared__ float s_w[kRW][kMaxTB], s_in[kRW][kMaxTB];
    __shared__ __attribute__((aligned(16))) char s_dump[kRW][1024];       // the code warm-up's LDS-DMA lands here (vdn_common.h); never read
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRW + wave;
    if (r >= fa.B) return;
    // (the launch sits alone on the step's critical path between the forward and the backward kernels, on caches full of their
    // planes: its 47 KB of code arrive as data while the ray's planes are loaded - vdn_common.h)
    warm_code_issue(kWarmCodeCompositeTrain, gridDim.x, 2048, s_dump[wave]);
    const RowOut ro = composite_row(fa, r, lane, CompositeGlobalSrc{fa.sdf, fa.normals, fa.color}, s_w[wave], s_in[wave]);
    CompositeBwdOvr ov;
    ov.on = true;
    const float mask_sum = (float)fa.B + 1e-5f;                       // dpt_runner.py:213 with mask = ones
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float m = 1.0f;
        const float diff = ro.c[k] - true_rgb[r * 3 + k];
        const float e = diff * m;
        const float sgn = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f);
        ov.gc[k] = sgn * m / mask_sum * grad_scale;
        if (lane == 0) g_color[r * 3 + k] = ov.gc[k];
    }
    ov.g_eik = igr_weight;
    ov.eik_den = (float)(*fg_count) + 1e-5f;
    composite_bwd_row(ba, r, wave, lane, ov);
    warm_l2_wait();
}

// composite_bwd_kernel with the upstream gradients made on the spot, as in composite_train_kernel: the colour term's from the colour
// the forward left in memory (loss_kernel's expressions, mask = 1), the eikonal denominator from the foreground work list's length.
// The feature channels' gradient a.g_feat was written by the forward (vdn_composite_fwd_train).
__global__ __launch_bounds__(kRW * 64) void composite_bwd_train_kernel(CompositeBwdArgs a, const float* color_out, const float* true_rgb,
                                                                       float* g_color, const int32_t* fg_count, float igr_weight, float grad_scale) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRW + wave;
    if (r >= a.B) return;
    CompositeBwdOvr ov;
    ov.on = true;
    const float mask_sum = (float)a.B + 1e-5f;                        // dpt_runner.py:213 with mask = ones
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float m = 1.0f;
        const float diff = color_out[r * 3 + k] - true_rgb[r * 3 + k];
        const float e = diff * m;
        const float sgn = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f);
        ov.gc[k] = sgn * m / mask_sum * grad_scale;
        if (lane == 0) g_color[r * 3 + k] = ov.gc[k];
    }
    ov.g_eik = igr_weight;
    ov.eik_den = (float)(*fg_count) + 1e-5f;
    composite_bwd_row(a, r, wave, lane, ov);
}

// Adjoint of the ray geometry (include/vdn_render.h: VdnRayAdjointArgs), one wave per ray, sample i = kE * lane + e.
//   d o = sum_i d pts_i;  d d = sum_i (d pts_i mid_i + d dirs_i) + d dir_cos;  d mid_i = d pts_i . d
//   mid_i = (z_i + z_{i+1}) / 2 and dists_i = z_{i+1} - z_i for i < n-1;  mid_{n-1} = z_{n-1} + sample_dist / 2, dists_{n-1} const
//   =>  d z_i = (d mid_i + d mid_{i-1}) / 2 - d dists_i + d dists_{i-1}      (terms with index -1 absent; i = n-1: d mid_i whole)
__global__ __launch_bounds__(kRW * 64) void ray_adjoint_kernel(RayAdjointArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * kRW + wave;
    if (r >= a.B) return;
    const float d0 = a.rays_d[r * 3], d1 = a.rays_d[r * 3 + 1], d2 = a.rays_d[r * 3 + 2];
    double so[3] = {0.0, 0.0, 0.0}, sd[3] = {0.0, 0.0, 0.0};
    auto pass = [&](const float* dpts, const float* ddirs, const float* ddists, const float* mid, int n, float* dz_a, int n_a,
                    float* dz_b) VDN_INL {
        float dm[kE], dl[kE];
#pragma unroll
        for (int e = 0; e < kE; ++e) {
            const int i = kE * lane + e;
            dm[e] = 0.0f; dl[e] = 0.0f;
            if (i < n) {
                const long q = (long)r * n + i;
                const float px = dpts[q * 3], py = dpts[q * 3 + 1], pz = dpts[q * 3 + 2];
                const float m = mid[q];
                so[0] += px; so[1] += py; so[2] += pz;
                sd[0] += (double)(px * m + ddirs[q * 3]); sd[1] += (double)(py * m + ddirs[q * 3 + 1]); sd[2] += (double)(pz * m + ddirs[q * 3 + 2]);
                dm[e] = px * d0 + py * d1 + pz * d2;
                dl[e] = ddists[q];
            }
        }
        // values of sample i-1: the previous element, or the previous lane's last element
        const float pm = __shfl_up(dm[kE - 1], 1), pl = __shfl_up(dl[kE - 1], 1);
#pragma unroll
        for (int e = 0; e < kE; ++e) {
            const int i = kE * lane + e;
            if (i >= n) continue;
            const float dm_prev = i == 0 ? 0.0f : (e == 0 ? pm : dm[e - 1]);
            const float dl_prev = i == 0 ? 0.0f : (e == 0 ? pl : dl[e - 1]);
            const float v = i == n - 1 ? dm[e] + 0.5f * dm_prev + dl_prev : 0.5f * (dm[e] + dm_prev) - dl[e] + dl_prev;
            if (i < n_a) dz_a[(long)r * n_a + i] += v;
            else if (dz_b != nullptr) dz_b[(long)r * (n - n_a) + (i - n_a)] += v;
        }
    };
    const int N = a.N, T = a.T;
#pragma unroll
    for (int e = 0; e < kE; ++e) {
        const int i = kE * lane + e;
        if (i < N) a.d_z[(long)r * N + i] = 0.0f;
        if (a.d_z_out != nullptr && i >= N && i < T) a.d_z_out[(long)r * (T - N) + (i - N)] = 0.0f;
    }
    pass(a.d_pts, a.d_dirs, a.d_dists, a.mid_z, N, a.d_z, N, nullptr);
    if (a.d_bg_pts != nullptr) pass(a.d_bg_pts, a.d_bg_dirs, a.d_bg_dists, a.bg_mid, T, a.d_z, N, a.d_z_out);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double o = wsum_d(so[k]), dd = wsum_d(sd[k]);
        if (lane == 0) {
            a.d_rays_o[r * 3 + k] = (float)o;
            a.d_rays_d[r * 3 + k] = (float)dd + a.d_dir_cos[r * 3 + k];
        }
    }
}

__global__ void variance_reduce_kernel(const float* partial, int B, float* out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < B; i += 64) s += (double)partial[i];
    s = wsum_d(s);
    if (threadIdx.x == 0) out[0] = (float)s;
}

}  // namespace vdn

extern "C" int vdn_alpha_composite_bwd(const VdnCompositeBwdArgs* a, void* stream) {
    using namespace vdn;
    if (!a || a->B <= 0 || a->N <= 0 || a->T < a->N || a->T > kMaxTB) return -1;
    if (!a->rays_o || !a->rays_d || !a->sdf || !a->normals || !a->dists || !a->mid_z || !a->color || !a->variance ||
        !a->alpha || !a->weights || !a->eik) return -2;
    if (!a->d_sdf || !a->d_normals || !a->d_color || !a->d_var_partial) return -3;
    if (a->T > a->N && (!a->bg_density || !a->bg_rgb || !a->bg_dists || !a->d_bg_density || !a->d_bg_rgb)) return -4;
    if (a->d_feat && (!a->feat || a->feat_ch <= 0 || a->feat_ch > 128)) return -5;
    if ((a->d_dists != nullptr) != (a->d_dir_cos != nullptr) || (a->d_bg_dists && !a->d_dists)) return -6;
    const bool ext_feat = a->d_feat && a->g_feat && a->feat_scratch;
    const int row_blocks = (int)(((long)a->B * a->T + 3) / 4);
    if (ext_feat) hipLaunchKernelGGL(feat_dot_kernel, dim3(row_blocks), dim3(256), 0, (hipStream_t)stream, *a);
    hipLaunchKernelGGL(composite_bwd_kernel, dim3((a->B + kRW - 1) / kRW), dim3(kRW * 64), 0, (hipStream_t)stream, *a);
    if (ext_feat) hipLaunchKernelGGL(feat_outer_kernel, dim3(row_blocks), dim3(256), 0, (hipStream_t)stream, *a);
    if (a->d_variance != nullptr)
        hipLaunchKernelGGL(variance_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->d_var_partial, a->B, a->d_variance);
    return (int)hipGetLastError();
}

extern "C" int vdn_composite_train(const VdnCompositeArgs* f, const VdnCompositeBwdArgs* b, const float* true_rgb, float* g_color,
                                   const int32_t* fg_count, float igr_weight, float grad_scale, void* stream) {
    using namespace vdn;
    if (!f || !b || !true_rgb || !g_color || !fg_count) return -1;
    if (f->B <= 0 || f->N <= 0 || f->T < f->N || f->T > kMaxTB || b->B != f->B || b->N != f->N || b->T != f->T) return -1;
    if (!f->rays_o || !f->rays_d || !f->sdf || !f->normals || !f->dists || !f->mid_z || !f->color || !f->variance) return -2;
    if (!f->weights || !f->alpha_out || !f->cdf || !f->inside_sphere || !f->color_out || !f->weight_sum || !f->weight_max || !f->eik_partial) return -3;
    if (f->T > f->N && (!f->bg_density || !f->bg_rgb || !f->bg_dists || !b->d_bg_density || !b->d_bg_rgb)) return -4;
    // the plain configuration only: no feature channels, no extra upstream gradients, no ray adjoints
    if (f->feat_out || b->d_feat || b->g_feat || b->g_weights || b->g_cdf || b->d_dists || b->d_dir_cos) return -10;
    if (b->alpha != f->alpha_out || b->weights != f->weights || !b->d_sdf || !b->d_normals || !b->d_color || !b->d_var_partial) return -3;
    hipLaunchKernelGGL(composite_train_kernel, dim3((f->B + kRW - 1) / kRW), dim3(kRW * 64), 0, (hipStream_t)stream, *f, *b, true_rgb,
                       g_color, fg_count, igr_weight, grad_scale);
    if (b->d_variance != nullptr)
        hipLaunchKernelGGL(variance_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, b->d_var_partial, b->B, b->d_variance);
    return (int)hipGetLastError();
}

extern "C" int vdn_composite_bwd_train(const VdnCompositeBwdArgs* a, const float* color_out, const float* true_rgb, float* g_color,
                                       const int32_t* fg_count, float igr_weight, float grad_scale, void* stream) {
    using namespace vdn;
    if (!a || !color_out || !true_rgb || !g_color || !fg_count || a->B <= 0 || a->N <= 0 || a->T < a->N || a->T > kMaxTB) return -1;
    if (!a->rays_o || !a->rays_d || !a->sdf || !a->normals || !a->dists || !a->mid_z || !a->color || !a->variance ||
        !a->alpha || !a->weights) return -2;
    if (!a->d_sdf || !a->d_normals || !a->d_color || !a->d_var_partial) return -3;
    if (a->T > a->N && (!a->bg_density || !a->bg_rgb || !a->bg_dists || !a->d_bg_density || !a->d_bg_rgb)) return -4;
    if (a->d_feat && (!a->feat || a->feat_ch <= 0 || a->feat_ch > 128)) return -5;
    // the Trainer's configuration only: no extra upstream gradients, no ray adjoints
    if (a->g_weights || a->g_cdf || a->d_dists || a->d_dir_cos) return -10;
    const bool ext_feat = a->d_feat && a->g_feat && a->feat_scratch;
    const int row_blocks = (int)(((long)a->B * a->T + 3) / 4);
    if (ext_feat) hipLaunchKernelGGL(feat_dot_kernel, dim3(row_blocks), dim3(256), 0, (hipStream_t)stream, *a);
    hipLaunchKernelGGL(composite_bwd_train_kernel, dim3((a->B + kRW - 1) / kRW), dim3(kRW * 64), 0, (hipStream_t)stream, *a, color_out, true_rgb,
                       g_color, fg_count, igr_weight, grad_scale);
    if (ext_feat) hipLaunchKernelGGL(feat_outer_kernel, dim3(row_blocks), dim3(256), 0, (hipStream_t)stream, *a);
    if (a->d_variance != nullptr)
        hipLaunchKernelGGL(variance_reduce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->d_var_partial, a->B, a->d_variance);
    return (int)hipGetLastError();
}

extern "C" int vdn_ray_adjoint(const VdnRayAdjointArgs* a, void* stream) {
    using namespace vdn;
    if (!a || a->B <= 0 || a->N <= 0 || a->T < a->N || a->T > kMaxTB) return -1;
    if (!a->rays_d || !a->mid_z || !a->d_pts || !a->d_dirs || !a->d_dists || !a->d_dir_cos || !a->d_rays_o || !a->d_rays_d || !a->d_z) return -2;
    if (a->T > a->N && a->d_bg_pts && (!a->bg_mid || !a->d_bg_dirs || !a->d_bg_dists || !a->d_z_out)) return -3;
    hipLaunchKernelGGL(ray_adjoint_kernel, dim3((a->B + kRW - 1) / kRW), dim3(kRW * 64), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}
