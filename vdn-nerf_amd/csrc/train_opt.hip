// Loss + output gradients (dpt_runner.py:208-243) and Adam (dpt_runner.py:144,254) as single launches.
#include "vdn_common.h"
#include "vdn_kernels.h"

namespace vdn {

__device__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}

// one block; rays strided over threads
__global__ __launch_bounds__(1024) void loss_kernel(LossArgs a) {
    __shared__ double sh[16];
    double msum = 0.0;
    for (int r = threadIdx.x; r < a.B; r += blockDim.x) msum += a.mask ? (double)a.mask[r] : 1.0;
    const float mask_sum = (float)block_sum(msum, sh) + 1e-5f;                 // dpt_runner.py:213
    double l1 = 0.0, sq = 0.0, dl1 = 0.0, bce = 0.0;
    for (int r = threadIdx.x; r < a.B; r += blockDim.x) {
        const float m = a.mask ? a.mask[r] : 1.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float diff = a.color[r * 3 + k] - a.true_rgb[r * 3 + k];
            const float e = diff * m;
            l1 += (double)fabsf(e);
            sq += (double)(diff * diff * m);
            const float sgn = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f);
            a.g_color[r * 3 + k] = sgn * m / mask_sum * a.grad_scale;
        }
        if (a.mask_weight != 0.0f && a.g_weights != nullptr) {
            float ws = 0.0f;
            for (int i = 0; i < a.T; ++i) ws += a.weights[(long)r * a.T + i];
            const float wc = fminf(fmaxf(ws, 1e-3f), 1.0f - 1e-3f);
            bce += -(double)(m * logf(wc) + (1.0f - m) * logf(1.0f - wc));
            const bool inside = ws >= 1e-3f && ws <= 1.0f - 1e-3f;
            const float gws = inside ? (-m / wc + (1.0f - m) / (1.0f - wc)) / (float)a.B * a.mask_weight * a.grad_scale : 0.0f;
            for (int i = 0; i < a.T; ++i) a.g_weights[(long)r * a.T + i] = gws;
        }
    }
    if (a.feats != nullptr && a.g_feats != nullptr) {
        // the depth-feature term (dpt_runner.py:239-243) over the flat [B * C] index: consecutive threads read consecutive
        // addresses (a thread per ray walked its own 384-byte row: 123 us for 512 x 96 values)
        const long n = (long)a.B * a.C;
        for (long base = threadIdx.x; base < n; base += 4L * blockDim.x) {
            float fv[4], gv[4], mv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {           // four independent load pairs in flight
                const long idx = base + (long)k * blockDim.x;
                const bool in = idx < n;
                fv[k] = in ? a.feats[idx] : 0.0f;
                gv[k] = in ? a.gt_feats[idx] : 0.0f;
                mv[k] = in ? (a.mask ? a.mask[(int)(idx / a.C)] : 1.0f) : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const long idx = base + (long)k * blockDim.x;
                if (idx >= n) break;
                const float e = (fv[k] - gv[k]) * mv[k];
                dl1 += (double)fabsf(e);
                const float sgn = e > 0.0f ? 1.0f : (e < 0.0f ? -1.0f : 0.0f);
                a.g_feats[idx] = sgn * mv[k] / mask_sum * a.depth_weight * a.grad_scale;
            }
        }
    }
    l1 = block_sum(l1, sh);
    sq = block_sum(sq, sh);
    dl1 = block_sum(dl1, sh);
    bce = block_sum(bce, sh);
    if (threadIdx.x == 0) {
        const float color_loss = (float)l1 / mask_sum;
        const float psnr = 20.0f * log10f(1.0f / sqrtf((float)sq / (mask_sum * 3.0f)));     // dpt_runner.py:230
        const float eik = a.eik[0];
        const float depth_loss = (float)dl1 / mask_sum;
        const float mask_loss = (float)bce / (float)a.B;
        float loss = color_loss + eik * a.igr_weight;
        if (a.mask_weight != 0.0f) loss += mask_loss * a.mask_weight;
        if (a.feats != nullptr) loss += depth_loss * a.depth_weight;
        a.out_scalars[0] = loss; a.out_scalars[1] = color_loss; a.out_scalars[2] = psnr;
        a.out_scalars[3] = eik; a.out_scalars[4] = depth_loss; a.out_scalars[5] = mask_loss;
        a.g_eik[0] = a.igr_weight;
    }
}

// elements [b0, e0) and [b1, e1) of the flat buffers (two ranges: parameters that share a step count need not be contiguous)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long b0, long e0, long b1r, long e1, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt) {
    const float step_size = lr / bc1;
    const long n0 = e0 - b0, n = n0 + (e1 - b1r);
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) {
        const long i = j < n0 ? b0 + j : b1r + (j - n0);
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);           // lerp form, as torch's _single_tensor_adam
        const float vi = v[i] * b2 + gi * gi * (1.0f - b2);
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);
    }
}

}  // namespace vdn

extern "C" int vdn_loss_fwd_bwd(const VdnLossArgs* a, void* stream) {
    if (!a || a->B <= 0 || !a->color || !a->true_rgb || !a->eik || !a->g_color || !a->g_eik || !a->out_scalars) return -1;
    if (a->feats && (!a->gt_feats || a->C <= 0)) return -2;
    if (a->mask_weight != 0.0f && (!a->weights || !a->g_weights || a->T <= 0)) return -3;
    hipLaunchKernelGGL(vdn::loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *a);
    return (int)hipGetLastError();
}

extern "C" int vdn_adam_step_ranges(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t begin0, int64_t end0,
                                    int64_t begin1, int64_t end1, float lr, float beta1, float beta2, float eps, int32_t step, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || begin0 < 0 || end0 < begin0 || begin1 < 0 || end1 < begin1 || step < 1) return -1;
    const int64_t n = (end0 - begin0) + (end1 - begin1);
    if (n <= 0) return -1;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(vdn::adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (long)begin0, (long)end0, (long)begin1, (long)end1, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2));
    return (int)hipGetLastError();
}

extern "C" int vdn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                             float lr, float beta1, float beta2, float eps, int32_t step, void* stream) {
    return vdn_adam_step_ranges(param, grad, exp_avg, exp_avg_sq, 0, n, 0, 0, lr, beta1, beta2, eps, step, stream);
}
