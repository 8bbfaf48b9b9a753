// Weight preparation kernels: weight-norm materialisation and MFMA chunk images.
// Stands in for torch.nn.utils.weight_norm's per-forward recompute (reference fields.py:65-66,
// 141-142): done once per optimizer step here.
#include "vdn_common.h"
#include "vdn_kernels.h"

namespace vdn {

// one wave per row: w_eff[r,:] = v[r,:] * (g[r] / ||v[r,:]||)
__global__ void weightnorm_kernel(const WeightNormDesc* descs, int n_layers) {
    const WeightNormDesc d = descs[blockIdx.x];
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= d.rows) return;
    const float* v = d.v + (long)row * d.cols;
    float* w = d.w_eff + (long)row * d.cols;
    if (d.g == nullptr) {
        for (int c = lane; c < d.cols; c += 64) w[c] = v[c];
        return;
    }
    float ss = 0.0f;
    for (int c = lane; c < d.cols; c += 64) ss += v[c] * v[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float inv = 1.0f / sqrtf(ss);
    const float s = d.g[row] * inv;
    for (int c = lane; c < d.cols; c += 64) w[c] = v[c] * s;
    if (lane == 0 && d.inv_norm != nullptr) d.inv_norm[row] = inv;
}

__device__ inline unsigned short f32_to_bf16_rn(float f) {
    // round-to-nearest-even on the f32 bits (weights are finite)
    const unsigned u = __float_as_uint(f);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// kImgSplit blocks per chunk (a chunk is 1 - 1.2 K 16-byte items behind two dependent index loads each: one 256-thread block per
// chunk left the launch latency-bound at ~280 blocks - 15.9 us for the SDF network's 4 MB of images)
constexpr int kImgSplit = 4;
__global__ void build_images_kernel(const ChunkDesc* descs) {
    const ChunkDesc d = descs[blockIdx.x];
    const int tid = blockIdx.y * blockDim.x + threadIdx.x, nthr = gridDim.y * blockDim.x;
    const int kt = d.k_pad / 32;
    const int kt0 = d.kt_count > 0 ? d.kt_begin : 0;
    const int ktn = d.kt_count > 0 ? d.kt_count : kt;
    auto val = [&](int i, int k) -> float {
        const int r = d.nmap[d.n0 + i];
        const int c = d.kmap[k];
        return (r < 0 || c < 0) ? 0.0f : d.scale * d.src[(long)r * d.row_stride + (long)c * d.col_stride];
    };
    if (d.fmt == 0) {
        // [kt*4 groups][64 lanes][4] f32, then 32 bias floats, zero pad to 1 KiB
        float* out = reinterpret_cast<float*>(d.dst) + (long)kt0 * 1024;
        const int n4 = ktn * 4 * 64;
        for (int idx = tid; idx < n4; idx += nthr) {
            const int g = idx >> 6, lane = idx & 63, i = lane & 31, h = lane >> 5;
            float4 o;
            o.x = val(i, 8 * g + 4 * h + 0);
            o.y = val(i, 8 * g + 4 * h + 1);
            o.z = val(i, 8 * g + 4 * h + 2);
            o.w = val(i, 8 * g + 4 * h + 3);
            reinterpret_cast<float4*>(out)[idx] = o;
        }
        float* b = reinterpret_cast<float*>(d.dst) + (long)kt * 1024;
        for (int i = tid; i < 256 && d.write_bias; i += nthr) {
            float bv = 0.0f;
            if (i < 32 && d.bias != nullptr) {
                const int r = d.nmap[d.n0 + i];
                if (r >= 0) bv = d.bias_scale * d.bias[r];
            }
            b[i] = bv;
        }
    } else {
        // bf16: [kt*2 k-steps][64 lanes][8 bf16] (lane (i,h), element j: k = 16s + 8(j>>2) + 4h + (j&3)),
        // then 32 f32 bias, zero pad to 1 KiB
        unsigned short* out = reinterpret_cast<unsigned short*>(d.dst) + (long)kt0 * 1024;
        const int n8 = ktn * 2 * 64;
        for (int idx = tid; idx < n8; idx += nthr) {
            const int s = idx >> 6, lane = idx & 63, i = lane & 31, h = lane >> 5;
            unsigned short o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f32_to_bf16_rn(val(i, 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)));
            uint4 pk;
            pk.x = o[0] | ((unsigned)o[1] << 16);
            pk.y = o[2] | ((unsigned)o[3] << 16);
            pk.z = o[4] | ((unsigned)o[5] << 16);
            pk.w = o[6] | ((unsigned)o[7] << 16);
            reinterpret_cast<uint4*>(out)[idx] = pk;
        }
        float* b = reinterpret_cast<float*>(d.dst + (long)kt * 2048);
        for (int i = tid; i < 256 && d.write_bias; i += nthr) {
            float bv = 0.0f;
            if (i < 32 && d.bias != nullptr) {
                const int r = d.nmap[d.n0 + i];
                if (r >= 0) bv = d.bias_scale * d.bias[r];
            }
            b[i] = bv;
        }
    }
    if (d.tail != nullptr && d.write_bias) {
        float* t = reinterpret_cast<float*>(d.dst + d.tail_off);
        const long ts = d.tail_stride > 1 ? d.tail_stride : 1;
        for (int i = tid; i < d.tail_n; i += nthr) t[i] = d.tail[i * ts];
    }
}

}  // namespace vdn

namespace vdn {
// one thread per (point, input component): x, then sin / cos of every octave (ocml sincosf: the fp32 parity path's trig)
__global__ void posenc_kernel(const float* __restrict__ in, float* __restrict__ out, long P, int d, int L) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * d) return;
    const long p = i / d;
    const int c = (int)(i - p * d);
    const float x = in[i];
    float* o = out + p * (long)(d * (1 + 2 * L));
    o[c] = x;
    for (int k = 0; k < L; ++k) {
        float s, co;
        sincosf(x * (float)(1 << k), &s, &co);
        o[d + 2 * d * k + c] = s;
        o[d + 2 * d * k + d + c] = co;
    }
}
}  // namespace vdn

extern "C" int vdn_posenc(const float* in, float* out, int64_t P, int32_t d, int32_t n_freqs, void* stream) {
    if (!in || !out || P <= 0 || d <= 0 || n_freqs < 0 || n_freqs > 24) return -1;
    const long n = (long)P * d;
    hipLaunchKernelGGL(vdn::posenc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, (long)P, d, n_freqs);
    return (int)hipGetLastError();
}

extern "C" int vdn_abi_version(void) { return VDN_ABI_VERSION; }

extern "C" int vdn_weightnorm_materialize(const VdnWeightNormDesc* descs_dev, int n_layers, int max_rows, void* stream) {
    if (descs_dev == nullptr || n_layers <= 0 || max_rows <= 0) return -1;
    dim3 grid(n_layers, (max_rows + 3) / 4);
    hipLaunchKernelGGL(vdn::weightnorm_kernel, grid, dim3(256), 0, (hipStream_t)stream, descs_dev, n_layers);
    return (int)hipGetLastError();
}

extern "C" int vdn_build_images(const VdnChunkDesc* descs_dev, int n_chunks, void* stream) {
    if (descs_dev == nullptr || n_chunks <= 0) return -1;
    hipLaunchKernelGGL(vdn::build_images_kernel, dim3(n_chunks, vdn::kImgSplit), dim3(256), 0, (hipStream_t)stream, descs_dev);
    return (int)hipGetLastError();
}
