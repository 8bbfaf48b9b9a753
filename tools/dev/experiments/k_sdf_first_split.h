// EXPERIMENT (round 5, measured, not shipped): the sampler's first SDF pass with the features split over 8 waves and four 32-point
// blocks per workgroup (sdf_first_split_kernel), in place of k_sdf_fwd2.h's MODE 0 kernel. Bit-identical values
// (tests/test_gpu_parity.py::test_fused_sampler_rounds_equal_the_two_launch_rounds passed with it) and NOT faster: same-box A/B, three
// alternating runs (tools/dev/sampler_probe.py): sampler 156.3 / 156.5 / 156.2 us against 154.1 / 159.4 / 155.1 us with MODE 0,
// render() 1.342 - 1.344 against 1.344 - 1.353 M rays/s, training step 1 107 - 1 112 against 1 102 - 1 108 us. Why: per CU and layer both
// forms move the same 512 KiB through LDS (MODE 0: every wave reads all weights; here: every wave reads all four blocks'
// activations) against 128 B / clk = 4 096 cycles, the same as the layer's MFMA time on a SIMD - the pass is co-bound by the LDS
// array and the matrix pipe, not by the latency of one wave's chain. Halving the LDS traffic needs each activation fragment to feed
// two MFMAs (4 waves x 2 tiles, weights for two tiles in registers: 288 registers double-buffered), i.e. one wave per SIMD again.
// This fragment was part of csrc/k_sdf_fwd0_split.h (namespace vdn::sdf0s) behind vdn_sdf_upsample_bf16.
// ---------------------------------------------------------------------------------------------------------------------------
// The sampler's FIRST pass (renderer.py:369-370: the sdf at the 64 coarse samples of every ray) + first up-sampling round, the
// same way (round 5): one workgroup = two rays = 128 points = FOUR 32-point blocks, 8 waves, wave w = output tile w of every
// layer for all four blocks - its layer's weights are read once into registers and serve 64 MFMAs. 512 rays = 256 workgroups =
// one per CU, as with k_sdf_fwd2.h's MODE 0 kernel (4 waves, every wave all 8 tiles of its 32 points: 128 dependent MFMAs and
// 128 softplus values per lane and layer on ONE wave per SIMD, 31 us of latency for the pass); here a SIMD runs two waves of 64
// MFMAs per layer each, and one wave's epilogue issues beside the other's MFMAs.
//  * per block the LDS image of sdf_fwd0_split_kernel (encoded input 4 KiB + two 16-KiB activation buffers): 4 x 36 KiB;
//  * the last layer's sdf row is ONE f32 fma chain per point over the 256 unrounded activations of layer 7, in MODE 0's order
//    (tile by tile): tile T's activations are in wave T's registers, so the running sum is RELAYED through LDS from wave 0 to
//    wave 7 (8 stages, a workgroup barrier between them; 256 B per block) instead of parking 4 x 32 KiB of f32 activations;
//  * same weight stream, same arithmetic operation for operation: bit-identical sdf (tests/test_gpu_parity.py), and the
//    up-sampling round behind it is upsample_row on the two rays' rows in LDS, as in MODE 0's UPS form.
namespace first {
constexpr int kNB = 4;
constexpr int kBlk = kW8;                       // one block's image: encoded input + the two activation buffers (36 KiB)
constexpr int kW8f = kNB * kBlk;                // row 0 of the last layer, 256 f32
constexpr int kRelay = kW8f + 1024;             // the running sdf sums: [block][lane] f32
constexpr int kLdsF = kRelay + kNB * 256;
constexpr int kRowsF = 0;                       // UPS: per ray 3 x kMaxT floats (z, sdf, cdf) where block 0's image was
}  // namespace first

template <bool UPS>
__global__ __launch_bounds__(kWaves * 64, 1) void sdf_first_split_kernel(SdfArgs a, UpsampleArgs up) {
    using P = BF16;
    using namespace first;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
    const long n_wg = (a.P + 127) / 128;
    if (a.cold_start) {         // (vdn_common.h; wave-uniform condition: every wave takes the barrier)
        warm_l2_issue(a.blob, PG::total * sdf2::kStride, n_wg, 256, smem + wave * 1024);
        warm_l2_sync();
    }
    WSet WA, WB;                                    // even / odd layers
    load_weights<0>(WA, a.blob, wave, lane);
    load_weights<1>(WB, a.blob, wave, lane);
    __builtin_amdgcn_sched_barrier(0);

    // the encoded input: wave w writes tile (w & 1) of block (w >> 1); point = 128 * workgroup + 32 * block + c
    float b0 = 0.0f;
    {
        const int blk = wave >> 1;
        long pd = (long)blockIdx.x * 128 + blk * 32 + c;
        if (pd >= a.P) pd = a.P - 1;                // (a ragged last workgroup works on duplicates and stores nothing for them)
        const long r = pd / a.n_per_ray;
        const long sidx = pd - r * a.n_per_ray;
        const float z = a.z[r * a.z_ld + sidx];
        float xin[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) xin[d] = (a.rays_o[r * 3 + d] + a.rays_d[r * 3 + d] * z) * a.scale;
        float pe39[39], pe[64];
        posenc<3, 6, false>(xin, pe39);
        // scaled units and the bf16 residue slots, exactly as k_sdf_fwd2.h forms them
#pragma unroll
        for (int i = 0; i < 39; ++i) pe[i] = pe39[i] * sdf2::kC1;
#pragma unroll
        for (int i = 0; i < 25; ++i) pe[39 + i] = fmaf(pe39[i], sdf2::kC1, -bf16_lo(pack_bf16x2(pe[i], 0.0f)));      // (the residue of the exact product)
        typename P::template Act<1> X;
        X.set(0, (wave & 1) == 0 ? vals_tile<64>(pe, h, 0) : vals_tile<64>(pe, h, 1));
        char* pb = smem + blk * kBlk + kPeb;
        *reinterpret_cast<bf16x8*>(pb + (2 * (wave & 1)) * 1024 + lane * 16) = X.r[0];
        *reinterpret_cast<bf16x8*>(pb + (2 * (wave & 1) + 1) * 1024 + lane * 16) = X.r[1];
        if (wave == 7) b0 = *reinterpret_cast<const float*>(a.blob + (long)PG::first_chunk(8) * sdf2::kStride + 8 * 2048);
        if (wave == 2) {                            // row 0 of W8 (f32, in every chunk's tail) -> LDS
            const f32x4 v = *(reinterpret_cast<const f32x4*>(a.blob + sdf2::kTail) + lane);
            *reinterpret_cast<f32x4*>(smem + kW8f + lane * 16) = v;
        }
    }
    lds_barrier();

    f32x16 g7[kNB];                                 // layer 7's activations of this wave's tile (f32, unrounded), per block
    auto hidden = [&](auto l_c, WSet& W) VDN_INL {
        constexpr int L = decltype(l_c)::value;
        using IO = LayerIO<L>;
        static_for<kNB>([&](auto b_c) VDN_INL {
            constexpr int blk = decltype(b_c)::value;
            char* sb = smem + blk * kBlk;
            const f32x16 acc = layer_mma<L>(W, sb, lane);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (blk == kNB - 1 && L + 2 <= 7) {
                load_weights<L + 2>(W, a.blob, wave, lane);      // this set is free again
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (L < 7) {
                u32x4 o[2];
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) {
                    const float g0 = sdf2::softplus_sigma(acc[2 * pr]).g, g1 = sdf2::softplus_sigma(acc[2 * pr + 1]).g;
                    o[pr >> 2][pr & 3] = pack_bf16x2(g0, g1);
                }
                if (wave < IO::nt) {
                    *reinterpret_cast<u32x4*>(sb + IO::out_base + (2 * wave) * 1024 + lane * 16) = o[0];
                    *reinterpret_cast<u32x4*>(sb + IO::out_base + (2 * wave + 1) * 1024 + lane * 16) = o[1];
                }
            } else {
#pragma unroll
                for (int t = 0; t < 16; ++t) g7[blk][t] = sdf2::softplus_sigma(acc[t]).g;
            }
        });
        lds_barrier();
    };
    hidden(std::integral_constant<int, 0>{}, WA);
    hidden(std::integral_constant<int, 1>{}, WB);
    hidden(std::integral_constant<int, 2>{}, WA);
    hidden(std::integral_constant<int, 3>{}, WB);
    hidden(std::integral_constant<int, 4>{}, WA);
    hidden(std::integral_constant<int, 5>{}, WB);
    hidden(std::integral_constant<int, 6>{}, WA);
    hidden(std::integral_constant<int, 7>{}, WB);

    // sdf = W8[0,:] . h8 + b8[0]: MODE 0's f32 chain (k_sdf_fwd2.h, layer 7's epilogue), tile by tile, pair by pair - stage T is
    // wave T's, the running sums travel through LDS
    float* relay = reinterpret_cast<float*>(smem + kRelay);
    float dots[kNB];
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        if (wave == T) {
#pragma unroll
            for (int blk = 0; blk < kNB; ++blk) {
                float sdf_dot = T == 0 ? 0.0f : relay[blk * 64 + lane];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 w8 = *reinterpret_cast<const f32x4*>(smem + kW8f + (8 * T + 2 * q + h) * 16);
                    sdf_dot = fmaf(g7[blk][4 * q + 0], w8[0], fmaf(g7[blk][4 * q + 1], w8[1], sdf_dot));
                    sdf_dot = fmaf(g7[blk][4 * q + 2], w8[2], fmaf(g7[blk][4 * q + 3], w8[3], sdf_dot));
                }
                dots[blk] = sdf_dot;
                if (T < 7) relay[blk * 64 + lane] = sdf_dot;
            }
        }
        if (T < 7) lds_barrier();
    }
    float* rows = reinterpret_cast<float*>(smem + kRowsF);
    if (wave == 7) {
#pragma unroll
        for (int blk = 0; blk < kNB; ++blk) {
            const float dot = dots[blk] + __shfl_xor(dots[blk], 32);
            const float sdf = fmaf(dot, 1.0f / sdf2::kC1, b0) * (1.0f / a.scale);
            const long pd = (long)blockIdx.x * 128 + blk * 32 + c;
            if (pd < a.P && h == 0) {
                const long r = pd / a.n_per_ray;
                const long sidx = pd - r * a.n_per_ray;
                a.sdf[r * a.sdf_ld + sidx] = sdf;
                if constexpr (UPS) {                // ray (blk >> 1) of this workgroup, samples 32 (blk & 1) + c: z and sdf rows for the round
                    float* rr = rows + (blk >> 1) * 3 * kMaxT;
                    rr[(blk & 1) * 32 + c] = a.z[r * a.z_ld + sidx];
                    rr[kMaxT + (blk & 1) * 32 + c] = sdf;
                }
            }
        }
    }
    if constexpr (UPS) {
        lds_barrier();                              // (block 0's image is free: every wave is past the last layer)
        if (wave < 2) {
            const int r = blockIdx.x * 2 + wave;
            float* rr = rows + wave * 3 * kMaxT;
            if (r < up.B) upsample_row(up, r, lane, 64, rr, rr + kMaxT, rr + 2 * kMaxT);
        }
    }
}

template <bool UPS>
int launch_first(const VdnSdfArgs* args, hipStream_t stream, const VdnUpsampleArgs* up = nullptr) {
    static bool once = (allow_big_lds(sdf_first_split_kernel<UPS>, first::kLdsF), true);
    (void)once;
    static_assert(first::kRowsF + 2 * 3 * kMaxT * 4 <= first::kBlk, "the round's rows fit block 0's image");
    const int grid = (args->P + 127) / 128;
    hipLaunchKernelGGL((sdf_first_split_kernel<UPS>), dim3(grid), dim3(kWaves * 64), first::kLdsF, stream, *args, up != nullptr ? *up : VdnUpsampleArgs{});
    return (int)hipGetLastError();
}

