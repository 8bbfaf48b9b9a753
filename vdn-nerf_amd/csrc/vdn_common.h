// Shared device helpers for the gfx950 (MI355X / CDNA4) NeuS render kernels.
//
// Register layout used by every MLP kernel ("activation tile"): one wave owns 32 points; a tile of
// 32 features x 32 points lives in 16 f32 registers per lane exactly as an MFMA 32x32 accumulator:
//   lane = c + 32*h  (c = point column 0..31, h = lane half)
//   reg t (0..15) holds feature  rho(t,h) = (t&3) + 8*(t>>2) + 4*h  of point c.
// A layer is  Y[out x pts] = W[out x in] . X[in x pts]: weights are the MFMA A operand (streamed
// through LDS), activations the B operand, so a layer's accumulators are directly the next layer's
// B operands and activations never leave registers between layers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#include <utility>

#define VDN_DEV __device__ __forceinline__
// lambdas inside kernels capture register arrays by reference: they must always be inlined, or the
// arrays escape to scratch memory
#define VDN_INL __attribute__((always_inline))

namespace vdn {

constexpr int kWave = 64;

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [0, N) - independent of the
// optimizer's unroll heuristics, so register arrays indexed by I never fall to scratch.
template <class F, int... Is>
VDN_DEV void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
VDN_DEV void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

VDN_DEV int rho(int t, int h) { return (t & 3) + 8 * (t >> 2) + 4 * h; }

// async global -> LDS copy of 16 B per lane; LDS destination = wave-uniform base + lane*16.
// Inline asm, not __builtin_amdgcn_global_load_lds: with the builtin, hipcc's wait-count pass marks a pending FLAT access
// and from then on emits s_waitcnt lgkmcnt(0) in front of every LDS consumer - a full LDS round trip before each MFMA
// group instead of a counted wait (measured on the SDF forward kernel, profiles/README.md round 2). The kernels count
// these loads themselves (WStream::acquire's vmcnt); the compiler does not see them. M0 (the LDS destination base) is
// declared clobbered.
VDN_DEV void glds16(const void* gsrc_lane, void* lds_wave_base) {
    const unsigned lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc_lane), "s"(lds) : "memory", "m0");
}

// Hardware transcendental forms (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each): what the MLP
// epilogues use - the ocml expf/log1pf expansions cost more VALU time than the layer's MFMAs.
VDN_DEV float hw_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
VDN_DEV float hw_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// Softplus(beta=100, threshold=20) and its derivative sigmoid(100 a) from one exponential
// (reference fields.py:70; derivative in the z/(z+1) form of ATen's backward), written to minimise VALU
// instructions - the epilogue, not the MFMAs, bounds the bf16 kernels:
//   t = a * 100 log2(e);  e = 2^-|t|;  L = log2(1 + e)                      (no overflow for any a)
//   softplus = max(a, 0) + L * ln2/100      (equals the thresholded form to < 1e-10 beyond 100 a > 20)
//   sigmoid(100 a) = a >= 0 ? 1/(1+e) : e/(1+e)
// max(a, 0) in ONE instruction: fmaxf() on an MFMA result costs two (hipcc first canonicalises the operand with
// v_max_f32 x, x, x because it cannot prove it is not a signalling NaN). As signed integers, negative floats (and -0)
// are negative and non-negative floats keep their order, so v_max_i32 against 0 is the same clamp.
// (Not inline asm: the hazard recognizer does not see inside it, and a VALU read right behind an MFMA needs wait states.)
VDN_DEV float relu0(float a) {
    const int i = __builtin_bit_cast(int, a);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}
VDN_DEV void softplus100_both(float a, float& hval, float& sval) {
    const float t = a * 144.26950408889634f;
    const float e = __builtin_amdgcn_exp2f(-fabsf(t));
    const float u = 1.0f + e;
    hval = fmaf(__builtin_amdgcn_logf(u), 0.0069314718055994529f, relu0(a));
    const float r = __builtin_amdgcn_rcpf(u);
    sval = a >= 0.0f ? r : e * r;
}
VDN_DEV float softplus100_fast(float a) {
    const float t = a * 144.26950408889634f;
    const float e = __builtin_amdgcn_exp2f(-fabsf(t));
    return fmaf(__builtin_amdgcn_logf(1.0f + e), 0.0069314718055994529f, relu0(a));
}

VDN_DEV float softplus100(float a) {
    // torch.nn.Softplus(beta=100), threshold 20  (reference fields.py:70)
    const float z = a * 100.0f;
    return z > 20.0f ? a : log1pf(expf(z)) * 0.01f;
}
VDN_DEV float softplus100_grad(float a) {
    // d/da, the form ATen's backward uses: z/(z+1), 1 past the threshold
    const float z = a * 100.0f;
    const float e = expf(z);
    return z > 20.0f ? 1.0f : e / (e + 1.0f);
}
VDN_DEV float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
VDN_DEV float softplus1(float x) {
    // F.softplus default: beta 1, threshold 20 (reference renderer.py:124)
    return x > 20.0f ? x : log1pf(expf(x));
}

// Scatter a per-point vector vals[NF] (compile-time indexed, identical on both lane halves) into
// NT activation tiles: X[tile*16+t] = vals[32*tile + rho(t,h)], zero beyond NF.
template <int NF, int NT>
VDN_DEV void vals_to_tiles(const float (&vals)[NF], int h, float* X) {
#pragma unroll
    for (int tile = 0; tile < NT; ++tile) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int f0 = 32 * tile + (t & 3) + 8 * (t >> 2);
            const int f1 = f0 + 4;
            const float a = f0 < NF ? vals[f0 < NF ? f0 : 0] : 0.0f;
            const float b = f1 < NF ? vals[f1 < NF ? f1 : 0] : 0.0f;
            X[tile * 16 + t] = h ? b : a;
        }
    }
}

// Inverse: gather NT tiles into a full per-point vector (every lane gets all values of its point).
template <int NF, int NT>
VDN_DEV void tiles_to_vals(const float* X, int h, float (&vals)[NF]) {
#pragma unroll
    for (int tile = 0; tile < NT; ++tile) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int f0 = 32 * tile + (t & 3) + 8 * (t >> 2);
            const int f1 = f0 + 4;
            const float mine = X[tile * 16 + t];
            const float other = __shfl_xor(mine, 32);
            if (f0 < NF) vals[f0 < NF ? f0 : 0] = h ? other : mine;
            if (f1 < NF) vals[f1 < NF ? f1 : 0] = h ? mine : other;
        }
    }
}

// sin/cos for the positional encoding. ACCURATE: ocml sincosf (full range reduction; the fp32 parity path).
// Otherwise the hardware v_sin_f32 / v_cos_f32 (argument in revolutions, |x| <= 256 rev): ~3e-6 abs error for the
// SDF's octaves and ~3e-5 at 2^9 x (NeRF), far below the bf16 rounding the encoded value then receives - and ~100x
// fewer VALU instructions, which matters because the bf16 kernels are VALU-issue-bound.
template <bool ACCURATE>
VDN_DEV void sincos_pe(float x, float& s, float& c) {
    if constexpr (ACCURATE) {
        sincosf(x, &s, &c);
    } else {
        const float r = x * 0.15915494309189535f;
        s = __builtin_amdgcn_sinf(r);
        c = __builtin_amdgcn_cosf(r);
    }
}

// Positional encoding of a D-vector with L log-spaced octaves, reference order
// [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...] (embedder.py:27-36).
template <int D, int L, bool ACCURATE = true>
VDN_DEV void posenc(const float (&v)[D], float (&pe)[D * (1 + 2 * L)]) {
#pragma unroll
    for (int d = 0; d < D; ++d) pe[d] = v[d];
#pragma unroll
    for (int k = 0; k < L; ++k) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float s, c;
            sincos_pe<ACCURATE>(v[d] * f, s, c);
            pe[D + 2 * D * k + d] = s;
            pe[D + 2 * D * k + D + d] = c;
        }
    }
}

}  // namespace vdn
