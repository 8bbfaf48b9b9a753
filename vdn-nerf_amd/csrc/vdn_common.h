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

// Cache policy of the big streaming stores. The activation planes a training-step forward saves are read next by another kernel
// hundreds of microseconds and gigabytes later: stored write-through AND non-temporal they do not push the data the step is about to
// re-read (weight streams, code, the planes written just before) out of L2 and the 256-MB memory-side cache. The fused SDF
// forward's saves (k_sdf_fwd2.h: plane_store16): same-box A/B of two builds, 1 219 / 1 210 / 1 216 -> 1 172 / 1 175 / 1 169 us per step.
#ifndef VDN_SAVE_NT
#define VDN_SAVE_NT 1
#endif
// The same for every chain kernel's plane stores (mlp_engine.h: store_tile - the background network's and the heads' saves, the
// backward kernels' deltas: 3 = write-through + non-temporal), the one-launch SDF backward's stores (k_sdf_bwd_split.h: bit 1 of the
// buffer instructions' cache-policy operand = non-temporal) and the weight-gradient GEMM's operand loads, which read every plane
// exactly once (train_dw_bf16.hip). Same-box A/B of builds, three alternating runs, against the build with only the line above:
// plane stores 1 171 / 1 179 / 1 216 -> 1 144 / 1 146 / 1 148 us with the SDF backward's, -> 1 124 / 1 124 / 1 129 us with the GEMM's
// loads as well. What must NOT be non-temporal: the SDF backward's plane LOADS (it reads H twice: + 140 us), anything a kernel re-reads.
#ifndef VDN_PLANE_ST_MODE
#define VDN_PLANE_ST_MODE 3
#endif
#ifndef VDN_BS_ST_AUX
#define VDN_BS_ST_AUX 2
#endif
#ifndef VDN_DW_LD_NT
#define VDN_DW_LD_NT 1
#endif
// ... and, in the SDF backward, of the loads that ARE the last reading: the V planes (read once) and the H planes' second reading
// (fbar): 1 119 / 1 119 / 1 122 -> 1 113 / 1 115 / 1 112 us. The chain kernels' plane loads (mlp_engine.h: load_raw,
// -DVDN_PLANE_LD_NT) as non-temporal: nothing (1 127 / 1 120 / 1 123).
#ifndef VDN_BS_LDV_AUX
#define VDN_BS_LDV_AUX 2
#endif

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

// LDS-DMA of a weight chunk with IMMEDIATE offsets (round 5). A wave fetches G consecutive KiB of a chunk - pieces wave * G ..
// + G - 1 - so that one wave-uniform scalar base, ONE per-lane offset register and ONE M0 value serve all of them: the
// instruction's 13-bit signed immediate moves both addresses, global = base + lane offset + imm and LDS = M0 + 16 * lane + imm
// (measured: tools/dev/hw_probe.hip), and piece I sits at imm = (I % 8) KiB - 4096 around `centre` = first byte of the wave's
// range + 4096 (a wave with more than 8 pieces re-bases every 8). Per chunk and wave: one scalar write of M0 (+ its wait state);
// per piece: the load alone. (Rounds 1 - 4 gave every piece its own 64-bit base and its own s_mov m0 + s_nop: 4 scalar
// instructions per piece in kernels whose bound is the instruction issue of one or two waves per SIMD.) M0 is written by a
// group's first piece and must survive until its last one: nothing else in these kernels writes M0 (the compiler has no use for
// it here - no indirect register addressing, no LDS instruction that takes M0 on gfx9 - and tests/test_boundary_cpu.py checks the
// disassembly of the shipped library for foreign writes).
constexpr int glds_imm(int I) { return (I % 8) * 1024 - 4096; }
constexpr int glds_group_off(int I) { return (I / 8) * 8192; }
template <int IMM>
VDN_DEV void glds16_imm(const char* base_uniform, unsigned voff) {
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(voff), "s"(base_uniform), "n"(IMM) : "memory");
}
// ... the first piece of a group: M0 = m0v (wave-uniform LDS byte address of the group's centre)
template <int IMM>
VDN_DEV void glds16_imm_m0(const char* base_uniform, unsigned voff, unsigned m0v) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:%3" ::"v"(voff), "s"(m0v), "s"(base_uniform), "n"(IMM) : "memory", "m0");
}
// ... with the LDS address formed in the same instruction: M0 = m0_wave + ADD (a compile-time slot / group offset)
template <int ADD, int IMM>
VDN_DEV void glds16_imm_m0add(const char* base_uniform, unsigned voff, unsigned m0_wave) {
    asm volatile("s_add_u32 m0, %1, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2 offset:%4" ::"v"(voff), "s"(m0_wave), "s"(base_uniform), "n"(ADD), "n"(IMM)
                 : "memory", "m0", "scc");
}

// L2 warm-up of a kernel's weight stream. Inside a training step every MLP kernel starts on caches full of other kernels'
// planes: each chunk of its weight stream is then an HBM miss for the first workgroup of each XCD that asks for it, and as the
// workgroups of a launch walk the stream in lockstep, every one of its ~40-140 chunk steps waits that miss out (measured on the fused
// SDF kernel: 196 us in the step, 144 us repeated back to back with the stream L2-resident; 400 MB of unrelated stores in front
// of the back-to-back launch reproduce the 196: tools/dev/sdf_var_probe.py). So the FIRST ROUND of workgroups reads the whole
// stream once, up front and in parallel: blocks b, b + 8, b + 16 .. share an XCD (round-robin dispatch - a placement assumed for
// SPEED only), the j-th of them reads slice j; 8 L2s x 1-3 MB from HBM take a few microseconds instead of ~100 exposed misses.
//   n_wg: workgroups of this launch that have rows; resident: workgroups the chip holds at once (256 CUs x workgroups per CU).
#ifndef VDN_WARM_L2
#define VDN_WARM_L2 1
#endif
// Two halves, so that a kernel can do its own prologue (input loads, encodings) while the stream arrives: warm_l2_issue issues the
// loads, warm_l2_wait (s_waitcnt vmcnt(0)) waits for them. The loads are LDS-DMA (global_load_lds_dwordx4): their data lands in a
// 1-KiB LDS dump area, NOT in registers. (Round 4 loaded into a VGPR quadruple handed from an asm statement in _begin to one in
// _end: the compiler does not know that such registers are the target of loads still in flight, and any copy, coalescing or spill
// of the value in between would have freed the physical registers for live values that the returning data then overwrites - ADVICE
// round 4.) `lds_dump` (wave-uniform): 1 KiB that nothing else reads or writes until THIS wave has passed its next
// s_waitcnt vmcnt(0) - in the ring kernels the wave's own first DMA piece of ring slot 0 (lds + wave * 1024: written only by this
// wave's own DMA, which it issues behind that wait; read by the others only behind the chunk's barrier); in kernels whose LDS holds
// activations, any 1 KiB per wave with a workgroup barrier behind the wait (warm_l2_sync).
// Un-counted vector-memory operations in flight only make a counted or compiler-placed s_waitcnt vmcnt(N) return later.
VDN_DEV void warm_l2_issue(const char* blob, int bytes, long n_wg, int resident, char* lds_dump) {
#if VDN_WARM_L2
    const int first = n_wg < resident ? (int)n_wg : resident;          // workgroups of the first round
    if (bytes <= 0 || first < 128 || (int)blockIdx.x >= first) return;     // (a small launch's few workgroups would each read MBs)
    const int slices = first >> 3;
    const int j = blockIdx.x >> 3;
    if (j >= slices) return;
    const int slice = ((bytes + slices - 1) / slices + 255) & ~255;
    const int begin = j * slice, end = begin + slice < bytes ? begin + slice : bytes;
    for (int off = begin + (int)threadIdx.x * 16; off < end; off += (int)blockDim.x * 16) glds16(blob + off, lds_dump);
#endif
}
VDN_DEV void warm_l2_wait() {
#if VDN_WARM_L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
VDN_DEV void warm_l2(const char* blob, int bytes, long n_wg, int resident, char* lds_dump) {
    warm_l2_issue(blob, bytes, n_wg, resident, lds_dump);
    warm_l2_wait();
}
// ... for kernels whose LDS holds activations that ANY wave may write: every wave's dump has landed before anyone goes on
// (all waves of the workgroup must call it)
VDN_DEV void warm_l2_sync() {
#if VDN_WARM_L2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
}

// The kernel's own CODE is as cold as its weight stream: the MLP kernels are 40 - 140 KB of straight-line code (the chunk-step streams
// are fully unrolled), far beyond the instruction cache, so every workgroup streams it from L2 - and inside a training step, from
// HBM, at the head of every instruction-fetch miss. tools/dev/cold_probe.py: the step's fused SDF forward takes 116 us back to back,
// 155 us behind 400 MB of unrelated loads (weight stream warmed as above), and 124 us when the same kernel has just run on 8
// workgroups - one per XCD - in between: 31 of the 39 us are the code. So the first round of workgroups also reads the kernel's
// code into L2 as data, sliced like the stream: `bytes` from the current program counter on - a per-kernel constant chosen BELOW
// the kernel's code size (tests/test_boundary_cpu.py checks it against the symbol table of the built objects), so the reads stay
// inside the kernel's own code.
VDN_DEV void warm_code_issue(int bytes, long n_wg, int resident, char* lds_dump) {
#ifdef VDN_NO_CODE_WARM                     // (development A/B: VDN_BUILD_VARIANT="nocw:-DVDN_NO_CODE_WARM")
    bytes = 0;
#endif
    const char* pc = reinterpret_cast<const char*>(__builtin_amdgcn_s_getpc() & ~15L);
    warm_l2_issue(pc, bytes & ~255, n_wg, resident, lds_dump);
}
// bytes of code each kernel warms from its warm-up site on. One line per kernel: `// symbol: <substring of the mangled name>` is a regular expression for the
// mangled names the constant applies to (tests/test_boundary_cpu.py: constant + 4 KiB <= the smallest matching symbol's size)
constexpr int kWarmCodeSdfFwd2Save = 120 * 1024;      // symbol: sdf_fwd2_kernelILi1ELb1E
constexpr int kWarmCodeSdfFwd2 = 102 * 1024;          // symbol: sdf_fwd2_kernelILi1ELb0E
constexpr int kWarmCodeSdfFwd3 = 140 * 1024;          // symbol: sdf_fwd2_kernelILi3ELb1E
constexpr int kWarmCodeSdfFwd2Mode0 = 46 * 1024;      // symbol: sdf_fwd2_kernelILi0E
constexpr int kWarmCodeNerfFwd2 = 46 * 1024;          // symbol: nerf_fwd2_kernelILb.ELb1E
constexpr int kWarmCodeNerfBwd = 78 * 1024;           // symbol: nerf_bwd_kernelINS_4BF16E
constexpr int kWarmCodeRenderFwd = 28 * 1024;         // symbol: rendernet_fwd_kernelINS_4BF16E
constexpr int kWarmCodeRenderBwd = 36 * 1024;         // symbol: rendernet_bwd_kernelINS_4BF16E
constexpr int kWarmCodeSdfBwdSplit = 20 * 1024;       // symbol: sdf_bwd_split_kernel
constexpr int kWarmCodeCompositeTrain = 40 * 1024;    // symbol: composite_train_kernel

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
