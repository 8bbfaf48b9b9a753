// renderer.py:239-315 in ONE launch on the exact-fp32 kernels (the path that holds the reference's 1e-4): the SDF network with its
// analytic gradient sweep (k_sdf_fwd.h, fields.py:72-108), the colour head (k_render_fwd.h, fields.py:148-176) and the ray's NeuS
// alpha / background blend / transmittance scan / weighted sums / eikonal term (k_composite_row.h, renderer.py:262-315), for rays of
// 128 inside samples: a 128-point workgroup (4 waves x 32 samples) IS one ray, as in the bf16 kernel (k_sdf_fwd2.h MODE 2).
// The three bodies run back to back in the workgroup that owns the ray; what passes between them - the feature vector, the normal,
// the colour - goes through this workgroup's own rows of the caller's buffers (written and read by the same workgroup, a
// barrier in between; the f32 MFMA kernels are 16 x slower than the bf16 ones and the 1 KB per point is noise beside them): one
// launch instead of four, the same device code, the same values bit for bit.
#pragma once
#include "k_sdf_fwd.h"
#include "k_render_fwd.h"
#include "k_composite_row.h"

namespace vdn {

// the ray's eikonal partial sums -> gradient_error over ALL rays without a launch of its own (as in k_sdf_fwd2.h MODE 2): every ray
// publishes its pair (one 8-byte write-through store, drained) and counts itself in; the ray whose count comes back last reads all
// pairs (sc1 loads: never this CU's L1) and reduces them exactly as eikonal_reduce_kernel does. wave-uniform call (one wave per ray).
VDN_DEV void eikonal_by_the_last_ray(const CompositeArgs& cm, float* eik_partial, int ray, const RowOut& ep, int* ticket, int lane) {
    int old = 0;
    if (lane == 0) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 pv = {ep.num, ep.den};
        asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(eik_partial + 2 * ray), "v"(pv) : "memory");
        old = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    old = __builtin_amdgcn_readfirstlane(old);
    if (old == cm.B - 1) {
        double sn = 0.0, sd = 0.0;
        for (int i = lane; i < cm.B; i += 64) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 v;
            asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(eik_partial + 2 * i) : "memory");
            sn += (double)v[0];
            sd += (double)v[1];
        }
        sn = wave_sum(sn);
        sd = wave_sum(sd);
        if (lane == 0) {
            cm.eik_out[0] = (float)sn / ((float)sd + 1e-5f);
            cm.eik_out[1] = (float)sn;
            cm.eik_out[2] = (float)sd;
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(F32::kWaves * 64, 1) void shade_f32_kernel(SdfArgs sa, RenderNetArgs ca, CompositeArgs cm, int* ticket) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 1. fields.py:72-108: sdf, feature vector, normal of this workgroup's 128 samples -> sa.sdf / sa.feat / sa.normals
    sdf_fwd_body<F32, 1, F32::kWaves, false>(sa, smem);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // this wave's stores have reached L2; the ring is quiet
    __builtin_amdgcn_s_barrier();
    // 2. fields.py:148-176 on them -> ca.out (the colour; the rows were written by this very workgroup and never read before)
    rendernet_fwd_body<F32, 1, 0>(ca, smem);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // 3. renderer.py:262-315: wave 0 composites the ray
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wave != 0) return;
    const int ray = blockIdx.x;
    float* eik_partial = cm.eik_partial;
    cm.eik_partial = nullptr;                   // (published below, ahead of the arrival count)
    float* scratch = reinterpret_cast<float*>(smem);
    const RowOut ep = composite_row(cm, ray, lane, CompositeGlobalSrc{cm.sdf, cm.normals, cm.color}, scratch, scratch + kMaxT);
    eikonal_by_the_last_ray(cm, eik_partial, ray, ep, ticket, lane);
}

}  // namespace vdn
