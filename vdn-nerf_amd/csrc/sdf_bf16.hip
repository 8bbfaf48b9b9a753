// bf16-MFMA (throughput) SDF forward: the second-generation kernel - see k_sdf_fwd2.h
// (built with -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1, vdn_hip/build.py: packed f32 VALU forms are an
// anti-lever beside MFMAs, and accumulators in arch VGPRs spare the epilogue one v_accvgpr_read per value)
#include "k_sdf_fwd2.h"
#include "k_sdf_fwd0_split.h"
extern "C" int vdn_sdf_mlp_fwd_bf16(int mode, const VdnSdfArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || args->P <= 0 || args->blob == nullptr) return -1;
    if (args->pts == nullptr && (args->rays_o == nullptr || args->rays_d == nullptr || args->z == nullptr || args->n_per_ray <= 0 ||
                                 args->z_ld < args->n_per_ray || args->sdf_ld < args->n_per_ray)) return -2;
    if (args->sdf == nullptr) return -3;
    if (mode == 0) {
        // few points (the sampler's up-sampling passes): features split over the waves, 32 points per workgroup (k_sdf_fwd0_split.h);
        // otherwise 128 points per workgroup, 80 KiB of LDS: two workgroups per CU. Same values either way.
        static const long split_max = [] { const char* e = getenv("VDN_SDF0_SPLIT_MAX"); return e != nullptr ? atol(e) : 8192L; }();
        if (args->P <= split_max) return vdn::sdf0s::launch<false>(args, stream);
        return vdn::sdf2::launch<0, false, 4, 3>(args, stream);
    }
    if (mode != 1) return -4;
    if (!args->feat || !args->normals) return -3;
    if (args->H != nullptr) {
        if (args->V == nullptr) return -3;
        return vdn::sdf2::launch<1, true, 4, 3>(args, stream);
    }
    // (the inference launch: a 5-slot ring = one workgroup barrier per TWO chunk steps in front of the sweep, k_sdf_fwd2.h VDN_SDF2_B2;
    // the training launch gains nothing from it - its steps wait on their plane stores - and keeps the 4-slot ring)
    return vdn::sdf2::launch<1, false, 5, 3>(args, stream);
}

// One up-sampling round behind its SDF pass (renderer.py:201 + 372-386) in one launch: vdn_sdf_mlp_fwd_bf16(mode 0) on the new
// samples (ray form, 16 per ray), then vdn_merge_upsample on the rows of those rays. Same values as the two calls.
extern "C" int vdn_sdf_merge_upsample_bf16(const VdnSdfArgs* args, const VdnMergeArgs* m, const VdnUpsampleArgs* u, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || m == nullptr || u == nullptr || args->P <= 0 || args->blob == nullptr) return -1;
    if (args->rays_o == nullptr || args->rays_d == nullptr || args->z == nullptr || args->sdf == nullptr) return -2;
    // the shape the fusion covers: 16 new samples per ray (two rays per 32-point workgroup), no work list, the pass' input
    // and output ARE the merge's new rows
    if (args->pts != nullptr || args->active_idx != nullptr || args->n_per_ray != 16 || m->K != 16 || args->P != (int64_t)m->B * 16 ||
        args->z != m->new_z || args->sdf != m->new_sdf || args->z_ld != 16 || args->sdf_ld != 16) return -10;
    if (m->B <= 0 || !m->z || !m->z_out || !m->sdf || !m->sdf_out) return -3;
    if (m->M < 1 || m->M + m->K > vdn::kMaxT || m->ld < m->M || m->ld_out < m->M + m->K) return -4;
    if (u->B != m->B || u->M != m->M + m->K || u->weights || !u->rays_o || !u->rays_d || !u->u || !u->new_z || u->n_imp < 1 || u->n_imp > 64) return -5;
    return vdn::sdf0s::launch<true>(args, stream, m, u);
}

// The sampler's first pass and first up-sampling round (renderer.py:369-370 + 147-191) in one launch: vdn_sdf_mlp_fwd_bf16
// (mode 0) on the 64 coarse samples per ray, then vdn_upsample_round on those rows. Same values as the two calls.
extern "C" int vdn_sdf_upsample_bf16(const VdnSdfArgs* args, const VdnUpsampleArgs* u, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || u == nullptr || args->P <= 0 || args->blob == nullptr) return -1;
    if (args->rays_o == nullptr || args->rays_d == nullptr || args->z == nullptr || args->sdf == nullptr) return -2;
    // covered: ray form, 64 samples per ray = the rows to up-sample (two rays per 128-point workgroup), no work list
    if (args->pts != nullptr || args->active_idx != nullptr || args->n_per_ray != 64 || u->M != 64 || args->P != (int64_t)u->B * 64 ||
        args->z_ld < 64 || args->sdf_ld < 64) return -10;
    if (u->weights || !u->rays_o || !u->rays_d || !u->u || !u->new_z || u->n_imp < 1 || u->n_imp > 64) return -5;
    return vdn::sdf2::launch<0, false, 4, 3, 0, true>(args, stream, u);
}
