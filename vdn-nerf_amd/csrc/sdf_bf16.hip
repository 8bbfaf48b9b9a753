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
        if (args->P <= split_max) return vdn::sdf0s::launch<>(args, stream);
        return vdn::sdf2::launch<0, false, 4, 3>(args, stream);
    }
    if (mode != 1) return -4;
    if (!args->feat || !args->normals) return -3;
    if (args->H != nullptr) {
        if (args->V == nullptr) return -3;
        return vdn::sdf2::launch<1, true, 4, 3>(args, stream);
    }
    return vdn::sdf2::launch<1, false, 4, 3>(args, stream);
}
