// renderer.py:239-315 as one launch: fused SDF network + gradient sweep + colour head + per-ray compositing (k_sdf_fwd2.h MODE 2).
// (built with -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 like sdf_bf16.hip, vdn_hip/build.py)
#include "k_sdf_fwd2.h"

extern "C" int vdn_shade_fused_bf16(const VdnSdfArgs* sa, const void* color_blob, int32_t squeeze_out, const VdnCompositeArgs* cm,
                                    int32_t* ticket, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (sa == nullptr || cm == nullptr || color_blob == nullptr || ticket == nullptr || sa->blob == nullptr) return -1;
    if (sa->pts != nullptr || !sa->rays_o || !sa->rays_d || !sa->z || !sa->sdf || !sa->normals) return -2;
    if (sa->active_idx != nullptr || sa->H != nullptr || sa->V != nullptr || sa->U_pe != nullptr) return -3;
    // one workgroup = one ray of 128 samples
    if (sa->n_per_ray != 128 || cm->N != 128 || cm->B <= 0 || sa->P != 128 * cm->B || sa->z_ld < 128 || sa->sdf_ld < 128) return -10;
    if (cm->T < cm->N || cm->T > vdn::kMaxT || cm->feat_out != nullptr) return -10;
    if (!cm->rays_o || !cm->rays_d || !cm->dists || !cm->mid_z || !cm->variance) return -4;
    if (!cm->weights || !cm->cdf || !cm->inside_sphere || !cm->color_out || !cm->weight_sum || !cm->weight_max || !cm->eik_partial ||
        !cm->eik_out) return -4;
    if (cm->T > cm->N && (!cm->bg_density || !cm->bg_rgb || !cm->bg_dists)) return -4;
    vdn::sdf2::ShadeExtra ex;
    ex.color_blob = static_cast<const char*>(color_blob);
    ex.ticket = ticket;
    ex.squeeze_out = squeeze_out;
    ex.cm = *cm;
    // (the 4-slot ring: with a fifth slot - one barrier per two chunk steps, k_sdf_fwd2.h VDN_SDF2_B2 - 12 more softplus' tiles move
    // from LDS into registers, and this mode, which also holds the colour head's input fragments, then spills: 512 VGPRs + 3)
    if (sa->feat != nullptr) return vdn::sdf2::launch<2, true, 4, 3>(sa, stream, nullptr, &ex);
    return vdn::sdf2::launch<2, false, 4, 3>(sa, stream, nullptr, &ex);
}
