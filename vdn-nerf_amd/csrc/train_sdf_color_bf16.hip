// The training step's forward of the SDF network AND the colour head in one launch (k_sdf_fwd2.h MODE 3; fields.py:72-108 + 148-176):
// vdn_sdf_mlp_fwd_bf16(mode 1, with saves) + vdn_rendernet_fwd_bf16 without the feature plane's second trip and the second launch.
// (built with -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 like sdf_bf16.hip, vdn_hip/build.py)
#include "k_sdf_fwd2.h"

extern "C" int vdn_sdf_color_train_bf16(const VdnSdfArgs* sa, const void* color_blob, int32_t squeeze_out, void* col_h, void* col_small,
                                        float* col_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (sa == nullptr || color_blob == nullptr || sa->blob == nullptr || col_h == nullptr || col_small == nullptr || col_out == nullptr) return -1;
    if (sa->pts != nullptr || !sa->rays_o || !sa->rays_d || !sa->z || sa->n_per_ray <= 0 || sa->z_ld < sa->n_per_ray || sa->sdf_ld < sa->n_per_ray) return -2;
    if (!sa->sdf || !sa->feat || !sa->normals || !sa->H || !sa->V || sa->P <= 0) return -3;
    if (sa->U_pe != nullptr || sa->tail_max_rows != 0) return -10;          // (ray gradients / the tail split take the separate launches)
    vdn::sdf2::ShadeExtra ex{};
    ex.color_blob = static_cast<const char*>(color_blob);
    ex.squeeze_out = squeeze_out;
    ex.col_h = col_h;
    ex.col_small = col_small;
    ex.col_out = col_out;
    return vdn::sdf2::launch<3, true, 4, 3>(sa, stream, nullptr, &ex);
}
