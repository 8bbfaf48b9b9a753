// The tail rows of a training step's SDF forward, 32 rows per workgroup with the features split over the waves
// (k_sdf_fwd1_split.h). Built with sdf_bf16.hip's flags (vdn_hip/build.py): the two kernels must round alike.
#include "k_sdf_fwd1_split.h"

extern "C" int vdn_sdf_fwd_tail_bf16(const VdnSdfArgs* args, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (args == nullptr || args->P <= 0 || args->blob == nullptr) return -1;
    if (args->pts == nullptr && (args->rays_o == nullptr || args->rays_d == nullptr || args->z == nullptr || args->n_per_ray <= 0 ||
                                 args->z_ld < args->n_per_ray || args->sdf_ld < args->n_per_ray)) return -2;
    if (!args->sdf || !args->feat || !args->normals) return -3;
    if (args->tail_max_rows <= 0 || args->tail_row0 < 0 || (args->tail_row0 % 128) != 0) return -4;
    if (args->U_pe != nullptr) return -10;
    if (args->H != nullptr) {
        if (args->V == nullptr) return -3;
        return vdn::sdf1s::launch<true>(args, stream);
    }
    return vdn::sdf1s::launch<false>(args, stream);
}
