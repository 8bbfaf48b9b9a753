// SDF backward, both chains in one feature-split launch (k_sdf_bwd_split.h), bf16 policy.
#include "k_sdf_bwd_split.h"

extern "C" int vdn_sdf_bwd_split_bf16(const VdnSdfRbarArgs* ra, const VdnSdfFbarArgs* fa, void* stream) {
    if (!ra || !fa || ra->P <= 0 || !ra->blob || !fa->blob) return -1;
    if (!ra->g_normals || !ra->S || !ra->V || !ra->UB || !fa->g_sdf || !fa->g_feat || !fa->AB) return -1;
    if (!ra->pts && (!ra->rays_o || !ra->rays_d || !ra->z || ra->n_per_ray <= 0 || ra->z_ld < ra->n_per_ray)) return -2;
    if (fa->P != ra->P || fa->S != ra->S || fa->s_from_h != ra->s_from_h || fa->active_idx != ra->active_idx || fa->n_active != ra->n_active ||
        fa->scale != ra->scale) return -3;
    if (fa->d_pts != nullptr || ra->s_from_h != 2) return -10;           // differentiable rays, or saves not in the bf16 forward's units: the two-kernel path
    // every plane is addressed through a buffer descriptor with 32-bit byte offsets (k_sdf_bwd_split.h): the largest one, AB, is
    // rows x 2336 bf16. Beyond 4 GiB the offsets would wrap: decline, the caller falls back to the two-kernel path.
    const long rows = (ra->P + 31L) & ~31L;
    if (rows * 2336L * 2L > 0xffffffffL || rows * 256L * 8L * 2L > 0xffffffffL) return -10;
    return vdn::sdfbs::launch(ra, fa, (hipStream_t)stream);
}
