// renderer.py:239-315 as one launch on the exact-fp32 path (k_shade_f32.h): what vdn_shade_fused_bf16 is on the bf16 path.
#include "k_shade_f32.h"

extern "C" int vdn_shade_fused_f32(const VdnSdfArgs* sa, const VdnRenderNetArgs* ca, const VdnCompositeArgs* cm, int32_t* ticket, void* stream_) {
    using namespace vdn;
    hipStream_t stream = (hipStream_t)stream_;
    if (sa == nullptr || ca == nullptr || cm == nullptr || ticket == nullptr || sa->blob == nullptr || ca->blob == nullptr) return -1;
    if (sa->pts != nullptr || !sa->rays_o || !sa->rays_d || !sa->z || !sa->sdf || !sa->normals || !sa->feat || !sa->S || !sa->w8row) return -2;
    if (sa->active_idx != nullptr || sa->H != nullptr || sa->V != nullptr || sa->U_pe != nullptr || ca->active_idx != nullptr || ca->save_h != nullptr) return -3;
    // one workgroup = one ray of 128 samples; the three stages describe the same rows
    if (sa->n_per_ray != 128 || cm->N != 128 || cm->B <= 0 || sa->P != 128 * cm->B || sa->z_ld != 128 || sa->sdf_ld != 128) return -10;
    if (cm->T < cm->N || cm->T > kMaxT || cm->feat_out != nullptr || ca->extra != nullptr || ca->d_out != 3) return -10;
    if (ca->P != sa->P || ca->n_per_ray != 128 || ca->pts != nullptr || ca->dirs != nullptr || ca->feat != sa->feat || ca->normals != sa->normals ||
        ca->z != sa->z || ca->rays_o != sa->rays_o || ca->rays_d != sa->rays_d || !ca->out) return -4;
    if (cm->sdf != sa->sdf || cm->normals != sa->normals || cm->color != ca->out) return -4;
    if (!cm->rays_o || !cm->rays_d || !cm->dists || !cm->mid_z || !cm->variance) return -4;
    if (!cm->weights || !cm->cdf || !cm->inside_sphere || !cm->color_out || !cm->weight_sum || !cm->weight_max || !cm->eik_partial ||
        !cm->eik_out) return -4;
    if (cm->T > cm->N && (!cm->bg_density || !cm->bg_rgb || !cm->bg_dists)) return -4;
    const size_t l1 = VDN_NSLOT * F32::stride(9), l2 = 3 * F32::stride(10), l3 = 2 * kMaxT * sizeof(float);
    const size_t lds = l1 > l2 ? (l1 > l3 ? l1 : l3) : (l2 > l3 ? l2 : l3);
    static bool once = (allow_big_lds(shade_f32_kernel, lds), true);
    (void)once;
    hipLaunchKernelGGL(shade_f32_kernel, dim3(cm->B), dim3(F32::kWaves * 64), lds, stream, *sa, *ca, *cm, ticket);
    return (int)hipGetLastError();
}
