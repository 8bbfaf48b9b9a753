/* vdn_render.h - C ABI of libvdn_render.so, the MI355X (gfx950) kernel library behind the
 * drop-in `dpt_models` package (vdn-nerf_amd/dpt_models).
 *
 * The reference (BoifZ/VDN-NeRF) has no FFI layer: its boundary is the Python class API of
 * dpt_models/fields.py and dpt_models/renderer.py. Each entry point below names
 * the reference code it stands in for. All pointers are DEVICE pointers unless marked host;
 * `stream` is a hipStream_t passed as void*; every call is asynchronous on that stream, owns no
 * memory and keeps no global state. Return value: 0 = ok, <0 = argument error, >0 = hipError_t.
 *
 * Layout conventions: all tensors fp32 row-major contiguous. P = number of points, B = rays.
 */
#ifndef VDN_RENDER_H
#define VDN_RENDER_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VDN_ABI_VERSION 1

int vdn_abi_version(void);

/* ---- weight preparation ---------------------------------------------------------------------
 * replaces torch.nn.utils.weight_norm's per-forward recomputation (fields.py:65-66,141-142):
 * W_eff[r,:] = v[r,:] * (g[r] / ||v[r,:]||), materialised once per optimizer step. */
typedef struct {
    const float* g;      /* [rows]      weight_g (or NULL: plain Linear, W_eff = v) */
    const float* v;      /* [rows,cols] weight_v (or Linear.weight) */
    float* w_eff;        /* [rows,cols] out */
    float* inv_norm;     /* [rows] out: 1/||v_r|| (kept for the backward), may be NULL */
    int32_t rows, cols;
} VdnWeightNormDesc;
int vdn_weightnorm_materialize(const VdnWeightNormDesc* descs_dev, int n_layers, int max_rows, void* stream);

/* One 32-row MFMA weight chunk to build (see csrc/mlp_engine_f32.h for the chunk format).
 * value(i, k) = scale * src[nmap[n0+i]*row_stride + kmap[k]*col_stride]   (0 where a map entry is -1)
 * so a transposed image is just swapped strides. */
typedef struct {
    const float* src;        /* effective weight matrix */
    const float* bias;       /* [src rows] or NULL */
    const int32_t* kmap;     /* [k_pad] source column per padded input index, -1 = zero */
    const int32_t* nmap;     /* [>= n0+32] source row per padded output index, -1 = zero */
    char* dst;               /* chunk destination inside the blob */
    int64_t row_stride, col_stride;
    int32_t n0;              /* first padded output row of this chunk */
    int32_t k_pad;           /* padded input width (multiple of 32) */
    float scale;
    int32_t fmt;             /* 0 = fp32 chunk, 1 = bf16 chunk */
} VdnChunkDesc;
int vdn_build_images(const VdnChunkDesc* descs_dev, int n_chunks, void* stream);

/* ---- SDF network: fields.py:72-108 (SDFNetwork.forward / .sdf / .gradient) --------------------
 * mode 0: sdf only (fields.py:91-92; used by the sampler renderer.py:370,201 and the mesh lattice
 *         renderer.py:441-446).  mode 1: sdf + 256-d feature + analytic d sdf/d x. */
typedef struct {
    const char* blob;          /* weight chunk stream for this mode */
    const float* pts;          /* [P,3], or NULL to generate pts = rays_o[r] + rays_d[r]*z[p], r = p / n_per_ray */
    const float* rays_o;       /* [B,3] */
    const float* rays_d;       /* [B,3] */
    const float* z;            /* [B,z_ld], first n_per_ray columns used */
    int32_t n_per_ray;
    int32_t z_ld;              /* row stride of z (>= n_per_ray) */
    int32_t sdf_ld;            /* row stride of the sdf output, point (r,s) -> sdf[r*sdf_ld+s] (= n_per_ray when dense) */
    int32_t P;                 /* B * n_per_ray (or number of explicit points) */
    float scale;               /* SDFNetwork(scale=...) */
    float* sdf;                /* out, see sdf_ld */
    float* feat;               /* [P,256] out (mode 1) */
    float* normals;            /* [P,3] out (mode 1) */
    float* S;                  /* [8,P,256] workspace: softplus'(pre-activation) per hidden layer (mode 1) */
    const float* w8row;        /* [256] row 0 of the last layer's effective weight (mode 1) */
} VdnSdfArgs;
int vdn_sdf_mlp_fwd_f32(int mode, const VdnSdfArgs* args_host, void* stream);

/* ---- RenderingNetwork (colour head / 96-channel VDN head): fields.py:148-176, mode 'idr' ------
 * points are regenerated as rays_o[r] + rays_d[r]*z[p] (renderer.py:233), r = p / n_per_ray. */
typedef struct {
    const char* blob;
    const float* rays_o;       /* [B,3] */
    const float* rays_d;       /* [B,3] (also the view direction, renderer.py:234) */
    const float* z;            /* [P] section mid-points */
    const float* normals;      /* [P,3]  d sdf / d x */
    const float* feat;         /* [P,256] SDF feature vector */
    const float* pts;          /* [P,3] explicit points (standalone RenderingNetwork.forward) or NULL */
    const float* dirs;         /* [P,3] explicit view dirs or NULL (then rays_d[r]) */
    float* out;                /* [P,d_out] */
    int32_t n_per_ray;
    int32_t P;
    int32_t d_out;             /* 1..4 or 96 */
    int32_t squeeze_out;       /* 1: sigmoid (fields.py:170-171), 0: relu */
} VdnRenderNetArgs;
int vdn_rendernet_fwd_f32(const VdnRenderNetArgs* args_host, void* stream);

/* ---- background NeRF: fields.py:324-353 + the inverted-sphere points of renderer.py:112-115 ---- */
typedef struct {
    const char* blob;
    const float* rays_o;       /* [B,3] */
    const float* rays_d;       /* [B,3] */
    const float* z;            /* [P] mid z of the background pass */
    const float* pts4;         /* [P,4] explicit input_pts (standalone NeRF.forward) or NULL */
    const float* dirs;         /* [P,3] explicit input_views or NULL (then rays_d[r]) */
    float* density;            /* [P] raw alpha_linear output (before softplus) */
    float* rgb;                /* [P,3] */
    float* feat;               /* [P,96] dpt_linear output, or NULL when gen_depth_feats is off */
    int32_t n_per_ray;
    int32_t P;
} VdnNerfArgs;
int vdn_nerf_mlp_fwd_f32(const VdnNerfArgs* args_host, void* stream);

/* ---- per-ray stages of NeuSRenderer (one wavefront per ray) -----------------------------------*/

/* renderer.py:334-359: z = near + (far-near)*linspace(0,1,n) [+ (t_rand-0.5)*2/n];
 * z_out = far / flip(zo) + 1/n, zo = linspace(1e-3, 1-1/(n_out+1), n_out) or its stratified jitter.
 * The linspace vectors are passed in (generated once by the host with torch.linspace so their
 * rounding is the reference's). t_rand / t_rand_out NULL = no perturbation. */
typedef struct {
    const float* near;         /* [B] */
    const float* far;          /* [B] */
    const float* lin_samples;  /* [n_samples] */
    const float* lin_outside;  /* [n_outside] */
    const float* out_lower;    /* [n_outside] */
    const float* out_upper;    /* [n_outside] */
    const float* t_rand;       /* [B] or NULL */
    const float* t_rand_out;   /* [B,n_outside] or NULL */
    float* z;                  /* [B,z_ld] first n_samples columns written */
    float* z_out;              /* [B,n_outside] */
    int32_t B, n_samples, n_outside, z_ld;
} VdnCoarseArgs;
int vdn_coarse_z(const VdnCoarseArgs* args_host, void* stream);

/* renderer.py:147-191 (up_sample) + 44-74 (sample_pdf, det=True) for one round. */
typedef struct {
    const float* rays_o;       /* [B,3] */
    const float* rays_d;       /* [B,3] */
    const float* z;            /* [B,ld] sorted, first M valid */
    const float* sdf;          /* [B,ld] */
    const float* u;            /* [n_imp] = linspace(0.5/n, 1-0.5/n, n) */
    float* new_z;              /* [B,n_imp] */
    float inv_s;               /* 64 * 2^round (renderer.py:378) */
    int32_t B, M, ld, n_imp;
} VdnUpsampleArgs;
int vdn_upsample_round(const VdnUpsampleArgs* args_host, void* stream);

/* renderer.py:197-205 (cat + sort + sdf permuted alike); also z_feed of renderer.py:390-391 with
 * the sdf pointers NULL. z_out may alias z (in place). */
typedef struct {
    const float* z;            /* [B,ld] first M valid, sorted */
    const float* sdf;          /* [B,ld] or NULL */
    const float* new_z;        /* [B,K] */
    const float* new_sdf;      /* [B,K] or NULL */
    float* z_out;              /* [B,ld_out] first M+K written */
    float* sdf_out;            /* [B,ld_out] or NULL */
    int32_t B, M, K, ld, ld_out;
} VdnMergeArgs;
int vdn_merge_sorted(const VdnMergeArgs* args_host, void* stream);

/* renderer.py:228-230 / 107-109: dists = diff(z) with last = sample_dist; mid_z = z + dists/2. */
typedef struct {
    const float* z;            /* [B,ld] */
    float* dists;              /* [B,n] */
    float* mid_z;              /* [B,n] */
    float sample_dist;
    int32_t B, n, ld;
} VdnSectionArgs;
int vdn_sections(const VdnSectionArgs* args_host, void* stream);

/* renderer.py:262-315: NeuS alpha, inside-sphere blend with the background pass, weights =
 * alpha * exclusive-cumprod(1-alpha+1e-7), colour / 96-ch feature sums, eikonal term. */
typedef struct {
    const float* rays_o;       /* [B,3] */
    const float* rays_d;       /* [B,3] */
    const float* sdf;          /* [B*N] */
    const float* normals;      /* [B*N,3] */
    const float* dists;        /* [B,N] */
    const float* mid_z;        /* [B,N] */
    const float* color;        /* [B*N,3] sampled colour */
    const float* feat;         /* [B*N,C] sampled VDN feature or NULL */
    const float* variance;     /* [1] SingleVarianceNetwork.variance (device) */
    const float* bg_density;   /* [B*T] NeRF alpha_linear output, or NULL when n_outside == 0 */
    const float* bg_rgb;       /* [B*T,3] */
    const float* bg_feat;      /* [B*T,C] or NULL */
    const float* bg_dists;     /* [B,T] */
    const float* background_rgb; /* [3] or NULL */
    float cos_anneal_ratio;
    int32_t B, N, T, feat_ch;
    float* weights;            /* [B,T] */
    float* alpha_out;          /* [B,T] blended alpha (kept for the backward) or NULL */
    float* cdf;                /* [B,N] */
    float* inside_sphere;      /* [B,N] */
    float* color_out;          /* [B,3] */
    float* feat_out;           /* [B,C] or NULL */
    float* weight_sum;         /* [B] */
    float* weight_max;         /* [B] */
    float* s_val;              /* [B] 1/inv_s */
    float* eik_partial;        /* [B,2] workspace */
    float* eik_out;            /* [3]: gradient_error, numerator, denominator */
} VdnCompositeArgs;
int vdn_alpha_composite_fwd(const VdnCompositeArgs* args_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VDN_RENDER_H */
