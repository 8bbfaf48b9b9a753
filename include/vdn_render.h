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

#define VDN_ABI_VERSION 28

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

/* One 32-row MFMA weight chunk to build (see csrc/mlp_engine.h for the chunk format).
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
    int32_t k_pad;           /* padded input width of the whole chunk (multiple of 32) */
    float scale;
    int32_t fmt;             /* 0 = fp32 chunk, 1 = bf16 chunk */
    int32_t kt_begin;        /* this descriptor fills k-tiles [kt_begin, kt_begin+kt_count) of the chunk; */
    int32_t kt_count;        /* kmap is indexed from the start of that range. kt_count 0 = whole chunk    */
    int32_t write_bias;      /* 1: also write the chunk's bias/pad block (exactly one descriptor per chunk) */
    float bias_scale;        /* the bias block holds bias_scale * bias[row] */
    const float* tail;       /* optional: tail_n floats copied to dst + tail_off (behind the bias block, inside the chunk's   */
    int32_t tail_off;        /* stride): a small constant that rides along with every chunk's DMA - the bf16 SDF streams carry */
    int32_t tail_n;          /* row 0 of the last layer's weight there (csrc/k_sdf_fwd2.h)                                    */
    int32_t tail_stride;     /* distance between consecutive tail floats in `tail` (0 or 1: contiguous; the colour head's "c2" */
                             /* stream carries one COLUMN of its first layer: stride = that matrix's row length)               */
} VdnChunkDesc;
int vdn_build_images(const VdnChunkDesc* descs_dev, int n_chunks, void* stream);

/* ---- positional encoding as a stand-alone op: embedder.py:27-36 (Embedder.embed) ----------------------------
 * out[p, :] = [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), cos(2^1 x), ...]  (each function over all d inputs, then the next),
 * x = in[p, 0:d]; n_freqs octaves, frequencies 2^k exactly (embedder.py:23). The MLP kernels evaluate the same encoding
 * in registers; this entry point serves callers of embed_fn (e.g. sdf_network.embed_fn_fine). */
int vdn_posenc(const float* in, float* out, int64_t P, int32_t d, int32_t n_freqs, void* stream);

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
    void* feat;               /* [P,256] out (mode 1) */
    float* normals;            /* [P,3] out (mode 1) */
    void* S;                  /* [8,P,256] workspace: softplus'(pre-activation) per hidden layer (mode 1, f32 entry point).
                               * The bf16 entry point never touches it: it keeps softplus' on the chip (8-bit, LDS + registers) */
    const float* w8row;        /* [256] row 0 of the last layer's effective weight (f32 entry point, mode 1; the bf16 streams carry it) */
    /* training-mode saves (mode 1), optional (H == NULL = nothing saved; with H, V is required and PE optional).
     * The f32 entry point saves in the network's own units. The bf16 entry point saves all three in units of
     * 1/(100 log2 e) (H = 100 log2(e) softplus(a), V = 100 log2(e) v, PE = 100 log2(e) encoding): the units its
     * hidden layers compute in (csrc/k_sdf_fwd2.h). Consumers: s_from_h = 2 in the backward chains, and a factor
     * 1/(100 log2 e) in the finalize scale of the weight-gradient entries that contract over these planes. */
    void* H;                  /* [8,P,256] H[l] = softplus output of layer l (= input of layer l+1) */
    void* V;                  /* [8,P,256] V[l] = sweep value v_l = u_{l+1} * softplus'(a_l) */
    void* PE;                 /* [P,64] positional encoding of the point (39 valid) */
    /* optional work list (vdn_foreground_active; same contract as VdnNerfArgs.active_idx): per-point inputs / outputs
     * are addressed by the dense point id, training saves and deltas by the compact row */
    const int32_t* active_idx;
    const int32_t* n_active;
    /* optional (mode 1): d sdf / d(encoded input) in the network's own units, the 39 values u with
     * normal = scale * J_PE(x)^T u (fields.py:97-108) - what the ray adjoint needs for the explicit x-dependence of
     * J_PE (learnable poses, poses.py:198-208). Rows follow the saves (compact with a work list). */
    float* U_pe;               /* [P,39] or NULL */
    /* optional (bf16 entry points, mode 1 with a work list): the TAIL of the list. The 128-row kernel runs whole rounds of 256
     * workgroups (one per CU); when the list ends within tail_max_rows rows behind tail_row0 (a multiple of 128, normally one
     * full round = 32 768), vdn_sdf_mlp_fwd_bf16 leaves the rows from tail_row0 on alone and vdn_sdf_fwd_tail_bf16 - called with
     * the SAME struct - evaluates them with 32-row workgroups (csrc/k_sdf_fwd1_split.h): same planes, same values, bit for bit.
     * Both decide on the device-side row count, so both launches are always made. tail_max_rows = 0: off. */
    int32_t tail_row0, tail_max_rows;
    /* != 0: the launch is part of a training step, where every kernel starts on caches full of other kernels' planes: its first
     * round of workgroups then reads the weight stream into L2 up front (csrc/mlp_engine.h: warm_l2). The launches that save
     * activations do so on their own; a render() loop keeps its weights cached and leaves this 0. */
    int32_t cold_start;
} VdnSdfArgs;
int vdn_sdf_mlp_fwd_f32(int mode, const VdnSdfArgs* args_host, void* stream);
/* bf16-MFMA variant (csrc/k_sdf_fwd2.h): blob holds bf16 chunks (fmt 1) of the SCALED streams (vdn_hip/images.py:
 * sdf_streams(scaled=True): hidden biases x 100 log2 e, last layer's weights / (100 log2 e), sweep weights / 255);
 * feat / H / V / PE are bf16 arrays in the tile-blocked layout of csrc/mlp_engine.h; sdf / normals stay f32. */
int vdn_sdf_mlp_fwd_bf16(int mode, const VdnSdfArgs* args_host, void* stream);
/* mode 1 (+ training saves when H is given) on rows [tail_row0, n) of the work list - see VdnSdfArgs.tail_row0; with
 * tail_row0 = 0 and tail_max_rows >= P it evaluates every row (tests). U_pe must be NULL (-10). */
int vdn_sdf_fwd_tail_bf16(const VdnSdfArgs* args_host, void* stream);

/* ---- RenderingNetwork (colour head / 96-channel VDN head): fields.py:148-176, mode 'idr' ------
 * points are regenerated as rays_o[r] + rays_d[r]*z[p] (renderer.py:233), r = p / n_per_ray. */
typedef struct {
    const char* blob;
    const float* rays_o;       /* [B,3] */
    const float* rays_d;       /* [B,3] (also the view direction, renderer.py:234) */
    const float* z;            /* [P] section mid-points */
    const float* normals;      /* [P,3]  d sdf / d x */
    const void* feat;          /* [P,256] SDF feature vector (f32 or bf16, per entry point) */
    const float* pts;          /* [P,3] explicit points (standalone RenderingNetwork.forward) or NULL */
    const float* dirs;         /* [P,3] explicit view dirs or NULL (then rays_d[r]) */
    float* out;                /* [P,d_out] */
    void* save_h;              /* [4,P,256] post-ReLU hidden activations (training) or NULL */
    void* save_small;          /* [P,64] the non-feature inputs [points, PE(view), normals] (33 valid) or NULL */
    int32_t n_per_ray;
    int32_t P;
    int32_t d_out;             /* 1..4 or 96 */
    int32_t squeeze_out;       /* 1: sigmoid (fields.py:170-171), 0: relu */
    /* optional work list (vdn_foreground_active; same contract as VdnNerfArgs.active_idx): per-point inputs / outputs
     * are addressed by the dense point id, training saves and deltas by the compact row */
    const int32_t* active_idx;
    const int32_t* n_active;
    /* optional: 96 more feature columns appended to the feature vector - render(depth_before_color=True) feeds the colour
     * network cat([feature_vector, VDN output]) (renderer.py:247-248; a d_feature = 352 network, blob with a 13-k-tile first layer) */
    const float* extra;        /* [P,96] dense point id, or NULL */
    void* save_extra;          /* [P,96] plane of `extra` for the weight-gradient GEMM (training) or NULL */
} VdnRenderNetArgs;
int vdn_rendernet_fwd_f32(const VdnRenderNetArgs* args_host, void* stream);
int vdn_rendernet_fwd_bf16(const VdnRenderNetArgs* args_host, void* stream);   /* bf16-MFMA variant: bf16 chunk blob, bf16 activation workspaces */

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
    /* training-mode saves, optional: */
    void* save_h;             /* [8,P,256] post-ReLU outputs of pts_linears.0..7 */
    void* save_pe;            /* [P,96] PE10(pts4) (84 valid) */
    void* save_feature;       /* [P,256] feature_linear output */
    void* save_vpe;           /* [P,32] PE4(view) (27 valid) */
    void* save_hv;            /* [P,128] views_linears.0 output (post-ReLU) */
    int32_t n_per_ray;
    int32_t P;
    /* Optional active-point list (vdn_background_active): only points active_idx[0 .. *n_active) are evaluated; density /
     * rgb / feat are written at their dense positions (the others keep their previous, finite contents - the compositor
     * multiplies them by zero), the training saves are written in COMPACT order (row q holds point active_idx[q]). */
    const int32_t* active_idx; /* [P] or NULL = all points */
    const int32_t* n_active;   /* device scalar */
} VdnNerfArgs;
int vdn_nerf_mlp_fwd_f32(const VdnNerfArgs* args_host, void* stream);
int vdn_nerf_mlp_fwd_bf16(const VdnNerfArgs* args_host, void* stream);   /* bf16-MFMA variant: bf16 chunk blob, bf16 activation workspaces */

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
    /* optional: run sample_pdf(bins = z, weights, n_imp, det=True) (renderer.py:44-74) on GIVEN weights instead of the
     * ones up_sample derives from sdf (which, with rays_o / rays_d, may then be NULL) - the form the reference's own
     * sample_pdf vectors are stated in */
    const float* weights;      /* [B,w_ld], first M-1 valid, or NULL */
    int32_t w_ld, _pad;
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
/* vdn_merge_sorted (with sdf) followed by vdn_upsample_round on the merged rows (M = merge.M + merge.K; the upsample args'
 * z / sdf / ld are ignored: the rows are handed over on chip) - cat_z_vals of round i and up_sample of round i+1
 * (renderer.py:372-386) in one launch. Same results as the two calls. */
int vdn_merge_upsample(const VdnMergeArgs* merge_host, const VdnUpsampleArgs* upsample_host, void* stream);
/* vdn_sdf_mlp_fwd_bf16(mode 0) on the new samples of a round (renderer.py:201; ray form: sdf.z = merge.new_z [B,16],
 * sdf.sdf = merge.new_sdf [B,16]) followed by vdn_merge_upsample on those rays (renderer.py:372-386), in ONE launch: a
 * 32-point workgroup of the small-pass kernel is two rays, whose waves go on to merge and up-sample their rows. Same results
 * as the two calls, bit for bit. Covers 16 new samples per ray, no work list; returns -10 for any other shape (the caller
 * then makes the two calls). */
/* vdn_sdf_mlp_fwd_bf16(mode 0) on the coarse samples (ray form, 64 per ray: renderer.py:369-370) followed by
 * vdn_upsample_round on those rows (upsample.M = 64; its z / sdf / ld are ignored: the rows are handed over on chip), in one
 * launch. Same results as the two calls, bit for bit; -10 for any other shape. */
int vdn_sdf_upsample_bf16(const VdnSdfArgs* sdf_host, const VdnUpsampleArgs* upsample_host, void* stream);
int vdn_sdf_merge_upsample_bf16(const VdnSdfArgs* sdf_host, const VdnMergeArgs* merge_host, const VdnUpsampleArgs* upsample_host, void* stream);

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
    const float* cos_anneal_dev; /* [1] device scalar or NULL. Non-NULL: the kernels read cos_anneal_ratio from it (the by-value field is
                                  * ignored) - a launch captured once in a HIP graph then follows the runner's annealing schedule
                                  * (dpt_runner.py:304-308) without a re-capture (ABI 28) */
} VdnCompositeArgs;
int vdn_alpha_composite_fwd(const VdnCompositeArgs* args_host, void* stream);
/* d_feats = sum_i w_i * feature_i (renderer.py:306-308) alone, from the `weights` / `inside_sphere` a compositor launch has
 * written: the second launch of vdn_alpha_composite_fwd, for callers whose compositor ran inside vdn_shade_fused_bf16. */
int vdn_feat_composite(const VdnCompositeArgs* args_host, void* stream);

/* ---- renderer.py:239-315 in ONE launch (the north-star kernel; bf16 path, inference): render_core's SDF network + analytic
 * gradient (fields.py:72-108), the colour head (fields.py:148-176, mode 'idr', d_out = 3) and the NeuS alpha / background blend /
 * transmittance scan / weighted sums of vdn_alpha_composite_fwd, for rays of exactly 128 inside samples: one 128-point workgroup
 * evaluates one ray, keeps the 256-d feature vector in registers, runs the colour head on it and composites the ray from LDS; the
 * ray that finishes last reduces the eikonal partial sums of all rays (`ticket`: one int32 arrival counter, zero before the first
 * call; the kernel leaves it zero). sdf: mode-1 arguments in ray form (n_per_ray = 128, P = 128 * comp.B, no work list, no saves;
 * sdf.feat may be NULL - the feature plane is only written when given, for a VDN head launched afterwards); color_blob: the colour
 * network's "c2" stream; comp: as for vdn_alpha_composite_fwd with comp.N = 128 - comp.sdf / normals / color are not read and
 * comp.feat_out must be NULL (the feature channels keep their own launches). Outputs equal the three launches' up to the rounding
 * of the colour head's first layer (the normal's z component enters as an f32 term here, as a bf16 operand there). */
/* renderer.py:239-315 in ONE launch on the exact-fp32 kernels (csrc/k_shade_f32.h; the counterpart of vdn_shade_fused_bf16 on the path
 * that holds the reference's 1e-4): vdn_sdf_mlp_fwd_f32(mode 1) + vdn_rendernet_fwd_f32 (colour head, d_out = 3) +
 * vdn_alpha_composite_fwd + the eikonal reduction, for rays of exactly 128 inside samples (one workgroup per ray), with the very
 * argument blocks of those calls: sdf_host (its sdf / feat / normals / S buffers are required: what passes between the stages goes
 * through them), color_host (feat / normals / z / rays must be sdf_host's, out = the sampled colour [P,3]), comp_host (sdf / normals /
 * color must be those buffers). Same device code as the separate launches: bit-identical outputs. ticket: [1] int32, zero before
 * the first launch. -10 = shape not covered (make the separate calls). */
int vdn_shade_fused_f32(const VdnSdfArgs* sdf_host, const VdnRenderNetArgs* color_host, const VdnCompositeArgs* comp_host,
                        int32_t* ticket, void* stream);
/* The training step's forward of the SDF network and of the colour head in ONE launch (bf16; csrc/k_sdf_fwd2.h MODE 3):
 * vdn_sdf_mlp_fwd_bf16(mode 1) with its training saves (sdf_host->H, V, PE, feat: as there) followed, on the feature vector kept in
 * registers, by vdn_rendernet_fwd_bf16 of the colour network (fields.py:148-176, mode 'idr', d_out = 3) with ITS saves: col_h
 * [4, rows, 256] hidden activations, col_small [rows, 64] the 33 small inputs, col_out [P,3] the colour - the planes
 * VdnRenderNetArgs.save_h / save_small / out name, in the same layouts (compact rows of sdf_host's work list; col_out by dense
 * point id). color_blob: the colour network's "c2" chunk stream (vdn_hip/images.py). The colour differs from the two launches'
 * by <= 4e-6 (the normal's z component enters the first colour layer in f32 instead of bf16). Declines (-10, nothing launched)
 * ray-gradient saves (U_pe) and the tail split (tail_max_rows). */
int vdn_sdf_color_train_bf16(const VdnSdfArgs* sdf_host, const void* color_blob, int32_t squeeze_out, void* col_h, void* col_small,
                             float* col_out, void* stream);
int vdn_shade_fused_bf16(const VdnSdfArgs* sdf_host, const void* color_blob, int32_t squeeze_out, const VdnCompositeArgs* comp_host,
                         int32_t* ticket, void* stream);

/* The eikonal sums of renderer.py:313-315 alone - relax_inside_sphere * (|gradient| - 1)^2 and relax_inside_sphere, summed
 * per ray and over the batch, with the compositor's own expressions (same translation unit, bit-identical to the values
 * vdn_alpha_composite_fwd leaves in eik_out) - available as soon as the SDF normals exist: the data-parallel Trainer
 * launches it right behind the fused SDF kernel and all-reduces (numerator, denominator) over the ranks while the colour
 * head, the background network and the compositor still run. */
typedef struct {
    const float* rays_o;       /* [B,3] */
    const float* rays_d;       /* [B,3] */
    const float* mid_z;        /* [B,N] section mid-points */
    const float* normals;      /* [B*N,3] */
    int32_t B, N;
    float* eik_partial;        /* [B,2] workspace */
    float* eik_out;            /* [3]: gradient_error, numerator, denominator */
} VdnEikonalArgs;
int vdn_eikonal_terms(const VdnEikonalArgs* args_host, void* stream);

/* ======================= training step: backward of the render path ============================
 * The reference back-propagates with autograd (dpt_runner.py:253) through renderer.py:209-330 and,
 * twice, through fields.py:97-108. Here the adjoint is hand-derived (DESIGN.md 'Backward'). */

/* backward of RenderingNetwork (fields.py:148-176): delta chain through W^T with ReLU masks from the
 * saved activations; emits per-layer deltas for the weight-gradient GEMM and d feature / d normals. */
typedef struct {
    const char* blob;          /* 'bwd' stream: W4^T, W3^T, W2^T, W1^T, W0^T */
    const float* g_out;        /* [P,d_out] gradient wrt the network output */
    const float* out;          /* [P,d_out] forward output (post sigmoid / relu) */
    const void* save_h;       /* [4,P,256] from the forward */
    void* delta_out;          /* [P,32] (d_out<=4) or [P,96]: delta of the last layer */
    void* delta_h;            /* [4,P,256]: deltas of hidden layers 0..3 */
    void* d_feat;             /* [P,256] d loss / d feature_vector */
    float* d_normals;          /* [P,3] */
    int32_t acc_feat;          /* 0: overwrite d_feat, 1: add into it */
    int32_t acc_normals;       /* 0: overwrite d_normals, 1: add into it */
    int32_t P;
    int32_t d_out;
    int32_t squeeze_out;
    /* optional work list (vdn_foreground_active; same contract as VdnNerfArgs.active_idx): per-point inputs / outputs
     * are addressed by the dense point id, training saves and deltas by the compact row */
    const int32_t* active_idx;
    const int32_t* n_active;
    /* optional input adjoints for differentiable rays (poses.py:198-208): d loss / d points and d loss / d view_dirs
     * (through the 4-octave encoding). dirs / rays_d as in VdnRenderNetArgs (dirs [P,3] or rays_d [B,3] with n_per_ray). */
    const float* dirs; const float* rays_d;
    int32_t n_per_ray;
    int32_t acc_pts;           /* 0: overwrite d_pts / d_dirs, 1: add into them */
    float* d_pts;              /* [P,3] dense point id, or NULL */
    float* d_dirs;             /* [P,3] or NULL */
    /* d_feature = 352 network (VdnRenderNetArgs.extra): d loss / d extra is ADDED into d_extra (the VDN head's output adjoint,
     * which the compositor wrote first) */
    float* d_extra;            /* [P,96] dense point id, or NULL for the 256-feature network */
} VdnRenderNetBwdArgs;
int vdn_rendernet_bwd_f32(const VdnRenderNetBwdArgs* args_host, void* stream);
int vdn_rendernet_bwd_bf16(const VdnRenderNetBwdArgs* args_host, void* stream);   /* bf16-MFMA variant: bf16 chunk blob, bf16 activation workspaces */

/* backward of the background NeRF (fields.py:324-353). */
typedef struct {
    const char* blob;          /* 'bwd' stream: Wout^T, Wviews^T, Whead^T, W7^T .. W1^T, W0^T (the last only read with d_pts) */
    const float* g_density;    /* [P] */
    const float* g_rgb;        /* [P,3] */
    const float* g_feat;       /* [P,96] or NULL */
    const void* save_h;       /* [8,P,256] */
    const void* save_hv;      /* [P,128] */
    void* delta_o;            /* [P,128] ([P,32] without the dpt head): delta of [rgb | dpt] */
    void* delta_v;            /* [P,128] delta of views_linears.0 */
    void* delta_head;         /* [P,288] delta of [feature_linear (256) | alpha_linear (1)] */
    void* delta_h;            /* [8,P,256] deltas of pts_linears.0..7 */
    int32_t P;
    /* as in VdnNerfArgs: g_* are read at the dense positions, saves and deltas are in compact order */
    const int32_t* active_idx;
    const int32_t* n_active;
    /* optional input adjoints for differentiable rays: with d_pts, the kernel also runs W0^T and maps the adjoints of
     * the two encodings back through pts4 = [p / r, 1 / r], r = clip(|p|, 1, 1e10) (renderer.py:112-115) to
     * d loss / d p and d loss / d view_dir. The points are regenerated from the rays as in VdnNerfArgs. */
    const float* rays_o; const float* rays_d; const float* z;
    int32_t n_per_ray;
    float* d_pts;              /* [P,3] dense point id (rows off the work list are not written), or NULL */
    float* d_dirs;             /* [P,3] (required with d_pts) */
} VdnNerfBwdArgs;
int vdn_nerf_mlp_bwd_f32(const VdnNerfBwdArgs* args_host, void* stream);
int vdn_nerf_mlp_bwd_bf16(const VdnNerfBwdArgs* args_host, void* stream);   /* bf16-MFMA variant: bf16 chunk blob, bf16 activation workspaces */

/* backward of SDFNetwork.forward + .gradient (fields.py:72-108), i.e. including the double backward
 * through the input-gradient. Two chains (DESIGN.md 'Backward'):
 *   rbar: adjoint of the reverse sweep, ascending layers, uses the forward weight images;
 *   fbar: adjoint of the forward pass, descending layers, uses the transposed images.
 * Flat workspaces (row-major blocks, P rows each):
 *   UB = [ub0: 64][ub1: 256][ub2: 256][ub3: 256][ub4: 288][ub5: 256][ub6: 256][ub7: 256][ub8: 256]  (2144 cols total)
 *   EX = [8][P][256]   second-order term added to the forward adjoint of each hidden layer
 *   AB = [ab8: 288][ab7: 256] ... [ab0: 256]                                                (2336 cols total) */
typedef struct {
    const char* blob;          /* forward stream of the SDF network (hidden layers 0..7 are used) */
    const float* pts;          /* [P,3] or NULL -> rays */
    const float* rays_o;
    const float* rays_d;
    const float* z;            /* [B,z_ld] */
    int32_t n_per_ray;
    int32_t z_ld;
    int32_t P;
    float scale;
    const float* g_normals;    /* [P,3] d loss / d (d sdf/d x) */
    const void* S;            /* [8,P,256] from the forward: softplus' planes, or - with s_from_h - the saved activations H */
    const void* V;            /* [8,P,256] from the forward */
    void* UB;                 /* out, see above */
    void* EX;                 /* out */
    int32_t s_from_h;          /* 1: S points at the H planes; softplus' = 1 - exp(-100 h) is evaluated in the kernel.
                               * 2: H and V are in units of 1/(100 log2 e) (bf16 forward): softplus' = 1 - 2^-H */
    /* optional work list (vdn_foreground_active; same contract as VdnNerfArgs.active_idx): per-point inputs / outputs
     * are addressed by the dense point id, training saves and deltas by the compact row */
    const int32_t* active_idx;
    const int32_t* n_active;
} VdnSdfRbarArgs;
int vdn_sdf_bwd_rbar_f32(const VdnSdfRbarArgs* args_host, void* stream);
int vdn_sdf_bwd_rbar_bf16(const VdnSdfRbarArgs* args_host, void* stream);   /* bf16-MFMA variant: bf16 chunk blob, bf16 activation workspaces */

typedef struct {
    const char* blob;          /* 'fbar' stream: W8^T, W7^T .. W1^T, W0^T (the last only read with d_pts) */
    const float* g_sdf;        /* [P] */
    const void* g_feat;       /* [P,256] */
    const void* S;            /* [8,P,256] softplus' planes, or - with s_from_h - the saved activations H */
    const void* EX;           /* [8,P,256] from rbar */
    void* AB;                 /* out, see above */
    int32_t P;
    float scale;
    int32_t s_from_h;          /* as in VdnSdfRbarArgs */
    /* optional work list (vdn_foreground_active; same contract as VdnNerfArgs.active_idx): per-point inputs / outputs
     * are addressed by the dense point id, training saves and deltas by the compact row */
    const int32_t* active_idx;
    const int32_t* n_active;
    /* optional input adjoint for differentiable rays: d loss / d point through sdf, feature and normal
     * (incl. the explicit x-dependence of the encoding's Jacobian in normal = scale J_PE(x)^T u). Needs the points
     * (pts, or rays + z as in VdnSdfArgs), the upstream g_normals of rbar and U_pe saved by the forward. */
    const float* pts; const float* rays_o; const float* rays_d; const float* z;
    int32_t n_per_ray, z_ld;
    const float* g_normals;    /* [P,3] */
    const float* U_pe;         /* [P,39] */
    int32_t acc_pts;           /* 0: overwrite d_pts, 1: add into it */
    float* d_pts;              /* [P,3] dense point id, or NULL */
} VdnSdfFbarArgs;
int vdn_sdf_bwd_fbar_f32(const VdnSdfFbarArgs* args_host, void* stream);
int vdn_sdf_bwd_fbar_bf16(const VdnSdfFbarArgs* args_host, void* stream);   /* bf16-MFMA variant: bf16 chunk blob, bf16 activation workspaces */
/* Both chains in ONE launch (bf16, no ray gradients: fbar.d_pts must be NULL, else -10): rbar's second-order term ex_l stays
 * in the wave that produced it (csrc/k_sdf_bwd_split.h), rbar.EX / fbar.EX are not touched. UB and AB come out bit-identical to
 * vdn_sdf_bwd_rbar_bf16 followed by vdn_sdf_bwd_fbar_bf16. The two structs must describe the same rows (P, S, s_from_h, work list).
 * Planes are addressed with 32-bit byte offsets: batches whose AB plane (rows x 2336 bf16) would exceed 4 GiB (P > 919 296) are
 * declined with -10 and nothing is launched; callers then run the two kernels. */
int vdn_sdf_bwd_split_bf16(const VdnSdfRbarArgs* rbar_host, const VdnSdfFbarArgs* fbar_host, void* stream);


/* ---- weight gradients: dW[m,n] = sum_p A[p,m] * B[p,n] over one or two (A,B) segments ----------
 * (A = per-layer delta, B = the layer's input; the second segment carries the sweep term of the SDF
 * net). K is split across workgroups; partial slabs are reduced and scattered by vdn_dw_finalize. */
typedef struct {
    const void* A1; const void* B1;       /* [P,lda1], [P,ldb1]  (f32 or bf16, per entry point) */
    const void* A2; const void* B2;       /* second segment or NULL */
    int32_t lda1, ldb1, lda2, ldb2;
    int32_t P;
    int32_t m_tiles, n_tiles;             /* 32-wide column tiles of A (outputs) and B (inputs); n_tiles may be 0 */
    int32_t splits;
    int32_t wg_begin;                     /* prefix sum of vdn_dw_entry_wgs_*(m_tiles, n_tiles, splits) over the list */
    int32_t _pad;
    float* slab;                          /* [splits, m_tiles*32, n_tiles*32] partial products */
    float* colsum;                        /* [splits, m_tiles*32] partial column sums of A1, or NULL */
    const int32_t* P_dev;                 /* optional device scalar: contract over the first min(P, *P_dev) rows only */
} VdnDwDesc;
int vdn_dw_gemm_f32(const VdnDwDesc* descs_dev, int n_desc, int total_wgs, void* stream);
/* bf16 variant: A/B are bf16 planes in the tile-blocked layout of the bf16 chains, rows padded to a multiple of 32 (the
 * kernel reads whole 32-row blocks and zeroes the rows beyond P itself); with two segments `splits` must be even (first
 * half = segment 1). One workgroup contracts one K split of a 256 x 256 output block. */
int vdn_dw_gemm_bf16(const VdnDwDesc* descs_dev, int n_desc, int total_wgs, void* stream);
/* workgroups one descriptor occupies in the launch (the two kernels tile differently); total_wgs = their sum */
int vdn_dw_entry_wgs_f32(int m_tiles, int n_tiles, int splits);
int vdn_dw_entry_wgs_bf16(int m_tiles, int n_tiles, int splits);

/* reduce the K-splits and scatter from image coordinates to the parameter's own layout:
 * target[rmap[i]*t_stride + cmap[j]] (+)= scale * sum_s slab[s,i,j];  btarget[rmap[i]] (+)= bscale * sum_s colsum[s,i] */
typedef struct {
    const float* slab; const float* colsum;
    const int32_t* rmap; const int32_t* cmap;
    float* target; float* btarget;         /* either may be NULL */
    int64_t t_stride;
    int32_t splits, M, N;                  /* M = m_tiles*32, N = n_tiles*32 */
    int32_t accumulate;                    /* 0: '=' (phase 0), 1: '+=' (phase 1) */
    float scale, bscale;
    /* optional second source for ONE target row: target[xrow, cmap[j]] gets  + xscale * sum_s xsum[s, cmap[j]]  on top of the slab
     * sums (the sdf row of the last SDF layer also collects colsum(ub_8), k_sdf_bwd.h) - instead of a '+=' descriptor and
     * a second launch */
    const float* xsum;                     /* [xsplits, xM] or NULL */
    int32_t xsplits, xM, xrow;
    float xscale;
} VdnDwFinalizeDesc;
int vdn_dw_finalize(const VdnDwFinalizeDesc* descs_dev, int n_desc, int max_M, int phase, void* stream);

/* weight-norm backward: dg_r = <dW_r, v_r>/||v_r||;  dv_r = g_r/||v_r|| * (dW_r - <dW_r,v_r>/||v_r||^2 * v_r) */
typedef struct {
    const float* g; const float* v; const float* inv_norm; const float* dw_eff;
    float* dg; float* dv;
    int32_t rows, cols;
} VdnWeightNormBwdDesc;
int vdn_weightnorm_bwd(const VdnWeightNormBwdDesc* descs_dev, int n_layers, int max_rows, void* stream);

/* ---- adjoint of vdn_alpha_composite_fwd (renderer.py:262-315) ---------------------------------- */
typedef struct {
    /* forward inputs */
    const float* rays_o; const float* rays_d;
    const float* sdf; const float* normals; const float* dists; const float* mid_z;
    const float* color; const float* feat; const float* variance;
    const float* bg_density; const float* bg_rgb; const float* bg_feat; const float* bg_dists;
    const float* background_rgb;
    float cos_anneal_ratio;
    int32_t B, N, T, feat_ch;
    /* saved forward results */
    const float* alpha;        /* [B,T] blended alpha */
    const float* weights;      /* [B,T] */
    const float* eik;          /* [3] gradient_error, numerator, denominator */
    /* upstream gradients (any may be NULL = zero) */
    const float* g_color;      /* [B,3] */
    const float* g_feat;       /* [B,C] */
    const float* g_weights;    /* [B,T] */
    const float* g_eik;        /* [1] d loss / d gradient_error */
    const float* g_cdf;        /* [B,N] d loss / d cdf_fine (= prev_cdf, renderer.py:276,322), or NULL */
    /* outputs */
    float* d_sdf;              /* [B*N] */
    float* d_normals;          /* [B*N,3] alpha + eikonal parts */
    float* d_color;            /* [B*N,3] wrt the colour head's output */
    float* d_feat;             /* [B*N,C] wrt the VDN head's output, or NULL */
    float* d_bg_density;       /* [B*T] */
    float* d_bg_rgb;           /* [B*T,3] */
    float* d_bg_feat;          /* [B*T,C] or NULL */
    float* d_var_partial;      /* [B] per-ray d loss / d variance */
    float* d_variance;         /* [1], or NULL: the caller sums d_var_partial itself (e.g. as one more vdn_dw_finalize descriptor) */
    /* optional, for differentiable rays (all three or none): adjoints of the section lengths and of the ray direction
     * inside true_cos = rays_d . normal (renderer.py:265) */
    float* d_dists;            /* [B,N] */
    float* d_bg_dists;         /* [B,T] (NULL without a background) */
    float* d_dir_cos;          /* [B,3] */
    /* optional scratch for the feature channels, [B*(2T+N)] floats: with it the per-sample feature dot products and the outer
     * products d_feat / d_bg_feat run as two streaming launches around the per-ray kernel (which has only 2 waves per CU) */
    float* feat_scratch;
    const float* cos_anneal_dev; /* [1] device scalar or NULL: as VdnCompositeArgs.cos_anneal_dev */
} VdnCompositeBwdArgs;
int vdn_alpha_composite_bwd(const VdnCompositeBwdArgs* args_host, void* stream);

/* Training step, plain configuration (no mask loss, no depth-feature loss, one rank): vdn_alpha_composite_fwd, the colour term's
 * gradient of vdn_loss_fwd_bwd (g_color = sign(color - true_rgb) / (B + 1e-5) * grad_scale, dpt_runner.py:228-229 with mask = 1;
 * written to g_color [B,3] too) and vdn_alpha_composite_bwd in ONE launch, one wavefront per ray. fwd / bwd: the two calls' own
 * argument blocks (bwd.alpha / bwd.weights must be fwd.alpha_out / fwd.weights; bwd.g_color / g_eik / eik are not read); fg_count:
 * device scalar = the eikonal term's global denominator = the number of inside samples within the relaxed sphere (n_active of
 * vdn_foreground_active); igr_weight = d loss / d gradient_error. fwd.eik_out is NOT written: the loss scalars are left to
 * vdn_eikonal_reduce + vdn_loss_fwd_bwd, which may run later on any stream. Adjoints are bit-identical to the three calls'.
 * -10: a configuration it does not cover (feature channels, g_weights, g_cdf, ray adjoints). */
int vdn_composite_train(const VdnCompositeArgs* fwd_host, const VdnCompositeBwdArgs* bwd_host, const float* true_rgb, float* g_color,
                        const int32_t* fg_count, float igr_weight, float grad_scale, void* stream);
/* gradient_error = sum(num) / (sum(den) + 1e-5) from the per-ray partial sums [B,2] -> eik_out [3] (ratio, num, den). */
int vdn_eikonal_reduce(const float* eik_partial, int32_t B, float* eik_out, void* stream);
/* The same idea with the VDN head (womsk_white_wdepth, one rank, no mask), where the 96 feature channels keep their own streaming
 * launches: vdn_composite_fwd_train = the per-ray compositor + the features' weighted sums (vdn_feat_composite), which also write
 * d loss / d render_feats of the depth-feature term (dpt_runner.py:239-243, mask = 1) into g_feats [B,C]; no eikonal reduce.
 * vdn_composite_bwd_train = vdn_alpha_composite_bwd (bwd.g_feat = that g_feats) with the colour term's gradient made from
 * (color_out, true_rgb) and the eikonal denominator from the foreground work list's length, as in vdn_composite_train; bwd.g_color,
 * bwd.g_eik and bwd.eik are not read. Gradients bit-identical to vdn_alpha_composite_fwd + vdn_loss_fwd_bwd + vdn_alpha_composite_bwd;
 * the loss SCALARS stay with vdn_eikonal_reduce + vdn_loss_fwd_bwd, which the caller may run off the critical path. -10: a
 * configuration it does not cover (extra upstream gradients, ray adjoints). */
int vdn_composite_fwd_train(const VdnCompositeArgs* fwd_host, const float* gt_feats, float* g_feats, float depth_weight, float grad_scale, void* stream);
int vdn_composite_bwd_train(const VdnCompositeBwdArgs* bwd_host, const float* color_out, const float* true_rgb, float* g_color,
                            const int32_t* fg_count, float igr_weight, float grad_scale, void* stream);

/* ---- adjoint of the ray geometry of render_core / render_core_outside (renderer.py:107-115, 228-237): collects the
 * per-point input adjoints of the networks and the section-length adjoints of the compositor into
 * d loss / d rays_o, d rays_d and d loss / d z (inside samples) / d z_out (outside samples), for learnable poses
 * (poses.py:198-208). Geometry: pts = o + d * mid_z, dirs = d; inside: dists_i = z_{i+1} - z_i (last = sample_dist),
 * mid_i = z_i + dists_i / 2; background the same over z_feed = [z | z_out] (every outside sample lies beyond the inside
 * ones, renderer.py:359). One wave per ray. */
typedef struct {
    const float* rays_d;       /* [B,3] */
    const float* mid_z;        /* [B,N] */
    const float* bg_mid;       /* [B,T] or NULL */
    const float* d_pts;        /* [B*N,3] inside points */
    const float* d_dirs;       /* [B*N,3] */
    const float* d_dists;      /* [B,N] */
    const float* d_dir_cos;    /* [B,3] */
    const float* d_bg_pts;     /* [B*T,3] or NULL */
    const float* d_bg_dirs;    /* [B*T,3] */
    const float* d_bg_dists;   /* [B,T] */
    int32_t B, N, T;
    float* d_rays_o;           /* [B,3] */
    float* d_rays_d;           /* [B,3] */
    float* d_z;                /* [B,N] */
    float* d_z_out;            /* [B,T-N] or NULL */
} VdnRayAdjointArgs;
int vdn_ray_adjoint(const VdnRayAdjointArgs* args_host, void* stream);

/* ---- caller semantics of dpt_runner.py:208-243 fused into one launch: loss terms + d loss / d outputs.
 * loss = L1(color)/mask_sum + igr*eik + mask_w*BCE(clip(weight_sum)) + depth_w*L1(feats)/mask_sum.
 * out_scalars = [loss, color_loss, psnr, eikonal, depth_loss, mask_loss]. `grad_scale` multiplies the
 * per-ray gradients (1/world_size under data parallelism). */
typedef struct {
    const float* color;        /* [B,3] */
    const float* true_rgb;     /* [B,3] */
    const float* mask;         /* [B] or NULL (= ones, use_mask False) */
    const float* feats;        /* [B,C] or NULL */
    const float* gt_feats;     /* [B,C] or NULL */
    const float* weights;      /* [B,T] (for weight_sum) */
    const float* eik;          /* [3] gradient_error, num, den */
    float igr_weight, mask_weight, depth_weight, grad_scale;
    int32_t B, T, C, _pad;
    float* g_color;            /* [B,3] */
    float* g_feats;            /* [B,C] or NULL */
    float* g_weights;          /* [B,T] or NULL (only needed when mask_weight != 0) */
    float* g_eik;              /* [1] */
    float* out_scalars;        /* [6] */
} VdnLossArgs;
int vdn_loss_fwd_bwd(const VdnLossArgs* args_host, void* stream);

/* torch.optim.Adam (no weight decay, no amsgrad) over flat buffers: dpt_runner.py:144,254. */
int vdn_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  float lr, float beta1, float beta2, float eps, int32_t step, void* stream);
/* The same over the element ranges [begin0, end0) and [begin1, end1) only (the second may be empty), with their own step
 * count: torch.optim.Adam keeps a step per parameter and skips parameters whose grad is None - the VDN head and the
 * background network's dpt_linear until the depth-feature loss switches on (dpt_runner.py:239-243). */
int vdn_adam_step_ranges(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t begin0, int64_t end0,
                         int64_t begin1, int64_t end1, float lr, float beta1, float beta2, float eps, int32_t step, void* stream);

/* ---- on-device ray generator: poses.py:168-212 (fixed poses / intrinsics) + dataset.py:111-118 --------------
 * p = K^-1 [x,y,1]; v = p/|p|; rays_d = R v; rays_o = t; colour / mask / feature gathered from images resident in HBM;
 * near/far from the unit sphere. out row = [rays_o(3) rays_d(3) mask(1) rgb(3) feats(C)] exactly as gen_random_rays_at. */
typedef struct {
    const float* pixels_x;     /* [B] pixel coordinates (already integral-valued for random rays) */
    const float* pixels_y;     /* [B] */
    const float* intrinsic_inv;/* [3,3] row-major (upper-left of the 4x4) */
    const float* pose;         /* [3,4] row-major c2w (upper 3 rows of the 4x4) */
    const float* image;        /* [H,W,3] or NULL */
    const float* mask;         /* [H,W,mask_ch] or NULL (then mask = 1) */
    const float* feats;        /* [H,W,C] or NULL */
    float* out;                /* [B, out_ld] */
    float* near;               /* [B] or NULL */
    float* far;                /* [B] or NULL */
    int32_t B, H, W, C, mask_ch, out_ld;
} VdnGenRaysArgs;
int vdn_gen_rays(const VdnGenRaysArgs* args_host, void* stream);

/* ---- background samples that can reach the image -------------------------------------------------------------------
 * render_core blends the NeRF++ background into the first N (inside) samples with weight (1 - inside_sphere)
 * (renderer.py:284-299): where the section mid-point lies inside the unit sphere the background network's output is
 * multiplied by exactly zero, forward and backward. This kernel lists the other points - sample s < N of ray r is active
 * iff |rays_o + rays_d * mid_z[r,s]| >= 1 (the compositor's own test, same arithmetic), samples s >= N always - in
 * ascending dense order p = r*T + s (deterministic), so the background network can skip the rest without changing one
 * bit of render()'s outputs or gradients. */
typedef struct {
    const float* rays_o;       /* [B,3] */
    const float* rays_d;       /* [B,3] */
    const float* mid_z;        /* [B,N] section mid-points of the inside samples */
    int32_t B, N, T;
    int32_t* active_idx;       /* [B*T] out */
    int32_t* n_active;         /* [1] out */
    int32_t* ray_counts;       /* [B] scratch */
} VdnBackgroundActiveArgs;
int vdn_background_active(const VdnBackgroundActiveArgs* args_host, void* stream);

/* Foreground counterpart, used by the training step only: inside sample s of ray r is active iff
 * |rays_o + rays_d * mid_z[r,s]| < radius. With radius 1.2 the skipped points have inside_sphere = 0 AND
 * relax_inside_sphere = 0 (renderer.py:284-286): their sdf / normal / colour enter the loss only through factors that are
 * exactly zero, so the SDF, colour and VDN networks skip them in the training step (render() itself still evaluates them:
 * it returns `gradients` for every sample). Dense ids p = r*N + s, ascending. */
typedef struct {
    const float* rays_o;
    const float* rays_d;
    const float* mid_z;        /* [B,N] */
    int32_t B, N;
    float radius;
    int32_t complement;        /* nonzero: list the samples the test REJECTS instead (!(|p| < radius)): the two lists of one
                                * (rays, mid_z, radius) partition the B*N samples. render() under grad evaluates the SDF network
                                * with the training saves on the first and without them on the second, which only has to deliver
                                * `gradients` (dpt_models/renderer.py::_RenderCoreFn) */
    int32_t* active_idx;       /* [B*N] out */
    int32_t* n_active;         /* [1] out */
    int32_t* ray_counts;       /* [B] scratch */
} VdnForegroundActiveArgs;

/* The per-ray preparation of a training step in two launches instead of seven: z_feed = sort(cat(z, z_out))
 * (vdn_merge_sorted, renderer.py:390-391), the sections of the inside depths and of z_feed (vdn_sections twice), and both
 * work lists (vdn_foreground_active, vdn_background_active: count pass + fill pass each). Same results element for
 * element. fg_active_idx == NULL skips the foreground list (every inside sample is evaluated: the caller then sets its
 * own row count). */
typedef struct {
    const float* rays_o; const float* rays_d;
    float* z;                  /* [B,z_ld] sorted inside depths; with new_z: the first M_old of them, completed in place */
    const float* new_z;        /* optional [B,N-M_old]: the last up-sampling round's samples, still to be merged into z
                                * (the final cat_z_vals of renderer.py:379-385, which evaluates no sdf), or NULL */
    int32_t M_old, _pad;
    const float* z_out;        /* [B,T-N] outside depths */
    float* z_feed;             /* [B,T] out: sorted [inside | outside] depths */
    int32_t B, N, T, z_ld;
    float sample_dist, fg_radius;
    float* dists; float* mid_z;            /* [B,N] out */
    float* bg_dists; float* bg_mid;        /* [B,T] out */
    int32_t* fg_active_idx; int32_t* fg_n_active; int32_t* fg_ray_counts;     /* as VdnForegroundActiveArgs, or NULL */
    int32_t* bg_active_idx; int32_t* bg_n_active; int32_t* bg_ray_counts;     /* as VdnBackgroundActiveArgs */
} VdnTrainPrepArgs;
int vdn_train_prep(const VdnTrainPrepArgs* args_host, void* stream);
int vdn_foreground_active(const VdnForegroundActiveArgs* args_host, void* stream);

/* ---- iso-surface of the lattice u = -sdf (device marching tetrahedra) ---------------------------------------------
 * Stands in for mcubes.marching_cubes(u, threshold) of renderer.py:36 (PyMCubes is third-party and not part of the
 * reference tree): every lattice cube is cut into the 6 Kuhn tetrahedra (path 000 -> +e_a -> +e_b -> 111 for the 6
 * axis orders), which is translation invariant, so neighbouring cubes agree on their shared faces and the surface is
 * watertight. "Inside" = u > threshold. Two passes around a prefix sum the caller does:
 *   vdn_mesh_count: counts[cube] = triangles of that cube          (cube = (x*(R-1) + y)*(R-1) + z)
 *   vdn_mesh_emit:  triangle t of a cube -> tri_pos[offset[cube]+t][3 corners][xyz] in lattice index coordinates and
 *                   tri_key[..][3] = a * R^3 + b, the (ordered, a < b) lattice-vertex pair of the cut edge: equal keys
 *                   are the same vertex (bit-identical position), which is how the caller welds the soup.
 * Triangles are wound so that their normal points from inside (u > threshold) to outside. u is [R][R][R] fp32. */
typedef struct {
    const float* u;
    float threshold;
    int32_t R;
    int32_t* counts;           /* [(R-1)^3]   (count pass) */
    const int64_t* offsets;    /* [(R-1)^3] exclusive prefix sum of counts   (emit pass) */
    float* tri_pos;            /* [n_tri][3][3] */
    int64_t* tri_key;          /* [n_tri][3] */
} VdnMeshArgs;
int vdn_mesh_count(const VdnMeshArgs* args_host, void* stream);
int vdn_mesh_emit(const VdnMeshArgs* args_host, void* stream);

/* ---- iso-surface of the lattice as mcubes.marching_cubes(u, threshold) numbers it (renderer.py:36) -------------------------
 * Marching cubes on the classic 256-case tables with PyMCubes' sequential bookkeeping (mcubes/src/marchingcubes.h of
 * PyMCubes 0.1.2, the version the reference's README pins; restated in oracle/marching_cubes.py - the package itself is absent
 * here): cells visited x-major with z innermost, corner "set" when (double)u <= isovalue, one float64 vertex per cut lattice
 * edge created by the first visited cell containing it (in-cell order: edges 6, 5, 10, then 0, 1, 2, 3, 4, 7, 8, 9, 11 where no
 * earlier cell exists), interpolated from the edge's first corner to its second, triangles in table order. Two passes around two
 * exclusive prefix sums over the cells (cell = (x*(R-1) + y)*(R-1) + z) that the caller makes:
 *   vdn_mesh_mc_count: cube_case[cell], n_verts[cell] = vertices the cell creates, n_tris[cell];
 *   vdn_mesh_mc_emit:  vertices[vert_offsets[cell] ..][xyz] (lattice index coordinates) and triangles[tri_offsets[cell] ..][3]
 *                      (vertex numbers) - the arrays the sequential library returns, element for element. u is [R][R][R] fp32. */
typedef struct {
    const float* u;
    double isovalue;
    int32_t R;
    uint8_t* cube_case;            /* [(R-1)^3]   written by the count pass, read by the emit pass */
    int32_t* n_verts;              /* [(R-1)^3]   (count pass) */
    int32_t* n_tris;               /* [(R-1)^3]   (count pass) */
    const int64_t* vert_offsets;   /* [(R-1)^3] exclusive prefix sum of n_verts   (emit pass) */
    const int64_t* tri_offsets;    /* [(R-1)^3] exclusive prefix sum of n_tris    (emit pass) */
    double* vertices;              /* [V][3] */
    int64_t* triangles;            /* [F][3] */
} VdnMeshMcArgs;
int vdn_mesh_mc_count(const VdnMeshMcArgs* args_host, void* stream);
int vdn_mesh_mc_emit(const VdnMeshMcArgs* args_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VDN_RENDER_H */
