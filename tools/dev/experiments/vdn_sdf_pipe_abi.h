/* The C ABI of the layer-pipelined SDF backward, as include/vdn_render.h declared it up to ABI 23:
   the experiment left the product in round 5 (slower than the kernels it fuses); kept here with its source. */
/* The same backward (rbar chain, fbar chain and the weight gradients of the SDF network's hidden layers) as ONE persistent
 * launch, layer-pipelined over the CUs (bf16 path): stage s of 17 = one layer of one chain (rbar layers 0..7,
 * then fbar W8^T .. W1^T, then the layer-0 weight gradient), `lanes` workgroups per stage, each lane a contiguous range of
 * 32-point blocks that flows through the stages from workgroup to workgroup (write-through stores + a counter per stage and
 * lane). A workgroup keeps its layer's weights in LDS and its 256 x 256 weight-gradient partial in registers for the whole
 * launch, so the planes of one layer are read once - by the chain AND the weight gradient - instead of once by each chain
 * and once more by vdn_dw_gemm. Stage table entries (built by the host, in device memory): */
typedef struct {
    int32_t kind;              /* 0: rbar stage (forward image), 1: fbar stage (transposed image), 2: no chain (layer 0: gradient only) */
    int32_t kt_lds;            /* input tiles of the chain, staged through LDS per block: the first kt_lds - kt_extra from x_in, the rest from reg_out */
    int32_t kt_reg;            /* 0 or 2: the weights of the last kt_reg k-tiles (always k-tiles 7, 8) stay in registers, the others in LDS */
    int32_t nt;                /* output tiles of the chain = tiles of the `own` operand = waves with work (<= 8) */
    int32_t chunk0;            /* first chunk of the layer in `blob` (chunk = one output tile, mlp_engine.h format, 20-KiB stride) */
    int32_t kt_extra;          /* input tiles taken from the plane reg_out (tiles reg_tile0 ..) instead of x_in: fbar W8^T's sdf adjoint */
    int32_t in_stage;          /* stage whose x_out is this stage's x_in (-1: x_in is an input of the launch) */
    int32_t ex_stage;          /* stage that must have finished a block before this one reads its EX rows (-1: none) */
    int32_t has_dw;            /* this stage accumulates a weight gradient */
    int32_t n_dw;              /* input tiles 0 .. n_dw-1 take part in the weight gradient */
    int32_t split;             /* first K split of this stage's slabs: lane l writes split + l */
    int32_t in_ld, out_ld;     /* leading dimensions (elements) of the x_in / x_out planes */
    int32_t out_tile0;         /* x_out tile of output tile 0 */
    int32_t reg_tile0;         /* first of the kt_reg tiles in reg_out: UB(0) tile 0, UB(4) tile 7, AB(8) tile 8 */
    int32_t own_ld;            /* leading dimension of `own` */
    int32_t reg_ld;            /* leading dimension of `reg_out` */
    int32_t copy_in;           /* 1: the input tiles are also copied to reg_out tiles 0 .. kt_lds-1 (fbar W8^T: AB(8) = [g_feat | g_sdf / scale]) */
    const char* blob;          /* weight stream */
    const void* x_in;          /* bf16 plane (PT32), rows = compact work-list rows */
    void* x_out;               /* bf16 plane or NULL */
    void* reg_out;             /* plane of the kt_reg extra input tiles (written before the stages run, see ub0 / ub4 / ab8 below); with copy_in
                                * the x_in tiles are copied into it too */
    const void* S;             /* H plane of the layer whose softplus' multiplies the chain output (units of 1/(100 log2 e)) */
    const void* aux;           /* rbar: V plane of the layer (units of 1/(100 log2 e)); fbar: EX plane that is added */
    void* ex_out;              /* rbar: EX plane written; else NULL */
    const void* own;           /* weight-gradient operand of the wave's own tile: V (rbar), H (fbar), PE (kind 2) */
    float* slab;               /* [splits][M][N] f32 partial sums of this stage's entry (vdn_dw_finalize sums them) */
    float* colsum;             /* fbar stages: [splits][M] column sums of x_in (bias gradient), else NULL */
    int32_t slab_m, slab_n;    /* M, N of the slab (multiples of 32) */
} VdnSdfPipeStage;

typedef struct {
    const VdnSdfPipeStage* stages;  /* device */
    int32_t n_stages, lanes;        /* the launch has n_stages * lanes workgroups of 512 threads, 160 KiB of LDS each */
    int32_t* sync;                  /* device, 2 + n_stages * lanes words: [ticket, status, block counters ...]; zeroed by the call */
    const float* rays_o;            /* as VdnSdfRbarArgs */
    const float* rays_d;
    const float* z;
    int32_t n_per_ray, z_ld;
    int32_t P;
    float scale;
    const float* g_normals;         /* [P,3] */
    const float* g_sdf;             /* [P] */
    const int32_t* active_idx;      /* optional work list, as VdnSdfRbarArgs */
    const int32_t* n_active;
    /* written by the call's first launch from g_normals / g_sdf (the encoding's adjoint, 39 values; g_sdf / scale): */
    void* ub0;                      /* UB(0) plane [rows, 64] */
    void* ub4;                      /* UB(4) plane [rows, 288]: tiles 7, 8 */
    void* ab8;                      /* AB(8) plane [rows, 288]: tile 8 */
} VdnSdfPipeArgs;
/* status word (sync[1]) after the launch: 0 = ok, 1 = a wait for another workgroup's counter gave up (results invalid) */
int vdn_sdf_bwd_pipe_bf16(const VdnSdfPipeArgs* args_host, void* stream);
