/* libnudf -- C ABI of the MI355X-native NeuralUDF volume-rendering hot path.
 *
 * Plain pointers and sizes only (no torch types).  All pointers are DEVICE pointers to
 * contiguous fp32 unless noted; the caller owns every buffer (kernels never allocate);
 * every call is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant,
 * and returns 0 or a hipError_t value (text via nudf_last_error()).  No host syncs.
 *
 * Each entry point names the chain of PyTorch ops of the reference (xxlong0/NeuralUDF,
 * paths relative to the reference root) that it replaces -- the reference itself has no
 * native kernels and no FFI; INTEGRATION.md shows the ctypes binding a maintainer adds.
 */
#ifndef NUDF_H
#define NUDF_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int nudf_version(void);
const char* nudf_last_error(void);

/* ------------------------------------------------------------------------------------
 * Non-finite status word.  The reference stops on the host when a NaN appears -- sample_pdf's new samples
 * (models/udf_renderer_blending.py:97-101), up_sample_unbias / up_sample_no_occ_aware's (:265-269, :860-864), the eikonal
 * term of render_core (:543-544) -- each a `.cpu()` sync in the middle of a step.  Here the caller hands the library ONE
 * int32 of device memory (zeroed by the caller); nudf_upsample ORs NUDF_STATUS_NONFINITE_SAMPLES into it when a new sample
 * is not finite, nudf_composite_fwd NUDF_STATUS_NONFINITE_RENDER when a ray's composited outputs (weight sum, depth, the two
 * colours) or one of the three renderer scalars / their parameters (inv_s, beta, gamma) are not, and nudf_step_loss_fwd /
 * nudf_blend_loss_fwd (the fused single-process loss launches -- ONLY those: a ray-sharded job forms its loss from all-reduced
 * sums outside the library, and its host loop looks at that scalar itself, neuraludf_amd/train.py Trainer.check_finite)
 * NUDF_STATUS_NONFINITE_LOSS when the step's total loss is not.  (The per-sample alphas and weights themselves cannot leave
 * [0, 1]: the reference's clips are hardware min / max here, which drop a NaN operand -- a NaN network output or parameter
 * therefore shows up in the colours, the scalars and the loss, which is where the bits look.)  Nothing on the device reads the word
 * and no call waits for it: the caller polls it when it likes (UDFRendererBlending.status(), Trainer.iteration every
 * `status_every` iterations).  NULL (the default) switches the checks off.  The pointer is per process (one process per
 * GPU); it is passed to the kernels as an argument, so launches captured in a HIP graph keep the word they were captured with.
 * ---------------------------------------------------------------------------------- */
enum {
  NUDF_STATUS_NONFINITE_RENDER = 1,
  NUDF_STATUS_NONFINITE_SAMPLES = 2,
  NUDF_STATUS_NONFINITE_LOSS = 4
};
int nudf_set_status_flag(int32_t* device_word);
int32_t* nudf_status_flag(void);

/* ------------------------------------------------------------------------------------
 * Dense layer GEMMs (fp32 MFMA).  Replace F.linear + weight_norm + Softplus/ReLU and their
 * autograd (double-)backward:  models/fields.py:192-231 (UDFNetwork.forward/.gradient),
 * :452-495 (ResidualRenderingNetwork.forward), :599-628 (NeRF.forward).
 * ---------------------------------------------------------------------------------- */
enum {
  NUDF_EPI_NONE = 0,      /* C1 = (acc + bias) * scale                                   */
  NUDF_EPI_SOFTPLUS = 1,  /* C1 = softplus100(acc+bias)*scale ; C2 = softplus100'(..)     */
  NUDF_EPI_RELU = 2,      /* C1 = relu(acc+bias)*scale                                   */
  NUDF_EPI_MUL = 3,       /* C1 = acc * X1 * scale                                       */
  NUDF_EPI_MULMASK = 4,   /* C1 = (X1 > 0) ? acc*scale : 0      (ReLU backward)          */
  NUDF_EPI_TANGENT = 5,   /* s from X1 (as MULSP): C1 = acc*s*scale ; C2 = acc*X2*100*(1-s)  */
  NUDF_EPI_BWD = 6,       /* s from X1 (as MULSP): C1 = acc*scale*s + X2                   */
  NUDF_EPI_SIGMOID = 7,   /* cols < iparam: sigmoid -> C1 (and C2) ; others raw -> C3[row, col-iparam] (or C1) */
  NUDF_EPI_UDFHEAD = 8,   /* col 0: |v|*scale -> C2[row], sign -> C3[row]; col c>0 -> C1[row,c-1] */
  NUDF_EPI_SKIPSPLIT = 9, /* col < iparam: C1 = acc*s*scale (s from X1) ; else C2[col-iparam] = acc*scale */
  NUDF_EPI_RELU_DUAL = 10,/* C1 = C2 = relu(acc+bias)                                    */
  NUDF_EPI_ADDMASK = 11,  /* C1 = (X1 > 0) ? (acc + X2)*scale : 0   (ReLU backward at a join) */
  NUDF_EPI_MULSP = 12     /* C1 = acc * softplus'(.) * scale, softplus' recovered from X1 = stored softplus output * (1/xscale) */
};

typedef struct NudfGemmNN {
  const float* A; int32_t lda;      /* [M, K] row-major, K % 32 == 0, lda % 4 == 0        */
  const float* B; int32_t ldb;      /* [K, ldb] row-major (packed W^T or W), ldb % 4 == 0 */
  const float* bias;                /* [N] or NULL                                        */
  float* C1; int32_t ldc1;          /* [M, N]                                             */
  float* C2; int32_t ldc2;          /* second output or NULL                              */
  float* C3; int32_t ldc3;          /* third output (UDFHEAD: sign per row; SIGMOID: raw tail) or NULL */
  const float* X1; int32_t ldx1;    /* epilogue operand 1 [M, N] or NULL                  */
  const float* X2; int32_t ldx2;    /* epilogue operand 2 [M, N] or NULL                  */
  int32_t M, N, K;
  int32_t epi;                      /* NUDF_EPI_*                                         */
  int32_t iparam;
  float scale;
  float xscale;                     /* X1 holds softplus output / xscale (MULSP, TANGENT, BWD, SKIPSPLIT) */
} NudfGemmNN;

/* C[M,N] = epilogue(A[M,K] B[K,N]) */
int nudf_gemm_nn(const NudfGemmNN* args, void* stream);
/* tuning hook: 0 = auto tile choice (default), 1..4 force a tile/buffering variant; returns the old value */
int nudf_set_gemm_variant(int variant);

typedef struct NudfGemmTN {
  const float* A1; int32_t lda1; int32_t na1;   /* [M, na1]                               */
  const float* B1; int32_t ldb1;                /* [M, ldb1]                              */
  const float* A2; int32_t lda2; int32_t na2;   /* optional second pair (NULL to skip)    */
  const float* B2; int32_t ldb2;
  float* C; int32_t ldc;                        /* [NA, NB], accumulated with atomics     */
  float* dbias;                                 /* [NA] += column sums of A1, or NULL     */
  int32_t M, NA, NB;
  int32_t rows_per_block;                       /* 0 = choose                             */
  int32_t prec;                                 /* MFMA operand precision: 0 = fp32 (exact), 2 = bf16 operands converted
                                                   from the fp32 tiles on the fly, fp32 accumulate (config-5 mode),
                                                   3 = bf16x3: fp32 emulated on the bf16 pipe (both operands split exactly
                                                   into three bf16 parts, six products; see NudfChainStep.prec)      */
} NudfGemmTN;

/* C[NA,NB] += A1^T B1 (+ A2^T B2): weight gradients, reduction over the M points */
int nudf_gemm_tn(const NudfGemmTN* args, void* stream);

/* up to 12 single-pair problems over the same M points in one launch (all weight gradients of a layer chain); at most
 * 64 output tiles of 128 x 128 in total.  A1 / B1 16-byte aligned, lda1 / ldb1 multiples of 4. */
#define NUDF_TN_MAX_PROBLEMS 12
typedef struct NudfGemmTNProblem {
  const float* A1; const float* B1;             /* [M, lda1], [M, ldb1]                   */
  float* C; float* dbias;                       /* [NA, ldc] +=, [NA] += or NULL          */
  int32_t lda1, ldb1, ldc, NA, NB;
  int32_t tile_start;                           /* unused (kept for layout compatibility) */
  int32_t flags;                                /* NUDF_TN_A16 / NUDF_TN_B16: that operand is stored as bf16 (leading
                                                   dimension in elements, a multiple of 8); widened to fp32 on chip    */
} NudfGemmTNProblem;
#define NUDF_TN_A16 1
#define NUDF_TN_B16 2
#define NUDF_TN_A_BLK 4                         /* fp32 operand in the BLOCKED layout of NudfChainStep (rows padded to 32) */
#define NUDF_TN_B_BLK 8
#define NUDF_TN_A_P4 16                         /* with NUDF_TN_A16: the bf16 operand is 4-point packed like the chains'
                                                   16-bit stored state (NUDF_CH_STATE16): element (row, col) at
                                                   ((row / 4) ld + col) 4 + row % 4, rows padded to a multiple of 4    */
#define NUDF_TN_B_P4 32
typedef struct NudfGemmTNGroup {
  int32_t n_problems, M, rows_per_block, total_tiles;   /* rows_per_block 0 = choose      */
  int32_t prec;                                 /* as NudfGemmTN.prec; 4 = f16x2 (see amax_a / amax_b below) */
  NudfGemmTNProblem prob[NUDF_TN_MAX_PROBLEMS];
  float* workspace;                             /* NULL: partial tiles are accumulated with fp32 atomics.  Otherwise a
                                                   16-byte aligned scratch of >= nudf_gemm_tn_grouped_workspace()
                                                   floats: every workgroup stores its partial tile there and a second
                                                   kernel adds them to C / dbias in a FIXED order (run-to-run identical
                                                   results).  Must not be shared by launches that can overlap.  With a
                                                   workspace the problems of ONE group must write pairwise disjoint C
                                                   and dbias ranges (that reduction is not atomic; overlapping groups
                                                   are refused with hipErrorInvalidValue); consecutive launches may
                                                   accumulate into the same C.                                       */
  int64_t workspace_floats;
  int32_t assign;                               /* != 0 (workspace path only): the reduction ASSIGNS C and dbias (0 + the
                                                   ordered sum) instead of adding to them -- the first launch into a
                                                   gradient buffer then needs no zero fill.  Elements the group does not
                                                   cover (rows >= NA, columns >= NB) are left untouched.               */
  /* prec 4 (f16x2: three fp16 MFMA products per fp32 product, NudfChainStep.prec 4) only: device scalars holding max |x| over
     ALL A operands / ALL B operands of the group (written by the sweeps that produce them, NudfChain.absmax_out), or NULL for a
     side whose operands lie in fp16's range as they are (activations).  Each side is multiplied by
     2^(10 - floor(log2 max)) before the split and C by the reciprocals afterwards -- exact; a scaled element beyond +-60 000
     (a maximum that was not reported) is clamped. */
  const float* amax_a;
  const float* amax_b;
} NudfGemmTNGroup;
int nudf_gemm_tn_grouped(const NudfGemmTNGroup* args, void* stream);
/* floats of workspace this group needs (0 for an empty group, < 0 on an invalid one) */
int64_t nudf_gemm_tn_grouped_workspace(const NudfGemmTNGroup* args);
/* the launch plan, for tests (host code only): out[4 b ..] = {problem, tile row * 256 + tile column, row chunk, workspace
 * slot} of workgroup b, or {-1, -1, -1, -1} for a HOLE (a workgroup that exits at once); returns the number of workgroups
 * LAUNCHED (at most `capacity` are written), < 0 on an invalid group.  Workgroups b and b' with b % 8 == b' % 8 run on the
 * same XCD: the tiles of a problem that read the same row chunk are given the same b % 8 -- every chunk, the holes are what
 * that costs -- and no XCD receives more live workgroups than it holds at once (64). */
int nudf_gemm_tn_grouped_plan(const NudfGemmTNGroup* args, int32_t* out, int capacity);
/* tuning / measurement bits, returns the old value: 2 = drop the epilogue (TIMING ONLY: results are discarded), 4 = skip
 * the bias sums, 8 = ignore the workspace (atomics), 16 = equal row chunks for every tile instead of the cost-weighted
 * split, 32 = no 2 x 2 quadrant layout for full tiles, 64 = tile-major workgroup order instead of the XCD-aware one,
 * 128 = full fp32 tiles through the generic k-loop instead of the interleaved one, 256 = 16-bit MFMA mode (prec != 0)
 * through the generic kernel's fp32 LDS image instead of the packed k-pair image (bit-identical C either way), 512 = bf16x3
 * mode through the generic kernel (split on the way out of the fp32 image), 1024 = bf16x3 mode through the WIDE kernel (one
 * 8-wave workgroup per pair of vertically adjacent tiles, double-buffered split image; bit-identical on equal row chunks,
 * measured slower: opt-in).  Default 0 (env NUDF_TN_FLAGS). */
int nudf_set_tn_flags(int flags);
/* tuning only: device buffer of >= 4 int64 per workgroup receiving {start, end} (wall_clock64, 100 MHz), tile layout * 16
 * + live sub-tiles per wave | CU identity (HW_ID [15:8] << 8, XCC_ID << 32), k-steps | shader-clock ticks of the same
 * span << 16; NULL switches it off */
int nudf_set_tn_debug(void* buf);


/* ------------------------------------------------------------------------------------
 * Fused UDF -> density -> alpha -> composite (forward / backward), one wavefront per ray.
 * Replaces models/udf_renderer_blending.py:352-362, 370-423, 484-553 (+ :151-159, :292-320).
 * ---------------------------------------------------------------------------------- */
typedef struct NudfComposite {
  const float* rays_o; const float* rays_d;   /* [N,3]                                     */
  const float* z;                              /* [N,S] sorted sample positions             */
  const float* udf;                            /* [N,S] unsigned distance at the mid points */
  const float* grad;                           /* [N,S,3] d udf / d x                       */
  const float* color;                          /* [N,S,3] view-dependent colour (sigmoid'd) */
  const float* color_base;                     /* [N,S,3]                                   */
  const float* bg_z;                           /* [N,n_out] outside sample positions / NULL */
  const float* bg_sigma;                       /* [N,n_out] raw NeRF density                */
  const float* bg_color;                       /* [N,n_out,3]                               */
  const float* scal;                           /* [3] device: inv_s, beta, gamma (clipped)  */
  const float* sample_dist;                    /* [1] device                                */
  const float* background_rgb;                 /* [3] device or NULL                        */
  int32_t N, S, n_out, s_nominal;              /* s_nominal: #weights summed into weight_sum */
  int32_t has_anneal; float cos_anneal;        /* cos_anneal_ratio (None -> has_anneal = 0) */
  float flip_saturation;
  int32_t use_norm_grad;
  float sparse_scale;
  int32_t alpha_type;                          /* sdf2alpha_type: 0 'numerical' (every shipped conf), 1 'theorical'
                                                  (udf_renderer_blending.py:321-323); occupies what was padding   */
  /* outputs */
  float* weights;                              /* [N,S+n_out]                               */
  float* out_color; float* out_color_base;     /* [N,3]                                     */
  float* out_depth;                            /* [N]                                       */
  float* out_normals;                          /* [N,3]                                     */
  float* out_wsum; float* out_wsum_all;        /* [N]                                       */
  float* sums;                                 /* [5] {eik_num, eik_den, eikns_num, eikns_den, sparse_sum}: assigned when ws is
                                                  given, += (caller zeroes) on the atomic path */
  float* ws;                                   /* scratch [5 * ceil(N/4)] for per-block partial sums (two-stage,
                                                  deterministic; NULL = one atomic per block and sum, which serialises
                                                  at the memory side for large N) */
  /* optional diagnostics, [N,S] each or NULL */
  float* o_alpha; float* o_alpha_plus; float* o_alpha_minus; float* o_vis_prob;
  float* o_alpha_occ; float* o_raw_occ; float* o_true_cos; float* o_grad_mag;
  float* o_mid_z; float* o_dists; float* o_inside; float* o_flip;
  const float* sched;                          /* [2] device {cos_anneal_ratio, flip_saturation} overriding the two by-value
                                                  fields above, or NULL: lets a captured HIP graph of the train step (kernel
                                                  arguments frozen at capture) follow the per-iteration schedules
                                                  (exp_runner_blending.py:193-228); has_anneal stays by value            */
  /* the renderer scalars formed in this launch instead of by nudf_scalars_fwd (one launch less each way):
     p_variance != NULL -> `scal` is ignored; inv_s / beta / gamma come from the three 1-element parameters exactly as
     nudf_scalars_fwd forms them (beta_hi = 1 / beta_min), and the forward writes them to scal_out [3] / recip_out [2]
     (= 1 / inv_s, 1 / beta) when given */
  const float* p_variance; const float* p_beta; const float* p_gamma;
  float beta_hi;
  int32_t defer_sums;                          /* != 0 (needs ws): the per-block partials stay in ws and `sums` is NOT
                                                  written -- nudf_partial_sums(ws, ceil(N/4), 5, sums) or the ws argument
                                                  of nudf_step_loss_fwd finishes the reduction in the consumer's launch */
  float* scal_out; float* recip_out;
} NudfComposite;

typedef struct NudfCompositeGrad {
  /* upstream (any may be NULL = zero) */
  const float* d_color; const float* d_color_base;   /* [N,3]                              */
  const float* d_weights;                             /* [N,S+n_out]                        */
  const float* d_depth;                               /* [N]                                */
  const float* d_normals;                             /* [N,3]                              */
  const float* d_wsum; const float* d_wsum_all;       /* [N]                                */
  const float* d_sums;                                /* [5] device                         */
  /* results */
  float* o_d_udf;                                     /* [N,S]                              */
  float* o_d_grad;                                    /* [N,S,3]                            */
  float* o_d_color; float* o_d_color_base;            /* [N,S,3] or NULL                    */
  float* o_d_bg_sigma;                                /* [N,n_out] or NULL                  */
  float* o_d_bg_color;                                /* [N,n_out,3] or NULL                */
  float* o_d_scal;                                    /* [3] d inv_s, d beta, d gamma: assigned when ws is given, += otherwise */
  float* ws;                                          /* scratch [3 * ceil(N/4)] or NULL (see NudfComposite.ws) */
  float* o_d_param;                                   /* [3] d variance, d beta, d gamma (needs p_variance and ws): the
                                                         reduction of the partials and nudf_scalars_bwd as one launch */
} NudfCompositeGrad;

int nudf_composite_fwd(const NudfComposite* args, void* stream);
int nudf_composite_bwd(const NudfComposite* args, const NudfCompositeGrad* grads, void* stream);
/* out[k] = sum over the nblk rows of ws [nblk, K] (K <= 8), fixed order: the second stage of the composite kernels'
 * deterministic batch sums, for a caller that asked them to leave the partials (NudfComposite.defer_sums) */
int nudf_partial_sums(const float* ws, int nblk, int K, float* out, void* stream);
/* per-ray colours from the per-32-point partial sums a colour chain left (NudfChainStep.row_sums [N S / 32, 4]):
 * out[ray][c] = sum of the ray's S / 32 blocks (+ background_rgb[c] (1 - wsum_all[ray]) for `out_color`, :527-528).
 * NudfComposite.color == NULL makes nudf_composite_fwd compute everything BUT the two colour sums (weights first). */
int nudf_composite_colour_finish(const float* sums_color, const float* sums_color_base, int N, int S,
                                 const float* background_rgb, const float* wsum_all, float* out_color,
                                 float* out_color_base, void* stream);
/* Layout of the FULL case (S = 128 / 256 / 512 inside samples, no outside samples, no diagnostics): 1 = lane l owns
 * S/64 consecutive samples (16-byte vector accesses), 0 (default) = sample i in lane i % 64 for every shape.  Same
 * arithmetic, different association of the two product scans; A-B switch (the two measure the same on MI355X). */
void nudf_set_composite_blocked(int on);

/* ------------------------------------------------------------------------------------
 * Hierarchical importance re-sampling (no autograd), one wavefront per ray.
 *   nudf_upsample: up_sample_unbias (mode 0, models/udf_renderer_blending.py:197-272) or
 *                  up_sample_no_occ_aware (mode 1, :834-866) fused with sample_pdf(det=True) (:66-104)
 *   nudf_merge   : sorted merge of cat_z_vals (:278-288), carrying udf along (udf* may be NULL)
 * ---------------------------------------------------------------------------------- */
typedef struct NudfUpsample {
  const float* rays_o; const float* rays_d;   /* [N,3]                                     */
  const float* z; const float* udf;           /* [N,M] current samples and their udf       */
  const float* u;                              /* [K] quantiles linspace(.5/K, 1-.5/K, K)   */
  const float* sample_dist;                    /* [1] device                                */
  const float* gamma_dev;                      /* [1] device gamma (mix schedule) or NULL   */
  int32_t N, M, K, mode;                       /* mode & 0xff: 0 up_sample_unbias, 1 up_sample_no_occ_aware;
                                                  | NUDF_UP_THEORICAL: sdf2alpha_type 'theorical' in up_sample_unbias
                                                  | NUDF_UP_SERIAL | NUDF_UP_NOCONTRACT: see below */
  float inv_s, beta, gamma;
  float* z_new;                                /* [N,K] ascending                           */
  float* pts_new;                              /* [N*K,3] o + d*z_new, or NULL              */
  float* dbg;                                  /* NULL, or [N, NUDF_UP_DBG_ROWS, M]: the intermediates of up_sample_unbias per
                                                  section -- rows cos_val, vis_prob, alpha_plus, alpha_minus, alpha, weights
                                                  (before sample_pdf's + 1e-5), cdf -- for the stage-wise comparison with the
                                                  reference's tensors (scripts/upsample_first_diff.py); mode 0 only */
  /* the PREVIOUS round's merge (nudf_merge) folded into this launch: merge_K > 0 -> the current samples are the sorted
     merge of prev_z / prev_udf [N, M - merge_K] with add_z / add_udf [N, merge_K]; z / udf above are ignored, the merged
     lists are also written to z_merged / udf_merged [N, M] (same values as nudf_merge writes: data movement only) */
  const float* prev_z; const float* prev_udf;
  const float* add_z; const float* add_udf;
  float* z_merged; float* udf_merged;
  int32_t merge_K;
} NudfUpsample;
#define NUDF_UP_DBG_ROWS 7
#define NUDF_UP_THEORICAL 256
#define NUDF_UP_SERIAL 512      /* the three scans in torch-CPU order: one running DOUBLE accumulator per row, every output
                                   rounded to float (cumprod / cumsum of a float tensor on the CPU); default: wave-parallel
                                   fp32 scans (what the same torch ops do on a GPU) */
#define NUDF_UP_NOCONTRACT 1024 /* kernel build without floating-point contraction: every product / sum of the reference's
                                   op chain rounded separately, as separate torch ops round them */
#define NUDF_UP_SLEEF 2048      /* sigmoid as torch's CPU kernel forms it: 1 / (1 + exp(-x)) with a true division and
                                   Vectorized<float>::exp = Sleef expf u10 (Cody-Waite ln 2 split, degree-5 polynomial in FMAs,
                                   two-step ldexp), restated bit for bit (scripts/sleef_expf_check.py: 1 of 4.5 M sigmoids
                                   differs from torch's on an AVX-512 host).  torch.exp itself goes through MKL's vsExp on
                                   contiguous tensors and is NOT covered: those call sites keep libm's expf */
#define NUDF_UP_EXPCR 4096      /* the exp call sites (udf2logistic, alpha_occ: torch.exp = MKL vsExp HA on the CPU) evaluated in
                                   double and rounded once: the correctly rounded value, which MKL's result equals on 98.9 % of
                                   arguments; default: libm expf */
int nudf_upsample(const NudfUpsample* args, void* stream);
int nudf_merge(const float* z, const float* udf, const float* z_new, const float* udf_new, int N, int M,
               int K, float* z_out, float* udf_out, void* stream);
/* the LAST merge of the schedule (z only, :278-288) + the section mid points the renderer evaluates next
 * (nudf_ray_points mode 1, :352-357) in one launch: z_out [N, M + K], pts [N (M + K), 3].  xrows != NULL: the points are
 * also written to columns 0..2 of the [N (M + K), ldx] row array at xrows and columns 3..xcols-1 are zeroed -- the
 * [feat | pts | 0] input buffer of the colour network (what nudf_copy_cols + a pad fill did).  Same arithmetic as the
 * separate launches. */
int nudf_merge_points(const float* z, const float* z_new, int N, int M, int K, float* z_out, const float* rays_o,
                      const float* rays_d, const float* sample_dist, float* pts, float* xrows, int ldx, int xcols,
                      void* stream);

/* ------------------------------------------------------------------------------------
 * Pixel / patch blending and the SSIM patch loss (gradients reach the blending logits and the
 * compositing weights only, as in the reference).
 * ---------------------------------------------------------------------------------- */
typedef struct NudfPixelBlend {   /* patch_projector.py:21-43, projector_utils.py:8-85, fields.py:498-519 */
  const float* pts;               /* [P,3] sample points                                    */
  const float* logits; int32_t nl;/* [P,nl] blending logits (first V columns used)          */
  const float* proj;              /* [V,12] row-major 3x4 = K_v[:3,:3] @ w2c_v[:3,:]        */
  const float* imgs;              /* [V,3,H,W] source images (img_layout 0) or [V,H,W,3] (img_layout 1) */
  int32_t P, V, H, W;
  float* pix;                     /* [P,3] blended pixel colour per sample                  */
  int32_t img_layout;             /* 1: channel-interleaved texels = the memory behind the reference's
                                     images[src_idx].permute(0,3,1,2) view (dataset.py:147-149); one 12-byte load
                                     per texel instead of three plane loads                              */
} NudfPixelBlend;
int nudf_pixel_blend_fwd(const NudfPixelBlend* a, void* stream);
int nudf_pixel_blend_bwd(const NudfPixelBlend* a, const float* d_pix, float* d_logits, void* stream);

typedef struct NudfPixelComposite {   /* udf_renderer_blending.py:503-518 */
  const float* w;                 /* [N,S+n_out] compositing weights                        */
  const float* pix;               /* [N,S,3]                                                */
  const float* pts;               /* [N,S,3] (inside-sphere test), used when bg_in != NULL  */
  const float* bg_in;             /* [N,S,3] background colour at the inside samples / NULL */
  const float* bg_tail;           /* [N,n_out,3] / NULL                                     */
  int32_t N, S, n_out;
  float* out;                     /* [N,3]                                                  */
} NudfPixelComposite;
int nudf_pixel_composite_fwd(const NudfPixelComposite* a, void* stream);
int nudf_pixel_composite_bwd(const NudfPixelComposite* a, const float* d_out, float* d_w, float* d_pix,
                             float* d_bg_in, float* d_bg_tail, void* stream);

typedef struct NudfPatchBlend {   /* patch_projector.py:45-164, fields.py:521-535, udf_renderer_blending.py:520-524 */
  const float* pts;               /* [N,S,3]                                                */
  const float* grad;              /* [N,S,3] d udf/dx (normal = flip * grad/(|grad|+1e-5))  */
  const float* rays_d;            /* [N,3]                                                  */
  const float* uv;                /* [N,2] ray pixel in the reference image (pixel units)   */
  const float* logits; int32_t nl;/* [N,S,nl]                                               */
  const float* w; int32_t ldw;    /* [N,ldw] compositing weights (first S used)             */
  const float* ref_cam;           /* [24] K_ref^-1 | R_ref | t_ref | cam centre             */
  const float* src_cam;           /* [V,24] K_src | R_rel | t_rel | -R_rel^T t_rel          */
  const float* imgs;              /* [V,3,H,W] (img_layout 0) or [V,H,W,3] (img_layout 1)     */
  int32_t N, S, V, H, W, hps;
  float* patch_colors;            /* [N,(2h+1)^2,3]                                         */
  float* patch_mask;              /* [N] = sum_s w_s [any view sees the whole patch]        */
  int32_t img_layout;             /* as in NudfPixelBlend                                   */
} NudfPatchBlend;
int nudf_patch_blend_fwd(const NudfPatchBlend* a, void* stream);
int nudf_patch_blend_bwd(const NudfPatchBlend* a, const float* d_patch, float* d_logits, float* d_w, void* stream);

/* Un-fused warps = the reference's projector interface itself (PatchProjector.pixel_warp / .patch_warp,
 * models/patch_projector.py:21-43, 45-164): per-VIEW samples and validity masks, forward only (the reference builds
 * the homographies under no_grad and the sample positions carry no gradient to any parameter).  The training path
 * uses the fused kernels above and never materialises these tensors. */
int nudf_pixel_warp(const NudfPixelBlend* a /* logits, pix unused */, float* colors /* [P,V,3] */,
                    float* mask /* [P,V] 0/1 */, void* stream);
typedef struct NudfPatchWarp {
  const float* pts;               /* [N,S,3]                                                */
  const float* normals;           /* [N,S,3] plane normals in world space                   */
  const float* uv;                /* [N,2] ray pixel in the reference image (pixel units)   */
  const float* ref_cam;           /* [24], as in NudfPatchBlend                             */
  const float* src_cam;           /* [V,24]                                                 */
  const float* imgs;              /* [V,3,H,W] (img_layout 0) or [V,H,W,3] (img_layout 1)   */
  int32_t N, S, V, H, W, hps, img_layout;
  float* colors;                  /* [N,S,V,(2h+1)^2,3]                                     */
  float* mask;                    /* [N,S,V,(2h+1)^2] 0/1                                   */
} NudfPatchWarp;
int nudf_patch_warp(const NudfPatchWarp* a, void* stream);

/* loss/patch_metric.py:21-41, 76-84; d_out/d_pred NULL = forward only */
int nudf_ssim_patch(const float* pred, const float* gt, const float* window, int N, int Npx, float* out,
                    const float* d_out, float* d_pred, void* stream);
/* every patch error of ColorPatchLoss (loss/loss.py:66-73, loss/patch_metric.py:44-67): type 0 'ssim', 1 'l1', 2 'ssd',
 * 3 'ncc' (returned as 1 - ncc); pred / gt [N, Npx, 3], window [Npx] (Gaussian, used by ssim / ncc), out [N] */
int nudf_patch_metric(int type, const float* pred, const float* gt, const float* window, int N, int Npx, float* out,
                      const float* d_out, float* d_pred, void* stream);

/* ------------------------------------------------------------------------------------
 * The BLENDING step's loss (BASELINE config 3) in three launches around the caller's sort: ColorLoss with its pixel and
 * trimmed patch terms (loss/loss.py:105-133, :21-44, :66-84), the patch-mask algebra of the runner
 * (exp_runner_blending.py:313-315), the three regularisers from the composite sums and the weighted total (:330-371).
 *   nudf_blend_loss_prepare: m[i] = ((patch_mask[i] * (weight_sum[i] > 0.5)) > 0) as 0 / 1, err_masked[i] = err[i] m[i]
 *   caller: err_sorted, order = sort(err_masked, descending)   (the trimmed mean needs order statistics)
 *   nudf_blend_loss_fwd:     out[12] = {total, colour total, Lb, Lc, Lpix, Lpatch, gradient_error, gradient_error_near_surface,
 *                            sparse_error, den_pix, k = floor(trim_ratio * sum m), kept count}
 *   nudf_blend_loss_bwd:     d cb, d c, d pix [3 N], d err [N] (through the sort's permutation), d sums [5]
 * The weights are read from the device vector w_dev (NUDF_LW_*).  ORs NUDF_STATUS_NONFINITE_LOSS like nudf_step_loss_fwd.
 * ---------------------------------------------------------------------------------- */
typedef struct NudfBlendLoss {
  const float* cb; const float* c; const float* pix; const float* gt;   /* [N,3]                                     */
  const float* err;            /* [N] per-ray patch error (nudf_patch_metric)                                        */
  const float* patch_mask;     /* [N] the renderer's patch validity weight (render result 'patch_mask')              */
  const float* weight_sum;     /* [N]                                                                                */
  const float* err_sorted;     /* [N] descending sort of err_masked (fwd / bwd)                                      */
  const int64_t* order;        /* [N] its permutation                                                                */
  float* m;                    /* [N] prepare: out; fwd / bwd: in                                                     */
  float* err_masked;           /* [N] prepare: out                                                                   */
  float* sums;                 /* [5] composite sums: read -- or, with sums_ws, reduced here and written              */
  const float* sums_ws;        /* NULL, or the composite launch's per-block partials (NudfComposite.defer_sums)      */
  const float* w_dev;          /* [NUDF_LW_COUNT] device                                                             */
  float* out;                  /* [12] fwd: out; bwd: in                                                             */
  const float* d_total;        /* bwd: device scalar upstream gradient of out[0] (NULL = 1)                          */
  float* d_cb; float* d_c; float* d_pix;   /* [N,3] bwd out                                                          */
  float* d_err;                /* [N] bwd out (every entry written)                                                  */
  float* d_sums;               /* [5] bwd out                                                                        */
  int32_t N, sums_nblk;
  float n_rays, trim_ratio;
} NudfBlendLoss;
int nudf_blend_loss_prepare(const NudfBlendLoss* args, void* stream);
int nudf_blend_loss_fwd(const NudfBlendLoss* args, void* stream);
int nudf_blend_loss_bwd(const NudfBlendLoss* args, void* stream);

/* ------------------------------------------------------------------------------------
 * ray sampling helpers (models/udf_renderer_blending.py:605-630, 352-357, 164-173, 205)
 * ---------------------------------------------------------------------------------- */
int nudf_coarse_z(const float* near, const float* far, int nf_stride, const float* t_rand, int N, int S,
                  float* z, float* sample_dist, void* stream);
/* nudf_coarse_z + the points o + d z of those samples (nudf_ray_points mode 0) in ONE launch -- the start of the
 * hierarchical sampling.  center != 0: t_rand holds the raw U[0, 1) draws and the - 0.5 of :618 is applied here
 * (one rounded subtraction, as the separate torch op).  pts [N S, 3] or NULL. */
int nudf_coarse_start(const float* near, const float* far, int nf_stride, const float* t_rand, int center, int N, int S,
                      float* z, float* sample_dist, const float* rays_o, const float* rays_d, float* pts, void* stream);
int nudf_outside_z(const float* far, int f_stride, const float* lin, int N, int n_out, int n_samples,
                   float* z_out, void* stream);
int nudf_ray_points(const float* rays_o, const float* rays_d, const float* z, const float* sample_dist, int N,
                    int S, int mode, float* pts, void* stream);

/* positional encoding value / JVP (tangent != NULL) / VJP  (models/embedder.py:15-36) */
int nudf_posenc(const float* x, int xld, int xdiv, const float* tangent, int D, int L, float in_scale, int P,
                float* dst1, int ld1, float scale1, float* dst2, int ld2, float scale2, void* stream);
int nudf_posenc_vjp(const float* x, int xld, int D, int L, float in_scale, int P, const float* src1, int ld1,
                    float scale1, const float* src2, int ld2, float scale2, float* g, void* stream);
int nudf_copy_cols(const float* src, int lds, int sdiv, float* dst, int ldd, int ncols, int P, float scale,
                   void* stream);
int nudf_add_cols(const float* a, int lda, const float* b, int ldb, float* out, int ldo, int P, int C, void* stream);

/* heads of the MLP chains */
int nudf_udf_grad_seed(const float* sign, const float* w_row0, const float* h, int ldh, float hscale, int P, int C,
                       float inv_scale, float* out, int ldo, void* stream);
int nudf_udf_head_bwd(const float* sign, const float* dudf, const float* dfeat, int ldf, int P, int F,
                      float scale, float* out, int ldo, void* stream);
int nudf_signed_colsum(const float* sign, const float* R, int ldr, int P, int C, float scale, float* out,
                       void* stream);
int nudf_sigmoid_head_bwd(const float* y, const float* dy, const float* dy_extra, int ldx, int nsig,
                          const float* draw, int ldr, int nraw, int P, float* out, int ldo, void* stream);

/* weight_norm packing (torch.nn.utils.weight_norm at fields.py:175-176, 433-446) and its backward */
int nudf_weightnorm_pack(const float* v, const float* g, int out, int in, const int* perm, float* W, int ldw,
                         float* Wt, int ldwt, float* inv_norm, void* stream);
int nudf_weightnorm_unpack_grad(const float* dW, int ldw, const float* v, const float* g, const float* inv_norm,
                                int out, int in, const int* perm, float* dv, float* dg, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-tensor Adam (one launch for all parameters).  Replaces the foreach kernels of
 * torch.optim.Adam.step() as the runner uses it (exp_runner_blending.py:136-139, :373-375);
 * same arithmetic, same state (exp_avg, exp_avg_sq, step).  The table is passed by value.
 * ---------------------------------------------------------------------------------- */
#define NUDF_ADAM_MAX_TENSORS 64
#define NUDF_ADAM_MAX_GROUPS 4
typedef struct NudfAdamTensor {
  float* p; const float* g; float* m; float* v;  /* param, grad, exp_avg, exp_avg_sq (device) */
  int32_t n;                                     /* elements                                   */
  int32_t group;
  float neg_step_size;                           /* -lr / (1 - beta1^t), t = this tensor's step */
  float bc2_sqrt;                                /* sqrt(1 - beta2^t)                          */
} NudfAdamTensor;
typedef struct NudfAdamGroup {
  float one_minus_beta1, beta2, one_minus_beta2, eps;
} NudfAdamGroup;
typedef struct NudfAdam {
  int32_t n_tensors; int32_t pad_;
  NudfAdamTensor t[NUDF_ADAM_MAX_TENSORS];
  int32_t block_start[NUDF_ADAM_MAX_TENSORS + 1]; /* prefix sums of ceil(n / nudf_adam_chunk())  */
  int32_t pad2_;
  NudfAdamGroup group[NUDF_ADAM_MAX_GROUPS];
  const float* dyn;                               /* NULL, or device [2 * n_tensors]: {neg_step_size, bc2_sqrt} of tensor i at
                                                     dyn[2 i], dyn[2 i + 1], overriding the by-value fields: the two numbers
                                                     that change every step (learning-rate schedule, bias corrections), so
                                                     that a captured HIP graph of the step stays valid               */
} NudfAdam;
int nudf_adam_step(const NudfAdam* args, void* stream);
int nudf_adam_chunk(void);                         /* elements one block updates                  */

/* ------------------------------------------------------------------------------------
 * Fused MLP layer chains: a tile of points stays resident in LDS across all layers of a sweep
 * (csrc/mlp_chain.hip).  Replaces, for UDFNetwork (models/fields.py:192-231), the same chains of
 * F.linear + weight_norm + Softplus and their autograd (double) backward as nudf_gemm_nn, with one
 * launch per sweep.  Weights are passed in MFMA-B fragment order (nudf_pack_frag).
 * ROW PADDING: the kernel stores whole 64-point tiles, so every buffer it writes (C1, C2, G0) or reads in
 * an epilogue (X1, X2, r1_row) must hold roundup(P, 64) rows; rows >= P are scratch.  x, v, A0 of
 * INIT_LOAD and seed_sign are read with clamped row indices and need only P rows.  P * ld < 2^31.
 * ---------------------------------------------------------------------------------- */
enum {
  NUDF_CH_NONE = 0,      /* out = (acc + bias) * scale                                            */
  NUDF_CH_SOFTPLUS = 1,  /* out = softplus100(acc + bias) * scale                                 */
  NUDF_CH_MULSP = 2,     /* out = acc * softplus'(X1) * scale ; cols >= iparam > 0: C2[col-iparam] = acc*scale */
  NUDF_CH_TANGENT = 3,   /* out = acc * s * scale ; C2 = acc * X2 * 100 (1 - s),  s = softplus'(X1) */
  NUDF_CH_BWD = 4,       /* out = acc * scale * softplus'(X1) + X2                                */
  NUDF_CH_UDFHEAD = 5,   /* col 0: C2[row] = |acc + bias| * scale, C1[row] = sign                 */
  NUDF_CH_RELU = 6,      /* out = relu(acc + bias) ; C2 (optional) mirrors it                     */
  NUDF_CH_SIGMOIDN = 7,  /* cols < iparam: sigmoid -> C1 (and tile); cols >= iparam raw -> C2[col-iparam];
                            N <= iparam: C2 mirrors the sigmoid columns instead                    */
  NUDF_CH_MULMASK = 8,   /* out = (X1 > 0) ? acc * scale : 0            (ReLU backward)           */
  NUDF_CH_ADDMASK = 9,   /* out = (X1 > 0) ? (acc + X2) * scale : 0     (ReLU backward at a join) */
  NUDF_CH_RELUADD = 10   /* out = relu(acc + bias + X2)   (skip layer whose second input part was multiplied
                            by an earlier step: NeRF's cat([input_pts, h]) is wider than the LDS tile)  */
};
enum {
  NUDF_CH_INIT_LOAD = 0,   /* activation tile = A0[rows, 0:k0]                                    */
  NUDF_CH_INIT_POSENC = 1, /* activation tile = PE(x) (or its JVP with tangent v), zero-padded to k0 */
  NUDF_CH_INIT_SEED = 2    /* tile[r,c] = seed_sign[r] * seed_wrow[c] * seed_scale * softplus'(A0[r,c]) */
};
#define NUDF_CH_MAX_STEPS 14
typedef struct NudfChainStep {
  const float* Bp;                 /* packed weights (nudf_pack_frag) of the [K, N] operand          */
  const float* bias;               /* [N] or NULL                                                    */
  const float* X1; const float* X2;/* epilogue operands [P, ld] stored by an earlier sweep, or NULL   */
  float* C1; float* C2;            /* HBM copies of the outputs (NULL = keep in LDS only)            */
  const float* r1_row;             /* optional rank-1 term: acc += r1_row[row] * r1_col[col]         */
  const float* r1_col;
  int32_t K, N;                    /* K % 16 == 0, K <= 288, N <= 256                                */
  int32_t epi;                     /* NUDF_CH_*                                                      */
  int32_t iparam;
  int32_t ldx1, ldx2, ldc1, ldc2;
  int32_t ldr1;                    /* element stride of r1_row                                       */
  int32_t prec;                    /* MFMA operand precision: 0 = fp32 (v_mfma_f32_32x32x2_f32), 1 = fp16, 2 = bf16 (fp32
                                      accumulate; Bp then holds the 16-bit fragment layout of
                                      nudf_weightnorm_pack_multi), 3 = bf16x3: fp32 EMULATED on the bf16 matrix pipe --
                                      both operands split exactly into three bf16 parts (x = hi + mid + lo), the six
                                      products hi hi, hi mid, mid hi, hi lo, lo hi, mid mid on v_mfma_f32_32x32x16_bf16
                                      with fp32 accumulation; the dropped terms are <= 2^-23 |x| |y| per product, the size
                                      of one fp32 rounding (Bp: the three-plane layout of NudfPackFrag.dtype 3);
                                      4 = f16x2: fp32 EMULATED on the fp16 matrix pipe with THREE products -- both operands
                                      split into two fp16 parts, x = hi + 2^-11 lo (hi = fp16(x), lo = fp16((x - hi) 2^11):
                                      11 + 11 significant bits, the low part pre-scaled into fp16's normal range),
                                      acc0 += hi hi', acc1 += hi lo' + lo hi' on v_mfma_f32_32x32x16_f16, result
                                      acc0 + 2^-11 acc1 (the correction terms in their own fp32 accumulator); dropped:
                                      lo lo' <= 2^-22 |x| |y|.  For operands inside fp16's range (|x| < 65504; below
                                      6e-5 the split keeps an ABSOLUTE resolution of ~3e-11): the forward-order sweeps
                                      (encodings, activations, weight-normed weights) as they are; the backward sweeps,
                                      whose operands are adjoints of the loss, with NudfChain.tile_scale
                                      (Bp: the two-plane layout of NudfPackFrag.dtype 4)                                */
  int32_t act_write;               /* 1: the outputs become the next step's activation tile          */
  int32_t act_col0;                /* ... at tile columns [act_col0, act_col0 + N)                   */
  int32_t pe_tail_col;             /* >= 0: afterwards write PE(x)*pe_tail_scale at these tile columns ...       */
  int32_t ld_pe;
  float* pe_dst;                   /* ... and at the same columns of pe_dst [P, ld_pe] (or NULL)                */
  float pe_tail_scale;
  float scale, xscale;
  int32_t layout;                  /* NUDF_CH_BLK_* bits: which of this step's [P, ld] buffers use the BLOCKED layout */
  /* SIGMOIDN steps of the workgroup-shared kernel only -- the compositing sum of a colour head taken inside its epilogue
     (udf_renderer_blending.py:508-526: (colour * weights[..., None]).sum(dim=1)), so that the per-sample colours need not
     go to memory: row_w [rows padded to 64, zero tail] = the compositing weight of every point; row_sums [ceil(P / 32), 4]:
     row_sums[b][c] = sum over the 32 points of block b of row_w[r] * sigmoid(v[r][c]), c < min(iparam, 4) (fixed order).
     With rays of S % 32 == 0 samples the blocks of a ray are consecutive: nudf_composite_colour_finish adds them up.
     C1 / C2 may be NULL then. */
  const float* row_w;
  float* row_sums;
  /* NUDF_CH_BWD steps only (workgroup-shared kernel, split / 16-bit modes): a THIRD stored operand.  With X3 = NULL the step
     adds X2 = EX[l-1], the second-order term the tangent sweep stored (EX = (R W^T) * DA * softplus'' / softplus').  With X3 set
     the term is formed here instead, from arrays that exist anyway -- X2 = R[l] (the tangent sweep's activations, which the
     second-order weight gradient reads) and X3 = DA[l-1] (the input-gradient sweep's adjoints):
         EX = X2 * X3 * 100 (1 - s) / (s * scale),   s = softplus'(.) recovered from X1 as everywhere, 0 where s = 0
     (R = (R W^T) s scale, so R / (s scale) is the pre-activation tangent again).  The tangent sweep then neither reads DA nor
     writes EX: two of its four arrays per layer (fields.py:219-231 with create_graph=True is the reference's form of all of this). */
  const float* X3;
  int32_t ldx3;
} NudfChainStep;
/* Blocked layout of a [P, ld] buffer (P padded to 32 rows, ld % 4 == 0): element (r, c) lives at
 *   (r / 32) * 32 * ld + (c / 4) * 128 + (r % 32) * 4 + (c % 4)
 * i.e. inside every 32-row block the four-column quads are stored one after the other, 32 rows x 16 bytes each.  The
 * transposed-product chain kernel (tile_rows = 66: accumulator lane = point, registers = quads of features) then moves
 * 1 KB of CONTIGUOUS memory per wave instruction for its stored-state operands and outputs, and a [32 x 128] operand
 * tile of the weight-gradient GEMM is 16 KB contiguous.  Only that kernel and nudf_gemm_tn_grouped read the layout. */
#define NUDF_CH_BLK_X1 1
#define NUDF_CH_BLK_X2 2
#define NUDF_CH_BLK_C1 4
#define NUDF_CH_BLK_C2 8
#define NUDF_CH_BLK_PE 16           /* pe_dst */
/* config-5 mode: this step's stored-state arrays -- X1, X2, C1, the TANGENT mirror C2 and pe_dst -- hold bf16 (ld in
 * elements); SOFTPLUS / MULSP / TANGENT / BWD steps of the workgroup-shared kernels only.  Values are rounded to
 * nearest even when stored; everything on chip stays fp32.  The arrays are 4-POINT PACKED: element (row, col) of a
 * [R, ld] buffer (R a multiple of 64) sits at ((row / 4) ld + col) 4 + row % 4 -- the four consecutive points a lane of a
 * 32 x 32 accumulator tile holds for its column are one 8-byte access, and a dword is the row pair the 16-bit
 * weight-gradient GEMM contracts (NUDF_TN_A_P4 / NUDF_TN_B_P4).  The same holds for a bf16 A0 / G0 of the SEED
 * initialisation (init_state16 bits 1 / 2). */
#define NUDF_CH_STATE16 32
/* the ReLU family of the 16-bit mode (colour net): per array, because its steps also touch fp32 interface buffers (the
 * view-branch input, d VIN).  RELU: C1; MULMASK / ADDMASK: X1 and / or C1 -- bf16, 4-point packed as above. */
#define NUDF_CH_P4_X1 64
#define NUDF_CH_P4_C1 128
typedef struct NudfChain {
  int32_t P, n_steps;
  int32_t init;                    /* NUDF_CH_INIT_*                                                 */
  int32_t k0;                      /* initial tile width (multiple of 4, <= 288)                     */
  int32_t x_div;                   /* x row of point p is p / x_div (samples per ray for per-ray directions; >= 1) */
  int32_t tile_rows;               /* 0 = choose; 32 / 64 points per workgroup (shared tile); 66 = 64-point shared tile,
                                      transposed product (16-byte epilogue accesses); 128 = prefer the wave-private
                                      kernel (4 waves x 32 points); 130 = two 64-point tiles per workgroup run in
                                      anti-phase (mlp_chain_pair_kernel: opt-in, a measured counter-example -- NUDF_CHAIN_PAIR);
                                      66 / 128 / 130 need fp32 steps and 16-byte aligned rows and fall back
                                      to 64 otherwise */
  int32_t lda0, ldg0;
  int32_t pe_L, pe_jvp;            /* positional encoding: frequencies, 1 = JVP with tangent v       */
  int32_t init_state16;            /* INIT_SEED: bit 0 = A0 holds bf16, bit 1 = G0 receives bf16, bit 2 = A0 is in the
                                      BLOCKED layout, bit 3 = G0 is (fp32, transposed-product kernel only)          */
  float pe_in_scale;
  float seed_scale, seed_xscale;
  const float* A0;                 /* INIT_LOAD source / INIT_SEED stored activation                 */
  float* G0;                       /* optional HBM copy of the initial tile [P, ldg0]                */
  const float* x;                  /* [P,3] points (positional encoding), or NULL                    */
  const float* v;                  /* [P,3] tangent (JVP), or NULL                                   */
  const float* seed_sign;          /* [P]                                                            */
  const float* seed_wrow;          /* [k0]                                                           */
  unsigned long long* dbg;         /* NULL, or [blocks*4 waves][64] timeline: hw_id, t0, per step (K loop end,
                                      after barrier 1, epilogue end, after barrier 2) in s_memtime ticks */
  float* absmax_out;               /* NULL, or ONE device float (zeroed by the caller; split modes of the workgroup-shared kernel):
                                      the launch raises it (atomic max) to the largest |value| it puts into the stored arrays a
                                      weight-gradient GEMM will read -- the initial tile (INIT_LOAD values; for the JVP encoding the
                                      bound 2^(L-1) |v| in_scale), every C1 / pe-free output of its MULSP / BWD / MULMASK / ADDMASK /
                                      NONE steps, and the rank-1 operand r1_row.  It is what scales that side of an f16x2 GEMM
                                      (NudfGemmTNGroup.amax_a / amax_b).                                                      */
  /* Per-tile scaling of a LINEAR sweep (split modes of the workgroup-shared kernel; what lets the backward sweeps -- whose
     operands are adjoints of the LOSS, 1e-5 ... 1e-12 -- contract as f16x2, prec 4).  The tangent, adjoint and ReLU-backward
     sweeps map each point's seeds linearly to that point's outputs, so a tile of points may run multiplied by any power of two:
     tile_scale = 1 makes the kernel take m = the largest |seed| of its tile -- the initial tile (INIT_LOAD values; for the JVP
     encoding the bound 2^(L-1) |v| in_scale), the rank-1 operands of its steps, tile_amax_in -- and sigma = 2^(-6 - floor(log2 m));
     the activation tile then holds sigma times the sweep's values (largest seed in [2^-6, 2^-5): 2^21 of growth below fp16's
     largest number; elements down to 2^-8 of the seed keep all 22 bits of the split, smaller ones 2^-30 of the seed absolutely),
     operands that enter from memory (r1_row, X2 of BWD / ADDMASK) are multiplied by sigma and everything that goes to memory (G0,
     C1, C2, pe_dst) by 1 / sigma -- powers of two, exact: memory holds what it held without the option.  Requires a
     linear chain: INIT_LOAD or the JVP encoding, steps NONE / MULSP / TANGENT / BWD / MULMASK / ADDMASK without bias.
     A tile of zero seeds runs with sigma = 1.
     tile_amax_in: NULL or [rows padded to 64 / 32] floats, per 32 points a bound of what ENTERS later in the sweep (X2), merged
     into m; tile_amax_out: NULL or the same shape, receives per 32 points the largest |value| this launch stored for them (what
     absmax_out reduces over the whole launch) -- the tangent sweep's is the adjoint sweep's tile_amax_in. */
  const float* tile_amax_in;
  float* tile_amax_out;
  int32_t tile_scale;
  int32_t reserved0;
  NudfChainStep step[NUDF_CH_MAX_STEPS];
} NudfChain;
int nudf_mlp_chain(const NudfChain* args, void* stream);
/* 16-bit mode (prec 1 / 2), 64-point tiles: chains whose steps ALL contract in the same 16-bit type run on the 16-bit-tile
 * kernel -- the LDS activation tile holds that type (the MFMA operand itself: one ds_read_b128 per 32-row tile and k step, no
 * conversion in the K loop, 37 KB per workgroup = three workgroups per CU), the epilogue rounds the new activations once on
 * their way into the tile.  Bit-identical to the fp32-tile kernel (same values rounded to the same type).  0 switches it off
 * (A/B, tests; env NUDF_CHAIN_T16); returns the old setting. */
int nudf_set_chain_t16(int on);
/* out[((g*NT + T)*64 + lane)*4 + j] = B[8g + 4(lane>>5) + j][32T + (lane&31)], zero outside K x N;
 * out holds roundup(K,16)/8 * roundup(N,32)/32 * 256 floats */
int nudf_pack_frag(const float* B, int ldb, int K, int N, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * Multi-layer weight packing in ONE launch (table by value): weight_norm (W = g v/|v|), the two GEMM layouts
 * and the MFMA-fragment copies the fused chains read (nudf_pack_frag layout), for up to 16 layers; and the
 * matching multi-layer backward of the packing.  Same arithmetic as nudf_weightnorm_pack / _unpack_grad
 * (torch.nn.utils.weight_norm at fields.py:175-176, 433-446).
 * ---------------------------------------------------------------------------------- */
#define NUDF_PACK_MAX_LAYERS 16
#define NUDF_PACK_MAX_FRAGS 4
typedef struct NudfPackFrag {
  float* dst;                      /* fragment-ordered [K, N] operand (zero-initialised by the caller once)     */
  int32_t transpose;               /* 1: B[k][n] = W[o0 + n][i0 + k] (W^T, forward sweeps); 0: B[k][n] = W[o0 + k][i0 + n] */
  int32_t o0, i0;                  /* offsets into the packed [out, in] matrix                                  */
  int32_t K, N;
  int32_t dtype;                   /* 0: fp32 fragments (nudf_pack_frag layout); 1 / 2: fp16 / bf16 fragments for
                                      v_mfma_f32_32x32x16_*: dst16[((g*NT + T)*64 + lane)*8 + j] =
                                      B[16g + 8(lane>>5) + j][32T + (lane&31)], round-to-nearest-even;
                                      3: bf16x3 split, three planes hi / mid / lo of that layout stored back to back per
                                      (g, T): dst16[(((g*NT + T)*3 + plane)*64 + lane)*8 + j]  (3x the 16-bit size);
                                      4: f16x2 split, two fp16 planes hi = fp16(w), lo = fp16((w - hi) 2^11):
                                      dst16[(((g*NT + T)*2 + plane)*64 + lane)*8 + j]  (2x the 16-bit size)            */
} NudfPackFrag;
typedef struct NudfPackLayer {
  const float* v; const float* g;  /* weight_v [out,in], weight_g [out] (NULL: plain Linear)                    */
  const int32_t* perm;             /* input-column permutation or NULL                                          */
  float* W; float* Wt;             /* [out_pad, ldw], [in_pad, ldwt] (either may be NULL)                       */
  float* inv_norm;                 /* [out] 1/|v_row| (kept for the backward) or NULL                           */
  int32_t out, in, ldw, ldwt;
  int32_t nfrag, row_start;        /* row_start: prefix sum of `out` over the preceding layers                  */
  NudfPackFrag frag[NUDF_PACK_MAX_FRAGS];
} NudfPackLayer;
typedef struct NudfPackMulti {
  int32_t n_layers, total_rows;
  NudfPackLayer layer[NUDF_PACK_MAX_LAYERS];
} NudfPackMulti;
int nudf_weightnorm_pack_multi(const NudfPackMulti* args, void* stream);

typedef struct NudfUnpackLayer {
  const float* dW;                 /* [out_pad, ldw] packed weight gradient                                     */
  const float* v; const float* g; const float* inv_norm;
  const int32_t* perm;
  float* dv; float* dg;            /* [out,in], [out] (dg NULL for a plain Linear)                              */
  const float* db_in; float* db_out; /* optional [out] copy of the bias gradient out of the flat packed buffer  */
  int32_t out, in, ldw, row_start;
} NudfUnpackLayer;
typedef struct NudfUnpackMulti {
  int32_t n_layers, total_rows;
  NudfUnpackLayer layer[NUDF_PACK_MAX_LAYERS];
} NudfUnpackMulti;
int nudf_weightnorm_unpack_grad_multi(const NudfUnpackMulti* args, void* stream);

/* ------------------------------------------------------------------------------------
 * The three learnable scalars of the renderer in one launch (instead of ~15 one-element torch kernels):
 *   inv_s = exp(10 variance).clip(1e-6, 1e6)                      fields.py:654-655, udf_renderer_blending.py:373
 *   beta  = exp(10 beta).clip(0, beta_hi).clip(1e-6, 1e6)          fields.py:674-675, :376   (beta_hi = 1/beta_min)
 *   gamma = exp(10 gamma).clip(1e-6, 1e6)                          fields.py:677-678, :377
 * scal[3] = {inv_s, beta, gamma}; recip[2] = {1/inv_s, 1/beta} (the 'variance' / 'beta' entries of render()).
 * bwd: d_param[3] = d_scal * d scal / d param (torch.clip passes the gradient on the closed interval).
 * ---------------------------------------------------------------------------------- */
int nudf_scalars_fwd(const float* variance, const float* beta, const float* gamma, float beta_hi, float* scal,
                     float* recip, void* stream);
int nudf_scalars_bwd(const float* variance, const float* beta, const float* gamma, float beta_hi, const float* d_scal,
                     float* d_param, void* stream);

/* sum_i |pred_i - gt_i| (the numerator of ColorPixelLoss, loss/loss.py:37-43) and its backward
 * d_pred_i = d_out * sign(pred_i - gt_i).  out[0] = sum (one workgroup: these are [N,3] tensors). */
int nudf_l1_sum_fwd(const float* pred, const float* gt, int n, float* out, void* stream);
int nudf_l1_sum_bwd(const float* pred, const float* gt, int n, const float* d_out, float* d_pred, void* stream);

/* batch-global regularisers from the composite kernel's partial sums (udf_renderer_blending.py:531-536, 553):
 *   err[0] = sums[0] / (sums[1] + 1e-5), err[1] = sums[2] / (sums[3] + 1e-5), err[2] = sums[4] / n_rays
 * bwd: d_sums[5] (entries 1 and 3, the mask counts, included for completeness). */
int nudf_sums_errors_fwd(const float* sums, float n_rays, float* err, void* stream);
int nudf_sums_errors_bwd(const float* sums, float n_rays, const float* d_err, float* d_sums, void* stream);

/* Loss weights in device memory.  The runner's schedules move the colour weights and the regulariser weights from
 * iteration to iteration (adjust_color_loss_weights, exp_runner_blending.py:230-251; regularization_weights_schedule,
 * :199-211), and a step captured in a HIP graph replays the kernel ARGUMENTS of its capture: every loss entry point
 * below therefore takes `w_dev`, NULL or a device vector of NUDF_LW_COUNT floats laid out by the NUDF_LW_* indices,
 * whose entries override the by-value weights of the same call. */
enum { NUDF_LW_COLOR_BASE = 0, NUDF_LW_COLOR = 1, NUDF_LW_COLOR_PIXEL = 2, NUDF_LW_COLOR_PATCH = 3, NUDF_LW_IGR = 4,
       NUDF_LW_IGR_NS = 5, NUDF_LW_SPARSE = 6, NUDF_LW_MASK = 7,
       NUDF_LW_COLOR_SUM = 8,      /* color_base + color + color_pixel formed by the host in double (the host-side mirror's
                                      torch expressions divide by it; the kernels form the sum themselves in fp32) */
       NUDF_LW_COUNT = 16 };

/* ColorLoss in one launch when only the two L1 terms are active (loss/loss.py:105-133 with color_pixel = None,
 * patch_colors = None):  den = mask ? sum(mask) + 1e-4 : n ;  Lb = sum|cb - gt| / den ;  Lc = sum|c - gt| / den ;
 * out[3] = {(Lb w_b + Lc w_c) / (w_b + w_c + w_px), Lb, Lc}.
 * bwd: d_cb / d_c from the upstream gradients d_out[3] (any of which may be zero). */
int nudf_color_loss_fwd(const float* cb, const float* c, const float* gt, int n, const float* mask, int n_mask,
                        float w_b, float w_c, float w_px, const float* w_dev, float* out, float* den_out, void* stream);
int nudf_color_loss_bwd(const float* cb, const float* c, const float* gt, int n, const float* den, float w_b, float w_c,
                        float w_px, const float* w_dev, const float* d_out, float* d_cb, float* d_c, void* stream);
/* the same loss split at the ray-sharding exchange step (one process per GPU): local sums[3] = {sum|cb - gt|,
 * sum|c - gt|, sum(mask) or n} -> the caller all-reduces them (RCCL) -> finish forms out[3] / den exactly as above;
 * nudf_color_loss_bwd then runs on the local rays with the GLOBAL den. */
int nudf_color_loss_sums(const float* cb, const float* c, const float* gt, int n, const float* mask, int n_mask,
                         float* sums, void* stream);
int nudf_color_loss_finish(const float* sums, int has_mask, float w_b, float w_c, float w_px, const float* w_dev,
                           float* out, float* den_out, void* stream);

/* The whole loss assembly of a train step when only the two L1 colour terms and the three regularisers are active
 * (exp_runner_blending.py:330-371; loss/loss.py:105-133; udf_renderer_blending.py:531-536, 553): nudf_color_loss_fwd +
 * nudf_sums_errors_fwd + total = ((cl + gens w_igr_ns) + sparse w_sparse) + ge w_igr in one launch, each product / sum
 * of the last line rounded on its own like the runner's scalar torch ops.  out[8] = {total, cl, Lb, Lc, ge, gens,
 * sparse, 0}; den_out[1].  bwd: upstream d_total (device scalar, NULL = 1) and optionally d_extra[8] for the other
 * outputs (indexed like out[], NULL = none) -> d_cb / d_c [n] and d_sums[5].
 * sums_ws != NULL: the composite kernel left its per-block partial sums [sums_nblk, 5] there (NudfComposite.defer_sums);
 * this launch reduces them first (nudf_partial_sums' order) and WRITES the five sums to `sums` as well. */
int nudf_step_loss_fwd(const float* cb, const float* c, const float* gt, int n, const float* mask, int n_mask,
                       float* sums, float n_rays, float w_b, float w_c, float w_px, float w_igr, float w_igr_ns,
                       float w_sparse, const float* w_dev, float* out, float* den_out, const float* sums_ws,
                       int sums_nblk, void* stream);
int nudf_step_loss_bwd(const float* cb, const float* c, const float* gt, int n, const float* den, const float* sums,
                       float n_rays, float w_b, float w_c, float w_px, float w_igr, float w_igr_ns, float w_sparse,
                       const float* w_dev, const float* d_total, const float* d_extra, float* d_cb, float* d_c, float* d_sums,
                       void* stream);
/* out4 [P_pad, 4] (16-byte aligned): column 0 = sign[r] * d[r] * scale (d NULL: sign[r] * scale) for r < P, everything
 * else zero: the 4-wide column-0 operand of the UDF head's adjoint and second-order weight gradient (fields.py:184-231) */
/* out4_sign (NULL or like out4): column 0 = sign[r] * scale, written by the same launch */
int nudf_col0_seed4(const float* sign, const float* d, float scale, int P, int P_pad, float* out4, float* out4_sign,
                    void* stream);

/* ------------------------------------------------------------------------------------
 * GPU-resident ray / patch batch generation: Dataset.gen_random_rays_patches_at (dataset/dataset.py:228-294) and
 * Dataset.near_far_from_sphere (:329-335) in one call.  The pixel coordinates are drawn by the caller
 * (torch.randint keeps the reference's RNG semantics).
 * ---------------------------------------------------------------------------------- */
typedef struct NudfRayBatch {
  const float* image;              /* [H, W, 3] colours of the reference view (images[img_idx])                 */
  const float* mask;               /* [H, W, 3]                                                                  */
  const float* intrinsics_inv;     /* [4, 4] row-major                                                           */
  const float* pose;               /* [4, 4] row-major camera-to-world                                           */
  const int64_t* pixels_x; const int64_t* pixels_y;   /* [N]                                                     */
  int32_t N, H, W, h_patch_size;
  float* rays;                     /* [N, 10] = rays_o 3 | rays_v 3 | colour 3 | mask 1                          */
  float* ndc_uv;                   /* [N, 2] or NULL                                                             */
  float* xyz_cam;                  /* [N, 3] K^-1 (x, y, 1) or NULL                                              */
  float* near; float* far;         /* [N] each, or both NULL                                                     */
  float* patch_color;              /* [N, (2h+1)^2, 3] or NULL (crop_patch = False)                              */
  uint8_t* patch_mask;             /* [N] or NULL                                                                */
} NudfRayBatch;
int nudf_gen_ray_batch(const NudfRayBatch* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NUDF_H */
