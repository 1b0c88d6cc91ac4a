/* libmiseg_hip.so -- C ABI of the MI355X (gfx950) hot path for MI-Seg's 3D cross-modality nets.
 *
 * Nothing like this exists in the reference (it is pure Python on torch/cuDNN/cuBLAS);
 * each entry point below names the reference code whose device arithmetic it replaces, as
 * /root/reference-relative file:line.  The Python host in mi-seg_amd/ binds these with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain C symbols, POD structs, raw device pointers; no C++/torch types cross the boundary.
 *   - every call only ENQUEUES work on `stream` and returns; no implicit device synchronisation, no
 *     allocation: all buffers (incl. workspaces, sized by the *_workspace_bytes helpers) are the caller's.
 *   - return 0 on success, a negative MISEG_E_* otherwise; miseg_last_error() gives a thread-local message.
 *   - activations are channels-last rows: element (row r, channel c) lives at base[r * ld + c]; a 3D volume
 *     is rows in (b, d, h, w) row-major order.  dtype is MISEG_F32 or MISEG_BF16 (bf16 storage, fp32 math).
 *   - parameters and their gradients are always fp32.
 *   - the library keeps no state of the caller's and reads NO environment variable (round 5: the tuning switches of earlier rounds'
 *     sweeps are gone); per-process caches are limited to device attributes and kernel attributes set once per device.
 */
#ifndef MISEG_HIP_H
#define MISEG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* miseg_stream_t; /* hipStream_t */

enum { MISEG_F32 = 0, MISEG_BF16 = 1 };
enum { MISEG_OK = 0, MISEG_E_BADARG = -1, MISEG_E_UNSUPPORTED = -2, MISEG_E_LAUNCH = -3 };
enum { MISEG_ACT_NONE = 0, MISEG_ACT_LEAKY = 1, MISEG_ACT_GELU = 2, MISEG_ACT_PRELU = 3 };
#define MISEG_MAX_STYLES 4

/* bumped on EVERY change of a struct layout or prototype; bindings must refuse a library whose version differs from the header they mirror */
#define MISEG_ABI_VERSION 9
int miseg_abi_version(void);
const char* miseg_last_error(void);
/* writes e.g. "gfx950" for the code objects embedded in the library */
int miseg_device_arch(char* buf, size_t n);

/* ------------------------------------------------------------------------------------------------
 * (Conditional) instance norm over channels-last rows.
 * Replaces networks/norms/conditional_instance_norm.py:59-68 (per-sample style lookup + stack),
 * nn.InstanceNorm{1,3}d in networks/blocks/dynunet_block.py:80-81,98 and F.instance_norm in
 * networks/nets/swin_transformer.py:135-136, plus the LeakyReLU / residual add that follow them in
 * dynunet_block.py:100-126.
 *   x: [B][S][C] rows (ldx), statistics per (b, c) over the S rows, biased variance, eps inside sqrt.
 *   styles: device int32[B] or NULL (=> row 0); gamma/beta[s]: fp32[C] per style or NULL (no affine).
 *   y = act((x - mean) * rstd * gamma[s] + beta[s] + res)
 * ---------------------------------------------------------------------------------------------- */
/* statistics are kept as fp64 (sum x, sum x^2) pairs in R replicas: stat is double[R][B][C][2] with R * B * C * 16 =
 * miseg_instnorm_stat_bytes(B, C) bytes; the CALLER zero-fills it (the host pools all statistics buffers of a step behind
 * one fill), miseg_instnorm_stats adds each workgroup's partial sums to one replica with fp64 atomics (one row of
 * addresses serialises ~900 workgroups of a 96^3 tensor); apply / backward add the replicas up and derive mean and
 * 1/sqrt(var+eps). */
size_t miseg_instnorm_stat_bytes(int B, int C);
typedef struct {
  const void* x; int64_t ldx;
  int B, S, C, dtype;
  void* stat;                      /* in/out: miseg_instnorm_stat_bytes(B, C) bytes, zero on entry */
} miseg_instnorm_stats_params;
int miseg_instnorm_stats(const miseg_instnorm_stats_params* p, miseg_stream_t stream);

typedef struct {
  const void* x; int64_t ldx;
  const void* res; int64_t ldres;  /* optional residual added before the activation */
  void* y; int64_t ldy;
  int B, S, C, dtype;
  const void* stat; float eps;
  const int32_t* styles; int num_styles;
  const float* gamma[MISEG_MAX_STYLES]; const float* beta[MISEG_MAX_STYLES];
  int act; float slope;            /* MISEG_ACT_NONE | MISEG_ACT_LEAKY */
  /* optional: `res` is the raw input of a second instance norm (the block's shortcut branch, dynunet_block.py:118-124) and is normalised
   * on the fly with its own statistics (res_stat, layout as stat) and affine rows: y = act(norm(x) + norm_res(res)).  miseg_instnorm_apply only. */
  const void* res_stat;
  const float* res_gamma[MISEG_MAX_STYLES]; const float* res_beta[MISEG_MAX_STYLES];
  /* optional, with res_stat and res == NULL: the shortcut branch is the 1x1x1 convolution of a ONE-channel image (the stem block,
   * dynunet_block.py:87-97) and is never stored: res[row][c] = round(r1x[row] * r1w[c]) in the compute dtype.  r1x: [B*S] rows of one
   * element (row stride ldr1x), r1w: [C] in the compute dtype; res_stat from miseg_rank1_stats.  miseg_instnorm_apply only. */
  const void* r1x; int64_t ldr1x; const void* r1w;
} miseg_instnorm_apply_params;
int miseg_instnorm_apply(const miseg_instnorm_apply_params* p, miseg_stream_t stream);
/* statistics (into the zero-filled p->stat) + apply in one call; tensors of <= 512 rows per sample take ONE fused launch */
int miseg_instnorm_fwd(const miseg_instnorm_apply_params* p, miseg_stream_t stream);
/* miseg_instnorm_fwd whose input is still the `nslabs` fp32 partial slabs [B*S][C] (slab_stride elements apart) of a split convolution
 * (miseg_conv3_params.defer_slabs): x = round(sum of the slabs) is formed in registers and WRITTEN to p->x (the convolution's output, which
 * the backward pass reads), statistics, normalisation, residual and activation follow in the same launch.  S <= miseg_instnorm_fused_max_rows()
 * (2048) only: one workgroup owns every row of its channels there (dynunet_block.py:100-126 at the 12^3-and-smaller stages). */
int miseg_instnorm_fwd_slabs(const miseg_instnorm_apply_params* p, const float* slabs, int nslabs, int64_t slab_stride, miseg_stream_t stream);
int miseg_instnorm_fused_max_rows(void);

/* backward of the fused op above.  dy is the gradient w.r.t. y; when act != NONE, y (the saved output)
 * supplies the sign for the activation gradient.  Outputs: dx, optionally dres (= gradient flowing to the
 * residual input), dgamma/dbeta[s] (ACCUMULATED with atomics: caller zero-fills once per step).
 * dstat: scratch double [B][C][2], zero on entry. */
typedef struct {
  const void* dy; int64_t lddy;
  const void* y; int64_t ldy;
  const void* x; int64_t ldx;
  void* dx; int64_t lddx;
  void* dres; int64_t lddres;
  int B, S, C, dtype;
  const void* stat; float eps; void* dstat;
  const int32_t* styles; int num_styles;
  const float* gamma[MISEG_MAX_STYLES];
  float* dgamma[MISEG_MAX_STYLES]; float* dbeta[MISEG_MAX_STYLES];
  int act; float slope;
  const void* gadd; int64_t ldgadd;   /* optional: dx += gadd (the gradient of a skip branch forked off x: the fan-out sum rides here) */
  /* y may be NULL with MISEG_ACT_LEAKY when NO residual entered the activation: the sign is then recomputed from x with the forward's
   * scale / shift (needs beta; saves one read of the tensor in each backward kernel and keeping y alive) */
  const float* beta[MISEG_MAX_STYLES];
} miseg_instnorm_bwd_params;
int miseg_instnorm_bwd(const miseg_instnorm_bwd_params* p, miseg_stream_t stream);
/* miseg_instnorm_bwd whose incoming gradient is still the partial slabs of a split data-gradient convolution (as miseg_instnorm_fwd_slabs):
 * p->dy is ignored (the gradient is summed and rounded in registers and never written).  S <= miseg_instnorm_fused_max_rows() only. */
int miseg_instnorm_bwd_slabs(const miseg_instnorm_bwd_params* p, const float* slabs, int nslabs, int64_t slab_stride, miseg_stream_t stream);
/* the apply half alone (ABI 8): p->dstat already HOLDS the backward sums (sum dy, sum dy * xhat) - the GEMM that produced dy added them in
 * its epilogue (miseg_gemm_params.stat_mode 2, miseg_mlp_params.bs_dstat).  No activation.  dx, dgamma / dbeta as miseg_instnorm_bwd. */
int miseg_instnorm_bwd_apply(const miseg_instnorm_bwd_params* p, miseg_stream_t stream);
/* the reduction half alone (ABI 5): dstat[r][b][c] += (sum dy, sum dy * xhat) over the rows of sample b, xhat from `stat` - what the group /
 * batch norms of the reference's factory (networks/layers/factories.py:219-257) need besides the kernels above (their means run over channel
 * groups / the whole batch: mi-seg_amd/hip/functional.py::group_norm).  Reads dy, x, stat, eps, styles / gamma / beta only with an activation. */
int miseg_instnorm_bwd_reduce(const miseg_instnorm_bwd_params* p, miseg_stream_t stream);

/* Backward of y = LeakyReLU(norm_a(xa) + norm_b(xb)) (miseg_instnorm_apply with res_stat; dynunet_block.py:118-126): one reduction and one
 * apply pass for both norms.  dstat_a / dstat_b: scratch like miseg_instnorm_bwd's dstat (zero on entry); dgamma / dbeta accumulate. */
typedef struct {
  const void* dy; int64_t lddy;
  const void* y; int64_t ldy;
  const void* xa; int64_t ldxa; const void* xb; int64_t ldxb;
  void* dxa; int64_t lddxa; void* dxb; int64_t lddxb;
  int B, S, C, dtype;
  const void* stat_a; const void* stat_b; float eps; void* dstat_a; void* dstat_b;
  const int32_t* styles; int num_styles;
  const float* gamma_a[MISEG_MAX_STYLES]; const float* gamma_b[MISEG_MAX_STYLES];
  float* dgamma_a[MISEG_MAX_STYLES]; float* dbeta_a[MISEG_MAX_STYLES];
  float* dgamma_b[MISEG_MAX_STYLES]; float* dbeta_b[MISEG_MAX_STYLES];
  float slope;
  /* y == NULL: the activation's sign is recomputed from xa / xb (then the betas of both norms are read; NULL rows = 0) */
  const float* beta_a[MISEG_MAX_STYLES]; const float* beta_b[MISEG_MAX_STYLES];
  /* rank-1 shortcut (the stem block, dynunet_block.py:87-97 with one input channel): xb[row][c] = round(r1x[row] * r1w[c]) is not stored
   * (xb = dxb = y = NULL); r1x: [B*S] rows of one element (row stride ldr1x), r1w: [C] in the compute dtype; the weight gradient of that
   * 1x1x1 convolution, sum over rows of dxb[row][c] * r1x[row], is ADDED to r1dw [C] fp32 */
  const void* r1x; int64_t ldr1x; const void* r1w; float* r1dw;
} miseg_instnorm_pair_bwd_params;
int miseg_instnorm_pair_bwd(const miseg_instnorm_pair_bwd_params* p, miseg_stream_t stream);

/* LayerNorm over the channel dim of channels-last rows (vit_norm_name="layer", the reference default:
 * networks/blocks/swin_transformer_block.py:104-105, transformer_block.py:82-83).  gamma/beta may be NULL. */
typedef struct {
  const void* x; int64_t ldx; void* y; int64_t ldy;
  int64_t rows; int C, dtype; float eps;
  const float* gamma; const float* beta;
  float* mean; float* rstd;        /* out fp32 [rows], saved for backward */
} miseg_layernorm_fwd_params;
int miseg_layernorm_fwd(const miseg_layernorm_fwd_params* p, miseg_stream_t stream);
typedef struct {
  const void* dy; int64_t lddy; const void* x; int64_t ldx; void* dx; int64_t lddx;
  int64_t rows; int C, dtype;
  const float* gamma; const float* mean; const float* rstd;
  float* dgamma; float* dbeta;     /* accumulated (atomics); may be NULL */
} miseg_layernorm_bwd_params;
int miseg_layernorm_bwd(const miseg_layernorm_bwd_params* p, miseg_stream_t stream);

/* a (conditional) instance norm whose apply pass is folded into the operand load of a consumer kernel (ABI 8; one sample): its statistics
 * and affine rows.  stat == NULL: no fold. */
typedef struct {
  const void* stat;                 /* fp64 [16][1][C][2], miseg_instnorm_stats layout: (sum x, sum x^2) of the norm's raw input */
  const int32_t* styles;            /* device [1] (the sample's style id) or NULL = style 0 */
  int32_t num_styles; float eps;
  const float* gamma[MISEG_MAX_STYLES]; const float* beta[MISEG_MAX_STYLES];      /* rows per style; NULL = no affine */
} miseg_norm_ref;

/* ------------------------------------------------------------------------------------------------
 * GEMM on the matrix cores:  C[M][N] = A * B (+ bias[N]) (-> act).
 * Replaces every nn.Linear on the path (window_attention.py:92,94 qkv/proj; MONAI MLPBlock used at
 * swin_transformer_block.py:97; patch_merging.py:48 reduction; transformer_block.py:58-59), the 1x1x1
 * convolutions (dynunet_block.py:87-97,278-289) and, with gather/scatter kernels, ConvTranspose3d k2 s2
 * (unetr_block.py:51-59).
 *   ta == 0: A is [M][K] (lda)   ta == 1: A is stored [K][M] (lda)   -- likewise tb for B: 0 => [N][K], 1 => [K][N]
 *   only (ta,tb) = (0,0) "NT" and (1,1) "TN" are implemented (weights are re-packed once per step).
 *   out_dtype: dtype of C (MISEG_F32 for weight gradients).  accumulate != 0: C += result (fp32 C only).
 *   split_k = 0 lets the library choose kernel and split (TN: tall token streams are reduced through per-workgroup
 *   partial sums in `workspace` and a second kernel, no atomics); split_k > 1 partitions K over workgroups and reduces with
 *   fp32 atomics into a zero-filled or accumulate-mode C (fp32 C only).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* A; int64_t lda; const void* B; int64_t ldb; void* C; int64_t ldc;
  int M, N, K;
  int ta, tb;
  int dtype, out_dtype;
  const float* bias; int act;
  int accumulate, split_k;
  void* workspace;   /* miseg_gemm_workspace_bytes(p) bytes of scratch (uninitialised) or NULL when that is 0 */
  /* NT epilogue extras (all [M][N] row views in out_dtype): C = act(z) + res with z = A*B + bias;
   * epi_mode 1: aux = z (the pre-activation a GELU backward needs), 2: z *= gelu'(aux) (the GELU backward folded into the
   * data-gradient GEMM of the layer behind it).  MLP: swin_transformer_block.py:97, residuals :241-252. */
  const void* res; int64_t ldres;
  void* aux; int64_t ldaux;
  int epi_mode;
  int defer_reduce;  /* TN streaming path only: leave the per-split partial tiles in `workspace`; the caller sums them later with
                      * miseg_gemm_tn_reduce_batch (miseg_gemm_tn_splits(p) tells how many there are) */
  void* stat;        /* NT, optional: fp64 [16][1][N][2] (miseg_instnorm_stat_bytes(1, N), zero on entry): instance-norm statistics of the rounded
                      * output when all M rows are ONE sample and miseg_gemm_fuses_stat(p) says so (the tall-skinny bf16 path, N <= 96) -
                      * the linears / 1x1x1 convs whose output feeds an instance norm (swin_transformer_block.py:241-252, dynunet_block.py:118-124) */
  /* NT, optional (scat_cout > 0): the GEMM of a ConvTranspose3d(k2, s2) (unetr_block.py:80-85) stores straight into the upsampled tensor.
   * Rows are the voxels of a [.., scat_d, scat_h, scat_w] grid, the N = 8 * scat_cout columns are (j, co) with j = 4 jd + 2 jh + jw; element
   * (voxel (d,h,w), j, co) goes to row (2d+jd, 2h+jh, 2w+jw) of the [.., 2 scat_d, 2 scat_h, 2 scat_w] grid, column co, of C (row stride ldc -
   * e.g. the left half of a concat buffer).  Only where miseg_gemm_fuses_scatter(p) says so. */
  int scat_d, scat_h, scat_w, scat_cout;
  /* NT, optional (ABI 8), only where miseg_gemm_fuses_anorm(p) says so: A is the RAW input of a (conditional) instance norm over its M rows
   * (ONE sample) and is normalised as it is loaded - C = act(norm(A) B + bias) + res without the norm's apply pass (the Swin block's
   * norm1 -> qkv and norm2 -> fc1, swin_transformer_block.py:103,176-205).  an.stat: the norm's statistics (miseg_instnorm_stats layout, complete
   * when this launch starts).  an_out (optional): norm(A) [M][K] is stored as well, once, rounded exactly as miseg_instnorm_apply rounds
   * it - the operand of the weight-gradient product of the backward pass. */
  miseg_norm_ref an; void* an_out; int64_t ld_an_out;
  /* NT, optional (ABI 8), only where miseg_gemm_fuses_bstat(p) says so: stat_mode 2 - C [M][N] is the gradient with respect to the OUTPUT of an
   * instance norm (one sample = the M rows) whose raw input is bs_x [M][N] with forward statistics bs_stat: `stat` receives that norm's
   * backward sums (sum C, sum C * xhat) in the dstat layout of miseg_instnorm_bwd (zero on entry), from the rounded C - miseg_instnorm_bwd_apply
   * then needs no reduction launch.  stat_mode 0: `stat` = forward statistics as above. */
  int stat_mode;
  const void* bs_x; int64_t ld_bs_x; const void* bs_stat; float bs_eps;
  /* TN, optional (ABI 9), only where miseg_gemm_tn_fuses_colsum(p) says so (the streaming path): tn_colsum[m] += sum over the K rows of A[k][m]
   * (fp32 [M], atomics: zero or accumulating on entry) - with A = dy the bias gradient of the linear layer whose weight gradient dW = dy^T x
   * this product is (swin_transformer_block.py:97,103), from the A fragments the product already holds. */
  float* tn_colsum;
} miseg_gemm_params;
int miseg_gemm_fuses_stat(const miseg_gemm_params* p);  /* 1: miseg_gemm(p) with p->stat set is supported for this problem */
int miseg_gemm_fuses_anorm(const miseg_gemm_params* p); /* 1: ... with p->an.stat set (whatever an_out) */
int miseg_gemm_fuses_bstat(const miseg_gemm_params* p); /* 1: ... with p->stat, stat_mode 2 and the bs_* fields set */
/* instance-norm statistics (layout of miseg_instnorm_stats, one sample = all M rows, zero on entry) of the rank-1 product
 * round(x[m] * w[n]) WITHOUT storing it: the stem block's shortcut convolution (dynunet_block.py:87-97), consumed through the r1x / r1w
 * fields of miseg_instnorm_apply / miseg_instnorm_pair_bwd.  x: [M] rows of one element (stride ldx), w: [N] (stride ldw), N <= 128 */
int miseg_rank1_stats(const void* x, int64_t ldx, const void* w, int64_t ldw, int M, int N, int dtype, void* stat, miseg_stream_t stream);

/* Fused MLP of the high-resolution Swin blocks (ABI 3): y = W2 gelu(W1 x + b1) + b2 (+ res)  (MONAI MLPBlock as used at
 * networks/blocks/swin_transformer_block.py:97,176-205), one launch, the hidden activations never reach memory: the accumulator tile of the
 * first product is the operand of the second (v_mfma_f32_16x16x16_bf16).  Backward (miseg_mlp_bwd): recomputes the hidden pre-activation
 * from x, writes dz = (dy W2) * gelu'(z) and h = gelu(z) for the two weight-gradient products (miseg_gemm TN) and dx = dz W1.
 * bf16, C = 48, HID = 192, M >= 4096 (miseg_mlp_fused tells); weights in the compute dtype: w1 [HID][C], w2 [C][HID], w2t [HID][C] = w2
 * transposed, w1t [C][HID]; biases fp32 (may be NULL); stat: as miseg_gemm_params::stat (one sample, statistics of the rounded y). */
typedef struct {
  uint32_t struct_size;
  int32_t M, C, HID, dtype;
  const void* x; int64_t ldx;
  const void* w1; const float* b1; const void* w2; const float* b2;
  const void* res; int64_t ldres;          /* forward, optional */
  void* y; int64_t ldy; void* stat;        /* forward */
  const void* dy; int64_t lddy;            /* backward */
  const void* w2t; const void* w1t;
  void* dz; int64_t lddz; void* h; int64_t ldh; void* dx; int64_t lddx;
  /* ABI 8, optional.  Forward: an.stat != NULL - x is the RAW input of the (conditional) instance norm in front of the MLP (one sample; the
   * Swin block's norm2) and is normalised as it is loaded; an_out (optional) receives norm(x) [M][C] for the backward pass (which takes it
   * as its `x`).  Backward: bs_dstat != NULL - dx is the gradient with respect to that norm's OUTPUT; the norm's backward sums (sum dx,
   * sum dx * xhat; xhat from the raw input bs_x and the forward statistics bs_stat) are added to bs_dstat (dstat layout of miseg_instnorm_bwd,
   * zero on entry): miseg_instnorm_bwd_apply then needs no reduction launch. */
  miseg_norm_ref an; void* an_out; int64_t ld_an_out;
  const void* bs_x; int64_t ld_bs_x; const void* bs_stat; float bs_eps; void* bs_dstat;
} miseg_mlp_params;
int miseg_mlp_fused(int M, int C, int HID, int dtype);      /* 1: the two calls below support this problem */
int miseg_mlp_fwd(const miseg_mlp_params* p, miseg_stream_t stream);
int miseg_mlp_bwd(const miseg_mlp_params* p, miseg_stream_t stream);
int miseg_gemm_fuses_scatter(const miseg_gemm_params* p);  /* 1: miseg_gemm(p) with the scat_* fields set is supported for this problem */
int miseg_gemm_tn_splits(const miseg_gemm_params* p);   /* > 1: partial tiles [splits][M][N] fp32 in the workspace */
int miseg_gemm_tn_fuses_colsum(const miseg_gemm_params* p);   /* 1: this TN product takes the streaming path and can carry tn_colsum */
size_t miseg_gemm_workspace_bytes(const miseg_gemm_params* p);
int miseg_gemm(const miseg_gemm_params* p, miseg_stream_t stream);

/* up to MISEG_GEMM_GROUP independent TN problems C[M][N] += A[K][M]^T B[K][N] (fp32 C, accumulate mode) in ONE launch: the weight
 * gradients of the deep stages are a few dozen workgroups each; the host queues them during the backward pass and issues them
 * together.  `descs` is a HOST array (copied into the kernel arguments). */
/* regroup > 0 (ABI 8): column n = j * regroup + c of the summed product is stored at column c * (N / regroup) + j (see miseg_gemm_tn_desc) */
typedef struct { const float* partial; float* C; int64_t ldc; int32_t M, N, splits, block0, regroup, pad_; } miseg_tn_reduce_desc;
#define MISEG_TN_REDUCE_BATCH 32
/* C[m][n] += sum over splits of partial[s][m][n] for up to MISEG_TN_REDUCE_BATCH deferred reductions in one launch (HOST descriptors) */
int miseg_gemm_tn_reduce_batch(const miseg_tn_reduce_desc* descs_host, int n, miseg_stream_t stream);
#define MISEG_GEMM_GROUP 24
/* zeroed (ABI 5): 1 = C is known to hold zeros (a gradient slot no kernel has written since the step's fill) and no other problem of the
 * launch writes it: a problem whose reduction is not split then STORES its tiles instead of reading C back (the 85 M fp32 weight gradients
 * of C-UNETR's ViT: 340 MB less traffic per step). */
/* regroup > 0 (ABI 8): column n = j * regroup + c of the product is stored at column c * (N / regroup) + j - the weight gradient of a
 * ConvTranspose3d(k2, s2) as x^T dy8 (dy8's columns in (j, co) order, regroup = Cout) lands in the torch layout [Cin][Cout][2][2][2]
 * (unetr_block.py:51-59) without a permute pass */
typedef struct { const void* A; int64_t lda; const void* B; int64_t ldb; float* C; int64_t ldc; int32_t M, N, K, zeroed, regroup, pad_; } miseg_gemm_tn_desc;
int miseg_gemm_tn_group(const miseg_gemm_tn_desc* descs_host, int n, int dtype, miseg_stream_t stream);

/* fp32 re-layout: dst[i0][i1][i2] (+)= src[i0*s0 + i1*s1 + i2*s2]  (weight-gradient unpacking) */
int miseg_permute3(const float* src, float* dst, int n0, int n1, int n2, int64_t s0, int64_t s1, int64_t s2, int accumulate,
                   miseg_stream_t stream);

/* column sums: out[c] (+)= sum_r x[r][c]  (bias gradients) */
typedef struct { const void* x; int64_t ldx; int64_t rows; int C, dtype; float* out; int accumulate; } miseg_colsum_params;
int miseg_colsum(const miseg_colsum_params* p, miseg_stream_t stream);
/* up to MISEG_COLSUM_BATCH accumulate-mode column sums (all in `dtype`) in ONE launch: the bias gradients of a backward pass are
 * small launch-bound reductions, queued by the host and issued together; `descs` is a HOST array (copied into the kernel arguments). */
#define MISEG_COLSUM_BATCH 32
typedef struct { const void* x; int64_t ldx; int64_t rows; float* out; int32_t C, block0; } miseg_colsum_desc;
int miseg_colsum_batch(const miseg_colsum_desc* descs_host, int n, int dtype, miseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * 3x3x3 convolution, stride 1, zero padding 1, no bias, as an implicit GEMM on the matrix cores.
 * Replaces nn.Conv3d built by get_conv_layer (networks/blocks/dynunet_block.py:295-326) as used in
 * UnetResBlock/UnetBasicBlock (dynunet_block.py:55-76,100-126).
 *   x: [B][D][H][W][Cin] rows (ldx)   y: [B][D][H][W][Cout] rows (ldy)
 *   wpk: packed weights [Cout][27][Cin] in `dtype` (miseg_pack_conv3_weight); the data gradient is the
 *   same kernel run on dy with the flipped/transposed pack.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x; int64_t ldx; void* y; int64_t ldy; const void* wpk;
  int B, D, H, W, Cin, Cout, dtype;
  void* workspace;                 /* miseg_conv3_fwd_workspace_bytes (0 bytes for most shapes: may be NULL then) */
  /* optional epilogue pieces, only on the 96-byte-chunk path (channel rows of `dtype` a multiple of 96 bytes; else MISEG_E_UNSUPPORTED):
   * res (ldres): added to the result before rounding (the other gradient of a forked input when this is the data-gradient pass);
   * stat: fp64 [16][B][Cout][2] (miseg_instnorm_stat_bytes, zero on entry) - per-channel sum / sum of squares of the ROUNDED
   *       output, i.e. what miseg_instnorm_stats would compute from y (dynunet_block.py:105-107: every conv feeds a norm);
   *       with a workspace (split reduction) the second launch, which sums the partial slabs into y, computes them. */
  const void* res; int64_t ldres;
  void* stat;
  /* ABI 4: 1 = background launch (96-byte-chunk path): one workgroup per CU instead of two, so that the kernels of another stream find
   * registers, LDS and wave slots on every CU while this one runs (a branch of the step running beside its latency-bound chain) */
  int32_t background;
  /* ABI 5: 1 = when the launch splits its reduction (miseg_conv3_fwd_splits > 1) it stops after the partial slabs in `workspace`
   * ([splits][B*D*H*W][Cout] fp32): the caller's next launch sums them (miseg_instnorm_fwd_slabs) - y and stat are NOT written.  No `res`. */
  int32_t defer_slabs;
  /* ABI 9, only where miseg_conv3_fuses_shortcut(...) says so (bf16, 96-byte chunks, unsplit launch): y += sc_x * sc_w^T, a 1x1x1 term -
   * the data-gradient pass of a residual block's first convolution takes the gradient of the block's 1x1x1 shortcut convolution
   * along (dynunet_block.py:87-97,100-126: dx = dgrad3x3(g1) + g3 W3) instead of reading a [voxels][Cout] tensor that a GEMM wrote.
   * sc_x: [B][D][H][W][sc_C] rows (ld_sc_x), sc_C a multiple of 48; sc_w: [Cout][sc_C] in `dtype` (W3 transposed), contiguous. */
  const void* sc_x; int64_t ld_sc_x; const void* sc_w; int32_t sc_C;
  /* ABI 9, only where miseg_conv3_fuses_s2c(...) says so (96-byte chunks, unsplit launch, even D / H / W): output channels [0, s2c_C) are NOT
   * written to y but to s2c_out [B][D/2][H/2][W/2][8][s2c_C] - voxel (d, h, w) at block j = 4 (d&1) + 2 (h&1) + (w&1) of its coarse voxel: the
   * layout in which the ConvTranspose3d(k2, s2) in front of a decoder block reads the gradient of its output (unetr_block.py:80-85; the
   * data-gradient pass of that block's first convolution produces it, left half of the concat buffer).  Channels >= s2c_C go to y as usual. */
  void* s2c_out; int32_t s2c_C;
  /* ABI 9, only where miseg_conv3_fuses_fwd_shortcut(...) says so: fs_y = x * fs_w^T as a SECOND output of the launch - the 1x1x1 shortcut
   * convolution of a residual block beside its first 3x3x3 convolution (dynunet_block.py:87-97: both read the block's input).  fs_w: [Cout][Cin]
   * in `dtype`, contiguous; fs_y: [B][D][H][W][Cout] rows (ld_fs_y); fs_stat (optional): instance-norm statistics of fs_y, as `stat` for y. */
  const void* fs_w; void* fs_y; int64_t ld_fs_y; void* fs_stat;
} miseg_conv3_params;
/* small grids split the reduction over workgroups and need an fp32 staging buffer of the output */
size_t miseg_conv3_fwd_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout, int dtype);
/* number of partial slabs the launch leaves in `workspace` (1 = no split: y is written directly) */
int miseg_conv3_fwd_splits(int B, int D, int H, int W, int Cin, int Cout, int dtype);
/* 1 when miseg_conv3_fwd serves these shapes with its tiny-volume weight-streaming kernel (ABI 9: 3^3 / 6^3 voxels, hundreds of channels, bf16, aligned
 * operands) - information for a host that names / times launches; the call and its results are the same either way */
int miseg_conv3_fwd_tiny(int B, int D, int H, int W, int Cin, int Cout, int dtype);
/* 1 when miseg_conv3_fwd with these shapes can take a 1x1x1 shortcut term of sc_C channels along (miseg_conv3_params.sc_x) */
int miseg_conv3_fuses_shortcut(int B, int D, int H, int W, int Cin, int Cout, int sc_C, int dtype);
/* 1 when miseg_conv3_fwd with these shapes can produce a 1x1x1 convolution of its input as a second output (miseg_conv3_params.fs_w) */
int miseg_conv3_fuses_fwd_shortcut(int B, int D, int H, int W, int Cin, int Cout, int dtype);
/* 1 when miseg_conv3_fwd with these shapes can store its first s2c_C output channels in space-to-channel order (miseg_conv3_params.s2c_out) */
int miseg_conv3_fuses_s2c(int B, int D, int H, int W, int Cin, int Cout, int s2c_C, int dtype);
int miseg_conv3_fwd(const miseg_conv3_params* p, miseg_stream_t stream);

/* w: fp32 torch layout [Cout][Cin][3][3][3].  fwd_pack feeds miseg_conv3_fwd on x, bwd_pack (taps mirrored, channels swapped) feeds it
 * on dy; either may be NULL.  The layout is internal (row-major [Cout][27][CinP] or planar [27][CinP/k][CoutP16][k], by channel count);
 * buffers hold miseg_pack_conv3_elems(Cin, Cout, dtype, which) elements of `dtype` (which: 0 = fwd, 1 = bwd). */
size_t miseg_pack_conv3_elems(int Cin, int Cout, int dtype, int which);
/* K extent (elements; a whole number of 96-byte chunks) of the fast path for C channels on the K side of miseg_conv3_fwd, or 0 where the
 * generic row-major kernel runs.  Rows of >= 64 bytes that are neither a multiple of 96 bytes nor of a narrow chunk (64 / 32 bytes, bf16)
 * bytes are padded with zero weights to the next chunk (32 -> 48, 64 -> 96, 128 -> 144, 256 -> 288 bf16 channels; C-UNETR: 146.2 / 155.2 / 161.0 / 165.7 / 163.4 patches/s at 0 / 256 / 128 / 64 / 32).  Fused residual / statistics
 * (miseg_conv3_params.res / .stat) need a non-zero value. */
int miseg_conv3_k96(int C, int dtype);
/* 16x16 tiles miseg_pack_conv3_batch walks for one layer (the padded K groups of a fast-path pack get tiles of their own) */
int miseg_pack_conv3_tiles(int Cin, int Cout, int dtype);
typedef struct { const float* w; void* fwd_pack; void* bwd_pack; int Cin, Cout, dtype; } miseg_pack_conv3_params;
int miseg_pack_conv3_weight(const miseg_pack_conv3_params* p, miseg_stream_t stream);
/* every 3x3x3 weight of a model in ONE launch: `descs` is a DEVICE array of n descriptors sorted by tile0 = number of 16x16
 * (co, ci) tiles before the descriptor; total_tiles = the sum over descriptors of miseg_pack_conv3_tiles(Cin, Cout, dtype). */
typedef struct { const float* w; void* fwd_pack; void* bwd_pack; int32_t Cin, Cout, tile0, pad_; } miseg_pack_conv3_desc;
/* Versioned refresh (ABI 5; also miseg_param_cast_batch): `params_version` is a DEVICE int64 that counts the changes of the fp32 parameters
 * (miseg_opt_step bumps it through its `params_version` field; after any other change of a parameter the host adds to it with
 * miseg_counter_add), `state` a DEVICE int64[2] owned by this table of copies: state[0] = the version the copies were last made from,
 * state[1] = an arrival counter (zero between launches).  With both non-NULL the launch re-lays-out NOTHING when state[0] equals
 * *params_version (every workgroup reads the two words and exits) and otherwise records the new version when its last workgroup retires -
 * a replayed hipGraph thus refreshes the copies exactly in the steps that follow an optimiser step, with no host involvement.  NULL, NULL:
 * unconditional.  Set state[0] = -1 (miseg_fill32) after the table changed. */
int miseg_pack_conv3_batch(const miseg_pack_conv3_desc* descs_dev, int n, int total_tiles, int dtype, const int64_t* params_version, int64_t* state,
                           miseg_stream_t stream);

/* weight gradient: dw[Cout][Cin][27] (fp32, torch layout) (+)= sum_v dy[v][co] * x[v + tap][ci] */
typedef struct {
  const void* x; int64_t ldx; const void* dy; int64_t lddy; float* dw;
  int B, D, H, W, Cin, Cout, dtype;
  int accumulate;                  /* 0: dw = result; 1: dw += result; 2: dw is known to be zero on entry (no fill, no read-back) */
  void* workspace;                 /* miseg_conv3_wgrad_workspace_bytes */
  int32_t max_workgroups;          /* ABI 4: 0 = fill the chip (256); else a cap on the workgroups of the launch (a workgroup owns its CU's registers:
                                    * a background launch leaves the other CUs to the kernels of another stream).  miseg_conv3_wgrad only */
} miseg_conv3_wgrad_params;
size_t miseg_conv3_wgrad_workspace_bytes(int B, int D, int H, int W, int Cin, int Cout);
int miseg_conv3_wgrad(const miseg_conv3_wgrad_params* p, miseg_stream_t stream);
/* 1 when miseg_conv3_wgrad takes a layer of this shape with the tiny-volume kernel (ABI 6: bf16, 3^3 / 6^3 voxels, Cin % 16 == 0,
 * Cout % 48 == 0): a write-bound launch of its own that a host should not queue for miseg_conv3_wgrad_group */
int miseg_conv3_wgrad_tiny(int B, int D, int H, int W, int Cin, int Cout, int dtype);
/* Up to 24 layers of one dtype in one launch (+ one for the partial sums): the small-grid weight gradients of a backward pass
 * (dynunet_block.py:100-126 at 48^3 and below) fill the chip together instead of one after the other.  `workspace` of the
 * params is ignored; one shared buffer of miseg_conv3_wgrad_group_workspace_bytes is passed instead (may be NULL when that is 0). */
size_t miseg_conv3_wgrad_group_workspace_bytes(const miseg_conv3_wgrad_params* layers, int n);
int miseg_conv3_wgrad_group(const miseg_conv3_wgrad_params* layers, int n, void* workspace, miseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused 3D (shifted-)window attention core: zero pad -> cyclic roll -> window partition -> per-head
 * softmax(q*scale k^T + bias_table[index] + shift mask) v -> window reverse -> roll back -> crop, with the
 * scores never leaving the CU.  Replaces networks/blocks/window_attention.py:99-119 (everything between
 * the qkv and proj Linears) and the data movement of swin_transformer_block.py:116-169 +
 * networks/utils/swin_utils.py:15-143.
 *   qkv: [B][D][H][W][3*C] rows (ldq) holding qkv = Linear(norm1(x)) of the UNPADDED grid; rows of padded
 *        tokens are synthesised in-kernel as the qkv bias (a zero token through the Linear).
 *   out: [B][D][H][W][C] rows (ldo), head-major channels, ready for the proj Linear.
 *   window wd/wh/ww (already clamped), shift sd/sh/sw (0 for un-shifted blocks),
 *   rel-pos index always computed for a `tw`^3 table window (reference quirk: 7^3 index sliced [:n,:n]).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* qkv; int64_t ldq; void* out; int64_t ldo;
  const float* qkv_bias;           /* fp32 [3C] or NULL */
  const float* bias_table;         /* fp32 [(2tw-1)^3][heads] or NULL (global attention) */
  float* lse;                      /* out fp32 [B*nW][heads][n]: log-sum-exp per query row, saved for bwd */
  int B, D, H, W, C, heads, dtype;
  int wd, wh, ww, sd, sh, sw, tw;
  float scale;
  /* ABI 3: dropout on the attention probabilities (attn_drop of WindowAttention, swin_transformer_block.py:56-58,91; the SABlock of the
   * ViT): drop_p in [0, 1), 0 = off.  The mask is the counter-based one of miseg_dropout over the matrix [B*nW*heads*n rows][n columns]
   * (row = ((window * heads + head) * n + query), column = key) with the same (seed, stream_id, step_dev) key, so the backward call
   * re-creates it from the same three values; nothing is stored.  With drop_p > 0 the one-lane-per-query kernels run (any head_dim, both
   * dtypes): the matrix-core kernels take no mask. */
  float drop_p; uint64_t drop_seed, drop_stream; const uint64_t* drop_step_dev;
} miseg_winattn_params;
int miseg_winattn_fwd(const miseg_winattn_params* p, miseg_stream_t stream);
/* 1 when the head_dim-16 bf16 matrix-core kernels take this forward call (window attention with a bias table, <= 352 tokens per window;
 * since round 3 with or without attention dropout), 0 when the one-lane-per-query kernels do.  No launch. */
int miseg_winattn_on_matrix_cores(const miseg_winattn_params* p);

typedef struct {
  miseg_winattn_params f;          /* same geometry; f.out is the saved forward output */
  const void* dout; int64_t lddo;  /* gradient w.r.t. out */
  void* dqkv; int64_t lddq;        /* gradient w.r.t. qkv (unpadded grid) */
  float* dqkv_bias;                /* accumulated: contribution of padded tokens; may be NULL */
  float* dbias_table;              /* accumulated fp32 [(2tw-1)^3][heads]; may be NULL */
} miseg_winattn_bwd_params;
int miseg_winattn_bwd(const miseg_winattn_bwd_params* p, miseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Data-movement / small kernels (all channels-last rows)
 * ---------------------------------------------------------------------------------------------- */
/* y = a + b (elementwise over rows x C), any of the three may alias */
typedef struct { const void* a; int64_t lda; const void* b; int64_t ldb; void* y; int64_t ldy; int64_t rows; int C, dtype; } miseg_add_params;
int miseg_add(const miseg_add_params* p, miseg_stream_t stream);
/* y[b][s][c] = coef[b][c][0] * a[b][s][c] + coef[b][c][1] * x[b][s][c] + coef[b][c][2]   (ABI 5; coef: fp32 [B][C][3] on the device) - the input
 * gradient of a normalisation whose means run over several channels: dx = P dy + R x + Q with per-(sample, channel) coefficients */
typedef struct { uint32_t struct_size; const void* a; int64_t lda; const void* x; int64_t ldx; void* y; int64_t ldy; const float* coef; int B, S, C, dtype; } miseg_affine2_params;
int miseg_affine2(const miseg_affine2_params* p, miseg_stream_t stream);

/* strided 2D copy with dtype conversion: dst[r][c] = src[r][c] */
typedef struct { const void* src; int64_t lds; int sdtype; void* dst; int64_t ldd; int ddtype; int64_t rows; int C; } miseg_copy2d_params;
int miseg_copy2d(const miseg_copy2d_params* p, miseg_stream_t stream);

/* fp32 [R][C] -> dtype, optionally transposed to [C][R]  (weight re-packing for the NT GEMM) */
typedef struct { const float* src; void* dst; int R, C, dtype, transpose; } miseg_cast_params;
int miseg_cast_matrix(const miseg_cast_params* p, miseg_stream_t stream);

/* All per-step parameter re-layouts of a model in ONE launch (the per-call form above costs a ~4 us launch per
 * weight, ~100 per step of C-Swin-UNETR): descriptor i turns the fp32 matrix src [R][C] into dst in `dtype`,
 *   dst[r][m(c)] (transpose = 0)  or  dst[m(c)][r] (transpose = 1),   m(c) = (c % inner) * outer + c / inner
 * (inner = 1: identity; inner = 8, outer = Cout: the (co, tap) -> (tap, co) regrouping of ConvTranspose3d k2s2 weights,
 * unetr_block.py:80).  Descriptors live in device memory, sorted by tile0 = number of 32x32 tiles before them. */
typedef struct {
  const float* src; void* dst;
  int32_t R, C, transpose, inner, outer, tile0;
} miseg_cast_desc;
int miseg_param_cast_batch(const miseg_cast_desc* descs_dev, int ndesc, int total_tiles, int dtype, const int64_t* params_version, int64_t* state,
                           miseg_stream_t stream);      /* params_version / state: see miseg_pack_conv3_batch */

/* GELU (exact, erf): y = gelu(x); backward: dx = dy * gelu'(x)  (MONAI MLPBlock act, swin_transformer_block.py:97) */
typedef struct { const void* x; int64_t ldx; void* y; int64_t ldy; int64_t rows; int C, dtype; } miseg_gelu_fwd_params;
int miseg_gelu_fwd(const miseg_gelu_fwd_params* p, miseg_stream_t stream);
typedef struct { const void* dy; int64_t lddy; const void* x; int64_t ldx; void* dx; int64_t lddx; int64_t rows; int C, dtype; } miseg_gelu_bwd_params;
int miseg_gelu_bwd(const miseg_gelu_bwd_params* p, miseg_stream_t stream);

/* 2x2x2 space<->channel gathers.  `offsets` is 8 (dz,dy,dx) triples (host array): output channel block j of
 * the coarse voxel (d,h,w) is the fine voxel (2d+dz_j, 2h+dy_j, 2w+dx_j)  (zero beyond the fine grid).
 *   gather : fine [B][D][H][W][C] -> coarse [B][D2][H2][W2][8C]     (PatchMerging patch_merging.py:120-128 /
 *            PatchMergingV2 :69-71 slice tables; ConvTranspose k2s2 data gradient)
 *   scatter: coarse -> fine, fine[v] = sum of the blocks that reference v   (PatchMerging backward;
 *            ConvTranspose3d k2 s2 forward unetr_block.py:51-59 after the [Cin]x[8*Cout] GEMM)
 */
typedef struct {
  const void* src; int64_t lds; void* dst; int64_t ldd;
  int B, D, H, W, C, dtype;        /* D,H,W: FINE grid; coarse grid is ceil(./2) */
  int8_t offsets[24];
} miseg_s2c_params;
int miseg_space_to_channel(const miseg_s2c_params* p, miseg_stream_t stream);
int miseg_channel_to_space(const miseg_s2c_params* p, miseg_stream_t stream);

/* PatchEmbed Conv3d(k2,s2)+bias with Cin input channels given as NCDHW fp32 (the network input)
 * (networks/blocks/patch_embedding.py:167-169,204).  w: fp32 [Cout][Cin][2][2][2]. */
typedef struct {
  const float* x; void* y; int64_t ldy; const float* w; const float* bias;
  int B, Cin, D, H, W, Cout, dtype;  /* D,H,W: input grid (even) */
} miseg_patch_embed_params;
int miseg_patch_embed_fwd(const miseg_patch_embed_params* p, miseg_stream_t stream);
typedef struct {
  const float* x; const void* dy; int64_t lddy; float* dw; float* dbias;   /* accumulated */
  int B, Cin, D, H, W, Cout, dtype;
  void* workspace;   /* optional, miseg_patch_embed_bwd_workspace_bytes(p): per-workgroup partial sums + a second small launch instead of
                      * fp32 atomics from every workgroup onto the 9 * Cout outputs (46 of 53 us on the headline patch) */
} miseg_patch_embed_bwd_params;
size_t miseg_patch_embed_bwd_workspace_bytes(const miseg_patch_embed_bwd_params* p);
int miseg_patch_embed_bwd(const miseg_patch_embed_bwd_params* p, miseg_stream_t stream);

/* First encoder conv: 3x3x3, Cin in {1..4} NCDHW fp32 input -> channels-last Cout (dynunet_block.py:55-64 with
 * in_channels = image channels).  Weight gradient accumulated into dw fp32 [Cout][Cin][27]. */
typedef struct {
  const float* x; void* y; int64_t ldy; const float* w;
  int B, Cin, D, H, W, Cout, dtype;
} miseg_conv3_thin_params;
int miseg_conv3_thin_fwd(const miseg_conv3_thin_params* p, miseg_stream_t stream);
typedef struct {
  const float* x; const void* dy; int64_t lddy; float* dw;
  int B, Cin, D, H, W, Cout, dtype;
  void* workspace;   /* optional, miseg_conv3_thin_wgrad_workspace_bytes(p) (0: not used by this shape): per-workgroup partial sums + a second
                      * small launch instead of fp32 atomics from every workgroup onto the 27 * Cout outputs */
} miseg_conv3_thin_wgrad_params;
size_t miseg_conv3_thin_wgrad_workspace_bytes(const miseg_conv3_thin_wgrad_params* p);
int miseg_conv3_thin_wgrad(const miseg_conv3_thin_wgrad_params* p, miseg_stream_t stream);

/* Network input (NCDHW fp32, Cin <= 8 image channels) -> channels-last rows of CP channels (CP = 4 for fp32, 8 for bf16:
 * one 16-byte vector per voxel), channels >= Cin zero.  Feeds the first encoder convs (dynunet_block.py:55-64 with
 * in_channels = image channels) to the implicit-GEMM kernels above. */
int miseg_ncdhw_to_rows(const float* x, void* y, int B, int Cin, int64_t S, int CP, int dtype, miseg_stream_t stream);

/* ---- pieces of the MONAI residual UNet (networks/nets/unet.py:169-205, blocks/convolutions.py:173-179,323-329, acti_norm.py:104-110) ----
 * A stride-2 3x3x3 convolution is the stride-1 kernel followed by `resample2` dir 0 (keep the even voxels); ConvTranspose3d k3 s2 p1 op1 is
 * `resample2` dir 1 (zero insertion) followed by the stride-1 kernel on the mirrored / channel-swapped pack -- the two directions are each
 * other's adjoint, so the backward passes are the same two kernels.  D, H, W are the FINE grid; the coarse grid is ceil(./2). */
typedef struct { const void* x; int64_t ldx; void* y; int64_t ldy; int B, D, H, W, C, dtype, dir; } miseg_resample2_params;
int miseg_resample2(const miseg_resample2_params* p, miseg_stream_t stream);
/* y[r][c] = x[r][c] + bias[c]  (Conv3d / ConvTranspose3d bias, convolutions.py:115-139; the gradient is miseg_colsum) */
typedef struct { const void* x; int64_t ldx; const float* bias; void* y; int64_t ldy; int64_t rows; int C, dtype; } miseg_rowbias_params;
int miseg_rowbias_add(const miseg_rowbias_params* p, miseg_stream_t stream);
/* PReLU with ONE learnable slope (torch.nn.PReLU() as built by ADN, acti_norm.py:90-93): y = x > 0 ? x : a x;
 * backward: dx = dy (x > 0 ? 1 : a), dslope += sum dy x [x <= 0]  (dslope accumulated, may be NULL) */
typedef struct { const void* x; int64_t ldx; const float* slope; void* y; int64_t ldy; int64_t rows; int C, dtype; } miseg_prelu_fwd_params;
int miseg_prelu_fwd(const miseg_prelu_fwd_params* p, miseg_stream_t stream);
typedef struct {
  const void* dy; int64_t lddy; const void* x; int64_t ldx; const float* slope; void* dx; int64_t lddx; float* dslope; int64_t rows; int C, dtype;
  double* scratch;   /* DEVICE double[2], zero on entry (with dslope): the one-element slope gradient is a cancelling sum over every voxel - workgroup
                        partials meet here in float64 and the last workgroup adds the total to dslope (fp32 atomics were off by up to 60 %) */
} miseg_prelu_bwd_params;
int miseg_prelu_bwd(const miseg_prelu_bwd_params* p, miseg_stream_t stream);
/* channels-last rows [B][S][C] (ld) in `dtype`  <->  NCDHW fp32 [B][C][S]: dir 0 rows -> NCDHW (network output), dir 1 NCDHW -> rows */
int miseg_layout_ncdhw(const void* rows_, int64_t ld, float* ncdhw, int B, int C, int64_t S, int dtype, int dir, miseg_stream_t stream);

/* Output head: Conv3d 1x1x1 + bias from channels-last rows to NCDHW fp32 logits (dynunet_block.py:273-292),
 * and its backward (dx channels-last, dw/dbias accumulated). */
typedef struct {
  const void* x; int64_t ldx; float* y; const float* w; const float* bias;
  int B, S, Cin, Cout, dtype;
} miseg_head_params;
int miseg_head_fwd(const miseg_head_params* p, miseg_stream_t stream);
typedef struct {
  const void* x; int64_t ldx; const float* dy; void* dx; int64_t lddx; const float* w; float* dw; float* dbias;
  int B, S, Cin, Cout, dtype;
} miseg_head_bwd_params;
int miseg_head_bwd(const miseg_head_bwd_params* p, miseg_stream_t stream);

/* im2col for 3x3x3/pad 1 on small grids (<= 6^3): col[v][tap][ci]; and its adjoint col2im (dst zero-filled by
 * the kernel).  Used with miseg_gemm where the implicit-GEMM tile would be mostly halo. */
typedef struct { const void* src; int64_t lds; void* dst; int64_t ldd; int B, D, H, W, C, dtype; } miseg_im2col3_params;
int miseg_im2col3(const miseg_im2col3_params* p, miseg_stream_t stream);
int miseg_col2im3(const miseg_im2col3_params* p, miseg_stream_t stream);

/* fill a buffer of n 32-bit words with a value (gradient arenas, accumulators) */
#define MISEG_FILL_RANGES 16
/* dst[off .. off + len) = value for up to MISEG_FILL_RANGES (offset, length) pairs, in 32-bit words, of one 16-byte aligned buffer in ONE launch
 * (ABI 8; ranges_host: HOST array [n][2], copied into the kernel arguments; offsets multiples of 4 words): the per-step zero fill of the gradient
 * arena minus the slots whose weight-gradient kernel overwrites them whole (mi-seg_amd/runtime/arena.py) */
int miseg_fill32_ranges(void* dst, uint32_t value, const uint64_t* ranges_host, int n, miseg_stream_t stream);
int miseg_fill32(void* dst, uint32_t value, size_t n, miseg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * After the path, every training step: segmentation loss + d(loss)/d(logits), Dice metric,
 * multi-tensor optimiser step over the gradient arena, sliding-window stitching.  MONAI 1.1.0 arithmetic restated from its
 * public API (the reference calls it at networks/lightning_monai.py:46-67,68-79,86-93,190-195,255-278; MONAI itself is a
 * pinned dependency that is not vendored: PARITY UNPINNED by any reference test).
 * ---------------------------------------------------------------------------------------------- */
enum { MISEG_LABEL_F32 = 0, MISEG_LABEL_I32 = 1, MISEG_LABEL_I64 = 2, MISEG_LABEL_U8 = 3 };
enum { MISEG_LOSS_DICE_FOCAL = 0, MISEG_LOSS_DICE_CE = 1 };
/* DiceFocalLoss / DiceCELoss(to_onehot_y=True, softmax=True) on fp32 NCDHW logits [B][C][S] and class-id labels [B][1][S]
 * (lightning_monai.py:48-65, training_step :149-166).
 *   dice_focal, include_background = 0: channel 0 is stripped from logits AND target first, the softmax of the Dice term runs over the C-1
 *     foreground logits, the focal term (sigmoid form on raw logits, gamma) over the same channels;  = 1: all C channels.
 *   dice_ce: softmax over all C channels, Dice over channels >= (include_background ? 0 : 1), cross-entropy over all channels.
 *   Dice per (b, c): 1 - (2 sum(p t) + smooth_nr) / (sum(t) + sum(p^2 or p) + smooth_dr), mean over (b, c); focal: mean over (b, c, s); CE: mean over (b, s).
 * miseg_seg_loss_fwd: one pass over logits + labels -> per-workgroup fp64 partial sums in `workspace`, summed in a FIXED order by a
 *   one-workgroup launch (bit-reproducible) into sums[B][C][3] (sum p t, sum p^2|p, sum t) + sums[3 B C] (focal / CE total) and the scalar loss.
 * miseg_seg_loss_bwd: second pass -> dlogits [B][C][S] fp32 = gscale * d(loss)/d(logits) (gscale: device scalar or NULL = 1), from the saved sums. */
typedef struct {
  uint32_t struct_size;            /* sizeof(miseg_seg_loss_params): checked against the library's own */
  int kind;                        /* MISEG_LOSS_* */
  const float* logits; const void* label; int label_dtype;
  int B, C; int64_t S;
  int include_background, squared_pred;
  float smooth_nr, smooth_dr, gamma, lambda_dice, lambda_other;
  void* workspace;                 /* miseg_seg_loss_workspace_bytes(B, C, S) bytes, uninitialised */
  double* sums;                    /* out (fwd) / in (bwd): 3 B C + 1 doubles */
  float* loss;                     /* out (fwd): scalar */
  const float* gscale; float* dlogits;   /* bwd only */
} miseg_seg_loss_params;
size_t miseg_seg_loss_workspace_bytes(int B, int C, int64_t S);
int miseg_seg_loss_fwd(const miseg_seg_loss_params* p, miseg_stream_t stream);
int miseg_seg_loss_bwd(const miseg_seg_loss_params* p, miseg_stream_t stream);

/* DiceMetric(include_background=True, get_not_nans=True) after AsDiscrete(argmax=True, to_onehot=C) (lightning_monai.py:68-79,190-195):
 * dice[b][c] = 2 |pred == c & label == c| / (|pred == c| + |label == c|), NaN where |label == c| == 0; argmax takes the FIRST maximum.
 * counts: uint64 [B][C][3] scratch (zeroed by the call; integer atomics: bit-reproducible). */
typedef struct {
  uint32_t struct_size;
  const float* logits; const void* label; int label_dtype;
  int B, C; int64_t S;
  uint64_t* counts; float* dice;   /* out fp32 [B][C] */
} miseg_dice_metric_params;
int miseg_dice_metric(const miseg_dice_metric_params* p, miseg_stream_t stream);

/* One optimiser step for every parameter of a model in ONE launch (lightning_monai.py:255-278: AdamW / Adam / SGD-nesterov), over the flat
 * fp32 gradient arena the weight-gradient kernels accumulate into.  Descriptor i: parameter tensor `param` of n elements whose gradient, and
 * whose two state slots, live at element offset `off` of grad / state1 / state2 (state2 unused by SGD).  used[i] == 0 => the parameter had no
 * gradient this step (`grad is None`: an absent style's norm rows) and is skipped entirely - no update, no weight decay, no step count -
 * exactly like torch.optim skips `p.grad is None`.  steps[i] (int32, device) counts the updates of parameter i (Adam bias correction).
 * Descriptors live in device memory sorted by block0 = number of 4096-element blocks before them. */
enum { MISEG_OPT_ADAMW = 0, MISEG_OPT_ADAM = 1, MISEG_OPT_SGD_NESTEROV = 2 };
typedef struct { float* param; int64_t off; int32_t n, block0; } miseg_opt_desc;
typedef struct {
  uint32_t struct_size;
  int kind;
  const miseg_opt_desc* descs_dev; int ndesc, total_blocks;
  const float* grad; float* state1; float* state2;
  const int32_t* used; int32_t* steps;
  float lr, beta1, beta2, eps, weight_decay, momentum;
  const float* lr_dev;             /* optional device scalar overriding lr (schedulers under hipGraph replay) */
  int64_t* params_version;         /* optional DEVICE int64, incremented once per launch: the parameters changed (miseg_pack_conv3_batch) */
  /* ABI 8, a step in TWO launches over disjoint descriptor tables (the parameters whose gradients are final early - the big tiny-volume conv
   * weights of the headline net - are updated on a side stream beside the end of the backward pass, the rest after it):
   * index (optional, DEVICE int32 [ndesc]): descriptor d of THIS table is parameter index[d] of used / steps (NULL: d itself);
   * count_n: > 0 = after the update, steps[i] += used[i] for i < count_n and params_version is bumped (the LAST launch of a step, over all
   * parameters); 0 = an early launch: neither (every launch of a step must see the step counts of before it); < 0 = ndesc (one-launch step). */
  const int32_t* index; int32_t count_n;
} miseg_opt_step_params;
int miseg_opt_step(const miseg_opt_step_params* p, miseg_stream_t stream);

/* ABI 9.  The optimiser step of every 3x3x3 conv weight of a model TOGETHER WITH its re-layout (lightning_monai.py:255-278 + what the next
 * forward pass needs of dynunet_block.py:295-326): the launch walks the 16 x 16 (co, ci) tiles of the miseg_pack_conv3_batch table, updates a
 * tile's weights / state from their gradients (same arithmetic as miseg_opt_step, element offsets of `grad` / `state1` / `state2` = map[i].off
 * + the element's index in the weight) and writes the tile's forward and data-gradient packs from the NEW values.  The refresh launch of the
 * next step then has nothing to do for these weights: with `pack_state` (the `state` words of that table's versioned refresh) the launch
 * records the new parameter version there.  map[i] belongs to descs[i]; param_index = the weight's row in used / steps.  A tensor with
 * used[param_index] == 0 is left alone (weights, state, packs).  `p`: descs_dev / ndesc / total_blocks / index are ignored; count_n > 0 = this is
 * the LAST launch of the step (steps[i] += used[i] for i < count_n, version bump - see miseg_opt_step); 0 = another launch follows. */
typedef struct { int64_t off; int32_t param_index, pad_; } miseg_opt_pack_map;
int miseg_opt_step_pack_conv3(const miseg_opt_step_params* p, const miseg_pack_conv3_desc* descs_dev, const miseg_opt_pack_map* map_dev, int n, int total_tiles,
                              int dtype, int64_t* pack_state, miseg_stream_t stream);

/* Sliding-window stitching (MONAI sliding_window_inference, mode="constant", as used at lightning_monai.py:86-93,187): every window's logits
 * stay resident (win: fp32 [nd*nh*nw][C][rd][rh][rw], window (id, ih, iw) at index (id*nh + ih)*nw + iw, origin (start_d[id], start_h[ih],
 * start_w[iw]); 700 windows x 6 x 96^3 = 14.9 GB of the 288), and ONE gather pass writes out[c][d][h][w] = (sum over the windows covering the
 * voxel, in window-index order = the order MONAI accumulates them) / count.  No atomics, no read-modify-write of the 2.3 GB accumulator.
 * starts: HOST int32 arrays (copied into the kernel arguments; at most MISEG_STITCH_MAX_WINDOWS per axis); count (optional): uint16 [D][H][W].
 * Slab form (ABI 5; volumes whose window logits do not fit in memory at once): d_count > 0 writes only the depths [d_begin, d_begin + d_count)
 * of `out` / `count` (which still address the whole [C][D][H][W] volume) from the nd RESIDENT depth layers whose starts are start_d[0..nd):
 * every depth of the slab must be covered by resident layers only, i.e. start_d[0] <= d_begin, no gaps, and the layer after the last
 * resident one starts at or behind d_begin + d_count (the caller's promise; MONAI's accumulation order is kept inside the slab). */
#define MISEG_STITCH_MAX_WINDOWS 64
typedef struct {
  uint32_t struct_size;
  const float* win; float* out; uint16_t* count;
  int C, D, H, W, rd, rh, rw, nd, nh, nw;
  const int32_t* start_d; const int32_t* start_h; const int32_t* start_w;
  int d_begin, d_count;            /* d_count == 0: the whole volume */
} miseg_stitch_params;
int miseg_stitch_windows(const miseg_stitch_params* p, miseg_stream_t stream);

/* GPU-resident training augmentation (the random part of the MONAI chain at data/multi_modal.py:50-65, after the cached deterministic
 * part): for each of n <= MISEG_AUG_MAX_SAMPLES samples ONE gather pass produces a roi-sized image + label patch from the resident volume:
 *   RandCropByPosNegLabeld (crop origin, chosen by the host from cached foreground / background voxel lists), RandFlipd x3 (flip[a] on
 *   the PATCH axis a), RandRotate90d (rot_k quarter turns in the (0, 1) plane, roi_d == roi_h when rot_k is odd), RandScaleIntensityd
 *   (image * (1 + scale)), RandShiftIntensityd (image + shift) - applied in that order, intensity ops on the image only.
 * image: fp32 [C][D][H][W]; label: [D][H][W] elements of label_bytes (1, 4 or 8) bytes, copied verbatim; out_image fp32 [n][C][rd][rh][rw];
 * out_label [n][rd][rh][rw].  Sample descriptors are a HOST array (copied into the kernel arguments). */
#define MISEG_AUG_MAX_SAMPLES 16
typedef struct { int32_t origin[3]; int32_t flip[3]; int32_t rot_k; float scale, shift; } miseg_aug_sample;
typedef struct {
  uint32_t struct_size;
  const float* image; const void* label; int label_bytes;
  int C, D, H, W, rd, rh, rw, n;
  float* out_image; void* out_label;
  const miseg_aug_sample* samples_host;
} miseg_augment_params;
int miseg_augment_crop(const miseg_augment_params* p, miseg_stream_t stream);

/* Dropout / stochastic depth on channels-last rows (the `drop` / `dropout_path_rate` arguments of the reference's Swin stack:
 * networks/blocks/swin_transformer_block.py:90-97,205,247; MONAI Dropout / DropPath semantics): y = x * keep / (1 - p).
 *   rows_per_sample == 0: one Bernoulli(1 - p) draw per ELEMENT (nn.Dropout); > 0: one draw per SAMPLE (DropPath: rows r belong to
 *   sample r / rows_per_sample).  The mask is a pure function of (seed, stream_id, *step_dev, element / sample index) - a counter-based
 *   hash, nothing is stored: the backward pass is the same call on the gradient.  stream_id tells call sites apart within a step; step_dev
 *   (device uint64, may be NULL) tells steps apart under hipGraph replay, where seed and stream_id are baked into the graph
 *   (advance it once per step with miseg_counter_add). */
typedef struct {
  uint32_t struct_size;
  const void* x; int64_t ldx; void* y; int64_t ldy;
  int64_t rows; int C, dtype;
  int64_t rows_per_sample;
  float p;
  uint64_t seed, stream_id;
  const uint64_t* step_dev;
} miseg_dropout_params;
int miseg_dropout(const miseg_dropout_params* p, miseg_stream_t stream);
int miseg_counter_add(uint64_t* counter_dev, uint64_t value, miseg_stream_t stream);
/* *dst = *src on the device (a plain kernel: device-to-device copies recorded as memcpy nodes crashed hipStreamEndCapture on ROCm 7.2).
 * A dropout call snapshots the step counter so that its backward pass re-creates the same mask after the counter moved on. */
int miseg_counter_copy(uint64_t* dst_dev, const uint64_t* src_dev, miseg_stream_t stream);
/* (measurement and experiment entry points - miseg_debug_stamp, miseg_prof_*, miseg_flag_wait, miseg_graph_split_* - are declared in
 * include/miseg_hip_debug.h: no product path calls them) */

/* The resampling step of the cached, deterministic head of the data chain (Spacingd at data/multi_modal.py:41-44: image "bilinear", label
 * "nearest"): in [C][Di][Hi][Wi] -> out [C][Do][Ho][Wo], voxel centres aligned (src = (dst + 0.5) * in / out - 0.5), coordinates clamped to
 * the border.  mode 0: trilinear (fp32), mode 1: nearest (round half up).  elem_bytes 4: fp32 for mode 0; 1 / 4 / 8 for mode 1 (labels are
 * copied verbatim).  MONAI's own grid construction is not restated (parity unpinned). */
typedef struct {
  uint32_t struct_size;
  const void* in; void* out;
  int C, Di, Hi, Wi, Do, Ho, Wo, mode, elem_bytes;
} miseg_resample3d_params;
int miseg_resample3d(const miseg_resample3d_params* p, miseg_stream_t stream);

/* sizeof() of a params struct as this library was compiled ("miseg_gemm_params", ...), 0 for an unknown name: bindings compare it with
 * their own mirror at load time (together with miseg_abi_version) so that header and binding cannot drift silently. */
size_t miseg_abi_struct_size(const char* struct_name);
/* sha256 (hex) over the sources this library was compiled from (csrc/build.py); a binding that finds the sources beside the library
 * compares and refuses a stale build */
const char* miseg_source_digest(void);
/* MISEG_OK when HIP device `device` runs the code objects embedded in this library (gcnArchName starts with miseg_device_arch) */
int miseg_device_check(int device);

#ifdef __cplusplus
}
#endif
#endif
