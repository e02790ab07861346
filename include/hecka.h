/* libhecka — C-ABI of the MI355X (gfx950) HE-CKA diversity-loss hot path.
 *
 * The reference (Deep-Machine-Vision/he-cka-ensembles) is pure Python/PyTorch and has no FFI
 * for this path; each entry point below names the reference Python function
 * (file:line under /root/reference) whose device work it replaces.  The host-side mirror of the
 * reference plugin API (`hyper_amd.diversity.*`) binds these with ctypes; INTEGRATION.md shows
 * the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless said otherwise;
 *   - all tensors are fp32, row-major, contiguous; M = members, B = samples, P = features;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); work is enqueued,
 *     never synchronised; no entry point allocates, frees or copies to the host;
 *   - return value: 0 = ok, <0 = error (HECKA_E*); hecka_last_error() gives the message of the
 *     calling thread's last failure.  Nothing throws.
 */
#ifndef HECKA_H
#define HECKA_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HECKA_OK 0
#define HECKA_EINVAL (-1)    /* bad argument (null pointer, non-positive size, misuse) */
#define HECKA_ELAUNCH (-2)   /* HIP reported a launch error */
#define HECKA_EWORKSPACE (-3) /* workspace too small */

int hecka_version(void);
const char* hecka_last_error(void);
/* The library reads its (few, optional) HECKA_* environment knobs once, at load time - never on the call path;
 * hecka_config_reload() re-reads them (tests that switch a knob inside one process).  INTEGRATION.md 6 lists them. */
void hecka_config_reload(void);
/* Products per multiply of the LAYER GEMMs and convolutions on the fp16-split tile (hecka_bgemm_x3_f32, hecka_conv_*x3*_f32);
 * the loss path is not touched.  3 (default): x 2^e = h + l, products hh + hl + lh - fp32-grade.  1: hh only - the operands
 * (fp32 in HBM) are rounded to fp16 after their power-of-two scale, accumulation stays fp32: a 16-bit COMPUTE mode (11
 * significand bits, more than bf16's 8).  No counterpart in the reference, which is fp32-only: results
 * equal the fp64 product of the rounded operands (tests/test_gpu_parity.py), parity with the reference is unpinned.
 * Process-wide; returns the previous value, -1 for a value other than 1 or 3.  HECKA_LAYER_PRODUCTS sets the initial one. */
int hecka_set_layer_products(int n);
int hecka_get_layer_products(void);

/* Arithmetic of the two large GEMMs - the `gemm_mode` argument of hecka_gram_center_*, hecka_gram_bwd_*.  No counterpart
 * in the reference (torch.bmm picks its own algorithm).  It is a per-call argument: the library holds no process-wide
 * state, two callers in one process may use different modes.  Interior shapes (B % 128 == 0 and P % 16 == 0
 * resp. P % 128 == 0) run on the 16-bit matrix cores with fp32 operands split EXACTLY into 16-bit terms, partial
 * sums over 64 k folded into the running sum with round-to-nearest fp32 adds.  Ragged shapes made of whole quads (B % 4
 * == 0, P % 4 == 0 for the Gram; the reference's batch sizes 300 / 100 / 400) take the same tile with masked loads and
 * epilogues under HECKA_GEMM_AUTO when xmax / wmax are given and the grid of 128-row tiles fills the GPU (>= 384
 * workgroups; HECKA_RAGGED_X3=1 / 0 in the environment forces / forbids it); all other shapes use fp32 MFMA.
 *   HECKA_GEMM_AUTO (default): "f16x3" when the caller supplies the operands' absolute maxima (xmax / wmax below):
 *     x * 2^e = h + l, two fp16 terms (22 significand bits), three products hh + hl + lh per a*b; the power-of-two
 *     scale 2^e comes from the maximum, so an element below 2^-17 of its member's maximum loses relative (not
 *     absolute) precision.  Measured error <= 6e-7 of sum|a.b| at K = 8192 (fp32 MFMA: 2.5e-6).  Without maxima:
 *     "bf16x6" - three bf16 terms (24 bits), six products, no scaling, no range assumption; error <= 4e-7.
 *   HECKA_GEMM_BF16X6: never f16x3.     HECKA_GEMM_FP32: fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere.
 * The fp16-split layer GEMMs (hecka_bgemm_x3_f32, hecka_conv_*_x3_f32) are separate entry points: a caller that wants
 * fp32 MFMA there calls hecka_bgemm_f32 instead. */
#define HECKA_GEMM_AUTO 0
#define HECKA_GEMM_FP32 1
#define HECKA_GEMM_BF16X6 2

/* Column mean of each member's features: sbar[m,p] = (1/B) sum_b X[m,b,p].
 * First half of the fused replacement of center_gram (hyper/diversity/kernels.py:335-345):
 * C = K - rowmean(K) == X (X - 1 sbar^T)^T, so the centring is applied to the features.
 * xmax (may be NULL): [M] side output max |X[m]| for the f16x3 GEMMs (zeroed and filled by this call). */
int hecka_colmean_f32(const float* X, int M, int B, int P, float* sbar, float* xmax, void* stream);
/* hecka_colmean_f32 for L layers (same M and B) in ONE launch and one memset: X / sbar HOST arrays of L device pointers, P
 * a host array of widths (P % 4 == 0, 16-byte aligned), xmax_all [L, M] contiguous.  Same values as the per-layer call. */
int hecka_colmean_multi_f32(int L, const float* const* X, int M, int B, const int* P, float* const* sbar, float* xmax_all,
                            void* stream);

/* Centred linear Gram of every member, written once:
 *   C[m] = X_m X_m^T - rowmean(X_m X_m^T)   ([M,B,B]),   cvec[m,b] = sbar_m . (x_b - sbar_m)  ([M,B])
 * computed as Kt = (X-1 sbar^T)(X-1 sbar^T)^T on fp32 MFMA (upper-triangular tiles, mirrored) and
 * C = Kt + 1 cvec^T.  Replaces linear_gram + detach_diagonal(forward value) + center_gram
 * (hyper/diversity/kernels.py:107-127, 76-104, 335-345) as called from FeatureKernel.batched_kernel
 * (hyper/diversity/model_kernels.py:84-123) and HEModelKernel.forward (:369-377).
 * xmax (may be NULL): [M] max |X[m]| from hecka_colmean_f32; enables f16x3 (see HECKA_GEMM_AUTO). */
int hecka_gram_center_f32(const float* X, const float* sbar, const float* xmax, int M, int B, int P,
                          float* C, float* cvec, int gemm_mode, void* stream);

/* hecka_gram_center_f32 for L tracked layers (same M and B, their own P) in ONE launch: X / sbar / xmax / C / cvec are
 * HOST arrays of L device pointers (xmax, or any xmax[i], may be NULL), P a host array of L feature widths, layers
 * ordered deepest (largest P) first.  Split-precision path only, L <= 16: interior shapes (B % 128 == 0, every
 * P % 16 == 0), or - HECKA_GEMM_AUTO with every xmax given - ragged ones with B % 4 == 0 and every P % 4 == 0 (masked
 * tile, B * P * 4 < 1 GiB); callers fall back to per-layer hecka_gram_center_f32 otherwise.  Replaces the Python loop over layers of
 * ModelLayersKernel.forward (hyper/diversity/model_kernels.py:446-478) for the Gram stage: one layer alone leaves
 * the second scheduling round of the GPU three-quarters empty (M*10 workgroups on 512 slots at B = 512). */
int hecka_gram_center_multi_f32(int L, const float* const* X, const float* const* sbar, const float* const* xmax, int M,
                                int B, const int* P, float* const* C, float* const* cvec, int gemm_mode, void* stream);

/* hecka_colmean_multi_f32 + hecka_gram_center_multi_f32 WITHOUT the column-mean pass over X ("speculative" Gram; same
 * outputs: C, cvec, xmax_all - the replacement of linear_gram + center_gram, hyper/diversity/kernels.py:107-127, 335-345).
 * C = K - rowmean(K) holds for ANY shift t of the features: with xs = x - t the Gram kernel, handed t in place of the
 * means, writes Chat[a,b] = x_a . xs_b, and C = Chat - rowmean(Chat) 1^T exactly (a row-centring pass over the Grams,
 * B / P of the bytes of a pass over X), cvec = q + rho - mean(q) - mean(rho) with q = t . xs_b, rho the row means.  t = the
 * mean of 16 sample rows (near the true means, so nothing cancels); the f16x3 scale comes from the sampled maximum with
 * 16 x headroom and the Gram prologue tracks the exact max |x - t|: if a member does not fit (a tail the sample missed,
 * non-finite features) a device flag is raised and the exact two-pass path, enqueued behind and gated on that flag, runs
 * instead - nothing returns to the host, the call is stream-ordered and capturable like every other.
 * Eligible: fp32, HECKA_GEMM_AUTO arithmetic, 256 <= B <= 1024 with B % 128 == 0, every P % 32 == 0, layers deepest first,
 * a grid the all-layers launch does not split over K (hecka_gram_center_spec_eligible; HECKA_GRAM_SPEC=0 turns it off).
 * sbar[l] ([M,P_l], scratch: holds t, or the exact means after a fallback), xmax_all [L,M] (out: a bound on max |X_m|
 * for hecka_gram_bwd_*), ws of hecka_gram_center_spec_workspace_bytes bytes, 16-byte aligned; after the call the int at
 * byte offset 12 L M of ws is 1 if the fallback ran (tests read it). */
int hecka_gram_center_spec_eligible(int L, int M, int B, const int* P);
size_t hecka_gram_center_spec_workspace_bytes(int L, int M, int B, const int* P);
int hecka_gram_center_spec_multi_f32(int L, const float* const* X, float* const* sbar, float* xmax_all, int M, int B,
                                     const int* P, float* const* C, float* const* cvec, void* ws, size_t ws_bytes,
                                     void* stream);

/* Pairwise inner products of vectorised Grams: S[i,j] = <Ca[i,:], Cb[j,:]>, Ca [Ma,K], Cb [Mb,K],
 * K = B*B.  Split-K over workgroups with a fixed-order (deterministic) second-stage sum.
 * Replaces vectorize_grams + the matmul of unit_vector_rows outputs (the norms are the diagonal
 * of S) in pairwise_cossim / hyperspherical_energy (hyper/diversity/__init__.py:33-56, 79-108).
 * Ca == Cb, Ma == Mb on one GPU; with members sharded over GPUs Ca holds the local members and
 * Cb the all-gathered ones.  `ws` needs hecka_pairwise_workspace_bytes(Ma, Mb, K) bytes. */
size_t hecka_pairwise_workspace_bytes(int Ma, int Mb, long long K);
int hecka_pairwise_f32(const float* Ca, int Ma, const float* Cb, int Mb, long long K,
                       float* S, void* ws, size_t ws_bytes, void* stream);

/* hecka_pairwise_f32 for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): the same kernels and the same result, bit for bit, but the split-K partial sums live in
 * an 8 MiB array built into the library (one copy per GPU, held until the process ends) instead of a caller workspace.
 * No counterpart in the reference.  Calls no hipMalloc / hipFree / allocating routine.  Set-up: hecka_noalloc_init(), once
 * per device with that device current, looks up the device's copy of the array (hipGetSymbolAddress; idempotent, no
 * allocation).  A call on a device that was never set up does the set-up itself - except while `stream` is capturing,
 * where it returns HECKA_EINVAL: call hecka_noalloc_init() (or make one ordinary call) on the device BEFORE capturing.
 * `stream` must belong to the current device (checked; NULL = the current device's default stream): the kernels get the
 * current device's array.  Shapes whose partial sums exceed 8 MiB
 * (hecka_pairwise_workspace_bytes(Ma, Mb, K) > 8 MiB - not the case for any Ma x Mb <= 64 x 192 or 128 x 128 with the
 * default split) return HECKA_EWORKSPACE: use hecka_pairwise_f32.  All calls on one GPU share that GPU's array, so two
 * calls that overlap on one GPU (different streams, different threads) race; calls on one stream are ordered. */
int hecka_noalloc_init(void);
int hecka_pairwise_noalloc_f32(const float* Ca, int Ma, const float* Cb, int Mb, long long K, float* S, void* stream);

/* Pairwise head on the [Ma,Mb] matrix of inner products (rows = this GPU's members starting at global
 * index row_offset, columns = all members; nsq[Mb] = squared Frobenius norms of all members = diag S):
 *   cos_ij = S_ij / ((sqrt(nsq_i)+eps)(sqrt(nsq_j)+eps))                       unit_vector_rows, __init__.py:24-30
 *   kind 0 (cka): value = cos                                                   pairwise_cossim, __init__.py:33-56
 *   kind 1 (he) : theta = arccos([|cos|]/(1+arc_eps)); value = energy(theta)    hyperspherical_energy, __init__.py:108-144
 *                 energy: s==0 log(1/(theta+eps)+eps) | s==1 1/(theta+eps) | use_exp exp(-(s theta+eps)) | (1+eps)/(theta^s+eps)
 * fwd: val[Ma,Mb] (may be NULL) and *sum_out = scale * sum of values (diagonal pair skipped with REMOVE_DIAG;
 *      sum_out may be NULL) - the reduction of ModelLayersKernel.apply_reduction (model_kernels.py:439-444).
 * bwd: coefficient matrix A[Ma,Mb] of dC_i = sum_j A_ij C_j for upstream dLoss/dvalue_ij =
 *      (up_mat ? up_mat[i,j] : 1) * (up_scalar ? *up_scalar : 1) * scale; includes the Jacobian of the
 *      normalisation and DETACH_RIGHT (gradient through the left factor only).  Replaces autograd through
 *      __init__.py:24-30,108-144.  eps / arc_eps are passed by absolute value (model_kernels.py:291-300). */
#define HECKA_HEAD_USE_EXP 1
#define HECKA_HEAD_ABS_VALS 2
#define HECKA_HEAD_REMOVE_DIAG 4
#define HECKA_HEAD_DETACH_RIGHT 8
int hecka_head_fwd_f32(const float* S, const float* nsq, int Ma, int Mb, int row_offset, int kind, double s,
                       double eps, double arc_eps, int flags, float* val, float* sum_out, double scale, void* stream);
int hecka_head_bwd_f32(const float* S, const float* nsq, int Ma, int Mb, int row_offset, int kind, double s,
                       double eps, double arc_eps, int flags, const float* up_mat, const float* up_scalar,
                       double scale, float* A, void* stream);

/* gamma[i,b] = sum_j A[i,j] cvec[j,b]   (A [Ma,Mb], cvec [Mb,B], gamma [Ma,B]). */
int hecka_coef_gamma_f32(const float* A, const float* cvec, int Ma, int Mb, int B,
                         float* gamma, void* stream);

/* Gradient w.r.t. the (uncentred) Gram, symmetrised for the feature GEMM:
 *   dC_i = sum_j A[i,j] C_j ;   W_i = dC_i + dC_i^T = 2 dC_i + gamma_i 1^T - 1 gamma_i^T ;
 *   diag(W_i) = 0 when detach_diag.
 * A [Ma,Mb] is dLoss/dS symmetrised by the caller, C [Mb,B,B], gamma [Ma,B], W [Ma,B,B].
 * Replaces autograd's backward of matmul/normalise/center_gram/detach_diagonal
 * (ParticleMethods.backward, hyper/diversity/methods.py:139-152).
 * wmax (may be NULL): [Ma] side output max |W_i| for the f16x3 dX GEMM (zeroed and filled by this call). */
int hecka_combine_f32(const float* A, const float* C, const float* gamma, int Ma, int Mb, int B,
                      int detach_diag, float* W, float* wmax, void* stream);

/* The [M,M]-sized stages for ALL tracked layers in one launch each (the reference's Python loop over layers,
 * ModelLayersKernel.forward, model_kernels.py:446-478, is L launch trains of problems far smaller than a launch):
 * S [L,Ma,Mb], nsq [L,Mb], A [L,Ma,Mb], cvec [L,Mb,B], gamma [L,Ma,B], wbound [L,Ma] contiguous over the layer index;
 * kind / s / eps / arc_eps / flags / scale are HOST arrays of L per-layer settings (L <= 16 for the heads).
 * head_fwd_batched: totals[l] = scale[l] * sum of layer l's values (layer weight and 1/count folded into scale[l]).
 * head_bwd_batched: upstream (up_scalar ? *up_scalar : 1) * scale[l]. */
int hecka_head_fwd_batched_f32(const float* S, const float* nsq, int L, int Ma, int Mb, int row_offset, const int* kind,
                               const double* s, const double* eps, const double* arc_eps, const int* flags,
                               const double* scale, float* totals, void* stream);
int hecka_head_bwd_batched_f32(const float* S, const float* nsq, int L, int Ma, int Mb, int row_offset, const int* kind,
                               const double* s, const double* eps, const double* arc_eps, const int* flags,
                               const double* scale, const float* up_scalar, float* A, void* stream);
int hecka_coef_gamma_batched_f32(const float* A, const float* cvec, int L, int Ma, int Mb, int B, float* gamma,
                                 void* stream);
/* S_all[l] = C_all[l] C_all[l]^T for L layers in one launch (+ one reduction launch): C_all [L,M,K] contiguous;
 * ws >= L * hecka_pairwise_workspace_bytes(M, M, K).  M == 64 with K % 32 == 0 takes the streaming kernel, any other
 * shape the LDS-tiled one with the layer in blockIdx.y. */
int hecka_pairwise_self_batched_f32(const float* C_all, int L, int M, long long K, float* S_all, void* ws,
                                    size_t ws_bytes, void* stream);
/* The same two stages on the layouts of the member-sharded exchange (hyper_amd/parallel.py; no counterpart in the reference,
 * which scales with DDP over samples, hyper/experiments/training.py:686-695): one all_to_all_single for ALL layers delivers
 * a rank's column slices as [source rank][layer][members of that rank][n] - the 64 member rows of a layer lie in blocks of
 * blk_rows rows.  hecka_blocked_rows_eligible(M, n, blk_rows): M == 64, n % 64 == 0, blk_rows in {8, 16, 32, 64} (the
 * streaming kernels; callers repack otherwise).
 * hecka_pairwise_self_blocked_f32: C_all [M / blk_rows][L][blk_rows][K] -> S_all [L][M][M], bit-equal to the standard call.
 * hecka_combine_slice_blocked_f32: Cs_all [Mb / blk_rows][Lin][blk_rows][n] of which the layers l0 .. l0 + L - 1 are
 * combined (A_all [L,Ma,Mb], gamma_all [L,Ma,B] hold those L layers) into Ws_all [Ma / blk_rows][L][blk_rows][n] - the send
 * order of the returning all_to_all_single for that group of layers. */
int hecka_blocked_rows_eligible(int M, long long n, int blk_rows);
int hecka_pairwise_self_blocked_f32(const float* C_all, int L, int M, long long K, int blk_rows, float* S_all, void* ws,
                                    size_t ws_bytes, void* stream);
int hecka_combine_slice_blocked_f32(const float* A_all, const float* Cs_all, const float* gamma_all, const float* nsq_all, int L,
                                    int Lin, int l0, int Ma, int Mb, int B, long long n_start, long long n_count, int blk_rows,
                                    int detach_diag, float* Ws_all, void* stream);
int hecka_wbound_batched_f32(const float* A, const float* nsq, const float* gamma, int L, int Ma, int Mb, int B,
                             float* wbound, void* stream);

/* hecka_combine_f32 on a column slice of the flattened Gram: Cs [Mb, n_count] holds the elements
 * [n_start, n_start + n_count) of every C_j (row-major B*B), Ws [Ma, n_count] receives the same elements of every W_i.
 * For the member-sharded multi-GPU path (hyper_amd/parallel.py), where rank r holds a B*B/G slice of ALL members'
 * Grams after the all-to-all; the reference has no counterpart (it scales with DDP over samples,
 * hyper/experiments/training.py:686-695). */
int hecka_combine_slice_f32(const float* A, const float* Cs, const float* gamma, int Ma, int Mb, int B,
                            long long n_start, long long n_count, int detach_diag, float* Ws, void* stream);
/* hecka_combine_slice_f32 for L layers in one launch: A_all [L,Ma,Mb], Cs_all [L,Mb,n_count], gamma_all [L,Ma,B],
 * Ws_all [L,Ma,n_count], all contiguous, the same slice [n_start, n_start + n_count) of every layer (a member shard's
 * slice is 1/G of a Gram: one layer alone is a launch-bound 17 MB).
 * nsq_all [L,Mb] (may be NULL): nsq[j] = ||C_j||_F^2 of the WHOLE Gram of member j, the diagonal of hecka_pairwise_*'s S.
 * Given (and Mb == 64, Ma in {32, 64}, whole 64-column chunks), the 64-deep products run on the fp16 matrix cores as
 * three products of two-term fp16 splits (the arithmetic of the Gram and dX GEMMs; error <= 2^-21 of sum_j |A_ij C_j[n]|),
 * with exact power-of-two scales per member taken from sqrt(nsq[j]) >= max |C_j| - which moves the stage from the
 * fp32-MFMA floor to the HBM side.  NULL (or HECKA_COMBINE_X3=0): fp32 MFMA products.  Same argument in
 * hecka_combine_slice_blocked_f32. */
int hecka_combine_slice_batched_f32(const float* A_all, const float* Cs_all, const float* gamma_all, const float* nsq_all, int L,
                                    int Ma, int Mb, int B, long long n_start, long long n_count, int detach_diag, float* Ws_all,
                                    void* stream);

/* Upper bound of max |W_i| from what already exists: wbound[i] = 2 sum_j |A[i,j]| sqrt(nsq[j]) + 2 max_b |gamma[i,b]|
 * (nsq[j] = ||C_j||_F^2, the diagonal of hecka_pairwise_f32's S).  Accepted by hecka_gram_bwd_f32 in place of the exact
 * maximum (a looser power-of-two scale only lowers the fp16 low plane's subnormal floor), so callers that have nsq pass
 * wmax = NULL to hecka_combine_f32 and skip its pass over W.  No counterpart in the reference. */
int hecka_wbound_f32(const float* A, const float* nsq, const float* gamma, int Ma, int Mb, int B, float* wbound,
                     void* stream);

/* out = A V   (A [Ma,Mb], V [Mb,K], out [Ma,K]) — backward of the generic pairwise inner product
 * (autograd of the matmul in hyper/diversity/__init__.py:47,108 for Grams that did not come from
 * hecka_gram_center_f32). */
int hecka_rowmix_f32(const float* A, const float* V, int Ma, int Mb, long long K, float* out, void* stream);

/* Row centring of a [rows, cols] matrix: out = V - rowmean(V) (out may be NULL), mean[r] = rowmean
 * (mean may be NULL).  center_gram (hyper/diversity/kernels.py:335-345) and, being self-adjoint, its
 * own backward. */
int hecka_center_rows_f32(const float* V, long long rows, int cols, float* out, float* mean, void* stream);

/* W[m] = G[m] + G[m]^T, diagonal zeroed when detach_diag — turns dLoss/dK into the symmetric factor
 * of dX = W X; backward of torch.bmm(X, X^T) + detach_diagonal (hyper/diversity/kernels.py:122-125, 76-104). */
int hecka_symmetrize_f32(const float* G, int M, int B, int detach_diag, float* W, void* stream);

/* Non-linear Gram kernels on top of the raw Gram G = X X^T (hecka_gram_center_f32 with sbar = 0):
 *   kind 1 cossim: K'_ab = G_ab / sqrt(G_aa G_bb)                     cossim_gram, hyper/diversity/kernels.py:145-169
 *   kind 2 rbf   : d_ab = -2 G_ab + G_aa + G_bb, K' = exp(-d/(2 h_m))   rbf_gram, kernels.py:198-286 (Y=None, no EMA)
 * hecka_rbf_bandwidth_f32: h_m = param * median(d_m) / (2 log(B+1)), median = exact LOWER median of the B*B
 *   distances (torch.median), kernels.py:250-269; radix select, no gradient (as in the reference).
 * hecka_gram_transform_f32: C = K' - rowmean(K') when `center` (center_gram, kernels.py:335-345) else K';
 *   side outputs rmean[M,B] (row means, 0 if !center) and gdiag[M,B] (G_aa).  Out of place (C != G).
 * hecka_gram_transform_bwd_f32: in place on D = dK' + dK'^T with zero diagonal (hecka_combine_f32 with
 *   gamma = A rmean, or hecka_symmetrize_f32): D <- dG + dG^T, ready for hecka_gram_bwd_f32.  The diagonal of
 *   K' carries no gradient for either kernel (cossim: K'_aa = 1, rbf: d_aa = 0), detach_diag or not. */
int hecka_rbf_bandwidth_f32(const float* G, int M, int B, float param, float* hband, void* stream);
int hecka_gram_transform_f32(const float* G, int M, int B, int kind, const float* hband, int center, float* C,
                             float* rmean, float* gdiag, void* stream);
int hecka_gram_transform_bwd_f32(float* D, const float* Cown, const float* rmean, const float* gdiag,
                                 const float* hband, int M, int B, int kind, void* stream);

/* Batched-weight Linear as a grouped GEMM on fp32 MFMA:
 *   C[b] = alpha * A[b] op(B[b]) + beta * bias[b][col],  C [batch, rows, cols] contiguous
 *   A[b] = A + b*strideA : [rows, K];  B[b] = B + b*strideB : nn ? [K, cols] : [cols, K];  bias [batch, cols] or NULL.
 *   strideA / strideB = 0 share that operand across the batch (a 2-D input expanded over the ensemble).
 * forward  y = baddbmm(b', x, W'^T)   hyper/layers/linear.py:144-155: nn=0, A=x, B=W, alpha=scale, beta=bias_scale
 * dgrad    dx = dy W'                 nn=1, A=dy, B=W;   wgrad  dW = dy^T x: nn=0 on transposed operands. */
int hecka_bgemm_f32(const float* A, const float* B, float* C, int batch, int rows, int cols, int K,
                    long long strideA, long long strideB, int nn, float alpha, const float* bias, float beta,
                    void* stream);

/* Split-K form of hecka_bgemm_f32 (no bias) for a small output over a long K - the weight gradient of a convolution is
 * [O, C kh kw] summed over all N Ho Wo patches.  hecka_bgemm_splitk_workspace_bytes returns 0 when the shape should
 * not be split (then call hecka_bgemm_f32); otherwise `ws` takes the partial slabs, summed in a fixed order. */
size_t hecka_bgemm_splitk_workspace_bytes(int batch, int rows, int cols, int K);
int hecka_bgemm_splitk_f32(const float* A, const float* B, float* C, int batch, int rows, int cols, int K,
                           long long strideA, long long strideB, int nn, float alpha, void* ws, size_t ws_bytes,
                           void* stream);

/* hecka_bgemm_f32 (no bias) on the fp16-split tile, 2.5-3x the fp32-MFMA rate, for interior shapes (rows % 64 == 0,
 * cols % 128 == 0, K % 16 == 0; hecka_bgemm_x3_eligible): amaxA [batch] and amaxB [batch, or one value with
 * amaxB_stride = 0 for a shared operand] are max |A[b]| / max |B[b]| or upper bounds (hecka_absmax_rows_f32: out[r] =
 * max |V[r, 0..n)|).  Few tiles over a long K (convolution weight gradients) are split over K:
 * hecka_bgemm_x3_workspace_bytes is non-zero then, the slabs are summed in a fixed order.  The convolution GEMMs of the
 * batched-weight Conv2d at 128 and more channels; no counterpart in the reference (it calls F.conv2d per member). */
int hecka_absmax_rows_f32(const float* V, int rows, long long n, float* out, void* stream);
int hecka_bgemm_x3_eligible(int batch, int rows, int cols, int K);
size_t hecka_bgemm_x3_workspace_bytes(int batch, int rows, int cols, int K);
int hecka_bgemm_x3_f32(const float* A, const float* B, float* C, int batch, int rows, int cols, int K, long long strideA,
                       long long strideB, int nn, float alpha, const float* amaxA, const float* amaxB, int amaxB_stride,
                       void* ws, size_t ws_bytes, void* stream);

/* Implicit-GEMM convolution on the fp16-split tile, unit stride and dilation, output as large as the input:
 *   Y[b][n][r][h][w] = alpha * sum_k A[b][r][k] * X[b][n][c_k][h + di_k][w + dj_k]   (zero outside the image)
 * A [batch, rows, K] (strideA between batches), X [batch | shared, N, Cin, H, W] (strideX, 0 = shared), Y [batch, N, rows,
 * H, W]; tab [K][2] int32 per tap k: { (c_k H + di_k) W + dj_k , (di_k << 16) | (dj_k & 0xffff) }.  No patch matrix is
 * written or read.  Forward of Conv2d: A = weights [O, C kh kw], di = i - pad.  Input gradient: A = rotated weights
 * [C, O kh kw], X = dy, di = i - (kh - 1 - pad): replaces the dgrad GEMM and hecka_col2im_f32.  amaxA / amaxB as for
 * hecka_bgemm_x3_f32.  rows % 64 == 0, N H W % 128 == 0, K % 16 == 0 (hecka_conv_x3_eligible). */
int hecka_conv_x3_eligible(int batch, int rows, int K, int N, int Cin, int H, int W);
int hecka_conv_x3_f32(const float* A, const float* X, float* Y, const int* tab, int batch, int rows, int K,
                      long long strideA, long long strideX, int N, int Cin, int H, int W, float alpha, const float* amaxA,
                      const float* amaxB, int amaxB_stride, void* stream);

/* Weight gradient of the same convolutions with BOTH operands implicit (no patch matrix, no [O,Q] copy of dy):
 *   dW[b][o][r] = alpha * sum_{n,h,w} dY[b][n][o][h][w] * X[b][n][c_r][h + di_r][w + dj_r]
 * dY [batch,N,O,H,W], X [batch | shared,N,C,H,W], dW [batch,O,Rp] with Rp = C kh kw rounded up to a multiple of 128; tab
 * [Rp][2] as for hecka_conv_x3_f32 with rows r >= C kh kw marked never-in-range (di = -30000).  Split over K = N H W with a
 * fixed-order second stage: ws >= hecka_conv_wgrad_x3_workspace_bytes(...).  W % 16 == 0 or 16 % W == 0, O % 64 == 0. */
int hecka_conv_wgrad_x3_eligible(int batch, int O, int Rp, int N, int C, int H, int W);
size_t hecka_conv_wgrad_x3_workspace_bytes(int batch, int O, int Rp, int N, int C, int H, int W);
int hecka_conv_wgrad_x3_f32(const float* dY, const float* X, float* dW, const int* tab, int batch, int O, int Rp,
                            long long strideX, int N, int C, int H, int W, float alpha, const float* amaxA,
                            const float* amaxB, int amaxB_stride, void* ws, size_t ws_bytes, void* stream);

/* The two kernels above for ANY stride and zero padding (no dilation) - the stage transitions of a ResNet, 3 x 3 / stride 2
 * and the 1 x 1 / stride 2 shortcut (hyper/target/resnet.py:23-75; hyper/layers/conv.py:292-346): an OUTPUT position (ho, wo)
 * gathers its taps around input position (ho sh, wo sw); tab as above with di = i - ph, dj = j - pw.  Y / dY
 * [batch,N,rows | O,Ho,Wo], X [batch | shared,N,Cin,H,W].  N Ho Wo % 128 == 0 (forward), Wo % 16 == 0 or 16 % Wo == 0 (weight
 * gradient).  The unstrided entry points are these with Ho = H, Wo = W, sh = sw = 1.  (The INPUT gradient of a stride-2
 * convolution is hecka_conv_x3_placed_f32 below; other strides keep the GEMM + hecka_col2im_f32.) */
int hecka_conv_x3_strided_eligible(int batch, int rows, int K, int N, int Cin, int H, int W, int Ho, int Wo);
int hecka_conv_x3_strided_f32(const float* A, const float* X, float* Y, const int* tab, int batch, int rows, int K,
                              long long strideA, long long strideX, int N, int Cin, int H, int W, int Ho, int Wo, int sh, int sw,
                              float alpha, const float* amaxA, const float* amaxB, int amaxB_stride, void* stream);
/* hecka_conv_x3_strided_f32 whose (Ho x Wo) result grid is PLACED into a larger output image Y [batch,N,rows,OH,OW] at
 * (u esh + eoh, v esw + eow).  The input gradient of a stride-2 convolution is four such launches, one per parity class
 * (a, b) of the input pixels: dx[2u + a][2v + b] = sum over the taps i = a + ph, j = b + pw (mod 2) of the rotated weights
 * against dy[u + (a + ph - i) / 2][v + (b + pw - j) / 2] - a unit-stride convolution of dy with that class's taps, no
 * multiply by a structural zero, no patch matrix, no col2im. */
int hecka_conv_x3_placed_f32(const float* A, const float* X, float* Y, const int* tab, int batch, int rows, int K,
                             long long strideA, long long strideX, int N, int Cin, int H, int W, int Ho, int Wo, int sh, int sw,
                             int OH, int OW, int esh, int esw, int eoh, int eow, float alpha, const float* amaxA,
                             const float* amaxB, int amaxB_stride, void* stream);
/* hecka_conv_x3_placed_f32 with the K axis ordered (tap, channel) instead of (channel, tap): A [batch,rows,T Cin] with column
 * (t, c), Cin % 16 == 0, so a k-tile of 16 holds 16 channels of ONE tap and whether the tap falls inside the image is decided
 * once per k-tile and thread (the (c, i, j) order pays a table lookup and a range check per gathered element: the
 * gather, not the MFMAs, bounded those launches).  tab [T Cin / 16 + 1][4] int32 per k-tile: { byte offset of the tile's
 * first channel = c0 H W 4, byte offset of the tap = (di W + dj) 4, di, dj }; the last entry is padding (the kernel fetches one
 * entry ahead), 16-byte aligned.  Everything else as hecka_conv_x3_placed_f32 (dense output: OH = Ho, OW = Wo, 1, 1, 0, 0). */
int hecka_conv_x3_tapmajor_f32(const float* A, const float* X, float* Y, const int* tab, int batch, int rows, int K,
                               long long strideA, long long strideX, int N, int Cin, int H, int W, int Ho, int Wo, int sh, int sw,
                               int OH, int OW, int esh, int esw, int eoh, int eow, float alpha, const float* amaxA,
                               const float* amaxB, int amaxB_stride, void* stream);
/* The weight operands of hecka_conv_x3_tapmajor_f32 from W [M,O,C,T] (T = kh kw; the parameter layout of hyper/layers/conv.py:
 * 434-502): rot = 0 -> out [M,O,T,C] = W[m][o][c][t] (forward); rot = 1 -> out [M,C,T,O] = W[m][o][c][T-1-t] (input gradient:
 * channels swapped, taps rotated by 180 degrees).  C T <= 16384, out != W. */
int hecka_conv_weights_tapmajor_f32(const float* W, int M, int O, int C, int T, int rot, float* out, void* stream);
int hecka_conv_wgrad_x3_strided_eligible(int batch, int O, int Rp, int N, int C, int H, int W, int Ho, int Wo);
size_t hecka_conv_wgrad_x3_strided_workspace_bytes(int batch, int O, int Rp, int N, int C, int H, int W, int Ho, int Wo);
int hecka_conv_wgrad_x3_strided_f32(const float* dY, const float* X, float* dW, const int* tab, int batch, int O, int Rp,
                                    long long strideX, int N, int C, int H, int W, int Ho, int Wo, int sh, int sw, float alpha,
                                    const float* amaxA, const float* amaxB, int amaxB_stride, void* ws, size_t ws_bytes,
                                    void* stream);

/* Batched-weight Conv2d as grouped GEMMs (hyper/layers/conv.py:434-502 loops F.conv2d over the
 * members).  hecka_im2col_f32 gathers patches of x [Mx,N,C,H,W] (Mx = members, or 1 for an input shared by all
 * members) into cols [Mx,R,Q], R = C*kh*kw (+1 row of ones when ones_row: the bias then rides through the GEMM as an
 * extra weight column), Q = N*Ho*Wo, row r = (c*kh+i)*kw+j; zero padding.  Forward y[m] [O,Q] = W'[m] [O,R] cols[m],
 * wgrad dW'[m] = dy[m] cols[m]^T, dgrad dcols[m] = W[m]^T dy[m] are hecka_bgemm_f32 calls; hecka_col2im_f32 gathers
 * dcols [M,rows,Q] (rows >= C*kh*kw; extra rows ignored) back into dx [M,N,C,H,W] (deterministic, no atomics). */
int hecka_im2col_f32(const float* x, int Mx, int N, int C, int H, int W, int kh, int kw, int sh, int sw, int ph, int pw,
                     int dh, int dw, int ones_row, float* cols, void* stream);
int hecka_col2im_f32(const float* dcols, int M, int N, int C, int H, int W, int kh, int kw, int sh, int sw, int ph,
                     int pw, int dh, int dw, int rows, float* dx, void* stream);

/* Batched BatchNorm (hyper/layers/norm.py:143-224 loops F.batch_norm over the members).
 * x, y [G,N,C,S] (G = members, S = H*W or 1), weight / bias [G,C] (may be NULL), statistics per (member, channel).
 * training: batch statistics (biased variance) normalise; save_mean / save_rstd [G,C] are kept for backward and
 *   batch_var [G,C] (may be NULL) returns the biased variance for the caller's running-statistics update.
 * eval (training = 0): run_mean / run_var [G,C] normalise.
 * bwd: dx (may be NULL), dweight = sum dy xhat, dbias = sum dy ([G,C], may be NULL). */
int hecka_batchnorm_fwd_f32(const float* x, int G, int N, int C, long long S, const float* weight, const float* bias,
                            const float* run_mean, const float* run_var, int training, float eps, float* y,
                            float* save_mean, float* save_rstd, float* batch_var, void* stream);
int hecka_batchnorm_bwd_f32(const float* dy, const float* x, const float* save_mean, const float* save_rstd,
                            const float* weight, int G, int N, int C, long long S, int training, float* dx,
                            float* dweight, float* dbias, void* stream);

/* Batched BatchNorm fused with what follows it in a ResNet block (hyper/target/resnet.py:67-74: norm -> relu, and
 * norm -> + shortcut -> relu): y = [relu](batch_norm(x) [+ res]).  Backward takes the forward OUTPUT y as the relu mask
 * (NULL: no relu) and returns the masked gradient for the residual branch in dres (may be NULL).  S % 4 == 0.
 * ymax / dxmax [G] (may be NULL): max |y| / max |dx| per group as a by-product - the fp16-split convolution that reads the
 * tensor next needs exactly that as its operand scale, and would otherwise spend a pass over the tensor on it. */
int hecka_batchnorm_act_fwd_f32(const float* x, const float* res, int G, int N, int C, long long S, const float* weight,
                                const float* bias, const float* run_mean, const float* run_var, int training, float eps,
                                int relu, float* y, float* save_mean, float* save_rstd, float* batch_var, float* ymax,
                                void* stream);
int hecka_batchnorm_act_bwd_f32(const float* dy, const float* x, const float* y, const float* save_mean,
                                const float* save_rstd, const float* weight, int G, int N, int C, long long S,
                                int training, float* dx, float* dres, float* dweight, float* dbias, float* dxmax,
                                void* stream);

/* The same with the affine parameters read where the batched layers keep them - weight and bias of member g at
 * weight + g wstride and bias + g wstride (ONE flat [G, 2 C] row per member: bias = weight + C, wstride = 2 C; dweight /
 * dbias likewise, so their gradients return as one tensor and no copy of either half is made) - and with the running
 * statistics [G,C] updated by the kernel itself in training (run_mean_upd / run_var_upd, may be NULL):
 * rm = (1 - factor) rm + factor mean, rv = (1 - factor) rv + factor var n / (n - 1), n = N S (hyper/layers/norm.py:209-222
 * leaves that to F.batch_norm per member; here it was a dozen [G,C]-sized torch kernels per norm layer, 4 us each).
 * mask (may be NULL; G N C S / 4 bytes): the forward writes one byte per four outputs, bit e = y[4 i + e] > 0; the backward,
 * given it, does not read y (all it needs of y is the relu's sign: 1 / 16 of the bytes in each of its two sweeps). */
int hecka_batchnorm_act_fwd_packed_f32(const float* x, const float* res, int G, int N, int C, long long S, const float* weight,
                                       const float* bias, int wstride, const float* run_mean, const float* run_var,
                                       int training, float eps, int relu, float* y, float* save_mean, float* save_rstd,
                                       float* batch_var, float* ymax, float* run_mean_upd, float* run_var_upd, float factor,
                                       unsigned char* mask, void* stream);
int hecka_batchnorm_act_bwd_packed_f32(const float* dy, const float* x, const float* y, const float* save_mean,
                                       const float* save_rstd, const float* weight, int wstride, int G, int N, int C,
                                       long long S, int training, float* dx, float* dres, float* dweight, float* dbias,
                                       float* dxmax, const unsigned char* mask, void* stream);

/* Direct convolution for FEW channels (hyper/layers/conv.py:492-502 loops F.conv2d over the members; LeNet-5's convs are
 * 1 -> 6 and 6 -> 16 channels, 5x5): unit stride, no dilation, zero padding `pad` on both axes, C <= 16, O <= 16, kernel
 * 5x5 or 3x3 (hecka_conv_direct_eligible).  No patch matrix, y written as [M,N,O,Ho,Wo].
 *   x  [M|1, N, C, H, W]  (x_member_stride = N*C*H*W, or 0 when the members share the input)
 *   wt [M, C*KH*KW, OT]   weights of tap (c,i,j) for all output channels, OT = 8 if O <= 8 else 16 (zero padded), already
 *                         multiplied by the layer's weight scale
 *   y  [M, N, O, Ho, Wo] = conv + bias_scale * bias[m,o]  (bias may be NULL)
 * The input gradient is the same call on dy with wt built from the rotated weights ([M, O*KH*KW, OT >= C]) and
 * pad' = KH - 1 - pad.
 * wgrad: dw_partial [splits, M, O, C, KH*KW], db_partial [splits, M, O] (may be NULL) - partial sums over `splits` groups
 * of samples, to be summed in order by the caller (deterministic). */
int hecka_conv_direct_eligible(int C, int O, int KH, int KW);
int hecka_conv_direct_f32(const float* x, long long x_member_stride, const float* wt, const float* bias, float bias_scale,
                          int M, int N, int C, int H, int W, int O, int KH, int KW, int pad, float* y, void* stream);
int hecka_conv_direct_wgrad_f32(const float* x, long long x_member_stride, const float* dy, int M, int N, int C, int H, int W,
                                int O, int KH, int KW, int pad, int splits, float* dw_partial, float* db_partial,
                                void* stream);

/* y [Wd] = W^T u for a row-major W [H, Wd]: the power-iteration step v <- normalize(W^T u) of the hypernetwork's spectral
 * normalisation (hyper/net/spectral.py:92, `torch.mv(weight_mat.t(), u)`), HBM-bound, deterministic (row splits summed in
 * fixed order).  ws: hecka_gemv_t_workspace_bytes(H, Wd) bytes of device scratch. */
size_t hecka_gemv_t_workspace_bytes(int H, int Wd);
int hecka_gemv_t_f32(const float* W, const float* u, int H, int Wd, float* y, void* ws, size_t ws_bytes, void* stream);

/* ---- hypernetwork sampler side: shared-weight GroupNorm-MLPs over the sampled members ---------------
 * The Linear layers of those MLPs (hyper/net/models.py:60-127 `build_norm_block`: nn.Linear / spectral_norm_fc) are
 * hecka_bgemm_f32 calls with batch = 1 (forward nn = 0 with the bias epilogue, dgrad nn = 1, wgrad on transposed operands).
 * hecka_gemv_f32: y [H] = W v for a row-major W [H, Wd] - the power-iteration step u <- normalize(W v) and
 *   sigma = u . (W v) of the soft spectral normalisation (hyper/net/spectral.py:92-99 `torch.mv(weight_mat, v)`); one wave
 *   per row, deterministic.  (W^T u is hecka_gemv_t_f32 below.)
 * hecka_ger_f32: out [H, Wd] = alpha[0] u v^T - autograd's gradient of sigma with respect to the weight (u, v constants),
 *   alpha = device scalar dLoss/dsigma.
 * hecka_groupnorm_*: nn.GroupNorm on [N, C] rows (hyper/net/models.py:99-101, gpnorm.py:9-31; statistics per (row, group),
 *   biased variance) with the nn.GELU() that follows it in a norm block fused in (act = 1; 0 = none): forward keeps
 *   mean / rstd [N, G]; backward recomputes the activation's argument from x.  weight / bias [C] may be NULL (affine off);
 *   dx / dweight / dbias may be NULL. */
int hecka_gemv_f32(const float* W, const float* v, int H, int Wd, float* y, void* stream);
int hecka_ger_f32(const float* alpha, const float* u, const float* v, int H, int Wd, float* out, void* stream);
int hecka_groupnorm_fwd_f32(const float* x, int N, int C, int G, const float* weight, const float* bias, float eps, int act,
                            float* y, float* mean, float* rstd, void* stream);
int hecka_groupnorm_bwd_f32(const float* dy, const float* x, int N, int C, int G, const float* weight, const float* bias,
                            const float* mean, const float* rstd, int act, float* dx, float* dweight, float* dbias,
                            void* stream);

/* crater activation y = c x (1 + tanh((x + 1) / sqrt 2)) of the target nets (hyper/net/activation.py:8-12; the reference
 * composes it from five elementwise torch ops, each a pass over the tensor).  dy == NULL: out = crater(x); otherwise
 * out = dy * crater'(x).  n elements, 16-byte aligned. */
int hecka_crater_f32(const float* x, const float* dy, float* out, long long n, void* stream);
/* crater followed by the 2x2 / stride-2 max pool of a batched Conv2d (hyper/layers/conv.py:341-346: act -> pooling) in
 * one pass: x [R, H, W] (R = members * samples * channels).  dy == NULL: out [R, H/2, W/2] = maxpool2(crater(x));
 * otherwise out [R, H, W] = gradient w.r.t. x given dy [R, H/2, W/2] (the activated tensor is never written; ties keep
 * the first element of the window in row-major order, as torch.max_pool2d).  H % 2 == 0, W % 4 == 0. */
int hecka_crater_maxpool2_f32(const float* x, const float* dy, float* out, long long R, int H, int W, void* stream);

/* ---- 16-bit features (BASELINE config 3: bf16 networks) ------------------------------------------------------------
 * The reference runs the same Python path on bf16 feature tensors (torch.bmm in bf16, hyper/diversity/kernels.py:122);
 * here the three kernels that touch X read bf16 / fp16 features as they lie in HBM (half the bytes, no up-cast pass),
 * widen them to fp32 in registers and compute exactly as the fp32 entry points do; C, cvec, S, W stay fp32.
 * dtype: HECKA_DT_F32 forwards to the fp32 entry point.  Interior shapes only (B % 128 == 0; P % 16 == 0 for the Gram,
 * P % 128 == 0 for dX), HECKA_GEMM_AUTO, xmax / wmax required - otherwise HECKA_EINVAL: up-cast and use the fp32 calls.
 * hecka_gram_bwd_x16: 16-bit X times the power-of-two scale is exact in ONE fp16 term, so the product list is
 * W.h X + W.l X (two MFMA products per multiply); dX is written in the feature type (rounded once, to nearest). */
#define HECKA_DT_F32 0
#define HECKA_DT_BF16 1
#define HECKA_DT_F16 2
int hecka_colmean_x16(const void* X, int dtype, int M, int B, int P, float* sbar, float* xmax, void* stream);
/* ws: hecka_gram_center_multi_workspace_bytes(L, M, B, P) bytes (0 = none needed; NULL = never split).  With few
 * workgroups over deep layers (small ensembles, a member shard of a multi-GPU run) the launch splits layers over K into
 * partial Grams in ws and sums them in a fixed order - every stage of the epilogue is linear in the k range.
 * slice_rows (hecka_gram_center_multi_x16, fp32 features): 0 = C[l] is [M,B,B]; a power of two dividing B = C[l] is
 * written as [B / slice_rows][M][slice_rows][B], the send order of the member-sharded all-to-all (hyper_amd/parallel.py)
 * - no repacking copy between the Gram and the exchange; hecka_gram_bwd_multi_f32 reads W in the same order.
 * slice_members (with slice_rows; 0 = M): member rows per slice block.  With slice_members = L * M and C[l] = base +
 * l * M * slice_rows * B the L layers share ONE buffer [B / slice_rows][L][M][slice_rows][B]: every layer's rows for one
 * destination rank are contiguous, so the member-sharded exchange is a single all_to_all_single for all layers. */
size_t hecka_gram_center_multi_workspace_bytes(int L, int M, int B, const int* P);
int hecka_gram_center_multi_x16(int L, const void* const* X, int dtype, const float* const* sbar,
                                const float* const* xmax, int M, int B, const int* P, float* const* C,
                                float* const* cvec, void* ws, size_t ws_bytes, int slice_rows, int slice_members, int gemm_mode,
                                void* stream);
int hecka_gram_bwd_x16(const float* W, const void* X, int dtype, const float* wmax, const float* xmax, const float* gate,
                       int M, int B, int P, void* dX, int gemm_mode, void* stream);

/* dX[m] = W[m] X[m]   (W [M,B,B] symmetric, X [M,B,P]) — backward of torch.bmm(X, X^T)
 * (hyper/diversity/kernels.py:122).  wmax / xmax (may be NULL): [M] maxima from hecka_combine_f32 /
 * hecka_colmean_f32, or upper bounds of them (hecka_wbound_f32); both given enables f16x3 (see HECKA_GEMM_AUTO).
 * gate (may be NULL): device scalar; when *gate == 0 the call writes dX = 0 and reads neither W nor X.  RepulsiveKernel
 * drops a NaN repulsion term from the loss AND from the graph (hyper/diversity/methods.py:342-348: `if rep.isnan()` on the
 * host); here the caller sets *gate = 0 on the device instead, so that no 0 * NaN product reaches the feature gradients. */
int hecka_gram_bwd_f32(const float* W, const float* X, const float* wmax, const float* xmax, const float* gate, int M, int B,
                       int P, float* dX, int gemm_mode, void* stream);

/* hecka_gram_bwd_f32 for L tracked layers (same M and B, their own P) in ONE launch, as hecka_gram_center_multi_f32 is for
 * the forward: W / X / wmax / xmax / dX are HOST arrays of L device pointers, P a host array of widths (deepest first fills
 * the GPU best).  Replaces the per-layer backward of ModelLayersKernel.forward's loop (hyper/diversity/model_kernels.py:
 * 446-478): a layer alone leaves the tail of its last scheduling round idle, and a member shard's small layers do not
 * fill the GPU.  f16x3 only (HECKA_GEMM_AUTO, wmax and xmax required), B <= 1024, B % 4 == 0; ragged B / P as in
 * hecka_gram_bwd_f32.  slice_rows: 0, or a power of two dividing B - W then lies as the member-sharded all-to-all
 * delivers it, [B / slice_rows][M][slice_rows][B] (no repacking copy); slice_members as for hecka_gram_center_multi_x16
 * (the L layers of one exchange in ONE buffer [B / slice_rows][L][M][slice_rows][B], W[l] = base + l * M * slice_rows * B).
 * hecka_gram_bwd_multi_eligible says whether the call applies; callers use per-layer hecka_gram_bwd_f32 otherwise. */
int hecka_gram_bwd_multi_eligible(int L, int M, int B, const int* P, int slice_rows);
int hecka_gram_bwd_multi_f32(int L, const float* const* W, const float* const* X, const float* const* wmax,
                             const float* const* xmax, const float* gate, int M, int B, const int* P, float* const* dX,
                             int slice_rows, int slice_members, int gemm_mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HECKA_H */
