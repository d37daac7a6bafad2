/*
 * tfk.h -- C-ABI of libtfk.so: MI355X (gfx950) kernels for the torchflows
 * coupling-flow hot path (Bijection.forward / inverse with running log|det J|).
 *
 * The reference (davidnabergoj/torchflows v1.2.0) is pure Python over ATen and has
 * no FFI of its own; each entry point below replaces the chain of ATen ops issued by
 * the reference lines it cites (paths relative to the reference's torchflows/).
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - Every function returns 0 on success, a TFK_E* code otherwise, and never throws
 *    across the boundary; tfk_last_error() gives the text (thread-local).
 *  - All data pointers are CALLER-OWNED DEVICE pointers (hipMalloc'ed / torch
 *    tensor.data_ptr()); the library never allocates, frees or synchronises.
 *  - Work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default
 *    stream): calls are asynchronous and safe to issue from several host threads on
 *    different streams.
 *  - Tensors are fp32, contiguous, row-major: x/z are (N, D) with D = prod(event
 *    shape); log-dets are (N,).  N is int64, sizes inside a row are int32.
 *  - `accumulate` != 0: logdet[n] += this layer's log-det (the running sum of
 *    BijectiveComposition, bijections/base.py:210-222); 0: logdet[n] is overwritten.
 *  - `tgt_idx` (device int32[T], ascending flat event indices, the True positions of
 *    coupling.target_mask) may be NULL, meaning the contiguous tail [D-T, D) -- the
 *    HalfSplit mask (conditioning/coupling_masks.py:78-81).
 *  - z may alias x (in place): then only target positions are written.  Otherwise
 *    every element of z is written (z = x.clone() then z[..., target] = ...,
 *    layers_base.py:146-152).
 *  - N == 0 is a successful no-op.
 */
#ifndef TFK_H
#define TFK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFK_ABI_VERSION 29

enum {
    TFK_OK = 0,
    TFK_EINVAL = 1,   /* bad argument (null pointer, size, alignment contract) */
    TFK_ELAUNCH = 2,  /* HIP reported an error at launch */
    TFK_ENODEV = 3    /* no usable gfx950 device */
};

int tfk_abi_version(void);
const char *tfk_last_error(void);

/* Name of the device the calling thread is on + its CU count; TFK_ENODEV if none. */
int tfk_device_info(char *name, int32_t name_len, int32_t *compute_units);

/* ---- affine coupling ------------------------------------------------------
 * Replaces CouplingBijection.forward / inverse (layers_base.py:145-163) around
 * Affine.forward / inverse (transformers/linear/affine.py:33-59):
 *   u = h[n,t,0], beta = h[n,t,1]; alpha = expf(u/2 + log(1-1e-10)) + 1e-10;
 *   fwd: z_t = alpha*x_t + beta, ld = +sum_t logf(alpha)
 *   inv: x_t = (z_t - beta)/alpha, ld = -sum_t logf(alpha)
 * h is the conditioner output (N, T, 2), exactly the reference's layout. */
int tfk_affine_coupling_fwd(const float *x, const float *h, float *z, float *logdet,
                            int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T,
                            int32_t accumulate, void *stream);
int tfk_affine_coupling_inv(const float *z, const float *h, float *x, float *logdet,
                            int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T,
                            int32_t accumulate, void *stream);

/* ---- shift coupling (NICE) -------------------------------------------------
 * Shift.forward / inverse (affine.py:137-159): z_t = x_t +/- h[n,t,0]; log-det 0
 * (logdet is only zero-filled when accumulate == 0; it may be NULL otherwise). */
int tfk_shift_coupling_fwd(const float *x, const float *h, float *z, float *logdet,
                           int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T,
                           int32_t accumulate, void *stream);
int tfk_shift_coupling_inv(const float *z, const float *h, float *x, float *logdet,
                           int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T,
                           int32_t accumulate, void *stream);

/* ---- rational-quadratic spline coupling ------------------------------------
 * Replaces the same skeleton around MonotonicSpline.forward / inverse
 * (transformers/spline/base.py:53-72) and RationalQuadratic.rqs_forward_1d /
 * rqs_inverse_1d (transformers/spline/rational_quadratic.py:45-200).
 * h is (N, T, 3K-1) = [u_x(K) | u_y(K) | u_d(K-1)] per element; K = n_bins
 * (2 <= K <= 32); elements outside the strict box (-boundary, boundary) pass through
 * with zero log-det.  No host synchronisation (the reference's torch.any / assert
 * syncs are not reproduced; out-of-box handling is branch-free per element). */
int tfk_rqs_coupling_fwd(const float *x, const float *h, float *z, float *logdet,
                         int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T,
                         int32_t K, float boundary, int32_t accumulate, void *stream);
int tfk_rqs_coupling_inv(const float *z, const float *h, float *x, float *logdet,
                         int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T,
                         int32_t K, float boundary, int32_t accumulate, void *stream);

/* ---- invertible 1x1 convolution coupling (Glow) -------------------------------
 * The same skeleton around Invertible1x1ConvolutionTransformer
 * (transformers/linear/convolution.py:8-70) + LUTransformer (transformers/linear/matrix.py:
 * 11-99).  The T target positions are an image of n_channels x (T / n_channels) pixels,
 * channel-major; h is (N, n + n(n-1)) per SAMPLE = [diag logits | U above the diagonal
 * (triu row-major) | L below the diagonal (tril row-major)], U_ii = exp(h_i)/10 + 1,
 * off-diagonals h/10, unit-diagonal L.  fwd: y = L U x per pixel, log-det = sum log U_ii
 * once per sample (the reference does not scale it by the pixel count).  n_channels <= 16. */
int tfk_conv1x1_coupling_fwd(const float *x, const float *h, float *z, float *logdet,
                             int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T,
                             int32_t n_channels, int32_t accumulate, void *stream);
int tfk_conv1x1_coupling_inv(const float *z, const float *h, float *x, float *logdet,
                             int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T,
                             int32_t n_channels, int32_t accumulate, void *stream);
/* Reverse mode of the two entry points above (the reference differentiates its ATen graph, convolution.py:33-64):
 * x = the layer's INPUT rows, g (N, D) holds dL/d(out rows) on entry and dL/d(in rows) on return (only the T target
 * positions change: the other positions pass through the coupling), gld (N) = dL/d(logdet), gh (N, n + n(n-1)) is
 * OVERWRITTEN with dL/dh; one workgroup per sample, the sum over the pixels in a fixed order (deterministic).
 * inverse != 0: gradients of tfk_conv1x1_coupling_inv. */
int tfk_conv1x1_coupling_bwd(const float *x, const float *h, float *g, const float *gld, float *gh, int64_t N,
                             int32_t D, const int32_t *tgt_idx, int32_t T, int32_t n_channels, int32_t inverse,
                             void *stream);

/* ---- elementwise affine (ElementwiseAffine, ActNorm) -------------------------
 * Replaces ElementwiseBijection.forward / inverse (layers_base.py:300-318) with
 * global parameters value (D, 2) = [unconstrained alpha, beta] per element; the
 * reference's repeat of value to (N, D, 2) is not materialised.
 * inverse_affine == 0: Affine transformer (ElementwiseAffine, layers.py:19-26);
 * inverse_affine != 0: InverseAffine (ActNorm in eval mode, layers.py:29-69), i.e.
 *   _fwd computes (x - beta)/alpha with ld = -sum log alpha, _inv the affine map.
 * The log-det is the same for every row; it is added to / stored in logdet[n]. */
int tfk_elementwise_affine_fwd(const float *x, const float *value, float *z, float *logdet,
                               int64_t N, int32_t D, int32_t inverse_affine,
                               int32_t accumulate, void *stream);
int tfk_elementwise_affine_inv(const float *z, const float *value, float *x, float *logdet,
                               int64_t N, int32_t D, int32_t inverse_affine,
                               int32_t accumulate, void *stream);

/* ---- permutation -------------------------------------------------------------
 * PermutationMatrix.project_flat / solve_flat (matrix/permutation.py:19-23):
 * z[n, j] = x[n, perm[j]].  perm is a device int32[D]; NULL means the reversal
 * [D-1..0] (ReversePermutationMatrix, permutation.py:34-37).  Pass the inverse
 * permutation for Bijection.inverse.  Log-det is exactly 0 (not touched).
 * z must not alias x. */
int tfk_permute(const float *x, const int32_t *perm, float *z, int64_t N, int32_t D,
                void *stream);

/* ---- base density --------------------------------------------------------------
 * DiagonalGaussian.log_prob (base_distributions/gaussian.py:46-54) fused with the
 * final add of Flow.forward_with_log_prob (flows.py:647-648):
 *   out[n] = sum_d -(0.5*((z-loc)/exp(log_scale))^2 + 0.5*log(2pi) + log_scale)
 *            + (logdet_in ? logdet_in[n] : 0)
 * out may alias logdet_in (for D > 5461 -- column-tiled -- an aliased log-det is added with the first tile
 * instead of after the last). */
int tfk_diag_gauss_logprob(const float *z, const float *loc, const float *log_scale,
                           const float *logdet_in, float *out, int64_t N, int32_t D,
                           void *stream);

/* ---- reduction feeding the multi-GPU all-reduce --------------------------------
 * out_scalar[0] = sum_n in[n] accumulated in fp64 (the partial sum of log-likelihoods a rank contributes to the
 * single all-reduce; fp64 so that the 2^22-term sum of config 4 stays within 1e-5).  Both forms are deterministic
 * (fixed trees) and neither allocates:
 *   tfk_sum_f32     SURVEY.md 8(b)'s four-argument form: one workgroup, no scratch memory -- for short vectors;
 *   tfk_sum_f32_ws  two stages over the whole chip with a caller-owned device workspace of at least
 *                   tfk_sum_workspace_bytes(N) bytes -- what Flow / bench.py use at N >= 2^16. */
int tfk_sum_f32(const float *in, double *out_scalar, int64_t N, void *stream);
int64_t tfk_sum_workspace_bytes(int64_t N);
int tfk_sum_f32_ws(const float *in, double *out_scalar, void *workspace, int64_t N, void *stream);

/* ---- fused flow program (conditioner in-kernel, layers folded) -----------------
 * Runs a chain of layers on rows held in registers: ONE launch replaces the Python loop of
 * BijectiveComposition.forward / inverse (bijections/base.py:203-232) over
 * ElementwiseAffine / ActNorm (layers.py:19-69), ReversePermutationMatrix
 * (matrix/permutation.py:34-37, folded into the weight order) and Affine / Shift couplings
 * (layers.py:102-139) INCLUDING their FeedForward(tanh) conditioner
 * (conditioning/transforms.py:274-307): h never exists in HBM.
 * Supported: D a power of two in [16, 512], HalfSplit mask, no context.
 *
 * ops (HOST pointer): n_ops x int32[8] = {kind, src_plane, H, param_offset, K,
 *   boundary, scale, c} -- the last three are fp32 bit patterns and only used by RQS ops
 *   (K = n_bins = 8, boundary = spline box half-width, scale = 1 - 1e-3*K,
 *   c = log(expm1(1 - 1e-5)), rational_quadratic.py:20,36-38);
 * params (device, 16-byte aligned, n_params % 4 == 0) holds, at param_offset (floats,
 * multiple of 4), in PHYSICAL element order (plane A = positions [0, D/2), B = the rest):
 *   TFK_OP_EW_MULADD : alpha[D] | beta[D] | logdet_const | pad[3]
 *        z = alpha*x + beta;   logdet += logdet_const
 *   TFK_OP_EW_SUBDIV : alpha[D] | beta[D] | logdet_const | pad[3] | 1/alpha[D]
 *        z = (x - beta)/alpha (evaluated with the packed reciprocal + one residual
 *        correction: the IEEE quotient except on near-ties);   logdet += logdet_const
 *   coupling (affine P=2 / shift P=1)   : W1t[H][D/2] | b1[H padded to 4] | W2t[H][D/2*P] | b2[D/2*P]
 *        src_plane = which plane feeds the conditioner (the other one is transformed);
 *        W1t[k][m] multiplies physical source element m; W2t[k][m*P + p], b2[m*P + p]
 *        produce parameter p of physical target element m.
 *   RQS coupling (TFK_OP_RQS_FWD / _INV, layers.py:154-163 + spline/rational_quadratic.py):
 *        W1t[H][D/2] | b1[H padded to 4] | W2t[H][4][D/8][24] | b2[4][D/8][24]
 *        target element m = 4*j + e lives at [e][j]; its 23 parameters [u_x(8) | u_y(8) | u_d(7)]
 *        are padded to 24 floats; H <= 32.
 * Outputs (each may be NULL, at least one must not): z rows (physical order), logdet
 * (accumulate as elsewhere), logprob[n] = diag-Gaussian log-density of the final rows
 * (gauss_loc / gauss_log_scale given in physical order) + the chain's log-det. */
enum {
    TFK_OP_EW_MULADD = 0,
    TFK_OP_EW_SUBDIV = 1,
    TFK_OP_AFFINE_FWD = 2,
    TFK_OP_AFFINE_INV = 3,
    TFK_OP_SHIFT_FWD = 4,
    TFK_OP_SHIFT_INV = 5,
    TFK_OP_RQS_FWD = 6,
    TFK_OP_RQS_INV = 7,
    TFK_OP_MADE_FWD = 8,   /* MADE + Affine on the whole row, parallel map (tfk_flow_run_mfma only) */
    TFK_OP_MADE_INV = 9,   /* same with (x - beta) / alpha */
    TFK_OP_MADE_RQS = 10,  /* MADE + RQ spline (8 bins) on the whole row, parallel map; hidden <= 16, D <= 128 */
    TFK_OP_PLANE_SWAP = 11, /* tfk_flow_run_mfma: mask[D/2] floats, != 0 exchanges x[i] and x[D/2 + i] (D <= 128; odd event sizes,
                              padded so that every element keeps its index in both halves) */
    /* "lean" programs of tfk_flow_run_mfma (ABI v20): a chain of couplings of ONE kind and one hidden width (<= 16)
     * whose source plane alternates, optionally ended by one TFK_OP_EW_FMA; they run on a straight-line kernel
     * (csrc/tfk_flow_chain.h) and cannot be mixed with the kinds above.  The elementwise layers between the
     * couplings are folded by the packer (torchflows_amd/fused.py): into W1 / b1 where an element feeds the
     * conditioner, into the coupling's pre-affine (s, t) where it is transformed -- x_t <- fma(s, x_t, t) first.
     * Block: A1[D/32][64][4] | b1[4][4] | A2[nA2/4][64][4] | b2[T2][4][4] | pre_s[D/2] | pre_t[D/2], lane-major
     * A-operands (nA2 = T2 * gemm2_steps rounded up to 4; entry t * gemm2_steps + k = tile t, step k), with W1 / b1
     * multiplied by 2 log2(e) and the scale-logit rows of W2 / b2 by log2(e) / 2 (b2 += log(1 - 1e-10) log2(e)):
     * tanh = 1 - 2 / (exp2(.) + 1), alpha = exp2(.) + 1e-10, log-det accumulated in base 2.
     * Op record field K = 256 (D = 64 only): GEMM 2 in the bf16 x 3 operand format described at TFK_OP_RQS_*_LEAN --
     * block A1 | b1 | A23[T2][2][64][4 dwords] | pre_s | pre_t, A23[t] = {[W_hi | W_mid], [W_lo | W_hi]}, b2 as the
     * weight of hidden unit 15 (hidden width <= 15).
     * Bit 2 of src_plane (ABI v27, odd event sizes): HalfSplit then has one target more than sources and -- with the
     * reversals between the couplings -- the MIDDLE element is a target of EVERY coupling, so it has to sit in whichever
     * plane is being transformed: both planes reserve their last column (D/2 - 1 of the plane) for it, and a coupling whose
     * src_plane carries bit 2 first takes the element over from the other plane's last column and clears that one.
     * Resident operands only (no streaming), no context.  The lean spline couplings (TFK_OP_RQS_*_LEAN, TFK_OP_LRS_*_LEAN)
     * take the same bit. */
    TFK_OP_AFFINE_FWD_LEAN = 12,
    TFK_OP_AFFINE_INV_LEAN = 13,
    TFK_OP_SHIFT_FWD_LEAN = 14,
    TFK_OP_SHIFT_INV_LEAN = 15,
    TFK_OP_EW_FMA = 16,     /* s[D] | t[D] | logdet_const | pad[3]:  z = fma(s, x, t), logdet += logdet_const */
    /* lean SPLINE programs (ABI v20): a chain of RQ-spline couplings (8 bins) of one direction, one hidden width
     * (<= 16) and one spline box, source plane alternating, parameter blocks at a constant stride, optionally ended
     * by one TFK_OP_EW_FMA: ONE launch for the whole chain (csrc/tfk_flow_rqs_chain.h) -- the rows stay in registers
     * and the operands are streamed per layer from `params` (global memory: a chain does not have to fit the LDS).
     * Block: A1[D/32][64][4] | b1[4][4] | pre_s[D/2] | pre_t[D/2], then D/64 chunks of A2[48][64][4] | b2[48][4][4]
     * (tile 6 e + c of a chunk = parameters 4 c .. 4 c + 3 of the lane-group's target element 8 chunk + e, lane-major
     * over <= 4 k-steps).  Per element 24 parameters: [0, 8) width logits u_x, [8, 16) height logits u_x + u_y / 1000,
     * [16, 23) derivative logits c + u_d / 1000 -- ALL multiplied by log2(e) by the packer -- and one pad; W1 / b1
     * times 2 log2(e) as for the affine lean ops.  Op record as for TFK_OP_RQS_*: K = 8, boundary, scale, c.
     * K = 8 + 256 selects the bf16 x 3 operand format: GEMM 2 on the bf16 matrix pipe at fp32 accuracy -- every weight
     * is split by the packer into three bf16 pieces of 8 mantissa bits (hi = w & 0xffff0000, mid, lo likewise from the
     * exact remainders), the kernel splits the hidden activations the same way and keeps six of the nine piece products
     * (three v_mfma_f32_16x16x32_bf16 per tile).  Chunks then hold 4 target elements: A[24][2][64][4 dwords] with, per
     * tile and lane, the two operands [W_hi | W_mid] and [W_lo | W_hi] (4 bf16 each = hidden units 4 i + (lane >> 4));
     * no b2: the bias is the weight of hidden unit 15, which the kernel sets to 1 (hidden width <= 15); D/32 chunks
     * per layer. */
    TFK_OP_RQS_FWD_LEAN = 17,
    TFK_OP_RQS_INV_LEAN = 18,
    /* context-conditioned programs (tfk_flow_run_mfma_ctx): an elementwise affine layer whose (D, 2) parameters are
     * predicted from the row's context by a Linear conditioner (ElementwiseBijection with a context_shape,
     * layers_base.py:300-318) -- block Ac[D/8][cs][64] | bc[D/8][4][4], cs = ceil(C / 4) k-steps: tile t < D/16 holds
     * the parameters of the lane-group's elements 2 t, 2 t + 1 of plane A, the other tiles those of plane B -- and
     * couplings (TFK_OP_AFFINE_* / SHIFT_* / RQS_*, hidden width <= 16) whose conditioner sees [x_A || context]
     * (conditioning/context.py:46-60): src_plane bits 4..7 = cs, and cs further GEMM-1 A-operand steps A1c[cs][64]
     * follow the op's block. */
    TFK_OP_EWC_MULADD = 19,
    TFK_OP_EWC_SUBDIV = 20,
    /* lean MADE programs (the parallel map of MAF / IAF layers, cf. TFK_OP_MADE_*): a chain of MADE-based affine layers
     * of one kind and one hidden width (<= 16), optionally ended by one TFK_OP_EW_FMA, on the straight-line kernel.
     * Block: A1[D/16][64][4] (plane A's k-steps, then plane B's) | b1[4][4] | A2[nA2/4][64][4] | b2[D/8][4][4] |
     * pre_s[D] | pre_t[D], weights pre-scaled as for the lean couplings; every element takes its pre-affine. */
    TFK_OP_MADE_FWD_LEAN = 21,
    TFK_OP_MADE_INV_LEAN = 22,
    /* lean LINEAR rational spline programs (LinearRational, spline/linear_rational.py:9-182; CouplingLRS): as
     * TFK_OP_RQS_*_LEAN in the bf16 x 3 operand format (K = 8 + 256 only), 32 parameters = 8 tiles per element --
     * [0, 8) width logits, [8, 16) height logits u_x + u_y / 100, [16, 24) MINUS the lambda logits, [24, 31) derivative
     * logits c + u_d / 100, [31] the w0 logit, all times log2(e); chunks of 4 / HT elements = 16384 dwords.  Op record:
     * K = 8 + 256, boundary, scale = 1 - 1e-2 * 8, c = log(exp(1 - 1e-5) - 1). */
    TFK_OP_LRS_FWD_LEAN = 23,
    TFK_OP_LRS_INV_LEAN = 24,
    /* lean MADE SPLINE programs (the parallel map of MaskedAutoregressive / InverseAutoregressive RQNSF and LRS layers, cf.
     * TFK_OP_MADE_RQS): as the lean spline couplings in the bf16 x 3 operand format (K = 8 + 256, hidden width <= 15,
     * D = 64 or 128), but GEMM 1 reads both planes and every element of both planes is a target -- head
     * A1[D/16][64][4] (plane A's k-steps, then plane B's) | b1[4][4] | pre_s[D] | pre_t[D], then D/16 chunks (plane A's
     * elements first); MADE masks folded into the packed weights; src_plane = 0. */
    TFK_OP_MADE_RQS_FWD_LEAN = 25,
    TFK_OP_MADE_RQS_INV_LEAN = 26,
    TFK_OP_MADE_LRS_FWD_LEAN = 27,
    TFK_OP_MADE_LRS_INV_LEAN = 28
};
int tfk_flow_supported(int32_t D);
int tfk_flow_run(const float *x, float *z, float *logdet, const float *gauss_loc,
                 const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                 const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                 int32_t accumulate, void *stream);

/* ---- fused flow program, conditioner GEMMs on the matrix cores --------------------
 * Same semantics and outputs as tfk_flow_run for chains of elementwise ops and affine / shift
 * couplings, with the two conditioner GEMMs issued as v_mfma_f32_16x16x4_f32 (fp32 in / fp32
 * accumulate: numerically an fmaf chain).  D must be 64, 128 or 256 (RQS ops: 64 or 128),
 * hidden width <= 64 (RQS ops: <= 16).
 * ops (HOST pointer): n_ops x int32[8] = {kind, src_plane, gemm2_steps = ceil(H/4), param_offset,
 *   K, boundary, scale, c} (the last four as in tfk_flow_run, RQS ops only).
 * Elementwise ops use the parameter layout of tfk_flow_run; a coupling op holds
 *   A1[D/8][HT][64] | b1[HT][4][4] | A2[T2][gemm2_steps][64] | b2[T2][4][4],  HT = 1 / 2 / 4 for
 *   gemm2_steps <= 4 / 8 / 16,
 *   T2 = D/16 (affine), D/32 (shift), 6*D/8 (RQS, n_bins = 8: 6 tiles of 4 parameters per element)
 * accumulate: bit 0 = add to logdet instead of overwriting it; bit 1 = store the rows reversed
 *   (z[n, D-1-c] = column c: a ReversePermutationMatrix that follows the program, folded into the store);
 *   bit 2 = logprob is the base density of the rows as they come IN plus the log-det (Flow.sample with
 *   return_log_prob, flows.py:699-707: base_log_prob(z) + log_det of the inverse) instead of the rows going out;
 *   bit 3 = (lean affine / shift programs, D >= 128) stream the operands from `params` even if they fit the LDS -- a
 *   program whose blocks do NOT fit the 160 KiB LDS together is streamed anyway: one launch, two blocks resident.
 * A MADE op (TFK_OP_MADE_*: MaskedAutoregressiveBijection's parallel map, layers_base.py:201-206,
 * affine transformer, weights pre-multiplied by the MADE masks) reads BOTH halves of the row and
 * transforms both:  A1[2*D/8][HT][64] | b1[HT][4][4] | A2[2*D/16][gemm2_steps][64] | b2[2*D/16][4][4].
 * i.e. the MFMA A-operands per lane, with the row / column permutations that make the
 * accumulator layout of one GEMM the B-operand of the next (csrc/tfk_flow_mfma.hip;
 * packed by torchflows_amd/fused.py:_pack_mfma). */
int tfk_flow_mfma_supported(int32_t D);
/* Row widths the LEAN programs (TFK_OP_*_LEAN) run at: tfk_flow_mfma_supported's, 32 and 16 -- event sizes <= 32 are
 * padded to 32 instead of 64 (the straight-line kernels are instantiated for D / 8 = 4 as well; spline chains at
 * D = 32 in the bf16 x 3 operand format only), and chains of affine / shift couplings on event sizes <= 16 to 16
 * (D / 8 = 2 row elements per lane and plane: A1 is [64][2], a shift coupling's single GEMM-2 tile uses rows r < 2 of
 * every group of four; fp32 operands, no context, no MADE / spline ops). */
int tfk_flow_lean_supported(int32_t D);
int tfk_flow_run_mfma(const float *x, float *z, float *logdet, const float *gauss_loc,
                      const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                      const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                      int32_t accumulate, void *stream);
/* The same with input rows NARROWER than the kernel's row width (event sizes that are not 64 / 128 / 256): x is
 * (N, x_width), x_width even and <= D; its first half is read into the head of plane A, its second half into the head
 * of plane B, zeros behind them -- the layout the packer pads such flows to (zero weights make the padding an exact
 * identity), without a host-side padding pass over the rows.  Lean programs only; z / logdet / logprob as above
 * (z in the kernel's D-wide physical layout).
 * An ODD x_width (ABI v27; chains of affine / shift / spline couplings that carry the move bit, (x_width + 1) / 2 <= D / 2):
 * x_width / 2 sources into the head of plane A, the middle element into plane B's LAST column, the remaining x_width / 2
 * elements into the head of plane B. */
/* tfk_flow_run_mfma_in for a LEAN program that ends in the base density, plus the fp64 sum of the launch's N
 * log-probabilities in sum_out[0] (device) -- the per-rank term of SURVEY.md 8(e)'s one all-reduce -- without further
 * launches: every workgroup leaves the fp64 sum of its rows in the workspace, the last one to finish adds them in index
 * order (deterministic) and resets the workspace.  sum_workspace: tfk_flow_sum_workspace_bytes() bytes, ZERO before its
 * first use and not shared by launches that may overlap (one per stream).  Replaces: Flow.log_prob(x).sum() over a batch
 * (flows.py:646-658 + the reduction of the caller). */
int64_t tfk_flow_sum_workspace_bytes(void);
int tfk_flow_run_mfma_sum(const float *x, int32_t x_width, float *z, float *logdet, const float *gauss_loc,
                          const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                          const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                          int32_t accumulate, void *sum_workspace, double *sum_out, void *stream);

/* The same for context-conditioned flows: context (N, C) fp32, 1 <= C <= 16, one row per data row (Flow.log_prob(x,
 * context=...), flows.py:628-658).  Interpreter programs of elementwise ops, TFK_OP_EWC_* and couplings (no MADE ops), or
 * LEAN context programs (ABI v22, D >= 64):
 *  - spline chains (TFK_OP_RQS_*_LEAN / TFK_OP_LRS_*_LEAN, bf16 x 3 operand format): src_plane bits 4..7 = cs = ceil(C / 4)
 *    on every coupling, and every coupling's head ends with A1c[HT][64][4] -- lane (q, i), slot k: the weight of hidden
 *    unit 16 t + unit(i) for context element 4 k + q, times 2 log2(e) (zero beyond C) -- behind pre_t; at D <= 128 also the
 *    elementwise ops inside the program described for the affine chains below;
 *  - affine / shift chains (TFK_OP_AFFINE_*_LEAN / TFK_OP_SHIFT_*_LEAN, fp32 operands, D = 64 or 128): the
 *    same bits, A1c[64][4] behind each coupling's pre_t, AND elementwise ops inside the program: up to 3 in front of
 *    the first coupling and up to 3 behind the closing TFK_OP_EW_FMA, each either a TFK_OP_EW_FMA block (a constant
 *    x -> s x + t with its log-det) or a TFK_OP_EWC_* op (block laid out as in the interpreter, src_plane = cs << 4, but
 *    with the scale-logit rows times log2(e) / 2 and their bias (b / 2 + log(1 - 1e-10)) log2(e), as in the lean
 *    couplings).  A lean program is recognised by its first coupling op, not by its first op. */
int tfk_flow_run_mfma_ctx(const float *x, const float *context, int32_t C, float *z, float *logdet,
                          const float *gauss_loc, const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                          const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                          int32_t accumulate, void *stream);
int tfk_flow_run_mfma_in(const float *x, int32_t x_width, float *z, float *logdet, const float *gauss_loc,
                         const float *gauss_log_scale, float *logprob, int64_t N, int32_t D,
                         const int32_t *ops, int32_t n_ops, const float *params, int64_t n_params,
                         int32_t accumulate, void *stream);

/* Linear rational spline coupling: MonotonicSpline + LinearRational
 * (spline/base.py:53-72, spline/linear_rational.py:9-182) inside CouplingBijection.forward / inverse.
 * Same conventions as tfk_rqs_coupling_*; h is (N, T, 4*n_bins) = [u_x | u_y | u_lambda | u_d (K-1) |
 * u_w0] per element, 16-byte aligned; n_bins 4 or 8. */
int tfk_lrs_coupling_fwd(const float *x, const float *h, float *z, float *logdet, int64_t N, int32_t D,
                         const int32_t *tgt_idx, int32_t T, int32_t n_bins, float boundary,
                         int32_t accumulate, void *stream);
int tfk_lrs_coupling_inv(const float *z, const float *h, float *x, float *logdet, int64_t N, int32_t D,
                         const int32_t *tgt_idx, int32_t T, int32_t n_bins, float boundary,
                         int32_t accumulate, void *stream);

/* The SEQUENTIAL map of a MADE-based affine layer in one launch: MAF sampling
 * (MaskedAutoregressiveBijection.inverse, layers_base.py:208-221) and IAF density
 * (InverseMaskedAutoregressiveBijection.forward, :231-232), where the reference runs D conditioner
 * passes.  MADE with two masked linear layers (transforms.py:184-267); weights arrive multiplied by
 * their masks and zero-padded to hidden_padded (8, 16, 32 or 64) hidden units:
 *   W1t (D, hidden_padded): W1t[i][k] = (W1 * mask1)[k][i];  b1 (hidden_padded)
 *   W2  (D, 2, hidden_padded): W2[i][p][k] = (W2 * mask2)[2 i + p][k];  b2 (D, 2)
 * divide != 0: x_i = (z_i - beta_i) / alpha_i, log-det -= log alpha_i (Affine.inverse);
 * divide == 0: x_i = alpha_i z_i + beta_i, log-det += log alpha_i (InverseAffine.inverse).
 * logdet (N,) overwritten or accumulated; x may alias z. */
int tfk_made_affine_sequential(const float *z, float *x, float *logdet, int64_t N, int32_t D,
                               const float *W1t, const float *b1, const float *W2, const float *b2,
                               int32_t hidden_padded, int32_t divide, int32_t accumulate, void *stream);

/* The same walk for a MADE-based rational-quadratic spline layer (n_bins = 8): element i's 23 parameters
 * come from rows [23 i, 23 i + 23) of W2 (D, 23, hidden_padded), b2 (D, 23); the spline is inverted
 * (rational_quadratic.py:147-200).  hidden_padded in {8, 16}.  The log-det is the one the reference returns:
 * that of its LAST pass (layers_base.py:213-221), i.e. elements j < D-1 evaluated at their inverted values.
 * tfk_made_rqs_sequential_lds_bytes: LDS the launch needs at its smallest workgroup (must be <= 160 KiB). */
int64_t tfk_made_rqs_sequential_lds_bytes(int32_t D, int32_t hidden_padded, int32_t n_bins);
int tfk_made_rqs_sequential(const float *z, float *x, float *logdet, int64_t N, int32_t D,
                            const float *W1t, const float *b1, const float *W2, const float *b2,
                            int32_t hidden_padded, int32_t n_bins, float boundary, int32_t accumulate,
                            void *stream);
/* ... and for a MADE-based linear rational spline layer (n_bins = 8): W2 (D, 32, hidden_padded), b2 (D, 32). */
int64_t tfk_made_lrs_sequential_lds_bytes(int32_t D, int32_t hidden_padded, int32_t n_bins);
int tfk_made_lrs_sequential(const float *z, float *x, float *logdet, int64_t N, int32_t D,
                            const float *W1t, const float *b1, const float *W2, const float *b2,
                            int32_t hidden_padded, int32_t n_bins, float boundary, int32_t accumulate,
                            void *stream);

/* One block of the Glow ConvNet conditioner (multiscale/conditioning/classic.py, ConvNetBlock.forward) in
 * one launch: conv3x3 (padding 1) -> ReLU -> MaxPool2d(2) -> inference BatchNorm2d given as per-channel
 * scale / shift.  x (N, c_in, H, W) NCHW, weight (c_out, c_in, 3, 3), out (N, c_out, H/2, W/2);
 * c_in, c_out in {4, 8}; H, W even. */
int tfk_conv3x3_block_supported(int32_t c_in, int32_t c_out);
int tfk_conv3x3_relu_pool_affine(const float *x, const float *weight, const float *bias, const float *scale,
                                 const float *shift, float *out, int64_t N, int32_t c_in, int32_t c_out,
                                 int32_t H, int32_t W, void *stream);

/* ConvModifier with a 1x1 kernel (classic.py:8-42: conv2d whose padding exceeds kernel - 1): channel mixing
 * c_in -> c_out (1 or 4) of an (H, W) image placed in the middle of an (H_out, W_out) frame that holds the
 * bias (H_out - H and W_out - W even, >= 0).  x (N, c_in, H, W) with x_stride floats between images (a view
 * of wider rows is fine), weight (c_out, c_in), out (N, c_out, H_out, W_out) contiguous. */
int tfk_conv1x1_frame(const float *x, int64_t x_stride, const float *weight, const float *bias, float *out,
                      int64_t N, int32_t c_in, int32_t c_out, int32_t H, int32_t W, int32_t H_out, int32_t W_out,
                      void *stream);

/* Bounded conditioner output (conditioning/transforms.py:107-113, used by ConvNetConditioner with (-2, 2)):
 * out = lo + (hi - lo) * sigmoid(h) over n floats, the three roundings of the reference kept; h may alias out. */
int tfk_bounded_sigmoid(const float *h, float *out, int64_t n, float lo, float hi, void *stream);
/* Its reverse mode from the OUTPUT alone: g_in = g * (hi - lo) * s (1 - s), s = (out - lo) / (hi - lo); g_in may alias g. */
int tfk_bounded_sigmoid_bwd(const float *out, const float *g, float *g_in, int64_t n, float lo, float hi, void *stream);

/* ---- The ConvNet conditioner in TRAINING (classic.py:45-122 under Flow.fit: BatchNorm2d with batch statistics) ----
 * One forward and one reverse-mode launch per block (csrc/tfk_convtrain.hip); every sum over the batch is a
 * fixed-order sum of per-workgroup partials finished by the last workgroup of the SAME launch (no reduction launches).
 * workspace: tfk_convnet_train_workspace_bytes() bytes, ZEROED once by the caller (its first 8 bytes are the
 * ticket counter, left at zero by every launch), used by one stream at a time.
 *
 * BatchNorm tensors: `stats` (4 c floats: scale | shift | mean | 1/std of the normalisation the forward applied,
 * scale = weight / std, shift = bias - mean * scale); `coef` (3 c floats: d(loss)/d(input) = c1 g + c2 y + c3, the
 * batch-statistics backward folded into three per-channel numbers). */
int64_t tfk_convnet_train_workspace_bytes(void);
int tfk_convnet_train_block_supported(int32_t c_in, int32_t c_out, int32_t H);
/* conv3x3 (padding 1; input = in_affine applied to x inside the image, in_affine = scale | shift of the BatchNorm
 * in front or NULL) -> ReLU -> MaxPool2d(2): y (N, c_out, H/2, W/2) BEFORE normalisation, argmax (same shape, the
 * position 0..3 of the maximum in its 2x2 window), and the BatchNorm behind the block: training != 0 -> batch
 * statistics (biased variance), running statistics updated as torch.nn.functional.batch_norm does when
 * update_running != 0 (num_batches_tracked may be NULL); training == 0 -> the running statistics. */
int tfk_convnet_train_block_fwd(const float *x, const float *in_affine, const float *weight, const float *bias,
                                float *y, uint8_t *argmax, const float *bn_weight, const float *bn_bias,
                                float *running_mean, float *running_var, int64_t *num_batches_tracked, float eps,
                                float momentum, int32_t training, int32_t update_running, float *stats,
                                void *workspace, int64_t N, int32_t c_in, int32_t c_out, int32_t H, int32_t W,
                                void *stream);
/* Reverse mode of the block: gz = d(loss)/d(BatchNorm output) (N, c_out, H/2, H/2), coef of that BatchNorm, y /
 * argmax / x / in_affine / weight as in the forward (square planes, tfk_convnet_train_block_supported).
 * g_in (N, c_in, H, H) = d(loss)/d(the affine'd input).  sums: weight gradient (c_out x c_in x 9) | bias gradient
 * (c_out).  If the input came through a BatchNorm (bn_stats = its `stats`, else NULL): its coef, weight and bias
 * gradients are written too (bn_training: batch statistics were used). */
int tfk_convnet_train_block_bwd(const float *gz, const float *coef, const float *y, const uint8_t *argmax,
                                const float *x, const float *in_affine, const float *weight, float *g_in, float *sums,
                                const float *bn_stats, float *bn_coef, float *bn_dweight, float *bn_dbias,
                                int32_t bn_training, void *workspace, int64_t N, int32_t c_in, int32_t c_out, int32_t H,
                                void *stream);
/* ConvModifier (classic.py:8-42): ONE convolution with a kh x kw kernel (1 or 2 per axis; 2 where the size difference
 * is odd) whose padding (H_out - H + kh - 1) / 2 is >= kh - 1, so the (H, W) image lands inside an (H_out, W_out) frame
 * that holds the bias; weight (c_out, c_in, kh, kw), c_out 1 or 4; the BatchNorm in front applied on load (in_affine or
 * NULL).  Reverse mode: g_in (N, c_in, H, W) = d(loss)/d(the affine'd input); sums: weight gradient (c_out x c_in x kh
 * x kw) | sum g_in (c_in) | sum g_in x (c_in) | bias gradient (c_out); BatchNorm outputs as above. */
int tfk_convnet_train_frame_fwd(const float *x, const float *in_affine, const float *weight, const float *bias,
                                float *out, int64_t N, int32_t c_in, int32_t c_out, int32_t H, int32_t W,
                                int32_t H_out, int32_t W_out, int32_t kh, int32_t kw, void *stream);
int tfk_convnet_train_frame_bwd(const float *g_out, const float *x, const float *in_affine, const float *weight,
                                float *g_in, float *sums, const float *bn_stats, float *bn_coef, float *bn_dweight,
                                float *bn_dbias, int32_t bn_training, void *workspace, int64_t N, int32_t c_in,
                                int32_t c_out, int32_t H, int32_t W, int32_t H_out, int32_t W_out, int32_t kh,
                                int32_t kw, void *stream);
/* The WHOLE network in one call each way (the launches above in sequence + the dot product that completes the second
 * modifier's bias gradient): what a host in an interpreted language wants -- one FFI crossing per pass instead of seven.
 * The network is the reference's default ConvNet: modifier (4, c, kh, kw) -> blocks 4->8 @32, 8->8 @16, 8->4 @8 ->
 * modifier (1, 4, 1, 1) -> Linear (M, 100).  All pointers are device pointers the caller owns; `acts` are written by
 * the forward pass and read by the backward pass. */
typedef struct tfk_convnet_train_plan {
    const float *mod1_w, *mod1_b;
    const float *conv_w[3], *conv_b[3];
    const float *bn_w[3], *bn_b[3];
    float *bn_mean[3], *bn_var[3];          /* running statistics */
    int64_t *bn_count[3];                   /* num_batches_tracked, may be NULL */
    float bn_eps[3], bn_momentum[3];
    const float *mod2_w, *mod2_b;
    const float *lin_w, *lin_b;
    int32_t c, h, w, kh, kw, M;
    /* activations: a0 (N, 4, 32, 32); y[k] / amax[k] (N, 8, 16, 16), (N, 8, 8, 8), (N, 4, 4, 4); stats[k] 4 C_k floats;
     * a16 (N, 16); lin_fold 18 M floats (W16 | b_eff | w_frame) */
    float *a0, *y[3];
    uint8_t *amax[3];
    float *stats[3];
    float *a16, *lin_fold;
    void *workspace;                        /* tfk_convnet_train_workspace_bytes(), zeroed once */
} tfk_convnet_train_plan;
/* theta (N, M) = net(x), x (N, c, h, w). */
int tfk_convnet_train_forward(const tfk_convnet_train_plan *plan, const float *x, float *theta, int64_t N,
                              int32_t training, int32_t update_running, void *stream);
/* Reverse pass from g_theta (N, M).  g_x (N, c, h, w); scratch: N * 6736 floats; bn_out: 100 floats (per BatchNorm k:
 * coef 3 C_k | d weight C_k | d bias C_k, at offsets 0, 40, 80); sums: the launches' sums back to back --
 *   modifier 1 [dW 4 c kh kw | 2 c (ignore) | db 4]  block 1 [dW 288 | db 8]  block 2 [dW 576 | db 8]  block 3 [dW 288 | db 4]
 *   modifier 2 [dW 4 | 8 (ignore) | db 1]  Linear [dW 100 M | db M]
 * = tfk_convnet_train_sums_floats(c, kh, kw, M) floats. */
int64_t tfk_convnet_train_sums_floats(int32_t c, int32_t kh, int32_t kw, int32_t M);
int tfk_convnet_train_backward(const tfk_convnet_train_plan *plan, const float *x, const float *g_theta, float *g_x,
                               float *scratch, float *bn_out, float *sums, int64_t N, int32_t training, void *stream);

/* The Linear layer behind the second ConvModifier.  Its input equals the modifier's bias (*frame_bias) outside the 4 x 4
 * interior of the (H_out, W_out) frame, so the layer is a 16-term product: prep folds the weight (M, H_out * W_out) into
 * W16 (M, 16) = its interior columns, w_frame (M) = the sum of the others, b_eff (M) = bias + *frame_bias * w_frame;
 * fwd: out (N, M) = b_eff + a16 (N, 16) W16^T;  bwd_input: g16 (N, 16) = g (N, M) W16;  wgrad: dW (M, H_out * W_out)
 * (interior columns sum_n g a16, the others *frame_bias * db) and db (M) = sum_n g.  No GEMM-library call (a training
 * step with one cannot be captured into a hipGraph on this stack); rows in order, deterministic.
 * d(loss)/d(*frame_bias) = dot(db, w_frame) joins the modifier's own bias gradient. */
int tfk_convnet_train_linear_prep(const float *weight, const float *bias, const float *frame_bias, float *W16,
                                  float *b_eff, float *w_frame, int32_t M, int32_t H_out, int32_t W_out, void *stream);
int tfk_convnet_train_linear_fwd(const float *a16, const float *W16, const float *b_eff, float *out, int64_t N, int32_t M,
                                 void *stream);
int tfk_convnet_train_linear_bwd_input(const float *g, const float *W16, float *g16, int64_t N, int32_t M, void *stream);
int tfk_convnet_train_linear_wgrad(const float *g, const float *a16, const float *frame_bias, float *dW, float *db,
                                   int64_t N, int32_t M, int32_t H_out, int32_t W_out, void *stream);

/* ---- a whole convolutional coupling of the image / multiscale flows in ONE launch (config 5) -----------------
 * Replaces, for one coupling of multiscale/base.py:19-114 (CheckerboardCoupling, ChannelWiseCoupling,
 * Invertible1x1ConvolutionalCoupling with the ConvNet conditioner), IN PLACE on the rows:
 *   x_A = x[..., source_mask].view(constant_shape); h = ConvNetConditioner(x_A)   (multiscale/conditioning/
 *   classic.py:8-145: ConvModifier, 3 x [conv3x3, ReLU, MaxPool2d(2), BatchNorm2d], ConvModifier, Linear, and the
 *   (-2, 2) sigmoid bound of conditioning/transforms.py:107-113);  z[..., target_mask] = transformer(x_B, h)
 *   (Affine, transformers/linear/affine.py:33-59, or Invertible1x1ConvolutionTransformer, linear/convolution.py:
 *   33-64 + linear/matrix.py:11-99);  logdet[n] += the layer's log-det   (layers_base.py:145-163).
 * The rows keep ONE physical layout (N, D) through all layers: `src_idx` lists the physical positions of the
 * conditioner's input image (c_in, hi, wi) in its own row-major order, `tgt_idx` those of the transformer's targets in
 * the order of h -- so Squeeze / chunk (multiscale/base.py:117-175, 271-280) and the masks (multiscale/coupling.py:
 * 6-63) are folded into the two tables by the caller.  `src_st` / `tgt_st` hold a pending map per listed element,
 * (s, t) interleaved: the value the layer sees is s * rows[n, idx] + t (the deferred ActNorm layers in front of it;
 * (1, 0) = none); targets are stored in final form.
 * The caller evaluates everything that does not depend on the sample (see torchflows_amd/image_program.py):
 *   weights  tfk_glow_weight_floats(c_in * kh * kw) floats: first ConvModifier W (4, c_in, kh, kw), b (4) | conv1 W as [ci=4][3][3][co=8],
 *            b1 (8), BatchNorm-1 scale (8), shift (8) | conv2 W as [ci=8][3][3][co=8], b2, scale, shift (8 each) |
 *            conv3 W as [ci=8][3][3][co=4], b3 (4) | second ConvModifier's 4 weights times BatchNorm-3's scales, then its
 *            bias + sum_c weight_c shift_c   (BatchNorm in inference form: scale = gamma / sqrt(var + eps), shift = beta - mean * scale)
 *   bg1, bg2 (8, 16, 16) and (8, 8, 8): the outputs of conv blocks 1 and 2 for the all-bias image (every pixel of the
 *            (4, 32, 32) frame = the first ConvModifier's bias), i.e. what the blocks produce wherever the source image's
 *            receptive field does not reach; only the window it does reach is computed per sample
 *   w_eff    tiles of [64 lanes][4] floats: tile t, lane l, k-step ks = W[16 t + (l & 15)][4 ks + (l >> 4)], where the rows
 *            of W are log2(e) times the rows of W_eff (n_params, 16) -- the Linear layer's columns of the 4x4 interior of the (1, 10, 10)
 *            image -- in KERNEL ORDER.  1x1 convolution: ceil(n_params / 16) tiles, rows in the order of h.  Affine:
 *            2 * ceil(T / 16) tiles; for the target listed at position 16 m + j of tgt_idx, row 32 m + j is its scale
 *            logit (h[n, t, 0]) and row 32 m + 16 + j its shift (h[n, t, 1]); rows of padding are zero.  Shift:
 *            ceil(T / 16) tiles, row 16 m + j = the shift of the target at position 16 m + j of tgt_idx
 *   b_eff    16 * tiles floats in the same row order: log2(e) times (Linear bias + its 84 frame columns times the second
 *            ConvModifier's bias) -- the kernel's sigmoid is 1 / (1 + exp2(-h log2 e))
 * tgt_idx (affine, shift) may list the targets in any order -- ascending physical position makes the 16 targets of a tile pair
 * neighbours in the row -- padded, like tgt_st, to a multiple of 16 entries.
 * Supported: first ConvModifier as the reference builds it for images up to 32 pixels (classic.py:21-33: kernel 1 and padding
 * (32 - size) / 2 along an axis where 32 - size is even, kernel 2 and padding (33 - size) / 2 where it is odd -- e.g. the
 * 7-row conditioner images of 28x28 inputs), ConvNet kernels (8, 8, 4), 1x1 convolutions of <= 16 channels.
 * slots / block / cg1 / cg2 / grid = 0 let the library choose the launch shape (tfk_glow_plan reports it). */
typedef struct tfk_glow_layer {
    int32_t kind;              /* 0 affine, 1 invertible 1x1 convolution, 2 shift (z = x +/- h, log-det 0; affine.py:137-159) */
    int32_t c_in, hi, wi;      /* conditioner input image */
    int32_t oy, ox;            /* where the modifier's non-constant rectangle starts in its 32x32 frame: padding - (kernel - 1) */
    int32_t kh, kw;            /* first ConvModifier's kernel, 1 or 2 per axis (0 = 1); the rectangle is (hi + kh - 1, wi + kw - 1) */
    int32_t T;                 /* target elements */
    int32_t n_params;          /* 2 T (affine), T (shift) or n + n (n - 1) (1x1 convolution of n channels) */
    int32_t n_ch, hw;          /* 1x1 convolution: target channels, pixels per channel (T = n_ch * hw); else 0 */
    int32_t slots, block, cg1, cg2, grid;   /* launch shape overrides, 0 = default */
    const int32_t *src_idx;    /* device int32[c_in * hi * wi] */
    const float *src_st;       /* device float[2 * c_in * hi * wi] */
    const int32_t *tgt_idx;    /* device int32[T]; affine: padded to 16 * ceil(T / 16) entries */
    const float *tgt_st;       /* device float[2 * T], padded likewise (any values), 8-byte aligned */
    const float *weights, *bg1, *bg2, *w_eff, *b_eff;
} tfk_glow_layer;
int64_t tfk_glow_weight_floats(int32_t c_in_times_taps);
int tfk_glow_plan(const tfk_glow_layer *layer, int32_t D, int32_t *slots, int32_t *block, int32_t *cg1, int32_t *cg2,
                  int32_t *lds_bytes, int32_t *tile_rows);
int tfk_glow_coupling(float *rows, float *logdet, int64_t N, int32_t D, const tfk_glow_layer *layer, int32_t inverse,
                      void *stream);

/* ---- SEVERAL consecutive couplings of an image flow in ONE launch, the rows held in the LDS (round 4) -------------
 * The couplings of one level of a multiscale flow (multiscale/base.py:249-296: the checkerboard layers, then -- squeezed
 * -- the channel-wise layers) all work on the same set of row elements.  tfk_glow_level runs a list of them back to back
 * for a few samples at a time: the samples' elements of that set are read from HBM once into the LDS (`row_idx`:
 * their physical positions in ascending order, NULL = the whole row), every coupling of the list reads its sources
 * and transforms its targets there, and the elements are written back once -- instead of one read of the sources and one
 * read + write of the targets PER coupling (AffineGlow (3, 32, 32): 19 couplings = 240 KB per row; as three level
 * launches 72 KB).  Semantics = tfk_glow_coupling applied step by step (same tables, same pending maps, the per-sample
 * log-det added in the same order).
 * A step is a tfk_glow_layer plus the positions of its elements INSIDE the level's element list:
 *   src_loc   device uint16[c_in * hi * wi], the order of layer.src_idx
 *   tgt_loc   device uint16[T'], the order of layer.tgt_idx; affine / shift: ascending, T' = T padded to a multiple of 64
 *             (layer.tgt_st padded likewise); 1x1 convolution: T' = T
 *   w4, b4    affine / shift: the Linear layer for v_mfma_f32_4x4x1 (4 samples x 64 parameters per instruction, no padding
 *             of the sample axis): w4 = rows of log2(e) W_eff (R, 16) row-major, b4 = log2(e) b_eff (R), R = n_params padded to
 *             a multiple of 64, in the order [u of target 0, beta of target 0, u of target 1, ...] (affine; = the order of h
 *             after sorting the targets) or [shift of target 0, 1, ...] (shift).  1x1 convolution: NULL (layer.w_eff is used).
 *   weights_host  the floats of layer.weights in HOST memory (layer.weights itself is not read by the level kernel)
 * The launch shape and everything else derived from the geometry (LDS layout, which cells of the activation buffers hold the
 * sample-independent background) are packed ONCE on the host into a caller-owned blob (tfk_glow_level_blob_bytes,
 * tfk_glow_level_pack), which the caller uploads; tfk_glow_level takes both copies (the host copy for the launch shape).
 * samples = 0 / block = 0: the library's choice (samples in {4, 8} resident per workgroup, 256 / 512 / 1024 threads).
 * rows_in may differ from rows_out only when row_idx is NULL (the level covers every element: no clone of the input). */
typedef struct tfk_glow_level_step {
    tfk_glow_layer layer;
    int32_t inverse, reserved;
    const uint16_t *src_loc, *tgt_loc;
    const float *w4, *b4;
    const float *weights_host;      /* HOST copy of layer.weights: packed into the blob (the kernel reads the conv weights as scalar loads through it) */
} tfk_glow_level_step;
int64_t tfk_glow_level_blob_bytes(const tfk_glow_level_step *steps, int32_t n_steps, int32_t D, int32_t D_level,
                                  int32_t samples, int32_t block);
int tfk_glow_level_pack(const tfk_glow_level_step *steps, int32_t n_steps, int32_t D, int32_t D_level, int32_t samples,
                        int32_t block, void *blob_host, int64_t blob_bytes);
int tfk_glow_level_info(const void *blob_host, int32_t *samples, int32_t *block, int32_t *lds_bytes, int32_t *wgs_per_cu);
int tfk_glow_level(const float *rows_in, float *rows_out, float *logdet, int64_t N, int32_t D, const int32_t *row_idx,
                   const void *blob_host, const void *blob_dev, void *stream);

/* rows[n, d] = st[2 d] * rows[n, d] + st[2 d + 1] in place: the flush of the pending maps behind the last coupling. */
int tfk_rows_fma(float *rows, const float *st, int64_t N, int32_t D, void *stream);

/* out[n] = DiagonalGaussian(loc, exp(log_scale)).log_prob(st[2 d] * rows[n, d] + st[2 d + 1]) [+ logdet_in[n]]: the flush
 * of an image program's pending maps (tfk_rows_fma) and the base density of Flow.log_prob (tfk_diag_gauss_logprob; gaussian.py:
 * 46-54, flows.py:647-648) as ONE read of the rows, which are not written (the mapped values are rounded as tfk_rows_fma
 * rounds them).  D <= 3276 (20 D bytes of LDS). */
int tfk_rows_fma_gauss_logprob(const float *rows, const float *st, const float *loc, const float *log_scale,
                               const float *logdet_in, float *out, int64_t N, int32_t D, void *stream);

/* ---- reverse mode of the layer kernels ------------------------------------
 * The reference has no backward code; these replace what torch.autograd derives from
 * affine.py:36-59, spline/base.py:53-72 + rational_quadratic.py:45-200, layers_base.py:237-318
 * and gaussian.py:46-54.  Recompute, not store: the kernels take the layer INPUT rows x and the
 * conditioner output h of the forward call, nothing else is kept between the two launches.
 *   g   (N, D) in/out: the gradient row buffer of the composition.  On entry its target columns
 *       hold dL/d(out rows), on exit dL/d(x target); pass-through columns are untouched (the
 *       conditioner's contribution to them is the caller's GEMM backward).
 *   gld (N,)   dL/d(log-det of this layer) = dL/d(total log-det)
 *   gh  (N, T, P) out: dL/dh, same layout as h.
 * inverse != 0: backward of the *_inv entry point (x = the rows that call received). */
int tfk_affine_coupling_bwd(const float *x, const float *h, float *g, const float *gld, float *gh,
                            int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T, int32_t inverse,
                            void *stream);
int tfk_shift_coupling_bwd(const float *g, float *gh, int64_t N, int32_t D, const int32_t *tgt_idx,
                           int32_t T, int32_t inverse, void *stream);
int tfk_rqs_coupling_bwd_supported(int32_t n_bins);             /* 4, 8, 16 */
int tfk_rqs_coupling_bwd(const float *x, const float *h, float *g, const float *gld, float *gh,
                         int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T, int32_t n_bins,
                         float boundary, int32_t inverse, void *stream);
/* linear rational spline (linear_rational.py:33-182), n_bins in {4, 8}; h / gh (N, T, 4 n_bins) */
int tfk_lrs_coupling_bwd(const float *x, const float *h, float *g, const float *gld, float *gh,
                         int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T, int32_t n_bins,
                         float boundary, int32_t inverse, void *stream);
/* ElementwiseAffine / ActNorm with batch-constant value (D, 2): g updated in place over all D
 * columns; gvalue (D, 2) = sum over rows of dL/dvalue (deterministic two-stage sum), or NULL to
 * skip it (ActNorm's value does not train, layers.py:49) -- then x, gld, workspace may be NULL. */
int64_t tfk_elementwise_affine_bwd_workspace_bytes(int64_t N, int32_t D);
int tfk_elementwise_affine_bwd(const float *x, const float *value, float *g, const float *gld,
                               float *gvalue, float *workspace, int64_t N, int32_t D,
                               int32_t inverse, void *stream);
/* g (N, D) out = glp[row] * d log N(z; loc, exp(log_scale)) / dz */
int tfk_diag_gauss_logprob_bwd(const float *z, const float *loc, const float *log_scale,
                               const float *glp, float *g, int64_t N, int32_t D, void *stream);

/* Fused training backward of ONE affine coupling layer on the HalfSplit mask (source = first half
 * of the row, target = second half) with the default FeedForward(Linear, Tanh, Linear) conditioner,
 * hidden width <= 15, D = 64 or 128: conditioner re-evaluation, transform backward, MLP backward
 * and the weight-gradient sums over the N rows in one launch (h, dL/dh never exist in HBM).
 *   x (N, D) the rows that entered the layer; g (N, D) in/out: dL/d(out rows) -> dL/d(x rows), all
 *   D columns; gld (N,); inverse_form != 0: the layer evaluates (x - beta) / alpha (ActNorm-style).
 *   g_reversed != 0: g arrives as dL/d(rows AFTER a following reversal), i.e. column c is read from
 *   g[n, D-1-c]; gscale (D floats or NULL): column factors applied to g as it is read -- the reverse
 *   mode of an ActNorm / ElementwiseAffine with fixed parameters that follows the coupling
 *   (1/alpha for the inverse-affine form, alpha for the affine form).
 *   params: the weights as MFMA operands (layout in csrc/tfk_bwd.hip, packed by
 *   torchflows_amd/autograd.py:_TrainPack); out: tfk_coupling_train_bwd_out_floats(D) floats holding
 *   dW2 | dW1 | db1 in accumulator layout (db2 = hidden unit 15 of dW2); workspace:
 *   tfk_coupling_train_bwd_workspace_bytes(D).  The sums are deterministic. */
int tfk_coupling_train_bwd_supported(int32_t D);
int64_t tfk_coupling_train_bwd_out_floats(int32_t D);
int64_t tfk_coupling_train_bwd_workspace_bytes(int32_t D);
int tfk_affine_coupling_train_bwd(const float *x, float *g, const float *gld, const float *params,
                                  int64_t n_params, int32_t gemm2_steps, float *out, float *workspace,
                                  int64_t N, int32_t D, int32_t inverse_form, const float *gscale,
                                  int32_t g_reversed, void *stream);

/* Training backward of ONE RQ-spline coupling layer (HalfSplit, FeedForward(Linear, Tanh, Linear)
 * conditioner, hidden width <= 16, D = 64, 8 bins) with the conditioner re-evaluated in the kernel:
 * h never exists in HBM, dL/dh is written once.  x, g, gld, inverse, gscale, g_reversed as in
 * tfk_affine_coupling_train_bwd.  Outputs in accumulator order (the caller un-permutes):
 *   gh_perm   (N, 768): column (6 e + c) * 16 + 4 q + r = dL/d(parameter 4c + r of target element 8q + e)
 *   gpre_perm (N, 16):  column 4 q + r = dL/d(pre-activation of hidden unit 4 r + q)
 * params: A1[8][64] | b1[16] | A2[48][gemm2_steps][64] | b2[48][16] | A2T[48][4][64] | A1T[2][4][64]
 * (csrc/tfk_bwd.hip; packed by torchflows_amd/autograd.py:_RqsTrainPack). */
int tfk_rqs_coupling_train_bwd_supported(int32_t D, int32_t n_bins);
int tfk_rqs_coupling_train_bwd(const float *x, float *g, const float *gld, const float *params,
                               int64_t n_params, int32_t gemm2_steps, float *gh_perm, float *gpre_perm,
                               int64_t N, int32_t D, int32_t n_bins, float boundary, int32_t inverse,
                               const float *gscale, int32_t g_reversed, void *stream);
/* The same launch, also writing the hidden activations it re-evaluates:
 *   hid_perm (N, 16): column 4 q + r = tanh(pre-activation of hidden unit 4 r + q), column 15 == 1 (hidden width <= 15:
 *   the bias column of the weight-gradient products below). */
int tfk_rqs_coupling_train_bwd_hid(const float *x, float *g, const float *gld, const float *params,
                                   int64_t n_params, int32_t gemm2_steps, float *gh_perm, float *gpre_perm,
                                   float *hid_perm, int64_t N, int32_t D, int32_t n_bins, float boundary,
                                   int32_t inverse, const float *gscale, int32_t g_reversed, void *stream);

/* out[m][k] = sum_n A[n][m] * B[n][k], m < M, k < 16: the products of a training step that contract over the batch
 * ROWS (dW2 = dL/dh^T hidden with A = gh_perm, M = 768; dW1^T = x_A^T dL/dpre with A = x, lda = D, M = 32; db1 with
 * A = gpre_perm, M = 16, column 15 of B == 1) on v_mfma_f32_16x16x4_f32, deterministic (per-workgroup partial blocks
 * added in a fixed order), with no GEMM-library call -- what makes the spline training step capturable into a hipGraph.
 * A (N, lda) row-major, only its first M columns are read; M = 16, 32 or a multiple of 256 (<= 1024); B (N, 16).
 * out holds M * 16 floats in ACCUMULATOR order: tile t (16 columns of A), lane (q, j), register r ->
 *   out[(t * 64 + 16 q + j) * 4 + r] = sum_n A[n][col(t, 4 q + r)] * B[n][j],
 *   col(t, i) = 64 (t >> 2) + 4 i + (t & 3) for M >= 256, 16 t + i for M = 16 / 32.
 * workspace: tfk_rows_outer_workspace_bytes(M) bytes. */
int64_t tfk_rows_outer_workspace_bytes(int32_t M);
int tfk_rows_outer(const float *A, int32_t lda, int32_t M, const float *B, float *out, float *workspace,
                   int64_t N, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TFK_H */
