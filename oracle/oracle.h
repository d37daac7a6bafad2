/*
 * oracle.h -- CPU restatement of the torchflows coupling-flow hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 * The product (torchflows_amd/) never imports, links or calls it.
 *
 * Parity pin: every function below is checked against outputs of the real
 * reference (davidnabergoj/torchflows v1.2.0, imported in the build container)
 * stored as golden fixtures under tests/golden/ (generator:
 * tests/golden/make_golden.py).  See tests/test_oracle_golden.py.
 *
 * All arithmetic is scalar IEEE fp32, no FMA contraction (-ffp-contract=off),
 * in the op order of the reference lines cited at each function.
 * Paths are relative to the reference root (torchflows/...).
 */
#ifndef TFK_ORACLE_H
#define TFK_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- integer rules (must be bit-exact) -------------------------------- */

/* HalfSplit: source = flat index < D/2, target = the rest.
 * bijections/finite/autoregressive/conditioning/coupling_masks.py:78-81 */
void orc_halfsplit_mask(int D, uint8_t *source_mask, uint8_t *target_mask);

/* Index lists of a boolean mask in ascending flat order == what
 * x[..., mask] gathers.  Returns the count. layers_base.py:119-129 */
int orc_mask_to_index(const uint8_t *mask, int D, int32_t *idx);

/* ReversePermutationMatrix: forward_permutation = [D-1..0], inverse via
 * scatter of arange. bijections/finite/matrix/permutation.py:8-37 */
void orc_reverse_permutation(int D, int32_t *fwd, int32_t *inv);

/* Checkerboard (bijections/finite/multiscale/coupling.py:6-31): source = arange(h*w) % 2
 * reshaped (h, w), repeated over channels, inverted on request; target = ~source. */
void orc_checkerboard_mask(int C, int H, int W, int invert, uint8_t *source_mask,
                           uint8_t *target_mask);

/* ChannelWiseHalfSplit (multiscale/coupling.py:34-63): source = channel < C/2. */
void orc_channelwise_mask(int C, int H, int W, int invert, uint8_t *source_mask,
                          uint8_t *target_mask);

/* Squeeze.forward as a gather list over the flat (C,H,W) event: out[j] = in[idx[j]],
 * sub-lattices in the order (even,even),(even,odd),(odd,even),(odd,odd)
 * (multiscale/base.py:136-154). */
void orc_squeeze_index(int C, int H, int W, int32_t *idx);

/* ---- transformers ------------------------------------------------------ */

/* Affine.forward / inverse on (N,T) with h (N,T,2) interleaved.
 * transformers/linear/affine.py:33-59.  logdet (N,) is OVERWRITTEN. */
void orc_affine_fwd(const float *x, const float *h, float *z, float *logdet,
                    int64_t N, int T);
void orc_affine_inv(const float *z, const float *h, float *x, float *logdet,
                    int64_t N, int T);

/* MonotonicSpline.forward/inverse + RationalQuadratic.*_1d on (N,T) with
 * h (N,T,3K-1).  transformers/spline/base.py:53-72,
 * transformers/spline/rational_quadratic.py:45-200.
 * logdet_el (N,T) per-element (may be NULL), bin_idx (N,T) int32 (may be NULL;
 * -1 outside the box), logdet (N,) OVERWRITTEN with the row sums. */
void orc_rqs_fwd(const float *x, const float *h, float *z, float *logdet,
                 float *logdet_el, int32_t *bin_idx,
                 int64_t N, int T, int K, float boundary);
void orc_rqs_inv(const float *z, const float *h, float *x, float *logdet,
                 float *logdet_el, int32_t *bin_idx,
                 int64_t N, int T, int K, float boundary);

/* Reverse mode of the two transformers above (what torch.autograd derives from
 * affine.py:33-59 and rational_quadratic.py:45-200; the reference has no backward code).
 * gz (N,T) = dL/d out, gld (N,) = dL/d logdet; gx (N,T), gh (N,T,P) OVERWRITTEN.
 * inverse != 0: gradients of the inverse-direction maps (x = first argument of *_inv). */
void orc_affine_bwd(const float *x, const float *h, const float *gz, const float *gld,
                    float *gx, float *gh, int64_t N, int T, int inverse);
void orc_rqs_bwd(const float *x, const float *h, const float *gz, const float *gld,
                 float *gx, float *gh, int64_t N, int T, int K, float boundary, int inverse);

/* MonotonicSpline + LinearRational on (N,T) with h (N,T,4K)
 * (transformers/spline/linear_rational.py:9-182); logdet (N,) OVERWRITTEN. */
void orc_lrs_fwd(const float *x, const float *h, float *z, float *logdet,
                 int64_t N, int T, int K, float boundary);
void orc_lrs_inv(const float *z, const float *h, float *x, float *logdet,
                 int64_t N, int T, int K, float boundary);

/* Invertible1x1ConvolutionTransformer + LUTransformer
 * (transformers/linear/convolution.py:33-70, transformers/linear/matrix.py:20-82) on
 * x (N, n, HW) channel-major with h (N, n + n(n-1)); logdet (N,) OVERWRITTEN with
 * +/- sum log U_ii (not scaled by HW). */
void orc_conv1x1_fwd(const float *x, const float *h, float *y, float *logdet,
                     int64_t N, int n, int HW);
void orc_conv1x1_inv(const float *y, const float *h, float *x, float *logdet,
                     int64_t N, int n, int HW);

/* Knots of M spline elements: bin_x, bin_y, delta each (M, K+1).
 * rational_quadratic.py:45-54, :75-77 */
void orc_rqs_knots(const float *h, int64_t M, int K, float boundary,
                   float *bin_x, float *bin_y, float *delta);

/* DiagonalGaussian.log_prob. base_distributions/gaussian.py:46-54 */
void orc_diag_gauss_logprob(const float *z, const float *loc,
                            const float *log_scale, float *out,
                            int64_t N, int D);

/* ---- conditioner ------------------------------------------------------- */

/* FeedForward.predict_theta_flat: Linear, (Tanh, Linear)* .
 * conditioning/transforms.py:274-307.  Weights are nn.Linear layout
 * W[l] (out_l, in_l) row-major, b[l] (out_l).  dims has n_linear+1 entries.
 * in_row = [x_A || context] already concatenated (context.py:46-60). */
void orc_feedforward_row(const float *in_row, int n_linear, const int32_t *dims,
                         const float *const *W, const float *const *b,
                         float *out_row, float *scratch /* 2*max(dims) */);

/* ---- layers and the composition driver -------------------------------- */

enum {
    ORC_ELEMENTWISE_AFFINE = 0,         /* layers.py:19-26  (Affine)        */
    ORC_ELEMENTWISE_INVERSE_AFFINE = 1, /* layers.py:29-69  (ActNorm, eval) */
    ORC_PERMUTATION = 2,                /* matrix/permutation.py:8-26       */
    ORC_AFFINE_COUPLING = 3,            /* layers.py:102-113                */
    ORC_RQS_COUPLING = 4,               /* layers.py:154-163                */
    ORC_SHIFT_COUPLING = 5,             /* layers.py:130-139 (NICE)         */
    ORC_LRS_COUPLING = 6                /* layers.py:142-151                */
};

typedef struct {
    int32_t kind;
    /* elementwise: value (D,2) = [unconstrained alpha, beta] per element */
    const float *value;
    /* permutation: z[j] = x[fwd[j]];  inverse: x[j] = z[inv[j]] */
    const int32_t *perm_fwd;
    const int32_t *perm_inv;
    /* coupling: gather lists (ascending flat index) */
    const int32_t *src_idx;
    int32_t S;
    const int32_t *tgt_idx;
    int32_t T;
    /* conditioner MLP */
    int32_t n_linear;
    const int32_t *dims;       /* n_linear+1 entries; dims[0] = S + C */
    const float *const *W;
    const float *const *b;
    /* spline */
    int32_t K;
    float boundary;
    /* MADE-based layers (layers_base.py:166-234): 0 = coupling; 1 = forward parallel, inverse
     * sequential; 2 = exchanged.  S = T = D, src_idx = tgt_idx = identity, W pre-masked. */
    int32_t autoregressive;
    int32_t swap_transformer;   /* the transformer's forward / inverse exchanged (InverseAffine) */
} orc_layer;

/* BijectiveComposition.forward (bijections/base.py:203-224): layers in order,
 * log_det accumulated sequentially in fp32.  context (N,C) may be NULL.
 * trace_z (n_layers,N,D) / trace_ld (n_layers,N) may be NULL. */
void orc_composition_forward(const orc_layer *layers, int n_layers,
                             const float *x, const float *context, int C,
                             float *z, float *logdet,
                             float *trace_z, float *trace_ld,
                             int64_t N, int D);

/* BijectiveComposition.inverse (bijections/base.py:226-232): reversed order. */
void orc_composition_inverse(const orc_layer *layers, int n_layers,
                             const float *z, const float *context, int C,
                             float *x, float *logdet,
                             int64_t N, int D);

/* Flow.log_prob (flows.py:628-658) = base_log_prob(z) + log_det. */
void orc_flow_log_prob(const orc_layer *layers, int n_layers,
                       const float *loc, const float *log_scale,
                       const float *x, const float *context, int C,
                       float *z, float *log_prob,
                       int64_t N, int D);

/* ActNorm data-dependent initialisation (train mode, first forward).
 * layers.py:58-68: shift = mean over the batch, scale = unbiased std (1 when the
 * batch has one row), value = [unconstrain_scale(scale), shift] with
 * unconstrain_scale(s) = (log(s - m) - log(1 - m)) * 2  (affine.py:36-37).
 * Statistics are accumulated in double and rounded once to fp32. */
void orc_actnorm_init(const float *x, int64_t N, int D, float *value /* (D,2) */);

/* Threads the row loop uses (OpenMP); 1 when built without -fopenmp. */
int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
