/*
 * oracle.c -- scalar fp32 CPU restatement of the torchflows coupling-flow hot
 * path.  TEST INFRASTRUCTURE ONLY (see oracle.h).  Written from the reference's
 * behaviour; every function cites the reference lines whose op order it keeps.
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, optional -fopenmp).
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- constants (python doubles rounded once to fp32, as ATen does when a
 *      python scalar meets an fp32 tensor) -------------------------------- */

/* affine.py:19-23: m = 1e-10, identity_unconstrained_alpha = log(1 - m) */
#define AFF_MIN_SCALE ((float)1e-10)
static inline float aff_c0(void) { return (float)log(1.0 - 1e-10); }

/* rational_quadratic.py:36-38 */
#define RQS_MIN_BIN ((float)1e-3)
#define RQS_MIN_DELTA ((float)1e-5)
static inline double rqs_boundary_u_delta(void) { return log(expm1(1.0 - 1e-5)); }

#define ORC_MAX_BINS 64

/* torch.clip semantics: NaN passes through (comparisons are false). */
static inline float clipf(float v, float lo, float hi)
{
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

/* ---- integer rules ------------------------------------------------------ */

void orc_halfsplit_mask(int D, uint8_t *source_mask, uint8_t *target_mask)
{
    /* coupling_masks.py:78-81: mask = arange(D).view(event) < D // 2,
     * Coupling (:50-56): target = ~source. */
    const int half = D / 2;
    for (int i = 0; i < D; ++i) {
        source_mask[i] = (uint8_t)(i < half);
        target_mask[i] = (uint8_t)!(i < half);
    }
}

int orc_mask_to_index(const uint8_t *mask, int D, int32_t *idx)
{
    /* x[..., mask] keeps the True positions in ascending flat order
     * (layers_base.py:119-125). */
    int n = 0;
    for (int i = 0; i < D; ++i)
        if (mask[i]) idx[n++] = i;
    return n;
}

void orc_reverse_permutation(int D, int32_t *fwd, int32_t *inv)
{
    /* permutation.py:34-37: forward_permutation = arange(D-1, -1, -1);
     * :16-17: inverse_permutation[forward_permutation] = arange(D). */
    for (int j = 0; j < D; ++j) fwd[j] = D - 1 - j;
    for (int j = 0; j < D; ++j) inv[fwd[j]] = j;
}

void orc_checkerboard_mask(int C, int H, int W, int invert, uint8_t *source_mask,
                           uint8_t *target_mask)
{
    /* multiscale/coupling.py:17-22 */
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < H * W; ++i) {
            uint8_t m = (uint8_t)(i % 2);
            if (invert) m = (uint8_t)!m;
            source_mask[c * H * W + i] = m;
            target_mask[c * H * W + i] = (uint8_t)!m;
        }
}

void orc_channelwise_mask(int C, int H, int W, int invert, uint8_t *source_mask,
                          uint8_t *target_mask)
{
    /* multiscale/coupling.py:49-54 */
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < H * W; ++i) {
            uint8_t m = (uint8_t)(c < C / 2);
            if (invert) m = (uint8_t)!m;
            source_mask[c * H * W + i] = m;
            target_mask[c * H * W + i] = (uint8_t)!m;
        }
}

void orc_squeeze_index(int C, int H, int W, int32_t *idx)
{
    /* multiscale/base.py:148-153: cat of x[::2, ::2], x[::2, 1::2], x[1::2, ::2], x[1::2, 1::2] */
    int n = 0;
    for (int q = 0; q < 4; ++q) {
        const int di = q / 2, dj = q % 2;
        for (int c = 0; c < C; ++c)
            for (int i = di; i < H; i += 2)
                for (int j = dj; j < W; j += 2)
                    idx[n++] = (c * H + i) * W + j;
    }
}

/* ---- invertible 1x1 convolution --------------------------------------------- */

#define ORC_MAX_CH 64

/* matrix.py:20-50: L (unit lower), U from h */
static void lu_extract(const float *h, int n, float *L, float *U, float *log_diag)
{
    const int n_off = n * (n - 1) / 2;
    memset(L, 0, sizeof(float) * (size_t)n * n);
    memset(U, 0, sizeof(float) * (size_t)n * n);
    for (int i = 0; i < n; ++i) {
        const float ud = expf(h[i]) / 10.0f + 1.0f;          /* :31-32 */
        U[i * n + i] = ud;
        L[i * n + i] = 1.0f;
        log_diag[i] = logf(ud);
    }
    int k = 0;
    for (int r = 0; r < n; ++r)                               /* triu_indices(offset=1) */
        for (int c = r + 1; c < n; ++c) U[r * n + c] = h[n + k++] / 10.0f;
    k = 0;
    for (int r = 1; r < n; ++r)                               /* tril_indices(offset=-1) */
        for (int c = 0; c < r; ++c) L[r * n + c] = h[n + n_off + k++] / 10.0f;
}

void orc_conv1x1_fwd(const float *x, const float *h, float *y, float *logdet,
                     int64_t N, int n, int HW)
{
    const int P = n + n * (n - 1);
    float L[ORC_MAX_CH * ORC_MAX_CH], U[ORC_MAX_CH * ORC_MAX_CH], ldg[ORC_MAX_CH], t[ORC_MAX_CH];
    for (int64_t s = 0; s < N; ++s) {
        lu_extract(h + s * P, n, L, U, ldg);
        float ld = 0.0f;
        for (int i = 0; i < n; ++i) ld += ldg[i];              /* matrix.py:52-64 */
        logdet[s] = ld;
        for (int p = 0; p < HW; ++p) {
            const float *xs = x + s * (int64_t)n * HW + p;
            float *ys = y + s * (int64_t)n * HW + p;
            for (int r = 0; r < n; ++r) {                      /* t = U x */
                float acc = 0.0f;
                for (int c = r; c < n; ++c) acc += U[r * n + c] * xs[(int64_t)c * HW];
                t[r] = acc;
            }
            for (int r = 0; r < n; ++r) {                      /* y = L t  (matrix.py:66-73) */
                float acc = 0.0f;
                for (int c = 0; c <= r; ++c) acc += L[r * n + c] * t[c];
                ys[(int64_t)r * HW] = acc;
            }
        }
    }
}

void orc_conv1x1_inv(const float *y, const float *h, float *x, float *logdet,
                     int64_t N, int n, int HW)
{
    const int P = n + n * (n - 1);
    float L[ORC_MAX_CH * ORC_MAX_CH], U[ORC_MAX_CH * ORC_MAX_CH], ldg[ORC_MAX_CH], t[ORC_MAX_CH];
    for (int64_t s = 0; s < N; ++s) {
        lu_extract(h + s * P, n, L, U, ldg);
        float ld = 0.0f;
        for (int i = 0; i < n; ++i) ld += ldg[i];
        logdet[s] = -ld;                                       /* matrix.py:82 */
        for (int p = 0; p < HW; ++p) {
            const float *ys = y + s * (int64_t)n * HW + p;
            float *xs = x + s * (int64_t)n * HW + p;
            for (int r = 0; r < n; ++r) {                      /* L t = y  (:78) */
                float acc = ys[(int64_t)r * HW];
                for (int c = 0; c < r; ++c) acc -= L[r * n + c] * t[c];
                t[r] = acc;
            }
            for (int r = n - 1; r >= 0; --r) {                 /* U x = t  (:79) */
                float acc = t[r];
                for (int c = r + 1; c < n; ++c) acc -= U[r * n + c] * xs[(int64_t)c * HW];
                xs[(int64_t)r * HW] = acc / U[r * n + r];
            }
        }
    }
}

/* ---- affine ------------------------------------------------------------- */

/* affine.py:33-34 constrain_scale: exp(c0 + u / 2) + m  (u/2 first, then +c0) */
static inline float aff_alpha(float u)
{
    return expf(u / 2.0f + aff_c0()) + AFF_MIN_SCALE;
}

static void affine_row(const float *x, const float *h, float *out, float *ld,
                       int T, int inverse)
{
    float acc = 0.0f;
    for (int t = 0; t < T; ++t) {
        const float alpha = aff_alpha(h[2 * t]);  /* affine.py:40-41 / :51-52 */
        const float log_alpha = logf(alpha);      /* :42 / :53 */
        const float beta = h[2 * t + 1];          /* :44-45 */
        if (!inverse)
            out[t] = alpha * x[t] + beta;         /* :48 */
        else
            out[t] = (x[t] - beta) / alpha;       /* :59 */
        acc += log_alpha;                         /* sum_except_batch :47 / :58 */
    }
    *ld = inverse ? -acc : acc;
}

void orc_affine_fwd(const float *x, const float *h, float *z, float *logdet,
                    int64_t N, int T)
{
    for (int64_t n = 0; n < N; ++n)
        affine_row(x + n * T, h + n * (int64_t)T * 2, z + n * T, logdet + n, T, 0);
}

void orc_affine_inv(const float *z, const float *h, float *x, float *logdet,
                    int64_t N, int T)
{
    for (int64_t n = 0; n < N; ++n)
        affine_row(z + n * T, h + n * (int64_t)T * 2, x + n * T, logdet + n, T, 1);
}

/* ---- rational-quadratic spline ------------------------------------------ */

/* rational_quadratic.py:45-54 compute_bins(u, minimum, maximum).
 * torch.softmax over the last dim as ATen's CPU kernel evaluates it (the
 * reference's CPU path is the parity target): subtract the max, exp, sum the K
 * terms in index order, then MULTIPLY by the reciprocal of the sum
 * (measured in the build container: e * (1 / sum) reproduces torch.softmax
 * bit-for-bit on 89 % of rows given torch's own exp, e / sum on 70 %). */
static void rqs_bins(const float *u, int K, float minimum, float maximum,
                     float *bins /* K+1 */, float *sizes /* K */)
{
    float mx = u[0];
    for (int j = 1; j < K; ++j) mx = u[j] > mx ? u[j] : mx;
    float e[ORC_MAX_BINS];
    float sum = 0.0f;
    for (int j = 0; j < K; ++j) {
        e[j] = expf(u[j] - mx);
        sum += e[j];
    }
    const float rsum = 1.0f / sum;
    const float scale = (float)(1.0 - 1e-3 * (double)K); /* python double, cast once */
    const float span = (float)((double)maximum - (double)minimum);
    /* torch.cumsum on the CPU accumulates fp32 inputs in DOUBLE and rounds every prefix once (ATen
     * ReduceOpsKernel.cpp, cumsum_cpu_kernel: acc_type<float, false>; checked in the build container:
     * torch.cumsum(fp32) == (fp32) numpy.cumsum(fp64) on 100 % of 8-term rows, the fp32 running sum on 80 %) */
    double run = 0.0;
    bins[0] = 0.0f;                                  /* F.pad(..., (1, 0)) :49 */
    for (int j = 0; j < K; ++j) {
        const float sm = e[j] * rsum;                /* :46 */
        const float w = RQS_MIN_BIN + scale * sm;    /* :47 */
        run = run + (double)w;                       /* cumsum :48 */
        bins[j + 1] = (float)run;
    }
    for (int j = 0; j <= K; ++j)
        bins[j] = span * bins[j] + minimum;          /* :50 */
    bins[0] = minimum;                               /* :51 */
    bins[K] = maximum;                               /* :52 */
    for (int j = 0; j < K; ++j)
        sizes[j] = bins[j + 1] - bins[j];            /* :53 */
}

/* F.softplus(x) with beta=1, threshold=20 */
static inline float softplusf(float v)
{
    return v > 20.0f ? v : log1pf(expf(v));
}

/* rational_quadratic.py:56-63 */
static inline float rqs_log_det(float s, float dk, float dk1, float xi, float q,
                                float term1)
{
    const float one_m_xi = 1.0f - xi;
    const float inner = dk1 * (xi * xi) + (2.0f * s) * q + dk * (one_m_xi * one_m_xi);
    const float log_num = 2.0f * logf(s) + logf(inner);
    const float log_den = 2.0f * logf(s + term1 * q);
    return log_num - log_den;
}

/* one in-bounds element. rational_quadratic.py:65-110 (fwd) / :130-182 (inv);
 * parameter split :112-128 / :184-200. */
typedef struct {
    float bin_x[ORC_MAX_BINS + 1], bin_w[ORC_MAX_BINS];
    float bin_y[ORC_MAX_BINS + 1], bin_h[ORC_MAX_BINS];
    float delta[ORC_MAX_BINS + 1];
} rqs_knots;

/* knots, bin sizes and derivatives of one element.
 * rational_quadratic.py:75-77 (== :140-142), parameter split :124-127 */
static void rqs_build(const float *h, int K, float boundary, rqs_knots *kn)
{
    float ux[ORC_MAX_BINS] = {0}, uxy[ORC_MAX_BINS] = {0};
    memset(kn, 0, sizeof(*kn));
    const float c = (float)rqs_boundary_u_delta();
    for (int j = 0; j < K; ++j) {
        ux[j] = h[j];
        uxy[j] = h[j] + h[K + j] / 1000.0f;          /* :76 / :141  u_x + u_y / 1000 */
    }
    rqs_bins(ux, K, -boundary, boundary, kn->bin_x, kn->bin_w);   /* :75 */
    rqs_bins(uxy, K, -boundary, boundary, kn->bin_y, kn->bin_h);  /* :76 */
    for (int j = 0; j <= K; ++j) {
        /* u_d = pad(h[2K:], (1,1), value=c)  :127 ; deltas :77 */
        const float ud = (j == 0 || j == K) ? c : h[2 * K + j - 1];
        kn->delta[j] = RQS_MIN_DELTA + softplusf(c + ud / 1000.0f);
    }
}

void orc_rqs_knots(const float *h, int64_t M, int K, float boundary,
                   float *bin_x, float *bin_y, float *delta)
{
    const int P = 3 * K - 1;
    for (int64_t m = 0; m < M; ++m) {
        rqs_knots kn;
        rqs_build(h + m * P, K, boundary, &kn);
        memcpy(bin_x + m * (K + 1), kn.bin_x, sizeof(float) * (size_t)(K + 1));
        memcpy(bin_y + m * (K + 1), kn.bin_y, sizeof(float) * (size_t)(K + 1));
        memcpy(delta + m * (K + 1), kn.delta, sizeof(float) * (size_t)(K + 1));
    }
}

static void rqs_element(float v, const float *h, int K, float boundary,
                        int inverse, float *out, float *ld, int32_t *kout)
{
    rqs_knots kn;
    rqs_build(h, K, boundary, &kn);
    const float *bin_x = kn.bin_x, *bin_w = kn.bin_w;
    const float *bin_y = kn.bin_y, *bin_h = kn.bin_h;
    const float *delta = kn.delta;

    /* searchsorted(bins, v) - 1 with right=False: #(bins < v) - 1   :82 / :147 */
    const float *search = inverse ? bin_y : bin_x;
    int cnt = 0;
    for (int j = 0; j <= K; ++j) cnt += (search[j] < v);
    const int k = cnt - 1;
    *kout = k;

    const float by = bin_y[k], bx = bin_x[k];
    const float hk = bin_h[k], wk = bin_w[k];
    const float dk = delta[k], dk1 = delta[k + 1];
    const float s = hk / wk;                             /* :94 / :159 */
    const float term1 = dk1 + dk - 2.0f * s;             /* :97 / :162 */

    if (!inverse) {
        float xi = (v - bx) / wk;                        /* :99 */
        xi = clipf(xi, 0.0f, 1.0f);                      /* :100 */
        const float q = xi * (1.0f - xi);                /* :101 */
        const float num0 = hk * (s * (xi * xi) + dk * q);/* :104 */
        const float den0 = s + term1 * q;                /* :105 */
        *out = by + num0 / den0;                         /* :106 */
        *ld = rqs_log_det(s, dk, dk1, xi, q, term1);     /* :109 */
    } else {
        const float term0 = v - by;                      /* :164 */
        const float term2 = hk * dk;                     /* :165 */
        const float a = (hk * s - term2) + term0 * term1;/* :167 */
        const float b = term2 - term0 * term1;           /* :168 */
        const float cc = (-s) * term0;                   /* :169 */
        float r = sqrtf(b * b - (4.0f * a) * cc);        /* :171 */
        r = r < 0.0f ? 0.0f : r;                         /* clip(min=0) */
        float xi = (2.0f * cc) / ((-b) - r);             /* :173 */
        xi = clipf(xi, 0.0f, 1.0f);                      /* :174 */
        const float q = xi * (1.0f - xi);                /* :175 */
        *out = xi * wk + bx;                             /* :178 */
        *ld = -rqs_log_det(s, dk, dk1, xi, q, term1);    /* :181 */
    }
}

static void rqs_row(const float *x, const float *h, float *out, float *ld,
                    float *ld_el, int32_t *bin_idx, int T, int K, float boundary,
                    int inverse)
{
    const int P = 3 * K - 1;
    float acc = 0.0f;
    for (int t = 0; t < T; ++t) {
        const float v = x[t];
        float o = v, l = 0.0f;                  /* spline/base.py:54-55 / :66-67 */
        int32_t k = -1;
        /* strict bounds mask  spline/base.py:29-33 */
        if (v > -boundary && v < boundary)
            rqs_element(v, h + (int64_t)t * P, K, boundary, inverse, &o, &l, &k);
        out[t] = o;
        if (ld_el) ld_el[t] = l;
        if (bin_idx) bin_idx[t] = k;
        acc += l;                               /* sum_except_batch :59 / :71 */
    }
    *ld = acc;
}

void orc_rqs_fwd(const float *x, const float *h, float *z, float *logdet,
                 float *logdet_el, int32_t *bin_idx,
                 int64_t N, int T, int K, float boundary)
{
    const int P = 3 * K - 1;
    for (int64_t n = 0; n < N; ++n)
        rqs_row(x + n * T, h + n * (int64_t)T * P, z + n * T, logdet + n,
                logdet_el ? logdet_el + n * T : NULL,
                bin_idx ? bin_idx + n * T : NULL, T, K, boundary, 0);
}

void orc_rqs_inv(const float *z, const float *h, float *x, float *logdet,
                 float *logdet_el, int32_t *bin_idx,
                 int64_t N, int T, int K, float boundary)
{
    const int P = 3 * K - 1;
    for (int64_t n = 0; n < N; ++n)
        rqs_row(z + n * T, h + n * (int64_t)T * P, x + n * T, logdet + n,
                logdet_el ? logdet_el + n * T : NULL,
                bin_idx ? bin_idx + n * T : NULL, T, K, boundary, 1);
}

/* ---- linear rational spline ---------------------------------
 * transformers/spline/linear_rational.py:9-182.  4K parameters per element:
 * [u_x (K) | u_y (K) | u_lambda (K) | u_d (K-1) | u_w0]. */
#define LRS_MIN_BIN ((float)1e-2)
#define LRS_MIN_D ((float)1e-5)
#define LRS_EPS ((float)5e-10)
static inline double lrs_const(void) { return log(exp(1.0 - 1e-5) - 1.0); }   /* :23 */

/* compute_bins :68-76 (softmax as ATen's CPU kernel, see rqs_bins) */
static void lrs_bins(const float *u, int K, float minimum, float maximum, float *bins /* K+1 */)
{
    float mx = u[0];
    for (int j = 1; j < K; ++j) mx = u[j] > mx ? u[j] : mx;
    float e[ORC_MAX_BINS];
    float sum = 0.0f;
    for (int j = 0; j < K; ++j) {
        e[j] = expf(u[j] - mx);
        sum += e[j];
    }
    const float rsum = 1.0f / sum;
    const float scale = (float)(1.0 - 1e-2 * (double)K);
    const float span = (float)((double)maximum - (double)minimum);
    double run = 0.0;                                /* cumsum :71 accumulates in double on the CPU (see rqs_bins) */
    bins[0] = 0.0f;
    for (int j = 0; j < K; ++j) {
        run = run + (double)(LRS_MIN_BIN + scale * (e[j] * rsum));
        bins[j + 1] = (float)run;
    }
    for (int j = 0; j <= K; ++j) bins[j] = span * bins[j] + minimum;
    bins[0] = minimum;
    bins[K] = maximum;
}

static void lrs_element(float v, const float *h, int K, float boundary, int inverse,
                        float *out, float *ld)
{
    float ux[ORC_MAX_BINS] = {0}, uxy[ORC_MAX_BINS] = {0}, kx[ORC_MAX_BINS + 1], ky[ORC_MAX_BINS + 1];
    float kd[ORC_MAX_BINS + 1], w[ORC_MAX_BINS + 1];
    const float c = (float)lrs_const();
    for (int j = 0; j < K; ++j) {
        ux[j] = h[j];
        uxy[j] = h[j] + h[K + j] / 100.0f;                          /* :87 */
    }
    lrs_bins(ux, K, -boundary, boundary, kx);                       /* :86 */
    lrs_bins(uxy, K, -boundary, boundary, ky);
    kd[0] = 1.0f;                                                   /* pad value 1.0, :80 */
    kd[K] = 1.0f;
    for (int j = 1; j < K; ++j)
        kd[j] = softplusf(c + h[3 * K + j - 1] / 100.0f) + LRS_MIN_D;   /* :79, :89 */
    const float w0 = softplusf(h[4 * K - 1]);                       /* :41 */
    for (int j = 0; j <= K; ++j) w[j] = w0 * sqrtf(kd[0] / kd[j]);  /* :42 */

    const float *search = inverse ? ky : kx;                        /* searchsorted left :105 / :150 */
    int cnt = 0;
    for (int j = 0; j <= K; ++j) cnt += (search[j] < v);
    const int k = cnt - 1;
    const float lam = 1.0f / (1.0f + expf(-h[2 * K + k]));          /* torch.sigmoid :88 */
    const float wk = w[k], wk1 = w[k + 1];
    const float xk = kx[k], xk1 = kx[k + 1], yk = ky[k], yk1 = ky[k + 1];
    const float dk = kd[k], dk1 = kd[k + 1];
    const float one_m = 1.0f - lam;
    const float ym = (one_m * wk * yk + lam * wk1 * yk1) / (one_m * wk + lam * wk1);        /* :58-61 */
    const float wm = (lam * wk * dk + one_m * wk1 * dk1) * ((xk1 - xk) / (yk1 - yk));       /* :62-67 */
    const float dx = xk1 - xk;
    if (!inverse) {
        const float phi = (v - xk) / dx;                            /* :110 */
        if (!(phi > lam)) {                                         /* :113-121 */
            const float den = wk * (lam - phi) + wm * phi;
            *out = (wk * yk * (lam - phi) + wm * ym * phi) / den;
            *ld = logf(lam * wk * wm * (ym - yk)) - logf(den * den + LRS_EPS) - logf(dx);
        } else {                                                    /* :123-131 */
            const float den = wm * (1.0f - phi) + wk1 * (phi - lam);
            *out = (wm * ym * (1.0f - phi) + wk1 * yk1 * (phi - lam)) / den;
            *ld = logf(one_m * wm * wk1 * (yk1 - ym)) - logf(den * den + LRS_EPS) - logf(dx);
        }
    } else {
        if (!(v > ym)) {                                            /* :157-166 */
            const float den = wk * (yk - v) + wm * (v - ym);
            *out = (lam * wk * (yk - v)) / den * dx + xk;
            *ld = logf(lam * wk * wm * (ym - yk)) - logf(den * den + LRS_EPS) + logf(dx);
        } else {                                                    /* :168-176 */
            const float den = wk1 * (yk1 - v) + wm * (v - ym);
            *out = (lam * wk1 * (yk1 - v) + wm * (v - ym)) / den * dx + xk;
            *ld = logf(one_m * wm * wk1 * (yk1 - ym)) - logf(den * den + LRS_EPS) + logf(dx);
        }
    }
}

static void lrs_row(const float *x, const float *h, float *out, float *ld, int T, int K,
                    float boundary, int inverse)
{
    const int P = 4 * K;
    float acc = 0.0f;
    for (int t = 0; t < T; ++t) {
        const float v = x[t];
        float o = v, l = 0.0f;                  /* spline/base.py:54-55 */
        if (v > -boundary && v < boundary)      /* strict, base.py:29-33 */
            lrs_element(v, h + (int64_t)t * P, K, boundary, inverse, &o, &l);
        out[t] = o;
        acc += l;
    }
    *ld = acc;
}

void orc_lrs_fwd(const float *x, const float *h, float *z, float *logdet,
                 int64_t N, int T, int K, float boundary)
{
    for (int64_t n = 0; n < N; ++n)
        lrs_row(x + n * T, h + n * (int64_t)T * 4 * K, z + n * T, logdet + n, T, K, boundary, 0);
}

void orc_lrs_inv(const float *z, const float *h, float *x, float *logdet,
                 int64_t N, int T, int K, float boundary)
{
    for (int64_t n = 0; n < N; ++n)
        lrs_row(z + n * T, h + n * (int64_t)T * 4 * K, x + n * T, logdet + n, T, K, boundary, 1);
}

/* ---- reverse mode ------------------------------------------
 * The reference has no backward code of its own: its gradients are what torch.autograd
 * derives from the forward graphs restated above.  The functions below are the
 * hand-derived reverse mode of exactly those graphs (same masks, same clip
 * sub-gradients, same constant knots), pinned to the reference's autograd outputs in
 * tests/golden/grads*.npz.  Upstream: gz (N,T) = dL/d out, gld (N,) = dL/d logdet.
 * Outputs: gx (N,T) = dL/d x, gh (N,T,P) = dL/d h (both OVERWRITTEN). */

void orc_affine_bwd(const float *x, const float *h, const float *gz, const float *gld,
                    float *gx, float *gh, int64_t N, int T, int inverse)
{
    for (int64_t n = 0; n < N; ++n)
        for (int t = 0; t < T; ++t) {
            const int64_t i = n * T + t;
            const float e = expf(h[2 * i] / 2.0f + aff_c0());   /* affine.py:33-34 */
            const float alpha = e + AFF_MIN_SCALE;
            const float beta = h[2 * i + 1];
            float galpha;
            if (!inverse) {               /* out = alpha x + beta, ld = +sum log alpha */
                gx[i] = gz[i] * alpha;
                gh[2 * i + 1] = gz[i];
                galpha = gz[i] * x[i] + gld[n] / alpha;
            } else {                      /* out = (x - beta) / alpha, ld = -sum log alpha */
                const float r = gz[i] / alpha;
                gx[i] = r;
                gh[2 * i + 1] = -r;
                galpha = -r * ((x[i] - beta) / alpha) - gld[n] / alpha;
            }
            gh[2 * i] = galpha * e * 0.5f;      /* d alpha / d u = exp(c + u/2) / 2 */
        }
}

/* softmax weights of compute_bins (same arithmetic as rqs_bins) */
static void rqs_softmax(const float *u, int K, float *sm)
{
    float mx = u[0];
    for (int j = 1; j < K; ++j) mx = u[j] > mx ? u[j] : mx;
    float sum = 0.0f;
    for (int j = 0; j < K; ++j) {
        sm[j] = expf(u[j] - mx);
        sum += sm[j];
    }
    const float rsum = 1.0f / sum;
    for (int j = 0; j < K; ++j) sm[j] = sm[j] * rsum;
}

/* F.softplus'(t), beta = 1, threshold = 20 (ATen: z = exp(t); z / (z + 1)) */
static inline float softplus_grad(float t)
{
    if (t > 20.0f) return 1.0f;
    const float z = expf(t);
    return z / (z + 1.0f);
}

/* One in-box element.  F(x, theta) = rqs_forward_1d output, L(x, theta) = its log-det.
 * forward direction:  out = F(v),  ld = L(v):        upstream (A, B) on (out, ld).
 * inverse direction:  out = X with F(X) = v,  ld = -L(X)  (rational_quadratic.py:130-182
 *   solves the same rational quadratic in closed form); by implicit differentiation
 *   dX/dv = 1/F_x,  dX/dtheta = -F_theta/F_x,  so with G = A - B L_x the gradients are
 *   those of the forward direction for the upstream pair (-G/F_x, -B) at x = X, and
 *   gv = G/F_x. */
static void rqs_element_bwd(float v, const float *h, int K, float boundary, int inverse,
                            float A, float B, float *gv, float *gh)
{
    const int P = 3 * K - 1;
    const float c = (float)rqs_boundary_u_delta();
    const float scale = (float)(1.0 - 1e-3 * (double)K);
    const float span = (float)((double)boundary + (double)boundary);
    rqs_knots kn;
    rqs_build(h, K, boundary, &kn);
    for (int j = 0; j < P; ++j) gh[j] = 0.0f;

    float xin = v;
    int k;
    if (!inverse) {
        int cnt = 0;
        for (int j = 0; j <= K; ++j) cnt += (kn.bin_x[j] < v);
        k = cnt - 1;
    } else {
        float o, l;
        int32_t kk;
        rqs_element(v, h, K, boundary, 1, &o, &l, &kk);
        k = kk;
        xin = o;                                   /* X = F^-1(v) */
    }
    const float bx = kn.bin_x[k];
    const float wk = kn.bin_w[k], hk = kn.bin_h[k];
    const float dk = kn.delta[k], dk1 = kn.delta[k + 1];
    const float s = hk / wk;
    const float term1 = dk1 + dk - 2.0f * s;
    const float xi_raw = (xin - bx) / wk;
    const float xi = clipf(xi_raw, 0.0f, 1.0f);
    const int pass = (xi_raw >= 0.0f && xi_raw <= 1.0f);   /* torch.clip sub-gradient */
    const float omx = 1.0f - xi;
    const float q = xi * omx;
    const float inner2 = s * (xi * xi) + dk * q;
    const float num0 = hk * inner2;
    const float den0 = s + term1 * q;
    const float inner = dk1 * (xi * xi) + (2.0f * s) * q + dk * (omx * omx);

    /* d F / d xi and d L / d xi (everything else held fixed) */
    const float dq = 1.0f - 2.0f * xi;
    const float F_xi = (hk * (2.0f * s * xi + dk * dq)) / den0 - (num0 / (den0 * den0)) * (term1 * dq);
    const float L_xi = (2.0f * dk1 * xi + 2.0f * s * dq - 2.0f * dk * omx) / inner
                       - 2.0f * (term1 * dq) / den0;
    const float F_x = pass ? F_xi / wk : 0.0f;
    const float L_x = pass ? L_xi / wk : 0.0f;
    if (inverse) {
        const float G = A - B * L_x;               /* d loss / d X, with ld_inv = -L(X) */
        *gv = G / F_x;
        A = -(G / F_x);
        B = -B;
    } else {
        *gv = A * F_x + B * L_x;
    }

    /* reverse sweep of the forward graph for upstream (A, B) */
    float g_s = 0.0f, g_q = 0.0f, g_xi = 0.0f, g_d0 = 0.0f, g_d1 = 0.0f;
    float g_hk = 0.0f, g_wk = 0.0f, g_by = A;
    const float g_num0 = A / den0;
    float g_den0 = -A * num0 / (den0 * den0);
    g_s += B * 2.0f / s;
    const float g_inner = B / inner;
    g_den0 += -2.0f * B / den0;
    g_d1 += g_inner * (xi * xi);
    g_s += g_inner * 2.0f * q;
    g_q += g_inner * 2.0f * s;
    g_d0 += g_inner * (omx * omx);
    g_xi += g_inner * (2.0f * dk1 * xi - 2.0f * dk * omx);
    g_s += g_den0;
    const float g_t1 = g_den0 * q;
    g_q += g_den0 * term1;
    g_hk += g_num0 * inner2;
    const float g_in2 = g_num0 * hk;
    g_s += g_in2 * (xi * xi);
    g_xi += g_in2 * 2.0f * s * xi;
    g_d0 += g_in2 * q;
    g_q += g_in2 * dk;
    g_d1 += g_t1;
    g_d0 += g_t1;
    g_s -= 2.0f * g_t1;
    g_xi += g_q * dq;
    float g_bx = 0.0f;
    if (pass) {                                    /* xi = (x - bx) / wk */
        g_bx -= g_xi / wk;
        g_wk -= g_xi * xi_raw / wk;
    }
    g_hk += g_s / wk;                              /* s = hk / wk */
    g_wk -= g_s * s / wk;
    /* wk = bin_x[k+1] - bin_x[k], hk = bin_y[k+1] - bin_y[k]; knots 0 and K are constants */
    const float g_x0 = g_bx - g_wk, g_x1 = g_wk;
    const float g_y0 = g_by - g_hk, g_y1 = g_hk;
    const float gcx0 = (k >= 1) ? span * g_x0 : 0.0f, gcx1 = (k + 1 <= K - 1) ? span * g_x1 : 0.0f;
    const float gcy0 = (k >= 1) ? span * g_y0 : 0.0f, gcy1 = (k + 1 <= K - 1) ? span * g_y1 : 0.0f;
    float ux[ORC_MAX_BINS], uy[ORC_MAX_BINS], smx[ORC_MAX_BINS], smy[ORC_MAX_BINS];
    for (int j = 0; j < K; ++j) {
        ux[j] = h[j];
        uy[j] = h[j] + h[K + j] / 1000.0f;
    }
    rqs_softmax(ux, K, smx);
    rqs_softmax(uy, K, smy);
    float gwx[ORC_MAX_BINS], gwy[ORC_MAX_BINS], dotx = 0.0f, doty = 0.0f;
    for (int i = 0; i < K; ++i) {                  /* cumsum: knot j sums bins i < j */
        gwx[i] = scale * ((i < k ? gcx0 : 0.0f) + (i < k + 1 ? gcx1 : 0.0f));
        gwy[i] = scale * ((i < k ? gcy0 : 0.0f) + (i < k + 1 ? gcy1 : 0.0f));
        dotx += smx[i] * gwx[i];
        doty += smy[i] * gwy[i];
    }
    for (int i = 0; i < K; ++i) {                  /* softmax backward */
        const float gux = smx[i] * (gwx[i] - dotx);
        const float guy = smy[i] * (gwy[i] - doty);
        gh[i] = gux + guy;                         /* u_y enters as u_x + u_y / 1000 */
        gh[K + i] = guy / 1000.0f;
    }
    if (k >= 1)
        gh[2 * K + k - 1] = g_d0 * softplus_grad(c + h[2 * K + k - 1] / 1000.0f) / 1000.0f;
    if (k <= K - 2)
        gh[2 * K + k] = g_d1 * softplus_grad(c + h[2 * K + k] / 1000.0f) / 1000.0f;
}

void orc_rqs_bwd(const float *x, const float *h, const float *gz, const float *gld,
                 float *gx, float *gh, int64_t N, int T, int K, float boundary, int inverse)
{
    const int P = 3 * K - 1;
    for (int64_t n = 0; n < N; ++n)
        for (int t = 0; t < T; ++t) {
            const int64_t i = n * T + t;
            const float v = x[i];
            if (v > -boundary && v < boundary) {
                rqs_element_bwd(v, h + i * P, K, boundary, inverse, gz[i], gld[n], gx + i, gh + i * P);
            } else {                               /* identity outside the box, base.py:54-55 */
                gx[i] = gz[i];
                for (int j = 0; j < P; ++j) gh[i * P + j] = 0.0f;
            }
        }
}

/* ---- base distribution --------------------------------------------------- */

static float gauss_row(const float *z, const float *loc, const float *log_scale, int D)
{
    /* gaussian.py:46-54 */
    const float half_log_2pi = (float)(0.5 * log(2.0 * M_PI));
    float acc = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float scale = expf(log_scale[d]);        /* :37-38 */
        const float t = (z[d] - loc[d]) / scale;
        float e = 0.5f * (t * t);
        e = e + half_log_2pi;
        e = e + log_scale[d];
        acc += -e;                                     /* :53-54 */
    }
    return acc;
}

void orc_diag_gauss_logprob(const float *z, const float *loc,
                            const float *log_scale, float *out,
                            int64_t N, int D)
{
    for (int64_t n = 0; n < N; ++n)
        out[n] = gauss_row(z + n * D, loc, log_scale, D);
}

/* ---- conditioner ---------------------------------------------------------- */

void orc_feedforward_row(const float *in_row, int n_linear, const int32_t *dims,
                         const float *const *W, const float *const *b,
                         float *out_row, float *scratch)
{
    /* transforms.py:293-307: Linear, then (Tanh, Linear) pairs.
     * nn.Linear = addmm(bias, x, W^T). */
    int maxd = 0;
    for (int l = 0; l <= n_linear; ++l) maxd = dims[l] > maxd ? dims[l] : maxd;
    float *cur = scratch, *nxt = scratch + maxd;
    const float *src = in_row;
    for (int l = 0; l < n_linear; ++l) {
        const int in = dims[l], out = dims[l + 1];
        float *dst = (l == n_linear - 1) ? out_row : nxt;
        for (int o = 0; o < out; ++o) {
            float acc = 0.0f;
            const float *w = W[l] + (int64_t)o * in;
            for (int i = 0; i < in; ++i) acc += src[i] * w[i];
            acc = acc + b[l][o];
            dst[o] = (l == n_linear - 1) ? acc : tanhf(acc);
        }
        if (l != n_linear - 1) {
            float *tmp = cur; cur = nxt; nxt = tmp;
            src = cur;
        }
    }
}

/* ---- layers ---------------------------------------------------------------- */

typedef struct {
    float *cond_in;   /* S + C */
    float *h;         /* T * P */
    float *xb;        /* T */
    float *zb;        /* T */
    float *scratch;   /* 2 * max width */
    float *row_a;     /* D */
    float *row_b;     /* D */
} row_ws;

static int layer_P(const orc_layer *L)
{
    switch (L->kind) {
    case ORC_AFFINE_COUPLING: return 2;
    case ORC_RQS_COUPLING: return 3 * L->K - 1;
    case ORC_LRS_COUPLING: return 4 * L->K;
    case ORC_SHIFT_COUPLING: return 1;
    default: return 0;
    }
}

static void ws_alloc(row_ws *w, const orc_layer *layers, int n_layers, int D, int C)
{
    int max_in = 1, max_h = 1, max_t = 1, max_w = 1;
    for (int i = 0; i < n_layers; ++i) {
        const orc_layer *L = &layers[i];
        if (L->n_linear <= 0) continue;
        if (L->kind < ORC_AFFINE_COUPLING) {
            /* context-conditioned elementwise layer: h = Linear(context) (D,2) */
            if (2 * D > max_h) max_h = 2 * D;
            if (C > max_in) max_in = C;
        } else {
            if (L->S + C > max_in) max_in = L->S + C;
            if (L->T * layer_P(L) > max_h) max_h = L->T * layer_P(L);
            if (L->T > max_t) max_t = L->T;
        }
        for (int l = 0; l <= L->n_linear; ++l)
            if (L->dims[l] > max_w) max_w = L->dims[l];
    }
    w->cond_in = (float *)malloc(sizeof(float) * (size_t)max_in);
    w->h = (float *)malloc(sizeof(float) * (size_t)max_h);
    w->xb = (float *)malloc(sizeof(float) * (size_t)max_t);
    w->zb = (float *)malloc(sizeof(float) * (size_t)max_t);
    w->scratch = (float *)malloc(sizeof(float) * 2 * (size_t)max_w);
    w->row_a = (float *)malloc(sizeof(float) * (size_t)D);
    w->row_b = (float *)malloc(sizeof(float) * (size_t)D);
}

static void ws_free(row_ws *w)
{
    free(w->cond_in); free(w->h); free(w->xb); free(w->zb);
    free(w->scratch); free(w->row_a); free(w->row_b);
}

/* One layer on one row: in -> out (distinct buffers), returns the layer's
 * log-det.  direction 0 = Bijection.forward, 1 = Bijection.inverse. */
static float layer_row(const orc_layer *L, const float *in, const float *ctx, int C,
                       float *out, int D, int direction, row_ws *w)
{
    float ld = 0.0f;
    switch (L->kind) {
    case ORC_ELEMENTWISE_AFFINE:
    case ORC_ELEMENTWISE_INVERSE_AFFINE: {
        /* layers_base.py:300-318: h = value broadcast over the batch; the
         * transformer is Affine (ElementwiseAffine) or InverseAffine (ActNorm,
         * affine.py:62-70: forward <-> inverse swapped). */
        const int inv = (L->kind == ORC_ELEMENTWISE_INVERSE_AFFINE) ? !direction : direction;
        const float *h = L->value;
        if (!h) {
            /* context given at construction: h = conditioner_transform(None, context)
             * (layers_base.py:283-296, :304-310; default class Linear) */
            orc_feedforward_row(ctx, L->n_linear, L->dims, L->W, L->b, w->h, w->scratch);
            h = w->h;
        }
        affine_row(in, h, out, &ld, D, inv);
        break;
    }
    case ORC_PERMUTATION: {
        /* matrix/base.py:22-38 + permutation.py:19-23; log-det exactly 0 */
        const int32_t *p = direction ? L->perm_inv : L->perm_fwd;
        for (int j = 0; j < D; ++j) out[j] = in[p[j]];
        ld = 0.0f;
        break;
    }
    case ORC_AFFINE_COUPLING:
    case ORC_RQS_COUPLING:
    case ORC_LRS_COUPLING:
    case ORC_SHIFT_COUPLING: {
        if (L->autoregressive) {
            /* MaskedAutoregressiveBijection (layers_base.py:166-225): the conditioner is MADE
             * (masked weights arrive pre-multiplied), every element is transformed.  One map is a
             * single parallel pass; the other walks the D elements in order, re-running the
             * conditioner on the partially inverted row and keeping the LAST pass's log-det
             * (:213-221).  autoregressive == 1: forward parallel / inverse sequential;
             * == 2: exchanged (InverseMaskedAutoregressiveBijection :227-234).
             * swap_transformer: the transformer's own maps are exchanged (InverseAffine). */
            const int sequential = (L->autoregressive == 1) ? (direction == 1) : (direction == 0);
            const int tdir = (sequential ? 1 : 0) ^ (L->swap_transformer ? 1 : 0);
            memcpy(out, in, sizeof(float) * (size_t)D);
            const int passes = sequential ? D : 1;
            for (int i = 0; i < passes; ++i) {
                for (int s = 0; s < D; ++s) w->cond_in[s] = out[s];
                for (int c = 0; c < C; ++c) w->cond_in[D + c] = ctx[c];
                orc_feedforward_row(w->cond_in, L->n_linear, L->dims, L->W, L->b, w->h, w->scratch);
                for (int t = 0; t < D; ++t) w->xb[t] = out[t];
                if (L->kind == ORC_AFFINE_COUPLING)
                    affine_row(w->xb, w->h, w->zb, &ld, D, tdir);
                else if (L->kind == ORC_RQS_COUPLING)
                    rqs_row(w->xb, w->h, w->zb, &ld, NULL, NULL, D, L->K, L->boundary, tdir);
                else
                    lrs_row(w->xb, w->h, w->zb, &ld, D, L->K, L->boundary, tdir);
                if (sequential) out[i] = w->zb[i];
                else memcpy(out, w->zb, sizeof(float) * (size_t)D);
            }
            break;
        }
        /* layers_base.py:145-163 */
        memcpy(out, in, sizeof(float) * (size_t)D);                  /* clone :146/:156 */
        for (int s = 0; s < L->S; ++s) w->cond_in[s] = in[L->src_idx[s]]; /* :119-121 */
        for (int c = 0; c < C; ++c) w->cond_in[L->S + c] = ctx[c];   /* context.py:58-60 */
        orc_feedforward_row(w->cond_in, L->n_linear, L->dims, L->W, L->b,
                            w->h, w->scratch);                       /* :142 */
        for (int t = 0; t < L->T; ++t) w->xb[t] = in[L->tgt_idx[t]]; /* :123-125 */
        if (L->kind == ORC_AFFINE_COUPLING) {
            affine_row(w->xb, w->h, w->zb, &ld, L->T, direction);
        } else if (L->kind == ORC_RQS_COUPLING) {
            rqs_row(w->xb, w->h, w->zb, &ld, NULL, NULL, L->T, L->K, L->boundary,
                    direction);
        } else if (L->kind == ORC_LRS_COUPLING) {
            lrs_row(w->xb, w->h, w->zb, &ld, L->T, L->K, L->boundary, direction);
        } else {
            /* Shift: affine.py:137-159, log-det 0 */
            for (int t = 0; t < L->T; ++t)
                w->zb[t] = direction ? w->xb[t] - w->h[t] : w->xb[t] + w->h[t];
            ld = 0.0f;
        }
        for (int t = 0; t < L->T; ++t) out[L->tgt_idx[t]] = w->zb[t]; /* :127-129 */
        break;
    }
    default:
        break;
    }
    return ld;
}

static void composition_rows(const orc_layer *layers, int n_layers,
                             const float *x, const float *context, int C,
                             float *z, float *logdet, float *trace_z, float *trace_ld,
                             const float *loc, const float *log_scale, float *log_prob,
                             int64_t N, int D, int direction)
{
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        row_ws w;
        ws_alloc(&w, layers, n_layers, D, C);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int64_t n = 0; n < N; ++n) {
            const float *ctx = context ? context + n * C : NULL;
            float *cur = w.row_a, *nxt = w.row_b;
            memcpy(cur, x + n * D, sizeof(float) * (size_t)D);
            float ld = 0.0f;                                    /* base.py:210 / :227 */
            for (int i = 0; i < n_layers; ++i) {
                const int li = direction ? n_layers - 1 - i : i; /* base.py:228 */
                const float l = layer_row(&layers[li], cur, ctx, C, nxt, D, direction, &w);
                ld = ld + l;                                    /* base.py:222 / :230 */
                float *t = cur; cur = nxt; nxt = t;
                if (trace_z)
                    memcpy(trace_z + ((int64_t)i * N + n) * D, cur, sizeof(float) * (size_t)D);
                if (trace_ld) trace_ld[(int64_t)i * N + n] = l;
            }
            if (z) memcpy(z + n * D, cur, sizeof(float) * (size_t)D);
            if (logdet) logdet[n] = ld;
            if (log_prob)                                       /* flows.py:647-648 */
                log_prob[n] = gauss_row(cur, loc, log_scale, D) + ld;
        }
        ws_free(&w);
    }
}

void orc_composition_forward(const orc_layer *layers, int n_layers,
                             const float *x, const float *context, int C,
                             float *z, float *logdet,
                             float *trace_z, float *trace_ld,
                             int64_t N, int D)
{
    composition_rows(layers, n_layers, x, context, C, z, logdet, trace_z, trace_ld,
                     NULL, NULL, NULL, N, D, 0);
}

void orc_composition_inverse(const orc_layer *layers, int n_layers,
                             const float *z, const float *context, int C,
                             float *x, float *logdet,
                             int64_t N, int D)
{
    composition_rows(layers, n_layers, z, context, C, x, logdet, NULL, NULL,
                     NULL, NULL, NULL, N, D, 1);
}

void orc_flow_log_prob(const orc_layer *layers, int n_layers,
                       const float *loc, const float *log_scale,
                       const float *x, const float *context, int C,
                       float *z, float *log_prob,
                       int64_t N, int D)
{
    composition_rows(layers, n_layers, x, context, C, z, NULL, NULL, NULL,
                     loc, log_scale, log_prob, N, D, 0);
}

void orc_actnorm_init(const float *x, int64_t N, int D, float *value)
{
    /* layers.py:58-68 */
    for (int d = 0; d < D; ++d) {
        double mean = 0.0;
        for (int64_t n = 0; n < N; ++n) mean += (double)x[n * D + d];
        mean /= (double)N;
        float scale = 1.0f;                                  /* :63-64 */
        if (N > 1) {
            double ss = 0.0;
            for (int64_t n = 0; n < N; ++n) {
                const double t = (double)x[n * D + d] - mean;
                ss += t * t;
            }
            scale = (float)sqrt(ss / (double)(N - 1));       /* torch.std, unbiased :66 */
        }
        /* affine.py:36-37 unconstrain_scale */
        value[2 * d] = (logf(scale - AFF_MIN_SCALE) - aff_c0()) * 2.0f;
        value[2 * d + 1] = (float)mean;
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
