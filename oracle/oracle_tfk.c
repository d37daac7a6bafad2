/* oracle_tfk.c -- the C-ABI of include/tfk.h exported by the CPU restatement (TEST INFRASTRUCTURE).
 *
 * SURVEY.md 8(b), last sentence of the C-ABI row: "The same symbols are exported by the CPU restatement build
 * (host pointers, stream = NULL)."  This file gives liboracle.so the minimum set of that row under the SAME names
 * and SAME signatures as libtfk.so (the prototypes are taken from include/tfk.h itself, so a drift is a compile
 * error): tfk_affine_coupling_{fwd,inv}, tfk_shift_coupling_{fwd,inv}, tfk_rqs_coupling_{fwd,inv},
 * tfk_elementwise_affine_{fwd,inv}, tfk_permute, tfk_diag_gauss_logprob, tfk_sum_f32 (+ _ws).  All pointers are
 * HOST pointers, `stream` must be NULL, the work is done by the orc_* restatement (oracle.c, which cites the
 * reference lines).  Used only by tests (the checker side of parity tests written against the tfk_* names); the
 * product path never loads it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../include/tfk.h"
#include "oracle.h"

static const char *g_err = "";

int tfk_abi_version(void) { return TFK_ABI_VERSION; }
const char *tfk_last_error(void) { return g_err; }

static int bad(const char *msg)
{
    g_err = msg;
    return TFK_EINVAL;
}

/* the skeleton of CouplingBijection.forward / inverse (layers_base.py:145-163): z = x.clone(); the T target
 * columns (tgt_idx ascending, NULL = the contiguous tail [D - T, D)) are replaced by transformer(x_B, h);
 * logdet[n] is stored or, with accumulate != 0, added to. */
typedef void (*row_map)(const float *xb, const float *h, float *zb, float *ld, int64_t N, int T, const void *arg);

static int coupling(const float *x, const float *h, float *z, float *logdet, int64_t N, int32_t D,
                    const int32_t *tgt_idx, int32_t T, int32_t accumulate, void *stream, int P, row_map fn,
                    const void *arg, int need_logdet)
{
    if (stream) return bad("CPU restatement: stream must be NULL");
    if (N < 0 || D <= 0 || T <= 0 || T > D) return bad("bad N / D / T");
    if (N == 0) return TFK_OK;
    if (!x || !h || !z || (need_logdet && !logdet)) return bad("null pointer");
    float *xb = (float *)malloc((size_t)N * T * sizeof(float));
    float *zb = (float *)malloc((size_t)N * T * sizeof(float));
    float *ld = (float *)malloc((size_t)N * sizeof(float));
    if (!xb || !zb || !ld) { free(xb); free(zb); free(ld); return bad("out of memory"); }
    for (int64_t n = 0; n < N; ++n)
        for (int t = 0; t < T; ++t)
            xb[n * T + t] = x[n * D + (tgt_idx ? tgt_idx[t] : D - T + t)];
    (void)P;
    fn(xb, h, zb, ld, N, T, arg);
    if (z != x) memcpy(z, x, (size_t)N * D * sizeof(float));
    for (int64_t n = 0; n < N; ++n) {
        for (int t = 0; t < T; ++t)
            z[n * D + (tgt_idx ? tgt_idx[t] : D - T + t)] = zb[n * T + t];
        if (logdet) logdet[n] = accumulate ? logdet[n] + ld[n] : ld[n];
    }
    free(xb); free(zb); free(ld);
    return TFK_OK;
}

static void m_affine_fwd(const float *xb, const float *h, float *zb, float *ld, int64_t N, int T, const void *a)
{ (void)a; orc_affine_fwd(xb, h, zb, ld, N, T); }
static void m_affine_inv(const float *xb, const float *h, float *zb, float *ld, int64_t N, int T, const void *a)
{ (void)a; orc_affine_inv(xb, h, zb, ld, N, T); }
static void m_shift_fwd(const float *xb, const float *h, float *zb, float *ld, int64_t N, int T, const void *a)
{ (void)a; for (int64_t i = 0; i < N * T; ++i) zb[i] = xb[i] + h[i]; for (int64_t n = 0; n < N; ++n) ld[n] = 0.0f; }
static void m_shift_inv(const float *xb, const float *h, float *zb, float *ld, int64_t N, int T, const void *a)
{ (void)a; for (int64_t i = 0; i < N * T; ++i) zb[i] = xb[i] - h[i]; for (int64_t n = 0; n < N; ++n) ld[n] = 0.0f; }
struct rqs_arg { int K; float boundary; };
static void m_rqs_fwd(const float *xb, const float *h, float *zb, float *ld, int64_t N, int T, const void *a)
{ const struct rqs_arg *r = (const struct rqs_arg *)a; orc_rqs_fwd(xb, h, zb, ld, NULL, NULL, N, T, r->K, r->boundary); }
static void m_rqs_inv(const float *xb, const float *h, float *zb, float *ld, int64_t N, int T, const void *a)
{ const struct rqs_arg *r = (const struct rqs_arg *)a; orc_rqs_inv(xb, h, zb, ld, NULL, NULL, N, T, r->K, r->boundary); }

int tfk_affine_coupling_fwd(const float *x, const float *h, float *z, float *logdet, int64_t N, int32_t D,
                            const int32_t *tgt_idx, int32_t T, int32_t accumulate, void *stream)
{ return coupling(x, h, z, logdet, N, D, tgt_idx, T, accumulate, stream, 2, m_affine_fwd, NULL, 1); }
int tfk_affine_coupling_inv(const float *z, const float *h, float *x, float *logdet, int64_t N, int32_t D,
                            const int32_t *tgt_idx, int32_t T, int32_t accumulate, void *stream)
{ return coupling(z, h, x, logdet, N, D, tgt_idx, T, accumulate, stream, 2, m_affine_inv, NULL, 1); }
int tfk_shift_coupling_fwd(const float *x, const float *h, float *z, float *logdet, int64_t N, int32_t D,
                           const int32_t *tgt_idx, int32_t T, int32_t accumulate, void *stream)
{ return coupling(x, h, z, logdet, N, D, tgt_idx, T, accumulate, stream, 1, m_shift_fwd, NULL, !accumulate); }
int tfk_shift_coupling_inv(const float *z, const float *h, float *x, float *logdet, int64_t N, int32_t D,
                           const int32_t *tgt_idx, int32_t T, int32_t accumulate, void *stream)
{ return coupling(z, h, x, logdet, N, D, tgt_idx, T, accumulate, stream, 1, m_shift_inv, NULL, !accumulate); }
int tfk_rqs_coupling_fwd(const float *x, const float *h, float *z, float *logdet, int64_t N, int32_t D,
                         const int32_t *tgt_idx, int32_t T, int32_t K, float boundary, int32_t accumulate, void *stream)
{
    struct rqs_arg a = {K, boundary};
    if (K < 2 || K > 32 || !(boundary > 0.0f)) return bad("bad K / boundary");
    return coupling(x, h, z, logdet, N, D, tgt_idx, T, accumulate, stream, 3 * K - 1, m_rqs_fwd, &a, 1);
}
int tfk_rqs_coupling_inv(const float *z, const float *h, float *x, float *logdet, int64_t N, int32_t D,
                         const int32_t *tgt_idx, int32_t T, int32_t K, float boundary, int32_t accumulate, void *stream)
{
    struct rqs_arg a = {K, boundary};
    if (K < 2 || K > 32 || !(boundary > 0.0f)) return bad("bad K / boundary");
    return coupling(z, h, x, logdet, N, D, tgt_idx, T, accumulate, stream, 3 * K - 1, m_rqs_inv, &a, 1);
}

/* ElementwiseBijection.forward / inverse with global parameters (layers_base.py:300-318): the reference repeats
 * value (D, 2) over the batch and calls the transformer with all D positions as targets. */
static int elementwise(const float *x, const float *value, float *z, float *logdet, int64_t N, int32_t D,
                       int divide, int32_t accumulate, void *stream)
{
    if (stream) return bad("CPU restatement: stream must be NULL");
    if (N < 0 || D <= 0) return bad("bad N / D");
    if (N == 0) return TFK_OK;
    if (!x || !value || !z || !logdet) return bad("null pointer");
    float *row = (float *)malloc((size_t)D * sizeof(float));
    if (!row) return bad("out of memory");
    for (int64_t n = 0; n < N; ++n) {
        float ld;
        if (divide) orc_affine_inv(x + n * D, value, row, &ld, 1, D);
        else orc_affine_fwd(x + n * D, value, row, &ld, 1, D);
        memcpy(z + n * D, row, (size_t)D * sizeof(float));
        logdet[n] = accumulate ? logdet[n] + ld : ld;
    }
    free(row);
    return TFK_OK;
}

int tfk_elementwise_affine_fwd(const float *x, const float *value, float *z, float *logdet, int64_t N, int32_t D,
                               int32_t inverse_affine, int32_t accumulate, void *stream)
{ return elementwise(x, value, z, logdet, N, D, inverse_affine != 0, accumulate, stream); }
int tfk_elementwise_affine_inv(const float *z, const float *value, float *x, float *logdet, int64_t N, int32_t D,
                               int32_t inverse_affine, int32_t accumulate, void *stream)
{ return elementwise(z, value, x, logdet, N, D, inverse_affine == 0, accumulate, stream); }

int tfk_permute(const float *x, const int32_t *perm, float *z, int64_t N, int32_t D, void *stream)
{
    if (stream) return bad("CPU restatement: stream must be NULL");
    if (N < 0 || D <= 0) return bad("bad N / D");
    if (N == 0) return TFK_OK;
    if (!x || !z || x == z) return bad("null or aliased pointer");
    for (int64_t n = 0; n < N; ++n)
        for (int j = 0; j < D; ++j) z[n * D + j] = x[n * D + (perm ? perm[j] : D - 1 - j)];   /* permutation.py:19-23 */
    return TFK_OK;
}

int tfk_diag_gauss_logprob(const float *z, const float *loc, const float *log_scale, const float *logdet_in,
                           float *out, int64_t N, int32_t D, void *stream)
{
    if (stream) return bad("CPU restatement: stream must be NULL");
    if (N < 0 || D <= 0) return bad("bad N / D");
    if (N == 0) return TFK_OK;
    if (!z || !loc || !log_scale || !out) return bad("null pointer");
    float *lp = (float *)malloc((size_t)N * sizeof(float));
    if (!lp) return bad("out of memory");
    orc_diag_gauss_logprob(z, loc, log_scale, lp, N, D);
    for (int64_t n = 0; n < N; ++n) out[n] = logdet_in ? lp[n] + logdet_in[n] : lp[n];     /* flows.py:648 */
    free(lp);
    return TFK_OK;
}

int tfk_sum_f32(const float *in, double *out_scalar, int64_t N, void *stream)
{
    if (stream) return bad("CPU restatement: stream must be NULL");
    if (N < 0 || !out_scalar || (N > 0 && !in)) return bad("bad arguments");
    double acc = 0.0;
    for (int64_t i = 0; i < N; ++i) acc += (double)in[i];
    out_scalar[0] = acc;
    return TFK_OK;
}

int64_t tfk_sum_workspace_bytes(int64_t N) { (void)N; return 0; }

int tfk_sum_f32_ws(const float *in, double *out_scalar, void *workspace, int64_t N, void *stream)
{
    (void)workspace;
    return tfk_sum_f32(in, out_scalar, N, stream);
}
