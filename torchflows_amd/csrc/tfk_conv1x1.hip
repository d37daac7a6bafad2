// tfk_conv1x1.hip -- invertible 1x1 convolution coupling (Glow).
//
// Replaces CouplingBijection.forward / inverse (layers_base.py:145-163) around
// Invertible1x1ConvolutionTransformer (transformers/linear/convolution.py:8-70) and
// LUTransformer (transformers/linear/matrix.py:11-99): the target part of a row is an image
// (n channels x HW pixels, channel-major); per SAMPLE the conditioner gives n + n(n-1)
// numbers h = [diag logits | U above the diagonal, row-major | L below the diagonal,
// row-major]; U_ii = exp(h_i)/10 + 1, off-diagonals h/10, L has a unit diagonal.
//   fwd: y = L (U x) at every pixel, log-det = + sum_i log U_ii   (ONCE per sample: the
//   inv: U x = L^-1 y,               log-det = - sum_i log U_ii    reference does not
//                                                                  multiply by HW; kept)
// One lane per (sample, pixel): the n channel values sit in registers (n <= 16), the
// sample's L/U entries are broadcast loads.  Reads of a channel across lanes are contiguous
// (consecutive pixels), so the traffic is coalesced; bound by HBM: 4*(2T or 2D) B per row.
#include "tfk_common.h"

namespace tfk {

constexpr int kMaxCh = 16;

template <bool INVERSE>
__global__ __launch_bounds__(kBlock) void k_conv1x1_coupling(
    const float *x, const float *__restrict__ h, float *z, float *logdet, long long N, int D,
    const int *__restrict__ tgt_idx, int T, int n, int HW, int accumulate, int inplace)
{
    extern __shared__ unsigned char is_tgt[];
    const bool use_mask = (tgt_idx != nullptr) && !inplace;
    if (use_mask) {
        for (int e = threadIdx.x; e < D; e += kBlock) is_tgt[e] = 0;
        __syncthreads();
        for (int t = threadIdx.x; t < T; t += kBlock) is_tgt[tgt_idx[t]] = 1;
        __syncthreads();
    }
    const int P = n + n * (n - 1);
    const int n_off = n * (n - 1) / 2;
    const long long total = N * (long long)HW;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (long long)gridDim.x * kBlock) {
        const long long row = i / HW;
        const int p = (int)(i - row * HW);
        const float *xr = x + row * D;
        float *zr = z + row * D;
        const float *hr = h + row * P;
        float v[kMaxCh];
#pragma unroll
        for (int c = 0; c < kMaxCh; ++c)
            if (c < n) {
                const int t = c * HW + p;
                v[c] = xr[tgt_idx ? tgt_idx[t] : D - T + t];
            }
        // U entry (r, c), r < c: h[n + r*n - r(r+1)/2 + (c - r - 1)]   (triu_indices, offset 1)
        // L entry (r, c), c < r: h[n + n_off + r(r-1)/2 + c]           (tril_indices, offset -1)
        float ld = 0.0f;
        if (!INVERSE) {
            // t = U v  (upper triangular, row r uses v[r..n-1]; in place top-down is safe)
#pragma unroll
            for (int r = 0; r < kMaxCh; ++r)
                if (r < n) {
                    const float ud = expf(hr[r]) / 10.0f + 1.0f;        // matrix.py:31-32
                    ld += logf(ud);
                    float acc = ud * v[r];
                    const int base = n + r * n - (r * (r + 1)) / 2 - r - 1;
#pragma unroll
                    for (int c = 0; c < kMaxCh; ++c)
                        if (c > r && c < n) acc = fmaf(hr[base + c] / 10.0f, v[c], acc);
                    v[r] = acc;
                }
            // y = L t  (unit lower; bottom-up so the inputs of row r are still the old values)
#pragma unroll
            for (int rr = 0; rr < kMaxCh; ++rr) {
                const int r = kMaxCh - 1 - rr;
                if (r < n) {
                    float acc = v[r];
                    const int base = n + n_off + (r * (r - 1)) / 2;
#pragma unroll
                    for (int c = 0; c < kMaxCh; ++c)
                        if (c < r) acc = fmaf(hr[base + c] / 10.0f, v[c], acc);
                    v[r] = acc;
                }
            }
        } else {
            // L t = y: forward substitution, top-down
#pragma unroll
            for (int r = 0; r < kMaxCh; ++r)
                if (r < n) {
                    float acc = v[r];
                    const int base = n + n_off + (r * (r - 1)) / 2;
#pragma unroll
                    for (int c = 0; c < kMaxCh; ++c)
                        if (c < r) acc = fmaf(-(hr[base + c] / 10.0f), v[c], acc);
                    v[r] = acc;
                }
            // U x = t: back substitution, bottom-up
#pragma unroll
            for (int rr = 0; rr < kMaxCh; ++rr) {
                const int r = kMaxCh - 1 - rr;
                if (r < n) {
                    const float ud = expf(hr[r]) / 10.0f + 1.0f;
                    ld -= logf(ud);
                    float acc = v[r];
                    const int base = n + r * n - (r * (r + 1)) / 2 - r - 1;
#pragma unroll
                    for (int c = 0; c < kMaxCh; ++c)
                        if (c > r && c < n) acc = fmaf(-(hr[base + c] / 10.0f), v[c], acc);
                    v[r] = acc / ud;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < kMaxCh; ++c)
            if (c < n) {
                const int t = c * HW + p;
                zr[tgt_idx ? tgt_idx[t] : D - T + t] = v[c];
            }
        if (p == 0) logdet[row] = accumulate ? logdet[row] + ld : ld;
        if (!inplace) {                               // clone of the untouched part: this lane's share
            for (int e = p; e < D; e += HW) {
                const bool tgt = tgt_idx ? (is_tgt[e] != 0) : (e >= D - T);
                if (!tgt) zr[e] = xr[e];
            }
        }
    }
}


// ---------------------------------------------------------------------------
// Reverse mode (the reference differentiates its ATen graph of convolution.py:33-64 / matrix.py:53-99).
// One workgroup per SAMPLE: lanes walk the sample's HW pixels, recompute the forward intermediates from the layer's
// INPUT rows, turn the incoming dL/d(out) at the target positions into dL/d(in) IN PLACE in g, and reduce the
// n^2 per-pixel outer-product contributions to dL/dh over the pixels (wave butterflies, then the 4 waves in a fixed
// order: deterministic).  With U = unit-upper part + diag, L = unit lower:
//   forward  y = L t, t = U x:  gt = L^T gy,  dL = gy t^T (strict lower),  gx = U^T gt,  dU = gt x^T (upper + diagonal)
//   inverse  w = U^-1 t, t = L^-1 v:  gt = U^-T gw,  dU = -gt w^T,  gv = L^-T gt,  dL = -gv t^T
//   h:  off-diagonals dh = d(.)/10;  diagonal U_rr = e_r/10 + 1, e_r = exp(h_r):  dh_r = dU_rr e_r/10
//       +/- gld (e_r/10) / U_rr   (the log-det is sum_r log U_rr ONCE per sample, quirk Q9).
// ---------------------------------------------------------------------------
template <bool INVERSE>
__global__ __launch_bounds__(kBlock) void k_conv1x1_coupling_bwd(
    const float *__restrict__ x, const float *__restrict__ h, float *g, const float *__restrict__ gld,
    float *__restrict__ gh, long long N, int D, const int *__restrict__ tgt_idx, int T, int n, int HW)
{
    __shared__ float part[kBlock / kWave][kMaxCh * kMaxCh];
    const int P = n + n * (n - 1);
    const int n_off = n * (n - 1) / 2;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    for (long long row = blockIdx.x; row < N; row += gridDim.x) {
        const float *xr = x + row * D;
        float *gr = g + row * D;
        const float *hr = h + row * P;
        for (int i = threadIdx.x; i < (kBlock / kWave) * kMaxCh * kMaxCh; i += kBlock) (&part[0][0])[i] = 0.0f;
        __syncthreads();
        float ud[kMaxCh];
#pragma unroll
        for (int r = 0; r < kMaxCh; ++r) ud[r] = r < n ? expf(hr[r]) / 10.0f + 1.0f : 1.0f;     // matrix.py:31-32
        auto Uo = [&](int r, int c) { return hr[n + r * n - (r * (r + 1)) / 2 - r - 1 + c] / 10.0f; };   // r < c
        auto Lo = [&](int r, int c) { return hr[n + n_off + (r * (r - 1)) / 2 + c] / 10.0f; };          // c < r
        const int iters = (HW + kBlock - 1) / kBlock;
        for (int it = 0; it < iters; ++it) {
            const int p = it * kBlock + threadIdx.x;
            const bool on = p < HW;
            float v[kMaxCh], gy[kMaxCh], t[kMaxCh], w[kMaxCh], gt[kMaxCh];
#pragma unroll
            for (int c = 0; c < kMaxCh; ++c) {
                v[c] = 0.0f; gy[c] = 0.0f;
                if (c < n && on) {
                    const int e = c * HW + p;
                    const int col = tgt_idx ? tgt_idx[e] : D - T + e;
                    v[c] = xr[col];
                    gy[c] = gr[col];
                }
            }
            if (!INVERSE) {
                // t = U x
#pragma unroll
                for (int r = 0; r < kMaxCh; ++r) {
                    float acc = ud[r] * v[r];
#pragma unroll
                    for (int c = 0; c < kMaxCh; ++c)
                        if (c > r && c < n) acc = fmaf(Uo(r, c), v[c], acc);
                    t[r] = r < n ? acc : 0.0f;
                }
                // gt = L^T gy
#pragma unroll
                for (int c = 0; c < kMaxCh; ++c) {
                    float acc = gy[c];
#pragma unroll
                    for (int r = 0; r < kMaxCh; ++r)
                        if (r > c && r < n) acc = fmaf(Lo(r, c), gy[r], acc);
                    gt[c] = c < n ? acc : 0.0f;
                }
                // gx = U^T gt  (into w)
#pragma unroll
                for (int c = 0; c < kMaxCh; ++c) {
                    float acc = ud[c] * gt[c];
#pragma unroll
                    for (int r = 0; r < kMaxCh; ++r)
                        if (r < c && c < n) acc = fmaf(Uo(r, c), gt[r], acc);
                    w[c] = acc;
                }
            } else {
                // t = L^-1 v (top-down), w = U^-1 t (bottom-up)
#pragma unroll
                for (int r = 0; r < kMaxCh; ++r) {
                    float acc = v[r];
#pragma unroll
                    for (int c = 0; c < kMaxCh; ++c)
                        if (c < r && r < n) acc = fmaf(-Lo(r, c), t[c], acc);
                    t[r] = r < n ? acc : 0.0f;
                }
#pragma unroll
                for (int rr = 0; rr < kMaxCh; ++rr) {
                    const int r = kMaxCh - 1 - rr;
                    float acc = t[r];
#pragma unroll
                    for (int c = 0; c < kMaxCh; ++c)
                        if (c > r && c < n) acc = fmaf(-Uo(r, c), w[c], acc);
                    w[r] = r < n ? acc / ud[r] : 0.0f;
                }
                // gt = U^-T gw: U^T gt = gw, top-down
#pragma unroll
                for (int c = 0; c < kMaxCh; ++c) {
                    float acc = gy[c];
#pragma unroll
                    for (int r = 0; r < kMaxCh; ++r)
                        if (r < c && c < n) acc = fmaf(-Uo(r, c), gt[r], acc);
                    gt[c] = c < n ? acc / ud[c] : 0.0f;
                }
                // gv = L^-T gt: L^T gv = gt, bottom-up (into gy)
#pragma unroll
                for (int cc = 0; cc < kMaxCh; ++cc) {
                    const int c = kMaxCh - 1 - cc;
                    float acc = gt[c];
#pragma unroll
                    for (int r = 0; r < kMaxCh; ++r)
                        if (r > c && r < n) acc = fmaf(-Lo(r, c), gy[r], acc);
                    gy[c] = c < n ? acc : 0.0f;
                }
            }
            // dL/d(in) at the target positions
#pragma unroll
            for (int c = 0; c < kMaxCh; ++c)
                if (c < n && on) {
                    const int e = c * HW + p;
                    gr[tgt_idx ? tgt_idx[e] : D - T + e] = INVERSE ? gy[c] : w[c];
                }
            // contributions to dL/dh, reduced over this wave's pixels
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < n; ++c) {
                    float val;
                    if (!INVERSE) val = c >= r ? gt[r] * v[c] : gy[r] * t[c];
                    else val = c >= r ? -(gt[r] * w[c]) : -(gy[r] * t[c]);
                    val = on ? val : 0.0f;
                    val = group_sum(val, kWave);
                    if (lane == 0) part[wave][r * kMaxCh + c] += val;
                }
        }
        __syncthreads();
        // h layout: [diag (n) | U above the diagonal, row-major | L below the diagonal, row-major]
        for (int i = threadIdx.x; i < n * n; i += kBlock) {
            const int r = i / n, c = i % n;
            float s = 0.0f;
            for (int wv = 0; wv < kBlock / kWave; ++wv) s += part[wv][r * kMaxCh + c];
            if (c == r) {
                const float e10 = expf(hr[r]) / 10.0f;
                const float ldg = gld[row] * (e10 / (e10 + 1.0f));
                gh[row * P + r] = s * e10 + (INVERSE ? -ldg : ldg);
            } else if (c > r) {
                gh[row * P + n + r * n - (r * (r + 1)) / 2 - r - 1 + c] = s / 10.0f;
            } else {
                gh[row * P + n + n_off + (r * (r - 1)) / 2 + c] = s / 10.0f;
            }
        }
        __syncthreads();
    }
}

template <bool INVERSE>
static int conv1x1(const float *x, const float *h, float *z, float *logdet, int64_t N, int32_t D,
                   const int32_t *tgt_idx, int32_t T, int32_t n, int32_t accumulate, void *stream,
                   const char *fn)
{
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D <= 0 || T <= 0 || T > D) return fail(TFK_EINVAL, "%s: need 0 < T <= D, got T = %d, D = %d", fn, T, D);
    if (n < 1 || n > kMaxCh) return fail(TFK_EINVAL, "%s: channel count %d must be in [1, %d]", fn, n, kMaxCh);
    if (T % n != 0) return fail(TFK_EINVAL, "%s: T = %d is not a multiple of the channel count %d", fn, T, n);
    if (N == 0) return TFK_OK;
    if (!x || !h || !z || !logdet) return fail(TFK_EINVAL, "%s: null pointer", fn);
    const bool inplace = (x == z);
    const size_t lds = (tgt_idx && !inplace) ? (size_t)D : 0;
    if (lds > 64 * 1024) return fail(TFK_EINVAL, "%s: masked path supports D <= 65536", fn);
    const int HW = T / n;
    const int grid = grid_for(N * (int64_t)HW, kBlock);
    hipLaunchKernelGGL((k_conv1x1_coupling<INVERSE>), dim3(grid), dim3(kBlock), lds,
                       static_cast<hipStream_t>(stream), x, h, z, logdet, (long long)N, D, tgt_idx, T,
                       n, HW, accumulate, inplace ? 1 : 0);
    return check_launch(fn);
}

}  // namespace tfk

extern "C" {

int tfk_conv1x1_coupling_fwd(const float *x, const float *h, float *z, float *logdet, int64_t N,
                             int32_t D, const int32_t *tgt_idx, int32_t T, int32_t n_channels,
                             int32_t accumulate, void *stream)
{
    return tfk::conv1x1<false>(x, h, z, logdet, N, D, tgt_idx, T, n_channels, accumulate, stream,
                               "tfk_conv1x1_coupling_fwd");
}

int tfk_conv1x1_coupling_inv(const float *z, const float *h, float *x, float *logdet, int64_t N,
                             int32_t D, const int32_t *tgt_idx, int32_t T, int32_t n_channels,
                             int32_t accumulate, void *stream)
{
    return tfk::conv1x1<true>(z, h, x, logdet, N, D, tgt_idx, T, n_channels, accumulate, stream,
                              "tfk_conv1x1_coupling_inv");
}

/* Reverse mode of tfk_conv1x1_coupling_fwd (inverse == 0) / _inv (inverse != 0). */
int tfk_conv1x1_coupling_bwd(const float *x, const float *h, float *g, const float *gld, float *gh, int64_t N,
                             int32_t D, const int32_t *tgt_idx, int32_t T, int32_t n_channels, int32_t inverse,
                             void *stream)
{
    const char *fn = "tfk_conv1x1_coupling_bwd";
    using namespace tfk;
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D <= 0 || T <= 0 || T > D) return fail(TFK_EINVAL, "%s: need 0 < T <= D, got T = %d, D = %d", fn, T, D);
    if (n_channels < 1 || n_channels > kMaxCh) return fail(TFK_EINVAL, "%s: channel count %d must be in [1, %d]", fn, n_channels, kMaxCh);
    if (T % n_channels != 0) return fail(TFK_EINVAL, "%s: T = %d is not a multiple of the channel count %d", fn, T, n_channels);
    if (N == 0) return TFK_OK;
    if (!x || !h || !g || !gld || !gh) return fail(TFK_EINVAL, "%s: null pointer", fn);
    const int HW = T / n_channels;
    const int grid = (int)(N < (int64_t)max_grid() * 4 ? N : (int64_t)max_grid() * 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (inverse)
        hipLaunchKernelGGL((k_conv1x1_coupling_bwd<true>), dim3(grid), dim3(kBlock), 0, s, x, h, g, gld, gh,
                           (long long)N, D, tgt_idx, T, n_channels, HW);
    else
        hipLaunchKernelGGL((k_conv1x1_coupling_bwd<false>), dim3(grid), dim3(kBlock), 0, s, x, h, g, gld, gh,
                           (long long)N, D, tgt_idx, T, n_channels, HW);
    return check_launch(fn);
}

}  // extern "C"
