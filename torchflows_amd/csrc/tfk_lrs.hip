// tfk_lrs.hip -- linear rational spline coupling (Dolatabadi et al. 2020), SURVEY.md 8(f)-4.
//
// Replaces CouplingBijection.forward / inverse (layers_base.py:145-163) around
// MonotonicSpline (spline/base.py:53-72) + LinearRational (spline/linear_rational.py:9-182).
// h is (N, T, 4K): [u_x (K) | u_y (K) | u_lambda (K) | u_d (K-1) | u_w0] per element, i.e. a
// stream of 16-byte aligned records (128 B for K = 8).  The kernel is priced against HBM:
// 4*(D + 4K*T + D) + 8 bytes per row (in place: 4*(2T + 4K*T) + 8).
//   * a workgroup takes R = 256/T rows = up to 256 records, loads them with coalesced float4
//     reads and stores them in LDS at a stride of 4K + 1 floats, so that the per-lane reads
//     of "my record" are bank-conflict-free (4K is a power of two);
//   * one lane = one spline element: two K-way softmaxes in ATen's CPU form (e * (1/sum)),
//     knot cumsum, bin search by compare / select, the bin's lambda and derivative logits
//     fetched from LDS by index; IEEE divisions and square root (the build uses
//     -fhip-fp32-correctly-rounded-divide-sqrt), phi is not clipped (the reference does not);
//   * per-row log-det: __shfl_xor inside the T-lane group when T is a power of two <= 64,
//     LDS otherwise; rows longer than the tile are walked in chunks.  Deterministic.
#include "tfk_lrs.h"

namespace tfk {

// Dynamic LDS: [256 records x (4K + 1) floats | 256 log-dets | D bytes of target mask]
template <int KT, bool INVERSE>
__global__ __launch_bounds__(kBlock) void k_lrs_coupling(
    const float *x, const float *__restrict__ h, float *z, float *logdet, long long N, int D,
    const int *__restrict__ tgt_idx, int T, int T_shift, LrsConst C, int accumulate, int inplace)
{
    constexpr int P = 4 * KT, PS = P + 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *rec = lds;
    float *ld_s = lds + kBlock * PS;
    unsigned char *is_tgt = reinterpret_cast<unsigned char *>(ld_s + kBlock);
    const int tid = threadIdx.x;
    const bool use_mask = (tgt_idx != nullptr) && !inplace;
    if (use_mask) {
        for (int e = tid; e < D; e += kBlock) is_tgt[e] = 0;
        __syncthreads();
        for (int t = tid; t < T; t += kBlock) is_tgt[tgt_idx[t]] = 1;
    }
    const int R = T <= kBlock ? kBlock / T : 1;
    const int chunks = T <= kBlock ? 1 : (T + kBlock - 1) / kBlock;
    const long long n_tiles = (N + R - 1) / R;
    const bool shfl_reduce = (T_shift >= 0) && (T <= kWave);

    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long row0 = tile * R;
        const int rows = (int)((N - row0) < (long long)R ? (N - row0) : (long long)R);
        float ld_thread = 0.0f;
        for (int ch = 0; ch < chunks; ++ch) {
            const int cbase = ch * kBlock;
            const int E = (chunks == 1) ? rows * T : ((T - cbase) < kBlock ? (T - cbase) : kBlock);
            const float4 *src = reinterpret_cast<const float4 *>(h + (row0 * (long long)T + cbase) * P);
            __syncthreads();                    // previous tile's readers are done
            for (int i = tid; i < E * (P / 4); i += kBlock) {
                const float4 v4 = nt_load4(src + i);                // h is read once
                const int r = i / (P / 4), j = (i - r * (P / 4)) * 4;
                float *dst = rec + r * PS + j;
                dst[0] = v4.x; dst[1] = v4.y; dst[2] = v4.z; dst[3] = v4.w;
            }
            __syncthreads();
            float ld = 0.0f;
            long long row = row0;
            int t = cbase + tid;
            if (chunks == 1) {
                const int r = T_shift >= 0 ? (tid >> T_shift) : (tid / T);
                t = tid - r * T;
                row = row0 + r;
            }
            if (tid < E) {
                const int idx = tgt_idx ? tgt_idx[t] : D - T + t;
                const float v = x[row * D + idx];
                float o = v;                                        // spline/base.py:54-55
                if (v > C.minimum && v < C.maximum)                 // strict, base.py:29-33
                    lrs_eval<KT, INVERSE>(rec + tid * PS, v, C, o, ld);
                z[row * D + idx] = o;
            }
            if (chunks > 1) {
                ld_thread += ld;
            } else if (shfl_reduce) {
                const float sum = group_sum(ld, T);
                if (tid < E && t == 0) logdet[row] = accumulate ? logdet[row] + sum : sum;
            } else {
                ld_s[tid] = ld;
                __syncthreads();
                if (tid < rows) {
                    float sum = 0.0f;
                    for (int j = 0; j < T; ++j) sum += ld_s[tid * T + j];
                    logdet[row0 + tid] = accumulate ? logdet[row0 + tid] + sum : sum;
                }
            }
        }
        if (chunks > 1) {
            __syncthreads();
            ld_s[tid] = ld_thread;
            __syncthreads();
            for (int o = kBlock / 2; o > 0; o >>= 1) {
                if (tid < o) ld_s[tid] += ld_s[tid + o];
                __syncthreads();
            }
            if (tid == 0) logdet[row0] = accumulate ? logdet[row0] + ld_s[0] : ld_s[0];
        }
        if (!inplace) {                                            // clone, layers_base.py:146
            const int total = rows * D;
            for (int e = tid; e < total; e += kBlock) {
                const int r = e / D;
                const int c = e - r * D;
                const bool tgt = tgt_idx ? (is_tgt[c] != 0) : (c >= D - T);
                if (!tgt) z[(row0 + r) * D + c] = x[(row0 + r) * D + c];
            }
        }
    }
}

template <bool INVERSE>
static int lrs_coupling(const float *x, const float *h, float *z, float *logdet, int64_t N, int32_t D,
                        const int32_t *tgt_idx, int32_t T, int32_t K, float boundary, int32_t accumulate,
                        void *stream, const char *fn)
{
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D < 1 || T < 1 || T > D) return fail(TFK_EINVAL, "%s: need 1 <= T <= D (T = %d, D = %d)", fn, T, D);
    if (K != 4 && K != 8) return fail(TFK_EINVAL, "%s: n_bins = %d (kernels exist for 4 and 8)", fn, K);
    if (!(boundary > 0.0f)) return fail(TFK_EINVAL, "%s: boundary must be positive", fn);
    if (N == 0) return TFK_OK;
    if (!x || !h || !z || !logdet) return fail(TFK_EINVAL, "%s: null pointer", fn);
    if (!aligned16(h)) return fail(TFK_EINVAL, "%s: h must be 16-byte aligned", fn);
    LrsConst C;
    C.minimum = -boundary;
    C.maximum = boundary;
    C.span = (float)((double)boundary + (double)boundary);
    C.scale = (float)(1.0 - 1e-2 * (double)K);
    C.c = (float)log(exp(1.0 - 1e-5) - 1.0);
    const bool inplace = (x == z);
    int T_shift = -1;
    if ((T & (T - 1)) == 0) {
        T_shift = 0;
        while ((1 << T_shift) < T) ++T_shift;
    }
    const int R = T <= kBlock ? kBlock / T : 1;
    const int64_t n_tiles = (N + R - 1) / R;
    const size_t lds = ((size_t)kBlock * (4 * K + 1) + kBlock) * sizeof(float) + ((tgt_idx && !inplace) ? (size_t)D : 0);
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int64_t grid = n_tiles < (int64_t)cu_count() * per_cu ? n_tiles : (int64_t)cu_count() * per_cu;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (K == 8)
        hipLaunchKernelGGL((k_lrs_coupling<8, INVERSE>), dim3((unsigned)grid), dim3(kBlock), lds, s, x, h, z, logdet,
                           (long long)N, D, tgt_idx, T, T_shift, C, accumulate, inplace ? 1 : 0);
    else
        hipLaunchKernelGGL((k_lrs_coupling<4, INVERSE>), dim3((unsigned)grid), dim3(kBlock), lds, s, x, h, z, logdet,
                           (long long)N, D, tgt_idx, T, T_shift, C, accumulate, inplace ? 1 : 0);
    return check_launch(fn);
}

// ---------------------------------------------------------------------------------------------
// Reverse mode (SURVEY.md 8(f)-2 for the linear rational spline).  The reference has no backward
// code: this replaces what torch.autograd derives from linear_rational.py:33-182 for one in-box
// element, both directions differentiated through their explicit formulas.  p holds the element's 4K
// parameters on entry and their gradients on exit; A = dL/d out, B = dL/d log-det; gv = dL/d v.
// Recompute, not store: knots, lambda, derivatives and weights are rebuilt from p.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_of(float t) {             // softplus'(t) = sigmoid(t)
    return 1.0f / (1.0f + exp_noovf(-t));
}

template <int KT, bool INVERSE>
__device__ __forceinline__ void lrs_bwd_eval(float (&p)[4 * KT], float v, const LrsConst &C, float A, float B,
                                             float &gv)
{
    float smx[KT], smy[KT];
    float mx = 0.0f, my = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const float ux = p[j];
        const float uy = ux + p[KT + j] / 100.0f;
        smx[j] = ux;
        smy[j] = uy;
        mx = j ? fmaxf(mx, ux) : ux;
        my = j ? fmaxf(my, uy) : uy;
    }
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        smx[j] = exp_noovf(smx[j] - mx);
        smy[j] = exp_noovf(smy[j] - my);
        sx += smx[j];
        sy += smy[j];
    }
    const float rx = 1.0f / sx, ry = 1.0f / sy;
    int k = 0;
    float xk = C.minimum, xk1 = C.maximum, yk = C.minimum, yk1 = C.maximum;
    float runx = 0.0f, runy = 0.0f, prevx = C.minimum, prevy = C.minimum;
    bool prev_below = true;
#pragma unroll
    for (int j = 1; j <= KT; ++j) {
        smx[j - 1] = smx[j - 1] * rx;
        smy[j - 1] = smy[j - 1] * ry;
        runx = runx + (kLrsMinBin + C.scale * smx[j - 1]);
        runy = runy + (kLrsMinBin + C.scale * smy[j - 1]);
        const float kx = (j == KT) ? C.maximum : C.span * runx + C.minimum;
        const float ky = (j == KT) ? C.maximum : C.span * runy + C.minimum;
        const bool below = (INVERSE ? ky : kx) < v;
        const bool sel = prev_below && !below;
        k = sel ? j - 1 : k;
        xk = sel ? prevx : xk;
        xk1 = sel ? kx : xk1;
        yk = sel ? prevy : yk;
        yk1 = sel ? ky : yk1;
        prev_below = below;
        prevx = kx;
        prevy = ky;
    }
    float ulam = p[2 * KT], ud0 = 0.0f, ud1 = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) ulam = (k == j) ? p[2 * KT + j] : ulam;
#pragma unroll
    for (int j = 0; j < KT - 1; ++j) {
        ud0 = (k == j + 1) ? p[3 * KT + j] : ud0;
        ud1 = (k == j) ? p[3 * KT + j] : ud1;
    }
    const bool first = (k == 0), last = (k == KT - 1);
    const float lam = 1.0f / (1.0f + exp_noovf(-ulam));
    const float td0 = C.c + ud0 / 100.0f, td1 = C.c + ud1 / 100.0f;
    const float dk = first ? 1.0f : softplus20(td0) + kLrsMinD;
    const float dk1 = last ? 1.0f : softplus20(td1) + kLrsMinD;
    const float uw0 = p[4 * KT - 1];
    const float w0 = softplus20(uw0);
    const float sq0 = sqrtf(1.0f / dk), sq1 = sqrtf(1.0f / dk1);
    const float wk = w0 * sq0, wk1 = w0 * sq1;
    const float one_m = 1.0f - lam;
    const float Dy = one_m * wk + lam * wk1;
    const float ym = (one_m * wk * yk + lam * wk1 * yk1) / Dy;
    const float dx = xk1 - xk, dy = yk1 - yk;
    const float W = lam * wk * dk + one_m * wk1 * dk1;
    const float ratio = dx / dy;
    const float wm = W * ratio;

    float g_wk = 0.0f, g_wk1 = 0.0f, g_wm = 0.0f, g_ym = 0.0f, g_lam = 0.0f, g_onem = 0.0f;
    float g_xk = 0.0f, g_yk = 0.0f, g_yk1 = 0.0f, g_dx = 0.0f;
    gv = 0.0f;
    if (!INVERSE) {
        const float phi = (v - xk) / dx;
        float g_phi = 0.0f;
        if (!(phi > lam)) {
            const float t1 = lam - phi;
            const float den = wk * t1 + wm * phi;
            const float num = wk * yk * t1 + wm * ym * phi;
            const float a1 = lam * wk * wm * (ym - yk);
            const float g_num = A / den;
            const float g_den = -A * num / (den * den) - 2.0f * B * den / (den * den + kLrsEps);
            const float g_a1 = B / a1;
            float g_t1 = g_num * wk * yk + g_den * wk;
            g_wk += g_num * yk * t1 + g_den * t1 + g_a1 * lam * wm * (ym - yk);
            g_yk += g_num * wk * t1 - g_a1 * lam * wk * wm;
            g_wm += g_num * ym * phi + g_den * phi + g_a1 * lam * wk * (ym - yk);
            g_ym += g_num * wm * phi + g_a1 * lam * wk * wm;
            g_phi += g_num * wm * ym + g_den * wm - g_t1;
            g_lam += g_a1 * wk * wm * (ym - yk) + g_t1;
        } else {
            const float t2 = 1.0f - phi, t3 = phi - lam;
            const float den = wm * t2 + wk1 * t3;
            const float num = wm * ym * t2 + wk1 * yk1 * t3;
            const float a2 = one_m * wm * wk1 * (yk1 - ym);
            const float g_num = A / den;
            const float g_den = -A * num / (den * den) - 2.0f * B * den / (den * den + kLrsEps);
            const float g_a2 = B / a2;
            const float g_t2 = g_num * wm * ym + g_den * wm;
            const float g_t3 = g_num * wk1 * yk1 + g_den * wk1;
            g_wm += g_num * ym * t2 + g_den * t2 + g_a2 * one_m * wk1 * (yk1 - ym);
            g_ym += g_num * wm * t2 - g_a2 * one_m * wm * wk1;
            g_wk1 += g_num * yk1 * t3 + g_den * t3 + g_a2 * one_m * wm * (yk1 - ym);
            g_yk1 += g_num * wk1 * t3 + g_a2 * one_m * wm * wk1;
            g_onem += g_a2 * wm * wk1 * (yk1 - ym);
            g_phi += g_t3 - g_t2;
            g_lam -= g_t3;
        }
        g_dx += -B / dx;                                            // ld has - log(dx)
        gv = g_phi / dx;                                            // phi = (v - x_k) / dx
        g_xk -= g_phi / dx;
        g_dx -= g_phi * phi / dx;
    } else {
        const float s2 = v - ym;
        float g_s2 = 0.0f, g_r;
        float r;
        if (!(v > ym)) {
            const float s1 = yk - v;
            const float den = wk * s1 + wm * s2;
            const float nm = lam * wk * s1;
            const float a1 = lam * wk * wm * (ym - yk);
            r = nm / den;
            g_r = A * dx;
            const float g_nm = g_r / den;
            const float g_den = -g_r * nm / (den * den) - 2.0f * B * den / (den * den + kLrsEps);
            const float g_a1 = B / a1;
            const float g_s1 = g_nm * lam * wk + g_den * wk;
            g_lam += g_nm * wk * s1 + g_a1 * wk * wm * (ym - yk);
            g_wk += g_nm * lam * s1 + g_den * s1 + g_a1 * lam * wm * (ym - yk);
            g_wm += g_den * s2 + g_a1 * lam * wk * (ym - yk);
            g_s2 += g_den * wm;
            g_ym += g_a1 * lam * wk * wm;
            g_yk += g_s1 - g_a1 * lam * wk * wm;
            gv -= g_s1;
        } else {
            const float s3 = yk1 - v;
            const float den = wk1 * s3 + wm * s2;
            const float nm = lam * wk1 * s3 + wm * s2;
            const float a2 = one_m * wm * wk1 * (yk1 - ym);
            r = nm / den;
            g_r = A * dx;
            const float g_nm = g_r / den;
            const float g_den = -g_r * nm / (den * den) - 2.0f * B * den / (den * den + kLrsEps);
            const float g_a2 = B / a2;
            const float g_s3 = g_nm * lam * wk1 + g_den * wk1;
            g_lam += g_nm * wk1 * s3;
            g_wk1 += g_nm * lam * s3 + g_den * s3 + g_a2 * one_m * wm * (yk1 - ym);
            g_wm += g_nm * s2 + g_den * s2 + g_a2 * one_m * wk1 * (yk1 - ym);
            g_s2 += g_nm * wm + g_den * wm;
            g_onem += g_a2 * wm * wk1 * (yk1 - ym);
            g_yk1 += g_s3 + g_a2 * one_m * wm * wk1;
            g_ym -= g_a2 * one_m * wm * wk1;
            gv -= g_s3;
        }
        gv += g_s2;
        g_ym -= g_s2;
        g_dx += A * r + B / dx;                                     // out = r dx + x_k, ld has + log(dx)
        g_xk += A;
    }
    // w_m = W dx / dy,  W = lam w_k d_k + (1 - lam) w_{k+1} d_{k+1}                         (:62-67)
    float g_dk = 0.0f, g_dk1 = 0.0f;
    const float g_W = g_wm * ratio, g_ratio = g_wm * W;
    g_dx += g_ratio / dy;
    const float g_dy = -g_ratio * ratio / dy;
    g_lam += g_W * wk * dk;
    g_wk += g_W * lam * dk;
    g_dk += g_W * lam * wk;
    g_onem += g_W * wk1 * dk1;
    g_wk1 += g_W * one_m * dk1;
    g_dk1 += g_W * one_m * wk1;
    g_yk1 += g_dy;
    g_yk -= g_dy;
    // y_m = ((1 - lam) w_k y_k + lam w_{k+1} y_{k+1}) / ((1 - lam) w_k + lam w_{k+1})       (:58-61)
    const float g_Ny = g_ym / Dy, g_Dy = -g_ym * ym / Dy;
    g_onem += g_Ny * wk * yk + g_Dy * wk;
    g_wk += g_Ny * one_m * yk + g_Dy * one_m;
    g_yk += g_Ny * one_m * wk;
    g_lam += g_Ny * wk1 * yk1 + g_Dy * wk1;
    g_wk1 += g_Ny * lam * yk1 + g_Dy * lam;
    g_yk1 += g_Ny * lam * wk1;
    g_lam -= g_onem;
    // w = w_0 sqrt(1 / d)                                                                    (:41-42)
    const float g_w0 = g_wk * sq0 + g_wk1 * sq1;
    g_dk += -0.5f * g_wk * wk / dk;
    g_dk1 += -0.5f * g_wk1 * wk1 / dk1;
    // knots: dx = x_{k+1} - x_k; knots 0 and K are constants
    const float g_x0 = g_xk - g_dx, g_x1 = g_dx;
    const float sc = C.scale * C.span;
    const float gcx0 = first ? 0.0f : sc * g_x0, gcx1 = last ? 0.0f : sc * g_x1;
    const float gcy0 = first ? 0.0f : sc * g_yk, gcy1 = last ? 0.0f : sc * g_yk1;
    float dotx = 0.0f, doty = 0.0f;
#pragma unroll
    for (int i = 0; i < KT; ++i) {                                  // cumsum: knot j sums bins i < j
        const float gwx = (i < k ? gcx0 : 0.0f) + (i < k + 1 ? gcx1 : 0.0f);
        const float gwy = (i < k ? gcy0 : 0.0f) + (i < k + 1 ? gcy1 : 0.0f);
        dotx += smx[i] * gwx;
        doty += smy[i] * gwy;
    }
    const float g_ulam = g_lam * lam * one_m;                       // sigmoid'
    const float g_ud0 = first ? 0.0f : g_dk * (td0 > 20.0f ? 1.0f : sigmoid_of(td0)) / 100.0f;
    const float g_ud1 = last ? 0.0f : g_dk1 * (td1 > 20.0f ? 1.0f : sigmoid_of(td1)) / 100.0f;
    const float g_uw0 = g_w0 * (uw0 > 20.0f ? 1.0f : sigmoid_of(uw0));
#pragma unroll
    for (int i = 0; i < KT; ++i) {                                  // softmax backward
        const float gwx = (i < k ? gcx0 : 0.0f) + (i < k + 1 ? gcx1 : 0.0f);
        const float gwy = (i < k ? gcy0 : 0.0f) + (i < k + 1 ? gcy1 : 0.0f);
        const float gux = smx[i] * (gwx - dotx);
        const float guy = smy[i] * (gwy - doty);
        p[i] = gux + guy;                                           // u_y enters as u_x + u_y / 100
        p[KT + i] = guy / 100.0f;
        p[2 * KT + i] = (k == i) ? g_ulam : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < KT - 1; ++j)
        p[3 * KT + j] = (k == j + 1) ? g_ud0 : ((k == j) ? g_ud1 : 0.0f);
    p[4 * KT - 1] = g_uw0;
}

// Flat map over the N*T spline elements, 256 per workgroup; records (4K floats) go through LDS at a
// stride of 4K + 1 (conflict-free per-lane reads), gradients back the same way.
//   bytes per element: h 16K + gh 16K + x 4 + g 8 (+ gld per row)
template <int KT, bool INVERSE>
__global__ __launch_bounds__(kBlock) void k_lrs_coupling_bwd(
    const float *__restrict__ x, const float *__restrict__ h, float *g, const float *__restrict__ gld,
    float *__restrict__ gh, long long N, int D, const int *__restrict__ tgt_idx, int T, LrsConst C)
{
    constexpr int P = 4 * KT, PS = P + 1;
    __shared__ float rec[kBlock * PS];
    const int tid = threadIdx.x;
    const long long total = N * (long long)T;
    const long long n_tiles = (total + kBlock - 1) / kBlock;
    for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const long long e0 = tile * kBlock;
        const int E = (int)((total - e0) < (long long)kBlock ? (total - e0) : (long long)kBlock);
        __syncthreads();
        for (int i = tid; i < E * P; i += kBlock) rec[(i / P) * PS + (i % P)] = h[e0 * P + i];
        __syncthreads();
        if (tid < E) {
            const long long e = e0 + tid;
            const long long row = e / T;
            const int t = (int)(e - row * T);
            const int idx = tgt_idx ? tgt_idx[t] : D - T + t;
            const float v = x[row * D + idx];
            const float A = g[row * D + idx];
            float p[P];
#pragma unroll
            for (int j = 0; j < P; ++j) p[j] = rec[tid * PS + j];
            float gv = A;                      // identity outside the box, spline/base.py:54-55
            if (v > C.minimum && v < C.maximum) {
                lrs_bwd_eval<KT, INVERSE>(p, v, C, A, gld[row], gv);
            } else {
#pragma unroll
                for (int j = 0; j < P; ++j) p[j] = 0.0f;
            }
            g[row * D + idx] = gv;
#pragma unroll
            for (int j = 0; j < P; ++j) rec[tid * PS + j] = p[j];
        }
        __syncthreads();
        for (int i = tid; i < E * P; i += kBlock) gh[e0 * P + i] = rec[(i / P) * PS + (i % P)];
    }
}

}  // namespace tfk

extern "C" {

int tfk_lrs_coupling_bwd(const float *x, const float *h, float *g, const float *gld, float *gh,
                         int64_t N, int32_t D, const int32_t *tgt_idx, int32_t T, int32_t K,
                         float boundary, int32_t inverse, void *stream)
{
    using namespace tfk;
    const char *fn = "tfk_lrs_coupling_bwd";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D < 1 || T < 1 || T > D) return fail(TFK_EINVAL, "%s: need 1 <= T <= D (T = %d, D = %d)", fn, T, D);
    if (K != 4 && K != 8) return fail(TFK_EINVAL, "%s: n_bins = %d (kernels exist for 4 and 8)", fn, K);
    if (!(boundary > 0.0f)) return fail(TFK_EINVAL, "%s: boundary must be positive", fn);
    if (N == 0) return TFK_OK;
    if (!x || !h || !g || !gld || !gh) return fail(TFK_EINVAL, "%s: null pointer", fn);
    LrsConst C;
    C.minimum = -boundary;
    C.maximum = boundary;
    C.span = (float)((double)boundary + (double)boundary);
    C.scale = (float)(1.0 - 1e-2 * (double)K);
    C.c = (float)log(exp(1.0 - 1e-5) - 1.0);
    const int64_t n_tiles = (N * (int64_t)T + kBlock - 1) / kBlock;
    const int grid = (int)(n_tiles < (int64_t)cu_count() * 4 ? n_tiles : (int64_t)cu_count() * 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define TFK_LB(KT_, INV_)                                                                                     \
    hipLaunchKernelGGL((k_lrs_coupling_bwd<KT_, INV_>), dim3(grid), dim3(kBlock), 0, s, x, h, g, gld, gh,     \
                       (long long)N, D, tgt_idx, T, C)
    if (K == 8) { if (inverse) TFK_LB(8, true); else TFK_LB(8, false); }
    else { if (inverse) TFK_LB(4, true); else TFK_LB(4, false); }
#undef TFK_LB
    return check_launch(fn);
}

int tfk_lrs_coupling_fwd(const float *x, const float *h, float *z, float *logdet, int64_t N, int32_t D,
                         const int32_t *tgt_idx, int32_t T, int32_t K, float boundary, int32_t accumulate,
                         void *stream)
{
    return tfk::lrs_coupling<false>(x, h, z, logdet, N, D, tgt_idx, T, K, boundary, accumulate, stream,
                                    "tfk_lrs_coupling_fwd");
}

int tfk_lrs_coupling_inv(const float *z, const float *h, float *x, float *logdet, int64_t N, int32_t D,
                         const int32_t *tgt_idx, int32_t T, int32_t K, float boundary, int32_t accumulate,
                         void *stream)
{
    return tfk::lrs_coupling<true>(z, h, x, logdet, N, D, tgt_idx, T, K, boundary, accumulate, stream,
                                   "tfk_lrs_coupling_inv");
}

}  // extern "C"
