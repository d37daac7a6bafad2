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
#include "tfk_common.h"
#include "tfk_spline.h"

namespace tfk {

constexpr float kLrsMinBin = 1e-2f;
constexpr float kLrsMinD = 1e-5f;
constexpr float kLrsEps = 5e-10f;

struct LrsConst {
    float minimum, maximum, span;
    float scale;     // 1 - 1e-2 * K
    float c;         // log(exp(1 - 1e-5) - 1)
};

__device__ __forceinline__ float lrs_deriv(float u, float c) {       // linear_rational.py:79, :89
    return softplus20(c + u / 100.0f) + kLrsMinD;
}

template <int KT, bool INVERSE>
__device__ __forceinline__ void lrs_eval(const float *p, float v, const LrsConst &C, float &out, float &ld)
{
    float ex[KT], ey[KT];
    float mx = 0.0f, my = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const float ux = p[j];
        const float uy = ux + p[KT + j] / 100.0f;                   // :87
        ex[j] = ux;
        ey[j] = uy;
        mx = j ? fmaxf(mx, ux) : ux;
        my = j ? fmaxf(my, uy) : uy;
    }
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        ex[j] = exp_noovf(ex[j] - mx);
        ey[j] = exp_noovf(ey[j] - my);
        sx += ex[j];
        sy += ey[j];
    }
    const float rx = 1.0f / sx, ry = 1.0f / sy;
    int k = 0;
    float xk = C.minimum, xk1 = C.maximum, yk = C.minimum, yk1 = C.maximum;
    float runx = 0.0f, runy = 0.0f, prevx = C.minimum, prevy = C.minimum;
    bool prev_below = true;
#pragma unroll
    for (int j = 1; j <= KT; ++j) {
        runx = runx + (kLrsMinBin + C.scale * (ex[j - 1] * rx));    // :69-71
        runy = runy + (kLrsMinBin + C.scale * (ey[j - 1] * ry));
        const float kx = (j == KT) ? C.maximum : C.span * runx + C.minimum;
        const float ky = (j == KT) ? C.maximum : C.span * runy + C.minimum;
        const bool below = (INVERSE ? ky : kx) < v;                 // searchsorted left, :105 / :150
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
    const float lam = 1.0f / (1.0f + exp_noovf(-p[2 * KT + k]));    // sigmoid, :88
    // boundary derivatives are exactly 1 (pad value, :80); the discarded reads stay in the record
    const float dk = (k == 0) ? 1.0f : lrs_deriv(p[3 * KT + k - 1], C.c);
    const float dk1 = (k == KT - 1) ? 1.0f : lrs_deriv(p[3 * KT + k], C.c);
    const float w0 = softplus20(p[4 * KT - 1]);                     // :41
    const float wk = w0 * sqrtf(1.0f / dk);                         // :42 (d_0 = 1)
    const float wk1 = w0 * sqrtf(1.0f / dk1);
    const float one_m = 1.0f - lam;
    const float ym = (one_m * wk * yk + lam * wk1 * yk1) / (one_m * wk + lam * wk1);       // :58-61
    const float dx = xk1 - xk;
    const float wm = (lam * wk * dk + one_m * wk1 * dk1) * (dx / (yk1 - yk));              // :62-67
    if (!INVERSE) {
        const float phi = (v - xk) / dx;                            // :110
        if (!(phi > lam)) {                                         // :113-121
            const float den = wk * (lam - phi) + wm * phi;
            out = (wk * yk * (lam - phi) + wm * ym * phi) / den;
            ld = log_normal(lam * wk * wm * (ym - yk)) - log_normal(den * den + kLrsEps) - log_normal(dx);
        } else {                                                    // :123-131
            const float den = wm * (1.0f - phi) + wk1 * (phi - lam);
            out = (wm * ym * (1.0f - phi) + wk1 * yk1 * (phi - lam)) / den;
            ld = log_normal(one_m * wm * wk1 * (yk1 - ym)) - log_normal(den * den + kLrsEps) - log_normal(dx);
        }
    } else {
        if (!(v > ym)) {                                            // :157-166
            const float den = wk * (yk - v) + wm * (v - ym);
            out = (lam * wk * (yk - v)) / den * dx + xk;
            ld = log_normal(lam * wk * wm * (ym - yk)) - log_normal(den * den + kLrsEps) + log_normal(dx);
        } else {                                                    // :168-176
            const float den = wk1 * (yk1 - v) + wm * (v - ym);
            out = (lam * wk1 * (yk1 - v) + wm * (v - ym)) / den * dx + xk;
            ld = log_normal(one_m * wm * wk1 * (yk1 - ym)) - log_normal(den * den + kLrsEps) + log_normal(dx);
        }
    }
}

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
                const float4 v4 = src[i];
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
    const int64_t grid = n_tiles < (int64_t)kCUs * per_cu ? n_tiles : (int64_t)kCUs * per_cu;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (K == 8)
        hipLaunchKernelGGL((k_lrs_coupling<8, INVERSE>), dim3((unsigned)grid), dim3(kBlock), lds, s, x, h, z, logdet,
                           (long long)N, D, tgt_idx, T, T_shift, C, accumulate, inplace ? 1 : 0);
    else
        hipLaunchKernelGGL((k_lrs_coupling<4, INVERSE>), dim3((unsigned)grid), dim3(kBlock), lds, s, x, h, z, logdet,
                           (long long)N, D, tgt_idx, T, T_shift, C, accumulate, inplace ? 1 : 0);
    return check_launch(fn);
}

}  // namespace tfk

extern "C" {

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
