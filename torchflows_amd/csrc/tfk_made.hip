// tfk_made.hip -- the SEQUENTIAL map of a MADE-based affine layer in one launch.
//
// Replaces MaskedAutoregressiveBijection.inverse (layers_base.py:208-221; MAF sampling) and, with
// the maps exchanged, InverseMaskedAutoregressiveBijection.forward (:231-232; IAF density): the
// reference walks the D elements in order and re-runs the whole conditioner (two masked GEMMs +
// tanh) on the partially inverted batch for every element -- D passes, ~4 D launches per layer.
// Here one lane owns one row and keeps the H hidden pre-activations in registers: when element i
// becomes known they are updated with column i of the first masked weight (a_k += W1[k, i] x_i),
// and element i's two parameters are read off row i of the second masked weight.  O(D H) work per
// row instead of O(D^2 H), one launch instead of ~4 D.
//   conditioner: MADE with two masked linear layers, transforms.py:184-267 (weights arrive already
//                multiplied by their masks, zero-padded to HMAX hidden units)
//   transformer: Affine / InverseAffine, affine.py:36-70; alpha = exp(u / 2 + c0) + m
//   log-det:     the reference keeps the log-det of its LAST pass, which for an affine transformer is
//                the exact -/+ sum_i log alpha_i(x_<i) accumulated here
// The dot products run in a different order than the reference's GEMMs (both within a few ulp).
// Data movement: rows are staged through LDS (row stride D + 1: conflict-free column access),
// weights are LDS broadcasts.  bytes per row: 8 D + 4 (+4).
#include "tfk_common.h"
#include "tfk_spline.h"
#include "tfk_lrs.h"

namespace tfk {

// Dynamic LDS: W1t[D][HMAX] | b1[HMAX] | W2[D][2][HMAX] | b2[D][2] | rows[BLOCK][D + 1]
template <int HMAX, bool DIVIDE>
__global__ void k_made_affine_sequential(
    const float *__restrict__ z, float *__restrict__ x, float *logdet, long long N, int D,
    const float *__restrict__ W1t, const float *__restrict__ b1, const float *__restrict__ W2,
    const float *__restrict__ b2, int accumulate)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *w1 = lds;                         // [D][HMAX]: column i of the first layer
    float *bb1 = w1 + D * HMAX;
    float *w2 = bb1 + HMAX;                  // [D][2][HMAX]
    float *bb2 = w2 + 2 * D * HMAX;
    float *rows = bb2 + 2 * D;
    const int tid = threadIdx.x, B = blockDim.x;
    for (int i = tid; i < D * HMAX; i += B) w1[i] = W1t[i];
    for (int i = tid; i < HMAX; i += B) bb1[i] = b1[i];
    for (int i = tid; i < 2 * D * HMAX; i += B) w2[i] = W2[i];
    for (int i = tid; i < 2 * D; i += B) bb2[i] = b2[i];
    const int RS = D + 1;
    for (long long base = (long long)blockIdx.x * B; base < N; base += (long long)gridDim.x * B) {
        const int nrows = (int)((N - base) < (long long)B ? (N - base) : (long long)B);
        __syncthreads();
        for (int e = tid; e < nrows * D; e += B) {                  // coalesced load, transposed park
            const int r = e / D;
            rows[r * RS + (e - r * D)] = z[base * D + e];
        }
        __syncthreads();
        if (tid < nrows) {
            float a[HMAX];
#pragma unroll
            for (int k = 0; k < HMAX; ++k) a[k] = bb1[k];
            float ld = 0.0f;
            float *mine = rows + tid * RS;
            for (int i = 0; i < D; ++i) {
                float u = bb2[2 * i], be = bb2[2 * i + 1];
                const float *r0 = w2 + (2 * i) * HMAX, *r1 = r0 + HMAX;
#pragma unroll
                for (int k = 0; k < HMAX; ++k) {
                    const float e2 = __builtin_amdgcn_exp2f(a[k] * 2.8853900817779268f);
                    const float hk = fmaf(-2.0f, __builtin_amdgcn_rcpf(e2 + 1.0f), 1.0f);    // tanh
                    u = fmaf(r0[k], hk, u);
                    be = fmaf(r1[k], hk, be);
                }
                const float alpha = aff_alpha(u);                    // affine.py:33-34
                const float la = log_normal(alpha);
                float v = mine[i];
                if (DIVIDE) {                                        // (z - beta) / alpha, affine.py:59
                    v = div_fast(v - be, alpha);
                    ld -= la;
                } else {                                             // alpha z + beta, affine.py:48
                    v = alpha * v + be;
                    ld += la;
                }
                mine[i] = v;
                const float *c = w1 + i * HMAX;
#pragma unroll
                for (int k = 0; k < HMAX; ++k) a[k] = fmaf(c[k], v, a[k]);
            }
            const long long row = base + tid;
            logdet[row] = accumulate ? logdet[row] + ld : ld;
        }
        __syncthreads();
        for (int e = tid; e < nrows * D; e += B) {
            const int r = e / D;
            x[base * D + e] = rows[r * RS + (e - r * D)];
        }
    }
}

template <int HMAX>
static int launch_made(const float *z, float *x, float *logdet, int64_t N, int D, const float *W1t,
                       const float *b1, const float *W2, const float *b2, int divide, int accumulate,
                       hipStream_t s, const char *fn)
{
    const size_t weights = ((size_t)D * HMAX + HMAX + 2 * (size_t)D * HMAX + 2 * (size_t)D) * sizeof(float);
    int block = 256;
    while (block > 64 && weights + (size_t)block * (D + 1) * sizeof(float) > 150 * 1024) block >>= 1;
    const size_t lds = weights + (size_t)block * (D + 1) * sizeof(float);
    if (lds > 160 * 1024) return fail(TFK_EINVAL, "%s: D = %d, hidden <= %d needs %zu bytes of LDS", fn, D, HMAX, lds);
    const void *kern = divide ? reinterpret_cast<const void *>(&k_made_affine_sequential<HMAX, true>)
                              : reinterpret_cast<const void *>(&k_made_affine_sequential<HMAX, false>);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", fn, lds);
    }
    int64_t grid = (N + block - 1) / block;
    if (grid > max_grid()) grid = max_grid();
    if (divide)
        hipLaunchKernelGGL((k_made_affine_sequential<HMAX, true>), dim3((int)grid), dim3(block), lds, s, z, x, logdet,
                           (long long)N, D, W1t, b1, W2, b2, accumulate);
    else
        hipLaunchKernelGGL((k_made_affine_sequential<HMAX, false>), dim3((int)grid), dim3(block), lds, s, z, x, logdet,
                           (long long)N, D, W1t, b1, W2, b2, accumulate);
    return check_launch(fn);
}

// The same walk for a MADE-based RATIONAL-QUADRATIC SPLINE layer (MaskedAutoregressiveRQNSF sampling,
// InverseAutoregressiveRQNSF density; architectures.py): element i's 3K-1 spline parameters are read off
// rows [P i, P (i+1)) of the second masked weight and the spline is inverted for it
// (rational_quadratic.py:147-200, in-box elements only, spline/base.py:53-72).
//   log-det: the reference returns the log-det of its LAST pass (layers_base.py:213-221), in which the
//   elements j < D-1 already hold their inverted values: sum_{j<D-1} ld_inv(x_j; h_j) + ld_inv(z_{D-1};
//   h_{D-1}) -- for a spline this is NOT the log-det of the map; it is reproduced here (a second
//   evaluation at x_j), so that the drop-in returns what the reference returns.
// LRS: the linear rational spline instead (MaskedAutoregressiveLRS / InverseAutoregressiveLRS; 4K parameters
// per element, linear_rational.py:137-182).
// Dynamic LDS: W1t[D][HMAX] | b1[HMAX] | W2[D][P][HMAX] | b2[D][P] | rows[BLOCK][D + 1]
template <int KT, bool LRS, typename CT>
__device__ __forceinline__ void spline_inverse(const float (&p)[LRS ? 4 * KT : 3 * KT - 1], float v, const CT &C,
                                               float &out, float &l)
{
    if constexpr (LRS) lrs_eval<KT, true>(p, v, C, out, l);
    else rqs_eval<KT, true, false, float[3 * KT - 1]>(p, KT, v, C, out, l);
}

template <int HMAX, int KT, bool LRS, typename CT>
__global__ void k_made_rqs_sequential(
    const float *__restrict__ z, float *__restrict__ x, float *logdet, long long N, int D,
    const float *__restrict__ W1t, const float *__restrict__ b1, const float *__restrict__ W2,
    const float *__restrict__ b2, CT C, int accumulate)
{
    constexpr int P = LRS ? 4 * KT : 3 * KT - 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *w1 = lds;
    float *bb1 = w1 + D * HMAX;
    float *w2 = bb1 + HMAX;                  // [D][P][HMAX]
    float *bb2 = w2 + P * D * HMAX;
    float *rows = bb2 + P * D;
    const int tid = threadIdx.x, B = blockDim.x;
    for (int i = tid; i < D * HMAX; i += B) w1[i] = W1t[i];
    for (int i = tid; i < HMAX; i += B) bb1[i] = b1[i];
    for (int i = tid; i < P * D * HMAX; i += B) w2[i] = W2[i];
    for (int i = tid; i < P * D; i += B) bb2[i] = b2[i];
    const int RS = D + 1;
    for (long long base = (long long)blockIdx.x * B; base < N; base += (long long)gridDim.x * B) {
        const int nrows = (int)((N - base) < (long long)B ? (N - base) : (long long)B);
        __syncthreads();
        for (int e = tid; e < nrows * D; e += B) {
            const int r = e / D;
            rows[r * RS + (e - r * D)] = z[base * D + e];
        }
        __syncthreads();
        if (tid < nrows) {
            float a[HMAX];
#pragma unroll
            for (int k = 0; k < HMAX; ++k) a[k] = bb1[k];
            float ld = 0.0f;
            float *mine = rows + tid * RS;
            for (int i = 0; i < D; ++i) {
                float hk[HMAX];
#pragma unroll
                for (int k = 0; k < HMAX; ++k) {
                    const float e2 = __builtin_amdgcn_exp2f(a[k] * 2.8853900817779268f);
                    hk[k] = fmaf(-2.0f, __builtin_amdgcn_rcpf(e2 + 1.0f), 1.0f);             // tanh
                }
                float p[P];
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    const float *r = w2 + (P * i + j) * HMAX;
                    float acc = bb2[P * i + j];
#pragma unroll
                    for (int k = 0; k < HMAX; ++k) acc = fmaf(r[k], hk[k], acc);
                    p[j] = acc;
                }
                const float v = mine[i];
                float out = v, l = 0.0f;                             // identity outside the box
                if (v > C.minimum && v < C.maximum) spline_inverse<KT, LRS>(p, v, C, out, l);
                if (i < D - 1) {                                     // what the reference's last pass sees
                    float o2 = out;
                    l = 0.0f;
                    if (out > C.minimum && out < C.maximum) spline_inverse<KT, LRS>(p, out, C, o2, l);
                }
                ld += l;
                mine[i] = out;
                const float *c = w1 + i * HMAX;
#pragma unroll
                for (int k = 0; k < HMAX; ++k) a[k] = fmaf(c[k], out, a[k]);
            }
            const long long row = base + tid;
            logdet[row] = accumulate ? logdet[row] + ld : ld;
        }
        __syncthreads();
        for (int e = tid; e < nrows * D; e += B) {
            const int r = e / D;
            x[base * D + e] = rows[r * RS + (e - r * D)];
        }
    }
}

template <int HMAX, int KT, bool LRS, typename CT>
static int launch_made_rqs(const float *z, float *x, float *logdet, int64_t N, int D, const float *W1t,
                           const float *b1, const float *W2, const float *b2, const CT &C, int accumulate,
                           hipStream_t s, const char *fn)
{
    constexpr int P = LRS ? 4 * KT : 3 * KT - 1;
    const size_t weights = ((size_t)D * HMAX + HMAX + (size_t)P * D * HMAX + (size_t)P * D) * sizeof(float);
    int block = 256;
    while (block > 64 && weights + (size_t)block * (D + 1) * sizeof(float) > 150 * 1024) block >>= 1;
    const size_t lds = weights + (size_t)block * (D + 1) * sizeof(float);
    if (lds > 160 * 1024) return fail(TFK_EINVAL, "%s: D = %d, hidden <= %d needs %zu bytes of LDS", fn, D, HMAX, lds);
    const void *kern = reinterpret_cast<const void *>(&k_made_rqs_sequential<HMAX, KT, LRS, CT>);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return fail(TFK_ELAUNCH, "%s: cannot reserve %zu bytes of LDS", fn, lds);
    }
    int64_t grid = (N + block - 1) / block;
    if (grid > max_grid()) grid = max_grid();
    hipLaunchKernelGGL((k_made_rqs_sequential<HMAX, KT, LRS, CT>), dim3((int)grid), dim3(block), lds, s, z, x, logdet,
                       (long long)N, D, W1t, b1, W2, b2, C, accumulate);
    return check_launch(fn);
}

}  // namespace tfk

using namespace tfk;

extern "C" {

int tfk_made_affine_sequential(const float *z, float *x, float *logdet, int64_t N, int32_t D,
                               const float *W1t, const float *b1, const float *W2, const float *b2,
                               int32_t hidden_padded, int32_t divide, int32_t accumulate, void *stream)
{
    const char *fn = "tfk_made_affine_sequential";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D < 1 || D > 1024) return fail(TFK_EINVAL, "%s: D = %d not in [1, 1024]", fn, D);
    if (hidden_padded != 8 && hidden_padded != 16 && hidden_padded != 32 && hidden_padded != 64)
        return fail(TFK_EINVAL, "%s: hidden_padded = %d must be 8, 16, 32 or 64", fn, hidden_padded);
    if (N == 0) return TFK_OK;
    if (!z || !x || !logdet || !W1t || !b1 || !W2 || !b2) return fail(TFK_EINVAL, "%s: null pointer", fn);
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (hidden_padded) {
    case 8: return launch_made<8>(z, x, logdet, N, D, W1t, b1, W2, b2, divide, accumulate, s, fn);
    case 16: return launch_made<16>(z, x, logdet, N, D, W1t, b1, W2, b2, divide, accumulate, s, fn);
    case 32: return launch_made<32>(z, x, logdet, N, D, W1t, b1, W2, b2, divide, accumulate, s, fn);
    default: return launch_made<64>(z, x, logdet, N, D, W1t, b1, W2, b2, divide, accumulate, s, fn);
    }
}

int64_t tfk_made_lrs_sequential_lds_bytes(int32_t D, int32_t hidden_padded, int32_t n_bins)
{
    const int64_t P = 4 * (int64_t)n_bins;
    return 4 * ((int64_t)D * hidden_padded + hidden_padded + P * D * hidden_padded + P * D) + 4 * 64 * ((int64_t)D + 1);
}

int tfk_made_lrs_sequential(const float *z, float *x, float *logdet, int64_t N, int32_t D,
                            const float *W1t, const float *b1, const float *W2, const float *b2,
                            int32_t hidden_padded, int32_t n_bins, float boundary, int32_t accumulate,
                            void *stream)
{
    const char *fn = "tfk_made_lrs_sequential";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D < 1 || D > 1024) return fail(TFK_EINVAL, "%s: D = %d not in [1, 1024]", fn, D);
    if (hidden_padded != 8 && hidden_padded != 16) return fail(TFK_EINVAL, "%s: hidden_padded = %d must be 8 or 16", fn, hidden_padded);
    if (n_bins != 8) return fail(TFK_EINVAL, "%s: n_bins = %d (the kernel is built for 8)", fn, n_bins);
    if (!(boundary > 0.0f)) return fail(TFK_EINVAL, "%s: boundary must be positive", fn);
    if (N == 0) return TFK_OK;
    if (!z || !x || !logdet || !W1t || !b1 || !W2 || !b2) return fail(TFK_EINVAL, "%s: null pointer", fn);
    LrsConst C;
    C.minimum = -boundary;
    C.maximum = boundary;
    C.span = (float)((double)boundary + (double)boundary);
    C.scale = (float)(1.0 - 1e-2 * (double)n_bins);
    C.c = (float)log(exp(1.0 - 1e-5) - 1.0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hidden_padded == 8) return launch_made_rqs<8, 8, true>(z, x, logdet, N, D, W1t, b1, W2, b2, C, accumulate, s, fn);
    return launch_made_rqs<16, 8, true>(z, x, logdet, N, D, W1t, b1, W2, b2, C, accumulate, s, fn);
}

int64_t tfk_made_rqs_sequential_lds_bytes(int32_t D, int32_t hidden_padded, int32_t n_bins)
{
    const int64_t P = 3 * (int64_t)n_bins - 1;
    // weights + 64 staged rows (the smallest workgroup)
    return 4 * ((int64_t)D * hidden_padded + hidden_padded + P * D * hidden_padded + P * D) + 4 * 64 * ((int64_t)D + 1);
}

int tfk_made_rqs_sequential(const float *z, float *x, float *logdet, int64_t N, int32_t D,
                            const float *W1t, const float *b1, const float *W2, const float *b2,
                            int32_t hidden_padded, int32_t n_bins, float boundary, int32_t accumulate,
                            void *stream)
{
    const char *fn = "tfk_made_rqs_sequential";
    if (N < 0) return fail(TFK_EINVAL, "%s: N = %lld < 0", fn, (long long)N);
    if (D < 1 || D > 1024) return fail(TFK_EINVAL, "%s: D = %d not in [1, 1024]", fn, D);
    if (hidden_padded != 8 && hidden_padded != 16) return fail(TFK_EINVAL, "%s: hidden_padded = %d must be 8 or 16", fn, hidden_padded);
    if (n_bins != 8) return fail(TFK_EINVAL, "%s: n_bins = %d (the kernel is built for 8)", fn, n_bins);
    if (!(boundary > 0.0f)) return fail(TFK_EINVAL, "%s: boundary must be positive", fn);
    if (N == 0) return TFK_OK;
    if (!z || !x || !logdet || !W1t || !b1 || !W2 || !b2) return fail(TFK_EINVAL, "%s: null pointer", fn);
    RqsConst C;
    C.minimum = -boundary;
    C.maximum = boundary;
    C.span = (float)((double)boundary + (double)boundary);
    C.scale = (float)(1.0 - 1e-3 * (double)n_bins);
    C.c = (float)log(expm1(1.0 - 1e-5));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hidden_padded == 8) return launch_made_rqs<8, 8, false>(z, x, logdet, N, D, W1t, b1, W2, b2, C, accumulate, s, fn);
    return launch_made_rqs<16, 8, false>(z, x, logdet, N, D, W1t, b1, W2, b2, C, accumulate, s, fn);
}

}  // extern "C"
